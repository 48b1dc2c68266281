#!/bin/bash
# rocprofv3 kernel trace of one bench.py command line:  gpurun -- bash profiles/scripts/prof_trace.sh <outdir-under-gpurun_out> <tag> <bench.py flags...>
export TMPDIR=/tmp
O=gpurun_out/$1; tag=$2; shift 2; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python bench.py "$@" > $O/${tag}_prof.json 2> $O/${tag}_prof.err
cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv; rm -rf $O/prof_$tag
head -${LINES_SHOWN:-14} $O/${tag}_kernel_stats.csv | cut -c1-220
