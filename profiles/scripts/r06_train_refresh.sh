#!/bin/bash
# the training step's files of the round-6 pass again (after the consumer-side merge): bench line, kernel trace, both launch-structure A/Bs
set -x
export TMPDIR=/tmp
O=gpurun_out/r06t; rm -rf $O; mkdir -p $O
python bench.py --workload train --steps 30 --warmup 10 > $O/train.json 2> $O/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -- python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/train_profiled.json 2> $O/train_trace.err
cp $(find $O/train_trace -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv; rm -rf $O/train_trace
bash profiles/scripts/train_ab.sh > $O/train_ab.txt 2>&1
bash profiles/scripts/train_ab_merge.sh > $O/train_ab_merge.txt 2>&1
rm -f gpurun_out/train_ab/*.pt
