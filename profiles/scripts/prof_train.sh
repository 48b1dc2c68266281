set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/${SN_PROF_TAG:-r03train}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/train_profiled.json 2> $O/trace.err
find $O -name "*kernel_stats.csv" | head
