set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r02dgl
rm -rf $O; mkdir -p $O
python bench.py --workload dgl --steps 50 --warmup 10 > $O/dgl_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --workload dgl --steps 50 --warmup 10 --no-cpu-baseline > $O/dgl_profiled.json 2> $O/trace.err
python bench.py --workload train --steps 30 --warmup 10 > $O/train_bench.json 2> $O/train.err
python bench.py --config 4 --steps 20 --warmup 5 > $O/basisnet_bench.json 2> $O/bn.err
find $O -name "*kernel_stats.csv"
