set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r02sc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --workload scatter --steps 5 --warmup 2 > $O/f.json 2> $O/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --workload scatter --steps 5 --warmup 2 > $O/w.json 2> $O/w.err
