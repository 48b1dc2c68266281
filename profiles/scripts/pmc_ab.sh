#!/bin/bash
# scratch/pmc_ab.sh <file> "<flags>" ... : per variant one rocprofv3 --pmc pass over a short forward bench; prints the phi kernel's counters
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
C=signnet_basisnet_amd/csrc
f=$1; shift
SET="${PMCSET:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE}"
i=0
for flags in "$@"; do
  fl="$flags"; [ "$fl" = "base" ] && fl=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $fl -c $C/$f.hip -o $C/$f.o || { echo "build failed: $flags"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o signnet_basisnet_amd/libsignnet_hip.so $C/*.o
  O=gpurun_out/pmcab/p$i; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O -- python bench.py --steps 20 --warmup 5 --streams 1 --no-overlap --no-scatter --no-cpu-baseline > $O.json 2> $O.err
  python - "$flags" $O <<'P'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[2] + '/**/*counter_collection.csv', recursive=True)
if not fs: print('PMC', sys.argv[1], 'no csv'); sys.exit()
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'k_phi_fused' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('PMC', sys.argv[1], {n: round(sum(v) / len(v)) for n, v in sorted(acc.items())}, 'launches', len(next(iter(acc.values()), [])))
P
  i=$((i+1))
done 2>&1 | grep "^PMC\|failed"
