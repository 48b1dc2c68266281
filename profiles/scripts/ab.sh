#!/bin/bash
# A/B of build variants of one csrc file on one box:  scratch/ab.sh <file.hip> "<flags A>" "<flags B>" ...
# ("base" = no extra flags).  Each variant: rebuild the object, relink, run the forward bench twice (sequential pass only).
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab; mkdir -p $out
C=signnet_basisnet_amd/csrc
f=$1; shift
i=0
for flags in "$@"; do
  fl="$flags"; [ "$fl" = "base" ] && fl=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $fl -c $C/$f.hip -o $C/$f.o || { echo "build failed: $flags"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o signnet_basisnet_amd/libsignnet_hip.so $C/*.o
  for rep in $(seq 1 ${REPS:-2}); do
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-scatter --no-extras --no-overlap --streams 1 > $out/b_${i}_$rep.json 2> $out/b_${i}_$rep.err
    python - "$flags" $out/b_${i}_$rep.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k=d['kernels']
    print('AB', sys.argv[1], '| value', round(d['value']), 'seq ms', round(d['sequential']['ms_per_step'],4), '| phi us', round(k['sn_phi_fused_f32']['mean_us'],1), 'rho', round(k['sn_rho_fused_f32']['mean_us'],1), 'gnn', round(k['sn_gnn_fused_f32']['mean_us'],1), 'plan', round(k.get('sn_batch_plan',{}).get('mean_us',0),1))
except Exception as e:
    print('AB', sys.argv[1], 'FAILED', e)
P
  done
  i=$((i+1))
done 2>&1 | grep "^AB" | tee $out/summary.txt
