#!/bin/bash
# A/B of fused_phi.hip build variants on one box: scratch/ab_phi.sh "<flags A>" "<flags B>" ...
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab; mkdir -p $out
C=signnet_basisnet_amd/csrc
i=0
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $flags -c $C/fused_phi.hip -o $C/fused_phi.o || { echo "build failed: $flags"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o signnet_basisnet_amd/libsignnet_hip.so $C/*.o
  for rep in 1 2; do
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-scatter --streams 1 > $out/b_${i}_$rep.json 2> $out/b_${i}_$rep.err
    python - "$flags" $out/b_${i}_$rep.json <<'P'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d['kernels']
print(sys.argv[1], '| value', round(d['value']), 'seq ms', round(d['sequential']['ms_per_step'],4), '| phi us', round(k['sn_phi_fused_f32']['mean_us'],1), 'rho', round(k['sn_rho_fused_f32']['mean_us'],1), 'gnn', round(k['sn_gnn_fused_f32']['mean_us'],1), 'roof', round(d['roofline']['mean_launch_us'],1))
P
  done
  i=$((i+1))
done 2>&1 | grep "^-D" | tee $out/summary.txt
