#!/bin/bash
# A/B of one csrc file against ANOTHER SOURCE of it on one box:  scripts/ab_file.sh <file> <alternative.hip> [rounds]
# (e.g. the previous commit's version: `git show HEAD~1:signnet_basisnet_amd/csrc/fused_phi.hip > profiles/scripts/_old.hip`).
# Alternates  shipped, alternative, shipped, alternative ...; each build: the forward bench twice (sequential pass only).
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab; mkdir -p $out
C=signnet_basisnet_amd/csrc
f=$1; alt=$2; rounds=${3:-2}
i=0
for r in $(seq 1 $rounds); do
  for which in shipped alt; do
    src=$C/$f.hip; [ $which = alt ] && src=$alt
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I$C -x hip -c $src -o $C/$f.o || { echo "build failed: $which"; continue; }
    hipcc --offload-arch=gfx950 -shared -fPIC -o signnet_basisnet_amd/libsignnet_hip.so $C/*.o
    for rep in 1 2; do
      python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-scatter --no-extras --no-overlap --streams 1 > $out/f_${i}_$rep.json 2> $out/f_${i}_$rep.err
      python - "$which" $out/f_${i}_$rep.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k=d['kernels']
    print('AB', sys.argv[1], '| value', round(d['value']), 'ms', round(d['ms_per_step'],4), '| phi us', round(k['sn_phi_fused_f32']['mean_us'],1), 'rho', round(k['sn_rho_fused_f32']['mean_us'],1), 'gnn', round(k['sn_gnn_fused_f32']['mean_us'],1), 'plan', round(k.get('sn_batch_plan',{}).get('mean_us',0),1))
except Exception as e:
    print('AB', sys.argv[1], 'FAILED', e)
P
    done
    i=$((i+1))
  done
done 2>&1 | grep "^AB" | tee $out/summary_file.txt
