#!/bin/bash
# A/B of an environment switch on one box, alternating:  gpurun -- bash profiles/scripts/ab_env.sh VAR A B [rounds] [bench flags...]
# prints value / per-kernel event times of the forward bench (sequential pass) for VAR=A and VAR=B
var=$1; a=$2; b=$3; rounds=${4:-2}; shift 4
out=gpurun_out/ab_env; mkdir -p $out
for r in $(seq 1 $rounds); do for v in $a $b; do
  env $var=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-scatter --no-extras --no-overlap --streams 1 "$@" > $out/${var}_${v}_$r.json 2> $out/${var}_${v}_$r.err
  python - "$var=$v" $out/${var}_${v}_$r.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=d['kernels']
    print('AB', sys.argv[1], '| value', round(d['value']), 'ms', round(d['ms_per_step'],4), '|', ' '.join(f"{n[3:-10]} {v['mean_us']:.1f}" for n,v in k.items()), '| clock', d.get('device',{}).get('measured_clock_mhz'))
except Exception as e:
    print('AB', sys.argv[1], 'FAILED', e)
P
done; done
