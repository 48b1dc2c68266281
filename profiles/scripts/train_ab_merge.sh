#!/bin/bash
# A/B of the consumer-side coefficient merge (SN_TRAIN_MERGE) on one box, alternating: graphed ms per step, launches, bit comparison
out=gpurun_out/train_ab; mkdir -p $out
for r in 1 2; do for mm in 0 1; do
  SN_TRAIN_MERGE=$mm python bench.py --workload train --steps 30 --warmup 10 --no-cpu-baseline > $out/train_m${mm}_$r.json 2> $out/train_m${mm}_$r.err
  python - <<P
import json
d=json.load(open("$out/train_m${mm}_$r.json"))
print("merge=$mm graphed ms", round(d["graphed"]["ms_per_step"],4), "eager", round(d["eager"]["ms_per_step"],3), "launches", d["launches_per_step"])
P
done; done
for mm in 0 1; do SN_TRAIN_MERGE=$mm python profiles/scripts/train_ab_bits.py $out/bits_m$mm.pt > $out/bits_m$mm.log 2>&1; done
echo "== merge 0 vs 1"; python profiles/scripts/train_ab_bits.py --compare $out/bits_m0.pt $out/bits_m1.pt
rm -f $out/*.pt
