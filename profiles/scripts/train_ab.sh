#!/bin/bash
# A/B of the training step under the launch-structure switches (graphed ms per step + bit comparison); run on the GPU box from the repo root
out=gpurun_out/train_ab; mkdir -p $out
for ff in 0 1; do for dd in 0 1; do
  SN_TRAIN_FUSE_FINISH=$ff SN_TRAIN_DEFER_DW=$dd python profiles/scripts/train_ab_bits.py $out/bits_f${ff}_d${dd}.pt > $out/bits_f${ff}_d${dd}.log 2>&1
  SN_TRAIN_FUSE_FINISH=$ff SN_TRAIN_DEFER_DW=$dd python bench.py --workload train --steps 30 --warmup 10 --no-cpu-baseline > $out/train_f${ff}_d${dd}.json 2> $out/train_f${ff}_d${dd}.err
  python - <<P
import json
d=json.load(open("$out/train_f${ff}_d${dd}.json"))
print("fuse_finish=$ff defer_dw=$dd graphed ms", round(d["graphed"]["ms_per_step"],4), "eager", round(d["eager"]["ms_per_step"],3), "launches", d["launches_per_step"])
P
done; done
for t in f0_d1 f1_d0 f1_d1; do echo "== f0_d0 vs $t"; python profiles/scripts/train_ab_bits.py --compare $out/bits_f0_d0.pt $out/bits_$t.pt; done
rm -f $out/*.pt
