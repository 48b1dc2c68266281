#!/bin/bash
# extra SQ counter passes over the forward bench (sequential pass): instruction mix, LDS activity / bank conflicts, wait reasons
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/pmc_detail
rm -rf $O; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters.txt
B="python bench.py --steps 20 --warmup 5 --streams 1 --no-overlap --no-scatter --no-cpu-baseline --no-extras"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- $B > $O/p$i.json 2> $O/p$i.err || echo "pass $i failed" >> $O/fail.txt
  i=$((i+1))
done
python - <<'P'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_detail/p*/')):
    fs = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'k_phi_fused' in k or 'k_rho_fused' in k or 'k_gnn_coop' in k:
            acc[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in acc.items():
        print(d.split('/')[-2], k, {n: round(sum(v) / len(v)) for n, v in c.items()})
P
