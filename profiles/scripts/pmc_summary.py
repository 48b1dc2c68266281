"""per-kernel means of rocprofv3 --pmc counter_collection.csv files -> the compact CSVs kept under profiles/."""
import csv, glob, sys, collections, json
O, tag = sys.argv[1], sys.argv[2]
def load(d):
    import os
    f = max(glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    disp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        disp[k].add(r["Dispatch_Id"])
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
    return per, meta, disp
per, meta, disp = load("pmc_sq")
names = sorted({c for k in per for c in per[k]})
with open(f"profiles/{tag}_pmc_sq.csv", "w") as f:
    f.write("kernel,dispatches,VGPR,AGPR,SGPR,scratch,LDS,wg_size,grid," + ",".join(n + "_mean" for n in names) + "\n")
    for k in sorted(per):
        f.write('"%s",%d,%s,' % (k, len(disp[k]), ",".join(meta[k])) + ",".join("%.1f" % (sum(per[k][n]) / max(1, len(per[k][n]))) for n in names) + "\n")
pf, _, df = load("pmc_fetch"); pw, _, dw = load("pmc_write")
out = {}
with open(f"profiles/{tag}_pmc_hbm.csv", "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_bytes_2xFETCH_plus_WRITE\n")
    for k in sorted(pf):
        fe = sum(pf[k]["FETCH_SIZE"]) / len(pf[k]["FETCH_SIZE"]); wr = sum(pw[k]["WRITE_SIZE"]) / len(pw[k]["WRITE_SIZE"]) if k in pw else 0.0
        f.write('"%s",%d,%.1f,%.1f,%d\n' % (k, len(df[k]), fe, wr, int((2 * fe + wr) * 1024)))
        out[k] = (fe, wr)
print({k[:40]: v for k, v in out.items() if "fused" in k or "coop" in k})
sq = {k: {n: sum(v[n]) / len(v[n]) for n in v} for k, v in per.items()}
for k in sq:
    if "phi_fused" in k or "gnn_coop" in k or "rho_fused" in k:
        s = sq[k]; print(k[:50], "MFMA busy / (SIMDs x GUI_ACTIVE/8?) :", s.get("SQ_VALU_MFMA_BUSY_CYCLES"), s.get("GRBM_GUI_ACTIVE"), s.get("SQ_BUSY_CYCLES"))
