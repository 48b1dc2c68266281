"""gpurun_out/r03f (written by r03_final.sh) -> the compact files kept under profiles/ (r03_*)."""
import csv, glob, os, shutil, subprocess, sys, collections
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03f"
def newest(pat):
    return max(glob.glob(f"{O}/{pat}", recursive=True), key=os.path.getmtime)
for src, dst in (("bench.json", "r03_bench.json"), ("bench_profiled.json", "r03_bench_profiled.json"), ("train.json", "r03_train_bench.json"),
                 ("train_profiled.json", "r03_train_bench_profiled.json"), ("config0.json", "r03_config0_bench.json"),
                 ("config2.json", "r03_config2_bench.json"), ("config4.json", "r03_config4_bench.json"), ("dgl.json", "r03_dgl_bench.json"),
                 ("evd.json", "r03_evd_bench.json")):
    shutil.copy(f"{O}/{src}", f"profiles/{dst}")
shutil.copy(newest("trace/**/*kernel_stats.csv"), "profiles/r03_kernel_stats.csv")
shutil.copy(newest("train_trace/**/*kernel_stats.csv"), "profiles/r03_train_kernel_stats.csv")
subprocess.check_call([sys.executable, "profiles/scripts/pmc_summary.py", O, "r03"])
rows = list(csv.DictReader(open(newest("train_pmc_sq/**/*counter_collection.csv"))))
per = collections.defaultdict(lambda: collections.defaultdict(list)); disp = collections.defaultdict(set)
for r in rows:
    per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"])); disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
names = sorted({c for k in per for c in per[k]})
with open("profiles/r03_train_pmc_sq.csv", "w") as f:
    f.write("kernel,dispatches," + ",".join(n + "_mean" for n in names) + ",mfma_busy_frac\n")
    for k in sorted(per):
        m = {n: sum(per[k][n]) / max(1, len(per[k][n])) for n in names}
        f.write('"%s",%d,' % (k, len(disp[k])) + ",".join("%.1f" % m[n] for n in names)
                + ",%.3f\n" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, 128.0 * m.get("GRBM_GUI_ACTIVE", 0.0))))
