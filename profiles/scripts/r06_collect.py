"""gpurun_out/r06f (written by r06_final.sh) -> the compact files kept under profiles/ (r06_*) and profiles/hbm_traffic.json."""
import csv, glob, json, os, shutil, subprocess, sys, collections
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06f"
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
def newest(pat):
    return max(glob.glob(f"{O}/{pat}", recursive=True), key=os.path.getmtime)
def last_json_line(path):
    with open(path) as f:
        return [l for l in f.read().splitlines() if l.startswith("{")][-1]
for src, dst in (("bench.json", "r06_bench.json"), ("bench_profiled.json", "r06_bench_profiled.json"), ("bench_driver_cmd.json", "r06_bench_driver_cmd.json"),
                 ("train.json", "r06_train_bench.json"), ("train_profiled.json", "r06_train_bench_profiled.json"),
                 ("config0.json", "r06_config0_bench.json"), ("config2.json", "r06_config2_bench.json"), ("config4.json", "r06_config4_bench.json"),
                 ("dgl.json", "r06_dgl_bench.json"), ("evd.json", "r06_evd_bench.json"), ("scatter.json", "r06_scatter_bench.json")):
    with open(f"profiles/{dst}", "w") as f:
        f.write(last_json_line(f"{O}/{src}") + "\n")
shutil.copy(newest("trace/**/*kernel_stats.csv"), "profiles/r06_kernel_stats.csv")
shutil.copy(newest("train_trace/**/*kernel_stats.csv"), "profiles/r06_train_kernel_stats.csv")
shutil.copy(newest("alleig_trace/**/*kernel_stats.csv"), "profiles/r06_alleig_kernel_stats.csv")
with open("profiles/r06_alleig_bench_profiled.json", "w") as f:
    f.write(last_json_line(f"{O}/alleig_profiled.json") + "\n")
for src, dst in (("pmc_detail.txt", "r06_pmc_sq_detail.txt"), ("gnn_stamps.txt", "r06_gnn_stage_stamps.txt"), ("plan_stamps.txt", "r06_plan_stamps.txt")):
    with open(f"{O}/{src}") as f, open(f"profiles/{dst}", "w") as g:
        g.write("".join(l for l in f if "amdgpu.ids" not in l and not l.startswith("+")))
subprocess.check_call([sys.executable, "profiles/scripts/pmc_summary.py", O, "r06"])
rows = list(csv.DictReader(open(newest("train_pmc_sq/**/*counter_collection.csv"))))
per = collections.defaultdict(lambda: collections.defaultdict(list)); disp = collections.defaultdict(set)
for r in rows:
    per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"])); disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
names = sorted({c for k in per for c in per[k]})
with open("profiles/r06_train_pmc_sq.csv", "w") as f:
    f.write("kernel,dispatches," + ",".join(n + "_mean" for n in names) + ",mfma_busy_frac\n")
    for k in sorted(per):
        m = {n: sum(per[k][n]) / max(1, len(per[k][n])) for n in names}
        f.write('"%s",%d,' % (k, len(disp[k])) + ",".join("%.1f" % m[n] for n in names)
                + ",%.3f\n" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, 128.0 * m.get("GRBM_GUI_ACTIVE", 0.0))))
# ---- scatter PMC: FETCH_SIZE / WRITE_SIZE of the standalone aggregation launches against their algorithmic bytes
def load(d):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"{d}/**/*counter_collection.csv"))):
        if "gin_gather" in r["Kernel_Name"] or "gine_gather" in r["Kernel_Name"]:
            per[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per
fe, wr = load("sc_fetch"), load("sc_write")
sc = json.loads(last_json_line(f"{O}/scatter.json"))
alg = {("gine" if "gine" in k else "gin", v["graphs"]): v["algorithmic_bytes"] for k, v in sc["kernels"].items()}
launches = {}
with open("profiles/r06_scatter_pmc_hbm.csv", "w") as out:
    out.write("kernel,graphs,grid,dispatches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_bytes_2xFETCH_plus_WRITE,algorithmic_bytes,ratio\n")
    for kind in ("gin_gather", "gine_gather"):
        ks = sorted([k for k in fe if kind + "<" in k[0]], key=lambda k: k[1])
        for g, k in zip([128, 1024, 2048, 6144], ks):
            f = sum(fe[k]) / len(fe[k]); w = sum(wr[k]) / len(wr[k]); b = int((2 * f + w) * 1024); a = alg[("gine" if "gine" in kind else "gin", g)]
            out.write('"%s",%d,%d,%d,%.0f,%.0f,%d,%d,%.3f\n' % (k[0], g, k[1], len(fe[k]), f, w, b, a, b / a))
            launches[str(a)] = {"kernel": f"{k[0]}, {g} graphs", "bytes": b}
# ---- hbm_traffic.json: what bench.py quotes as roofline.traffic (recorded, with its source)
hb = {}
for r in csv.DictReader(open("profiles/r06_pmc_hbm.csv")):
    for ent, key in (("sn_phi_fused_f32", "k_phi_fused"), ("sn_rho_fused_f32", "k_rho_fused"), ("sn_gnn_fused_f32", "k_gnn_coop")):
        if key in r["kernel"]:
            hb[ent] = {"fetch_kib": float(r["FETCH_SIZE_KiB_mean"]), "write_kib": float(r["WRITE_SIZE_KiB_mean"]), "bytes": int(r["hbm_bytes_2xFETCH_plus_WRITE"])}
old = json.load(open("profiles/hbm_traffic.json"))
old.update({"source": "profiles/r06_pmc_hbm.csv", "kernels": hb, "commit": commit,
            "_comment": old["_comment"].replace("r05_pmc_hbm.csv", "r06_pmc_hbm.csv")})
old["scatter"].update({"source": "profiles/r06_scatter_pmc_hbm.csv", "launches": launches,
                       "_comment": old["scatter"]["_comment"].replace("r05_scatter_pmc_hbm.csv", "r06_scatter_pmc_hbm.csv").replace("profiles/scripts/r05_final.sh", "profiles/scripts/r06_final.sh")})
json.dump(old, open("profiles/hbm_traffic.json", "w"), indent=2)
print("collected", commit)
