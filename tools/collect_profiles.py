#!/usr/bin/env python3
"""Developer tool: copy the summaries of gpurun_out/round (tools/measure_round.sh) into profiles/ under round names
and derive profiles/traffic.json.    python3 tools/collect_profiles.py 2"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]); R = f"r{rnd:02d}"
src = os.path.join(ROOT, "gpurun_out", "round"); dst = os.path.join(ROOT, "profiles")
def one(pat):
    g = glob.glob(os.path.join(src, pat), recursive=True)
    assert g, pat
    return max(g, key=os.path.getmtime)      # (gpurun MERGES into gpurun_out/: an older run's files may still lie there)
shutil.copy(one("prof/**/*kernel_stats.csv"), os.path.join(dst, f"{R}_kernel_stats.csv"))
# the trace: k_encode rows only (the generator and histogram launches of the set-up are not the subject)
rows = list(csv.DictReader(open(one("prof/**/*kernel_trace.csv"))))
keep = [r for r in rows if "k_encode" in r["Kernel_Name"]]
with open(os.path.join(dst, f"{R}_kernel_trace.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
shutil.copy(one("pmc_fetch/**/*counter_collection.csv"), os.path.join(dst, f"{R}_pmc_fetch_size.csv"))
shutil.copy(one("pmc_write/**/*counter_collection.csv"), os.path.join(dst, f"{R}_pmc_write_size.csv"))
# SQ counters: one CSV, the k_encode rows of both passes
out = []
for p in ("sq1", "sq2"):
    for r in csv.DictReader(open(one(f"{p}/**/*counter_collection.csv"))):
        if "k_encode" in r["Kernel_Name"]:
            out.append({"pass": p, "Dispatch_Id": r["Dispatch_Id"], "Kernel_Name": "hvlc::k_encode", "Grid_Size": r["Grid_Size"],
                        "Workgroup_Size": r["Workgroup_Size"], "VGPR_Count": r.get("VGPR_Count", ""), "LDS_Block_Size": r.get("LDS_Block_Size", ""),
                        "Counter_Name": r["Counter_Name"], "Counter_Value": r["Counter_Value"]})
with open(os.path.join(dst, f"{R}_sq_counters.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(out[0].keys())); w.writeheader(); w.writerows(out)
for n in ("bench_n1", "bench_H2.2-256MiB", "bench_H4.0-4GiB", "bench_H1.0-4GiB", "next_rows", "hist", "mgpu_n1", "host_boundary"):
    p = os.path.join(src, n + ".json")
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{R}_{n}.json"))
if os.path.exists(os.path.join(src, "slow_path.txt")):
    shutil.copy(os.path.join(src, "slow_path.txt"), os.path.join(dst, f"{R}_slow_path.txt"))
b = json.load(open(os.path.join(src, "bench_n1.json")))
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), os.path.join(dst, f"{R}_pmc_fetch_size.csv"),
                os.path.join(dst, f"{R}_pmc_write_size.csv"), str(b["roofline"]["algorithmic_bytes"]), str(rnd)], check=True)
# per-wave-tile summary of the SQ counters
agg = {}
for r in out:
    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
tiles = b["config"]["bytes_per_gpu"] / 4096.0
per_part = {k: round(sum(v) / len(v) / tiles, 1) for k, v in sorted(agg.items())}
print(per_part, "per 4 KiB part")
# the secondary ceiling of SURVEY.md 8(d), VALU issue: instructions per launch from the SQ pass, next to the traffic entry
tj_path = os.path.join(dst, "traffic.json")
tj = json.load(open(tj_path))
tj["H2.2-4GiB"]["valu_wave_insts_per_launch"] = int(round(sum(agg["SQ_INSTS_VALU"]) / len(agg["SQ_INSTS_VALU"])))
tj["H2.2-4GiB"]["salu_wave_insts_per_launch"] = int(round(sum(agg["SQ_INSTS_SALU"]) / len(agg["SQ_INSTS_SALU"])))
tj["H2.2-4GiB"]["lds_wave_insts_per_launch"] = int(round(sum(agg["SQ_INSTS_LDS"]) / len(agg["SQ_INSTS_LDS"])))
tj["H2.2-4GiB"]["sq_source"] = f"profiles/{R}_sq_counters.csv"
json.dump(tj, open(tj_path, "w"), indent=1)
