#!/usr/bin/env python3
"""Developer tool: profiles/traffic.json from the two PMC passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py`.
usage: make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <algorithmic_bytes> [round]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def avg(path, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_encode" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v)
f, w, alg = sys.argv[1], sys.argv[2], int(sys.argv[3])
rnd = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fk, wk = avg(f, "FETCH_SIZE"), avg(w, "WRITE_SIZE")
out = {"H2.2-4GiB": {
    "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024)),
    "fetch_size_kb_raw": fk, "write_size_kb_raw": wk,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 for 16 B/lane coalesced streaming reads "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is; both x1024 (KB -> bytes)",
    "kernel": "hvlc::k_encode",
    "source": [f"profiles/r{rnd:02d}_pmc_fetch_size.csv", f"profiles/r{rnd:02d}_pmc_write_size.csv"],
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline (separate passes)",
    "algorithmic_bytes": alg, "round": rnd}}
out["H2.2-4GiB"]["ratio_to_algorithmic"] = round(out["H2.2-4GiB"]["hbm_bytes_per_launch"] / alg, 4)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["H2.2-4GiB"]))
