#!/usr/bin/env python3
"""Developer tool: average the rocprofv3 PMC rows and trace durations of hvlc::k_encode found under a directory tree."""
import collections, csv, glob, os, sys
root = sys.argv[1]
KERNEL = os.environ.get("KERNEL", "k_encode")
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(os.path.join(d, "**/*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(os.path.basename(d.rstrip("/")), "us=%.1f" % (sum(dur) / max(len(dur), 1)),
          " ".join(f"{k[3:]}={sum(v)/len(v)/1e6:.1f}M" for k, v in sorted(agg.items())))
