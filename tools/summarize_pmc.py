#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_passes.sh.  usage: summarize_pmc.py <dir> [kernel-substr...]"""
import csv, glob, os, sys
from collections import defaultdict

src = sys.argv[1]
want = sys.argv[2:] or ["edge_kernel", "node_kernel"]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        for w in want:
            if w in k:
                i = k.find(w)
                name = k[i:k.find("(", i)]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    print(f"== {name}")
    for c in sorted(acc[name]):
        v = acc[name][c]
        print(f"  {c:<34}{sum(v) / len(v):>18.1f}  (n={len(v)})")
