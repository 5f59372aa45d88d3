#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats CSV (…_kernel_stats.csv)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print(f"{r['Name'][:88]:88s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs']) / 1e6:9.2f} "
          f"avg_us={float(r['AverageNs']) / 1e3:9.1f} {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
print("total ms", tot / 1e6)
