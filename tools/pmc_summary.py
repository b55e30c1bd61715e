#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into the small tables committed under profiles/.
  python tools/pmc_summary.py <dir with *_kernel_stats.csv / *_counter_collection.csv> ...
Per kernel: dispatch count and the mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void bpgpu::", "").replace("bpgpu::", "")
    return name.split("(")[0]


for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"# {f}")
        print("kernel,dispatches,counter,mean_per_dispatch")
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                print(f"{k},{len(v)},{c},{sum(v) / len(v):.6g}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
        print(f"# {f}")
        print("kernel,calls,avg_us,total_ms,percent")
        for row in csv.DictReader(open(f)):
            print(f"{short(row['Name'])},{row['Calls']},{float(row['AverageNs']) / 1e3:.2f},{float(row['TotalDurationNs']) / 1e6:.3f},{row['Percentage']}")
