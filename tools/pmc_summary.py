#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 counter_collection CSVs (one or more --pmc passes).
Usage: python tools/pmc_summary.py gpurun_out/prof4 [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = defaultdict(float)
    names = {}
    for row in csv.DictReader(open(path)):
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])  # summed over XCDs / dimensions
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (disp, counter), v in per_dispatch.items():
        acc[names[disp]][counter].append(v)
for kernel, counters in acc.items():
    if want not in kernel:
        continue
    print(kernel[:120])
    for counter, vals in sorted(counters.items()):
        vals = vals[len(vals) // 4:]  # drop warm-up dispatches
        print(f"    {counter:24s} mean {sum(vals)/len(vals):16.1f}   (n={len(vals)})")
