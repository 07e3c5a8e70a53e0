#!/usr/bin/env python3
"""Mean of every PMC counter per dispatch for the kernels whose name contains the given substring (rocprofv3 --pmc CSV output)."""
import csv
import glob
import os
import sys

d, needle = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if needle not in name:
            continue
        key = (name.replace("(anonymous namespace)::", "").replace("jst::kernels::", "").replace("jst::dev::", "").split("(")[0][:80], r["Counter_Name"])
        s = acc.setdefault(key, [0.0, 0])
        s[0] += float(r["Counter_Value"])
        s[1] += 1
for (name, counter), (total, n) in sorted(acc.items()):
    print(f"{name:80s} {counter:32s} mean {total / n:16.1f} over {n} dispatches")
