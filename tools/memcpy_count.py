#!/usr/bin/env python3
"""Counts the records of rocprofv3 --memory-copy-trace CSVs (directories given on the command line) by direction and prints them side
by side: the reference on DeviceType::HIP must show the SAME number of copies for 10 and for 110 steady-state cycles."""
import csv
import glob
import os
import sys

rows = []
for d in sys.argv[1:]:
    count = {}
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = r.get("Direction") or r.get("Name") or "copy"
            count[key] = count.get(key, 0) + 1
    kernels = 0
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        kernels += sum(1 for _ in csv.DictReader(open(f)))
    rows.append((os.path.basename(d.rstrip("/")), count, kernels))
for name, count, kernels in rows:
    print(f"{name}: kernel dispatches {kernels}; memory copies {sum(count.values())} {dict(sorted(count.items()))}")
if len(rows) == 2:
    same = rows[0][1] == rows[1][1]
    print("copies independent of the number of cycles:", same)
