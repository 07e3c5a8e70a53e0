#!/usr/bin/env python3
"""Compact view of a rocprofv3 --stats kernel_stats.csv: python tools/kstats.py <dir or csv>"""
import csv, glob, os, re, sys
path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
for f in files:
    for row in csv.DictReader(open(f)):
        name = re.sub(r"jst::(kernels|dev)::(\(anonymous namespace\)::)?", "", row["Name"])
        name = re.sub(r"HIP_vector_type<float, 2u>", "f2", name)
        name = name.split("(")[0][:90]
        print(f"{float(row['AverageNs'])/1e3:9.2f} us x{int(row['Calls']):5d}  {float(row['Percentage']):5.1f}%  {name}")
