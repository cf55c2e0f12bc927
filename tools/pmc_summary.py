#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-48:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
