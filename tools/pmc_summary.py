#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel name: mean of each counter over dispatches."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-28s n=%-5d mean=%.4g" % (c, len(v), sum(v) / len(v)))
