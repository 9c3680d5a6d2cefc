#!/usr/bin/env python3
"""Per-kernel, per-launch averages of rocprofv3 --pmc counter_collection.csv files (any counters).
usage: sq_summary.py out.csv counter_collection.csv [more.csv ...]"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: [set(), 0.0])
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0].split("<")[0]
        key = (name, row["Counter_Name"])
        acc[key][0].add((path, row["Dispatch_Id"]))
        acc[key][1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as fh:
    fh.write("kernel,launches,counter,value_per_launch\n")
    for (name, counter), (ids, total) in sorted(acc.items()):
        fh.write("%s,%d,%s,%.6g\n" % (name, len(ids), counter, total / max(len(ids), 1)))
