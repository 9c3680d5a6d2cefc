#!/usr/bin/env python3
"""Idle time between consecutive kernels of the L-BFGS loop from a rocprofv3 --kernel-trace CSV:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python tools/experiments/iter_time.py
    python tools/experiments/gap_trace.py /tmp/kt/*/*kernel_trace.csv
Prints one iteration (between two consecutive plm_logits launches late in the run): kernel, duration, gap before it."""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [k for k, r in enumerate(rows) if "plm_logits" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
prev_end = int(rows[a - 1]["End_Timestamp"])
tot_gap = tot_k = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
    print("%-32s %9.1f us   gap before %7.1f us" % (name[:32], (e - s) / 1e3, (s - prev_end) / 1e3))
    tot_gap += s - prev_end
    tot_k += e - s
    prev_end = e
print("kernels %.3f ms, gaps %.3f ms" % (tot_k / 1e6, tot_gap / 1e6))
