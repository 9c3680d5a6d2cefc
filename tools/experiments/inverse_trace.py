#!/usr/bin/env python3
"""Per-size summary of the kernels of ONE SPD inverse (cholinv.hip) from a rocprofv3 kernel trace:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $REPO/tools/time_mf.py --reps 2
    python tools/experiments/inverse_trace.py /tmp/kt/*/*kernel_trace.csv

Takes the launches of the last repetition, groups the GEMMs by grid (tiles in x, tiles in y)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if "gemm_nt" in r["Kernel_Name"] or "leaf" in r["Kernel_Name"]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = g[len(g) * (reps - 1) // reps:]
t0 = min(int(r["Start_Timestamp"]) for r in g)
t1 = max(int(r["End_Timestamp"]) for r in g)
print("span %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(g)))
agg = collections.defaultdict(lambda: [0, 0])
for r in g:
    key = ("leaf" if "leaf" in r["Kernel_Name"] else "gemm", int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    agg[key][0] += 1
    agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-22s %4d launches %8.3f ms  avg %8.1f us" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
