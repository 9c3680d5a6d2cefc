#!/usr/bin/env python3
"""Chain queue of the LAST inverse in a rocprofv3 --kernel-trace CSV: every kernel of steps 5 and 6 with the gap before it, and totals
of kernel time / gap time over the whole chain."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
sym = [i for i, r in enumerate(rows) if "symmetrize_kernel" in r["Kernel_Name"]]
end = sym[-1] + 2
start = sym[-2] + 2 if len(sym) > 1 else 0
R = rows[start:end]
chainq = collections.Counter(r["Queue_Id"] for r in R if "leaf" in r["Kernel_Name"]).most_common(1)[0][0]
C = [r for r in R if r["Queue_Id"] == chainq]
t0 = int(C[0]["Start_Timestamp"])
busy = gap = 0.0
prev = None
leafs = 0
for r in C:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - prev) / 1e3 if prev else 0.0
    busy += (e - s) / 1e3; gap += max(g, 0.0)
    if "leaf" in r["Kernel_Name"]: leafs += 1
    if 8 <= leafs <= 12:
        print("%9.1f  gap %6.1f  dur %7.1f  %5d WGs x %4s  %s" % ((s - t0) / 1e3, g, (e - s) / 1e3, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_X"]) // max(1, int(r["Workgroup_Size_Y"])), r["Workgroup_Size_X"], short(r["Kernel_Name"])))
    prev = e
print("chain queue: %d kernels, busy %.0f us, gaps %.0f us, span %.0f us" % (len(C), busy, gap, (int(C[-1]["End_Timestamp"]) - t0) / 1e3))
