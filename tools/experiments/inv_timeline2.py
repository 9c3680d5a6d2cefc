#!/usr/bin/env python3
"""Timeline of the LAST SPD inverse of a rocprofv3 kernel trace (tools/time_inv.py): an inverse ends with its X^T X launch (the
largest grid), so the last one is what lies between the last two of those.  Prints per-queue busy time, the chain's panels
(time from leaf group to leaf group), every side-queue launch and the large launches.
    python inv_timeline2.py trace.csv [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
g = [r for r in rows if "gemm_" in r["Kernel_Name"] or "leaf" in r["Kernel_Name"]]
g.sort(key=lambda r: int(r["Start_Timestamp"]))
size = lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
big = max(size(r) for r in g)
ends = [k for k, r in enumerate(g) if size(r) == big]
g = g[(ends[-2] + 1 if len(ends) > 1 else 0):ends[-1] + 1]
t0 = int(g[0]["Start_Timestamp"])
us = lambda r: ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("span %.2f ms, %d launches" % ((int(g[-1]["End_Timestamp"]) - t0) / 1e6, len(g)))
busy = {}
for r in g:
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + us(r)[1]
print("busy ms per queue:", {q: round(v / 1e3, 2) for q, v in busy.items()})
main = max(busy, key=lambda q: sum(1 for r in g if r["Queue_Id"] == q))
def name(r):
    k = r["Kernel_Name"]
    return "leaf" if "leaf" in k else "reduce" if "reduce" in k else "dma4" if "dma_kernel<4, 4>" in k or "dma_kernel<4>" in k else "dma42" if "dma_kernel<4, 2>" in k else "dma2" if "dma_kernel<2" in k else "small" if "small" in k else "gemm"
def grid(r):
    return "%4d x %3d x %2d" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
leaves = [r for r in g if name(r) == "leaf"]
print("leaves %d, total %.2f ms, avg %.1f us" % (len(leaves), sum(us(r)[1] for r in leaves) / 1e3, sum(us(r)[1] for r in leaves) / max(1, len(leaves))))
last_leaf_end = us(leaves[-1])[0] + us(leaves[-1])[1]
print("last leaf ends at %.2f ms (end of the factorisation chain)" % (last_leaf_end / 1e3))
sm = [r for r in g if name(r) == "small" and r["Queue_Id"] == main and us(r)[0] < last_leaf_end]
print("chain small products: %d launches, total %.2f ms, avg %.1f us" % (len(sm), sum(us(r)[1] for r in sm) / 1e3, sum(us(r)[1] for r in sm) / max(1, len(sm))))
mg = [r for r in g if r["Queue_Id"] == main]
gaps = [(us(b)[0] - us(a)[0] - us(a)[1], us(a)[0] + us(a)[1]) for a, b in zip(mg, mg[1:])]
print("main queue gaps: total %.2f ms; gaps >= 30 us:" % (sum(x for x, _ in gaps) / 1e3), ["%.0f@%.0f" % (x, at) for x, at in gaps if x >= 30])
print("launches of at least %.0f us, and every side-queue launch:" % min_us)
for r in g:
    s, d = us(r)
    if d >= min_us or r["Queue_Id"] != main:
        print("%9.1f us  +%8.1f us  %-6s grid %s  queue %s" % (s, d, name(r), grid(r), r["Queue_Id"]))
