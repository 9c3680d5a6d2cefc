#!/usr/bin/env python3
"""Timeline of ONE SPD inverse from a rocprofv3 kernel trace (last repetition): every launch of at least MIN_US
microseconds with its start, duration, grid and queue, plus the busy time per queue -- to see what the side-stream
products of cholinv.hip overlap with.   python inverse_timeline.py trace.csv [reps] [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
g = [r for r in rows if "gemm_" in r["Kernel_Name"] or "leaf" in r["Kernel_Name"]]
g.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last repetition: launches after the last gap of more than 2 ms
cut = 0
for k in range(1, len(g)):
    if int(g[k]["Start_Timestamp"]) - int(g[k - 1]["End_Timestamp"]) > 2e6:
        cut = k
g = g[cut:]
t0 = int(g[0]["Start_Timestamp"])
print("span %.2f ms, %d launches" % ((max(int(r["End_Timestamp"]) for r in g) - t0) / 1e6, len(g)))
busy = {}
for r in g:
    q = r.get("Queue_Id", "?")
    busy[q] = busy.get(q, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("busy ms per queue:", {q: round(v / 1e6, 2) for q, v in busy.items()})
for r in g:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < min_us:
        continue
    name = "leaf" if "leaf" in r["Kernel_Name"] else ("dma4" if "dma_kernel<4>" in r["Kernel_Name"] else "dma2" if "dma_kernel<2>" in r["Kernel_Name"] else "small" if "small" in r["Kernel_Name"] else "gemm")
    print("%9.1f us  +%8.1f us  %-5s grid %4d x %3d  queue %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, name,
          int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), r.get("Queue_Id", "?")))

# gaps on the main queue (the one with the most launches) and what the other queues did meanwhile
main = max(busy, key=lambda q: sum(1 for r in g if r.get("Queue_Id", "?") == q))
mg = [r for r in g if r.get("Queue_Id", "?") == main]
print("main queue %s: gaps of at least 30 us" % main)
for a, b in zip(mg, mg[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if gap >= 30:
        print("   gap %8.1f us at %9.1f us" % (gap, (int(a["End_Timestamp"]) - t0) / 1e3))
print("side queues, every launch:")
for r in g:
    if r.get("Queue_Id", "?") != main:
        print("%9.1f us  +%8.1f us  grid %4d x %3d  queue %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
              int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), r.get("Queue_Id", "?")))
# the main queue's launches inside the first side-queue window, one by one
side = [r for r in g if r.get("Queue_Id", "?") != main]
if side:
    w0 = int(side[0]["Start_Timestamp"])
    w1 = w0
    for r in side:
        if int(r["Start_Timestamp"]) - w1 > 500e3:
            break
        w1 = int(r["End_Timestamp"])
    print("main queue inside the first side window (%.1f .. %.1f us) and 300 us before it:" % ((w0 - t0) / 1e3, (w1 - t0) / 1e3))
    for r in mg:
        if w0 - 300e3 <= int(r["Start_Timestamp"]) <= w1:
            name = "leaf" if "leaf" in r["Kernel_Name"] else ("dma4" if "dma_kernel<4>" in r["Kernel_Name"] else "dma2" if "dma_kernel<2>" in r["Kernel_Name"] else "small" if "small" in r["Kernel_Name"] else "gemm")
            print("%9.1f us  +%8.1f us  %-5s grid %4d x %3d%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, name,
                  int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), "   <- window starts" if abs(int(r["Start_Timestamp"]) - w0) < 30e3 else ""))
