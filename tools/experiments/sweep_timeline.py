#!/usr/bin/env python3
"""Chain kernels of the LAST inverse in a rocprofv3 --kernel-trace CSV (block sweep): average duration of each kernel of the chain's
queue while no update launch is running (alone) and while one is (loaded), plus the loaded timeline of one step."""
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
t0 = int(R[0]["Start_Timestamp"])
upd = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in R if "sweep_update_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 30000]
chainq = collections.Counter(r["Queue_Id"] for r in R if "leaf" in r["Kernel_Name"]).most_common(1)[0][0]
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for r in R:
    if r["Queue_Id"] != chainq: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    loaded = any(a <= s <= b for a, b in upd)
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_X"]))
    agg[k][2 if loaded else 0] += 1
    agg[k][3 if loaded else 1] += (e - s) / 1e3
print("%-44s %6s | alone: n  avg us | loaded: n  avg us" % ("chain kernel", "WGs"))
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    print("%-44s %6d | %5d %8.1f | %5d %8.1f" % (k[0], k[1], v[0], v[1] / max(1, v[0]), v[2], v[3] / max(1, v[2])))
if len(upd) > 8:
    a, b = upd[8]
    print("---- chain queue during update launch 8 (%.0f us long)" % ((b - a) / 1e3))
    for r in R:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if a <= s <= b and r["Queue_Id"] == chainq:
            print("%9.1f %8.1f  %5d WGs x %4s  %s" % ((s - a) / 1e3, (e - s) / 1e3, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_X"]), r["Workgroup_Size_X"], short(r["Kernel_Name"])))
