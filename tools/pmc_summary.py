#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per
launch.  FETCH_SIZE is reported in KB and under-counts by 2x on gfx950 (MI355X_MICROARCH.md, HBM
section; confirmed here on vec_dot_kernel), so it is doubled; WRITE_SIZE is in KB as is.
usage: pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.csv [traffic.json workload]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0].split("<")[0]
        acc[name][0] += 1
        acc[name][1] += float(row["Counter_Value"])
    return acc


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    rows = []
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n = fetch[k][0]
        f_kb = fetch[k][1] / n
        w_kb = write.get(k, [1, 0.0])[1] / max(write.get(k, [1, 0.0])[0], 1)
        rows.append((k, n, f_kb, 2 * f_kb * 1024, w_kb, w_kb * 1024, 2 * f_kb * 1024 + w_kb * 1024))
    with open(sys.argv[3], "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_raw_per_launch,fetch_bytes_x2_corrected,WRITE_SIZE_KB_per_launch,write_bytes,hbm_bytes_per_launch\n")
        for r in rows:
            fh.write("%s,%d,%.1f,%.4g,%.1f,%.4g,%.4g\n" % r)
    if len(sys.argv) > 5:
        path, wl = sys.argv[4], sys.argv[5]
        try:
            t = json.load(open(path))
        except Exception:
            t = {}
        d = {r[0]: r[6] for r in rows}
        n = {r[0]: r[1] for r in rows}
        # per EVALUATION (= per launch of the logits kernel): the scatter stage is two launches of plm_scatter_kernel (main +
        # left-over strips) and the column-range slab sum when the alignment has left-over strips
        evals = max(n.get("plm_logits_kernel", 1), 1)
        scatter = sum(d[k] * n[k] for k in ("plm_scatter_kernel", "plm_sum_slabs_cols_kernel") if k in d) / evals
        t[wl] = {"plm_scatter": scatter or None, "plm_logits": d.get("plm_logits_kernel")}
        import os
        t["measured_at_commit"] = os.environ.get("DCA_COMMIT", "unknown")      # bench.py reports it next to roofline.traffic
        # fingerprint of the kernel sources the counters were collected on: bench.py flags the figures stale when the tree differs
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_sources_fingerprint
        t["kernel_sources_sha256"] = kernel_sources_fingerprint()
        json.dump(t, open(path, "w"), indent=1)
    for r in rows[:8]:
        print("%-28s launches %4d  hbm bytes/launch %.4g" % (r[0], r[1], r[6]))


if __name__ == "__main__":
    main()
