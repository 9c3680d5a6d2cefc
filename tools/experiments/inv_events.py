#!/usr/bin/env python3
"""Table of the LAST inverse of a DCA_CHOLINV_TRACE=1 run (stderr of tools/time_inv.py): per panel the chain's and the bulk
stream's marks in microseconds from the start of the factorisation.   python inv_events.py trace.txt"""
import re
import sys

lines = [ln for ln in open(sys.argv[1]) if "cholinv trace" in ln]
starts = [k for k, ln in enumerate(lines) if ln.strip().endswith("chain: block begins 0")]
ev = []
for ln in lines[starts[-1]:]:
    m = re.match(r"cholinv trace\s+([\d.]+) us\s+(.*) (\d+)$", ln.strip())
    ev.append((float(m.group(1)), m.group(2), int(m.group(3))))


def get(what, j):
    for t, w, jj in ev:
        if w == what and jj == j:
            return t
    return None


f = lambda v: "%8.0f" % v if v is not None else "       -"      # noqa: E731
print(" j  blockBeg  factored panelDone |  bulkBeg  rowsDone  deepDone | chain: factor  panel | bulk: rows   deep")
nb = max(j for _, w, j in ev if w == "chain: block begins") + 1
tot = [0.0, 0.0, 0.0, 0.0]
for j in range(nb):
    bb, bf, pd = get("chain: block begins", j), get("chain: block factored", j), get("chain: panel of the factor done", j)
    sb, rd, dd = get("bulk: begins step", j), get("bulk: rows of the next panel done", j), get("bulk: deep update of panel j + 2 done", j)
    d = lambda a, b: (b - a) if a is not None and b is not None else None      # noqa: E731
    parts = [d(bb, bf), d(bf, pd), d(sb, rd), d(rd, dd)]
    for k, v in enumerate(parts):
        tot[k] += v or 0.0
    print("%2d %s %s %s | %s %s %s | %s %s | %s %s" % (j, f(bb), f(bf), f(pd), f(sb), f(rd), f(dd), f(parts[0]), f(parts[1]), f(parts[2]), f(parts[3])))
print("sums (ms): chain factor %.2f, chain panel %.2f, bulk rows %.2f, bulk deep %.2f" % tuple(t / 1e3 for t in tot))
for t, w, j in ev[-2:]:
    print("%9.1f us  %s" % (t, w))
