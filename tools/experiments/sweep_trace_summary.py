#!/usr/bin/env python3
"""Per-step summary of a DCA_CHOLINV_TRACE=1 run of the block sweep (stderr of tools/time_inv.py): the LAST inverse in the file."""
import re, sys
rows = [l for l in open(sys.argv[1]) if l.startswith("cholinv trace")]
runs, cur = [], []
for l in rows:
    m = re.match(r"cholinv trace\s+([\d.]+) us\s+(.*) (\d+)$", l.strip())
    t, what, p = float(m.group(1)), m.group(2), int(m.group(3))
    if what == "chain: pivot block begins" and p == 0 and cur:
        runs.append(cur); cur = []
    cur.append((t, what, p))
runs.append(cur)
last = sorted(runs[-1])
import collections
print("total %.0f us" % max(t for t, _, _ in last))
# per step: when each mark happened, relative to the step's 'rest: begins'
ev = {(w, p): t for t, w, p in last}
keys = ["chain: pivot block begins", "chain: P done", "chain: next pivot block formed", "side: step begins", "side: W done", "side: prio begins", "side: prio done", "rest: begins", "rest: done"]
print("step " + " ".join("%10s" % k.split(": ")[1][:10] for k in keys) + "   rest-len  gap-before-rest")
prev_done = None
for p in sorted({p for _, _, p in last}):
    if ("rest: begins", p) not in ev: continue
    row = ["%10.0f" % ev[(k, p)] if (k, p) in ev else "         -" for k in keys]
    rb, rd = ev[("rest: begins", p)], ev[("rest: done", p)]
    print("%3d  " % p + " ".join(row) + "   %7.0f  %7.0f" % (rd - rb, rb - prev_done if prev_done is not None else 0))
    prev_done = rd
