#!/usr/bin/env python3
"""Which kernel boundaries of a trace carry idle time: per (predecessor -> successor) pair on the busiest queue, the number of gaps and
their sum.  python tools/gap_pairs.py <results.db>"""
import sqlite3, sys, re, collections
only_ours = len(sys.argv) > 2 and sys.argv[2] == "--ours"      # boundaries between this library's kernels only
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select queue_id, start, end, name, grid_x, workgroup_x from kernels order by start").fetchall()
byq = collections.defaultdict(list)
for q, s, e, n, gx, wx in rows:
    byq[q].append((s, e, re.sub(r"\(anonymous namespace\)::|void |\(.*\)$", "", n)[:58] + f" [{gx // max(wx, 1)}]"))
q, ev = max(byq.items(), key=lambda kv: len(kv[1]))
agg = collections.defaultdict(lambda: [0, 0.0])
for i in range(len(ev) - 1):
    g = (ev[i + 1][0] - ev[i][1]) / 1e3
    if only_ours and ("at::" in ev[i][2] or "rocclr" in ev[i][2] or "at::" in ev[i + 1][2] or "rocclr" in ev[i + 1][2]):
        continue
    if 0.5 <= g < 200:
        a = agg[(ev[i][2], ev[i + 1][2])]
        a[0] += 1; a[1] += g
print(f"total of the listed class: {sum(v[1] for v in agg.values()) / 1e3:.3f} ms in {sum(v[0] for v in agg.values())} gaps")
print(f"queue {q}: {len(ev)} dispatches; boundaries with 0.5 us <= gap < 200 us, by (predecessor -> successor):")
for (a, b), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {n:5d} x {s / n:6.2f} us = {s / 1e3:7.3f} ms   {a}  ->  {b}")
