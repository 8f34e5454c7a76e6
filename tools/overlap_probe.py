#!/usr/bin/env python3
"""Do two kernels of different streams really run side by side?  Reads a rocprofv3 kernel trace (rocpd SQLite) and prints, for the
first few `corr_lookup` dispatches, every dispatch that overlaps them in time (name, queue, start / end relative to the lookup).
    python tools/overlap_probe.py <results.db>"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute(f"select name, {q}, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
print("columns:", cols)
look = [r for r in rows if "corr_lookup" in r[0]]
for L in look[8:9]:
    print(f"lookup queue {L[1]} wgs {L[4] // max(L[5], 1)} dur {(L[3] - L[2]) / 1e3:.1f} us")
    for r in rows:
        if r is not L and r[2] < L[3] + 600e3 and r[3] > L[2] - 200e3:
            nm = re.sub(r"\(anonymous namespace\)::|void ", "", r[0])[:60]
            print(f"    q{r[1]} {nm:<60} wgs {r[4] // max(r[5], 1):>6} start {(r[2] - L[2]) / 1e3:>8.1f} end {(r[3] - L[2]) / 1e3:>8.1f}")
