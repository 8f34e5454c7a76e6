#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) run into small text tables for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db  > profiles/r01_kernel_stats.txt
    python tools/rocpd_summary.py --pmc gpurun_out/pmc_fetch/pmc_results.db

Kernel-trace mode prints the `--stats`-style table (calls, total, average, share) per kernel plus a
per-(kernel, grid) breakdown so every convolution shape is visible.  PMC mode prints the per-kernel counter
sums and per-dispatch averages.
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration, vgpr_count, accum_vgpr_count, lds_size from kernels").fetchall()
    agg, shape = {}, {}
    for name, gx, gy, gz, wx, dur, vg, ag, lds in rows:
        n = short(name)
        a = agg.setdefault(n, [0, 0, vg, ag, lds])
        a[0] += 1
        a[1] += dur
        s = shape.setdefault((n, gx // max(wx, 1), gy, gz), [0, 0])
        s[0] += 1
        s[1] += dur
    total = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"# total kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches")
    print(f"{'kernel':<112} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'lds':>6}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:<112} {a[0]:>6} {a[1] / 1e6:>10.3f} {a[1] / a[0] / 1e3:>10.2f} {100 * a[1] / total:>6.2f} {a[2]:>5} {a[3]:>5} {a[4]:>6}")
    print("\n# per (kernel, workgroups x, grid y, grid z)")
    print(f"{'kernel':<112} {'wgs':>8} {'gy':>4} {'calls':>6} {'total_ms':>10} {'avg_us':>10}")
    for (n, gx, gy, gz), s in sorted(shape.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{n:<112} {gx:>8} {gy:>4} {s[0]:>6} {s[1] / 1e6:>10.3f} {s[1] / s[0] / 1e3:>10.2f}")


def gap_stats(db):
    """Idle time between consecutive dispatches of the same queue (= HIP stream): what the kernel boundaries of a latency-bound
    workload cost, as opposed to the kernels themselves.  Printed under the --stats tables when the trace has start / end columns."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    if not {"start", "end"} <= set(cols):
        print("\n# (no start / end columns in this trace: columns are", cols, ")")
        return
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute(f"select {qcol or '0'}, start, end, name from kernels order by start").fetchall()
    byq = {}
    for q, st, en, name in rows:
        byq.setdefault(q, []).append((st, en, short(name)))
    print("\n# gaps between consecutive dispatches of one queue (ns -> us); gaps > 200 us (host-side pauses between calls) excluded")
    print(f"{'queue':>8} {'dispatches':>10} {'busy_ms':>10} {'gaps':>8} {'gap_sum_ms':>11} {'median_us':>10} {'p90_us':>8} {'span_ms':>9}")
    for q, ev in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        gaps = sorted((ev[i + 1][0] - ev[i][1]) / 1e3 for i in range(len(ev) - 1) if 0 <= ev[i + 1][0] - ev[i][1] < 200e3)
        busy = sum(e - s_ for s_, e, _ in ev) / 1e6
        if not gaps:
            continue
        print(f"{str(q):>8} {len(ev):>10} {busy:>10.3f} {len(gaps):>8} {sum(gaps) / 1e3:>11.3f} {gaps[len(gaps) // 2]:>10.2f} "
              f"{gaps[int(len(gaps) * 0.9)]:>8.2f} {(ev[-1][1] - ev[0][0]) / 1e6:>9.3f}")


def pmc_stats(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    print(f"# rocprofv3 --pmc summary of {db}")
    print("# columns:", cols)
    q = "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by sum(value) desc"
    print(f"{'kernel':<112} {'counter':<14} {'dispatches':>10} {'sum':>18} {'avg/dispatch':>16}")
    for name, ctr, n, v in c.execute(q):
        print(f"{short(name):<112} {ctr:<14} {n:>10} {v:>18.1f} {v / n:>16.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        pmc_stats(sys.argv[2])
    else:
        kernel_stats(sys.argv[1])
        gap_stats(sys.argv[1])
