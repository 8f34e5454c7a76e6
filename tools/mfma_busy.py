#!/usr/bin/env python3
"""Matrix-pipe occupancy per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) joined
with the kernel durations of the same pass: busy = MFMA busy cycles per SIMD / kernel cycles, clock = kernel cycles / duration.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d DIR -o pmc -- python bench.py ...
    python tools/mfma_busy.py DIR/.../pmc_results.db

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the part; SQ_BUSY_CYCLES over its 32 shader engines (checked on the
A-stationary volume kernel, whose MFMA count is known: 32 cycles x MFMAs == the counter)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size_y, counter_name, value, dispatch_id, end - start from counters_collection").fetchall()
agg = {}
for name, gy, cn, v, did, dur in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
    if "igemm_kernel" in n:
        n = ("volume GEMM " if (gy or 1) > 1 else "conv ") + n
    a = agg.setdefault(n[:100], {"d": {}, "c": {}})
    a["c"][cn] = a["c"].get(cn, 0.0) + v
    a["d"][did] = dur
tot = {}
print(f"{'kernel':<102} {'calls':>5} {'ms':>9} {'mfma_busy':>9} {'GHz':>6}")
out = []
for n, a in agg.items():
    c = a["c"]
    ms = sum(a["d"].values()) / 1e6
    cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0
    if cyc <= 0:
        continue
    out.append((ms, n, len(a["d"]), busy / cyc, cyc / (ms * 1e6)))
    fam = "conv (all igemm instantiations)" if n.startswith("conv ") else None
    if fam:
        t = tot.setdefault(fam, [0.0, 0.0, 0.0, 0])
        t[0] += ms; t[1] += busy; t[2] += cyc; t[3] += len(a["d"])
for ms, n, calls, b, ghz in sorted(out, reverse=True)[:40]:
    print(f"{n:<102} {calls:>5} {ms:>9.3f} {b:>9.3f} {ghz:>6.2f}")
for fam, t in tot.items():
    print(f"{fam:<102} {t[3]:>5} {t[0]:>9.3f} {t[1] / t[2]:>9.3f} {t[2] / (t[0] * 1e6):>6.2f}")
