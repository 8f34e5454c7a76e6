#!/usr/bin/env python3
"""Per-kernel sums of the counters of one rocprofv3 --pmc pass (rocpd .db), with every counter also as a ratio to the LAST one named:
    python tools/pmc_ratio.py <results.db> SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
names = sys.argv[2:]
rows = db.execute("select kernel_name, grid_size_x, workgroup_size_x, counter_name, value, dispatch_id, end - start from counters_collection").fetchall()
agg = {}
for name, gx, wx, cn, v, did, dur in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
    a = agg.setdefault(n[:96], {"d": {}, "c": {}})
    a["c"][cn] = a["c"].get(cn, 0.0) + v
    a["d"][did] = dur
print(f"{'kernel':<98} {'calls':>5} {'ms':>9} " + " ".join(f"{n[-18:]:>18}" for n in names) + "   ratios to " + names[-1])
for n, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]["d"].values()))[:30]:
    c = a["c"]
    den = c.get(names[-1], 0.0)
    print(f"{n:<98} {len(a['d']):>5} {sum(a['d'].values()) / 1e6:>9.3f} " + " ".join(f"{c.get(k, 0.0):>18.4g}" for k in names) + "   " +
          " ".join(f"{c.get(k, 0.0) / den:.4f}" if den else "-" for k in names[:-1]))
