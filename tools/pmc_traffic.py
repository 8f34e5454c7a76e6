#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.

    python tools/pmc_traffic.py <fetch pmc_results.db> <write pmc_results.db> > profiles/rNN_pmc_traffic_b64.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB;
FETCH_SIZE reports half of the bytes read (128-byte requests tallied at 64 B) and is doubled; WRITE_SIZE is taken as
is.  Calibrated per access width on this pool (profiles/r02_pmc_calibration.txt, tools/pmc_calib.hip: 1 GiB of known
traffic per launch): 1-, 4-, 8- and 16-byte-per-lane coalesced reads ALL report exactly 0.5x, every store width reports
1.0x (byte stores 1.002x), and a gather of 40-byte rows reports 2.0x its useful bytes = 4.1x after doubling, i.e. whole
128-byte lines.  In our own access pattern the pooling kernel (reads the whole level-0 volume once, writes levels 1-3
once) comes out at its exact algorithmic byte count.
bytes/launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / dispatches.  Kernels are grouped into the
families bench.py reports.
"""
import json
import sqlite3
import sys

FAMILIES = [
    ("igemm_conv_all", lambda n, gy: "igemm_kernel" in n and gy <= 1),
    ("corr_volume_gemm", lambda n, gy: ("igemm_kernel" in n and gy > 1) or ("corr_vol_split_kernel<2" in n and ", true>" in n)),   # fp32: the generic batched GEMM / the A-stationary kernel's fp32 form
    ("corr_vol_split6", lambda n, gy: "corr_vol_split_kernel<3" in n),
    ("corr_vol_split3", lambda n, gy: "corr_vol_split_kernel<2" in n and ", false>" in n),
    ("corr_split_planes", lambda n, gy: "split_planes_kernel" in n or "frag_order_f32_kernel" in n),
    ("corr_lookup", lambda n, gy: "corr_lookup" in n),
    ("pyramid_pool", lambda n, gy: "pyramid_pool" in n),
    ("upsample_warp", lambda n, gy: "upsample_warp_kernel" in n),
    ("upsample", lambda n, gy: "upsample_kernel" in n),
    ("flow_head", lambda n, gy: "flow_head_kernel" in n),
    ("warp", lambda n, gy: "warp_u8c3_x4_kernel" in n or "warp_kernel" in n or "warp_bilinear" in n),
    ("mask", lambda n, gy: "mask_bits" in n or "mask_rows" in n),
]


def collect(db, counter):
    c = sqlite3.connect(db)
    out = {}
    q = "select kernel_name, grid_size_y, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name, grid_size_y"
    for name, gy, n, v in c.execute(q, (counter,)):
        for fam, match in FAMILIES:
            if match(name, gy or 1):
                a = out.setdefault(fam, [0, 0.0])
                a[0] += n
                a[1] += v
                break
    return out


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    res = {}
    for fam in fetch:
        n, f = fetch[fam]
        nw, w = write.get(fam, (n, 0.0))
        n = max(n, 1)
        fb, wb = f * 1024.0 / n, w * 1024.0 / max(nw, 1)
        res[fam] = {"launches": n, "fetch_bytes_per_launch_raw": fb, "write_bytes_per_launch": wb,
                    "hbm_bytes_per_launch_corrected": 2.0 * fb + wb}
    # what the profile was taken on: the round tag, the source revision (.build_rev, written by tools/profile_round.sh's caller
    # before the snapshot leaves for the GPU box -- .git does not travel) and the sha256 of the libofx.so that ran
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = {"tag": sys.argv[3] if len(sys.argv) > 3 else None, "git_rev": None, "libofx_sha256": None}
    try:
        meta["git_rev"] = open(os.path.join(root, ".build_rev")).read().strip()
    except OSError:
        pass
    try:
        meta["libofx_sha256"] = hashlib.sha256(open(os.path.join(root, "sd_animation_optical_flow_amd", "libofx.so"), "rb").read()).hexdigest()
    except OSError:
        pass
    res["_profile"] = meta
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
