#!/usr/bin/env python3
"""PCIe-inclusive latency of the drop-in host API: algo.calc(frame1_bgr, frame2_bgr) -> numpy (flow, conf, log_conf)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import pdcnet_of
algo = pdcnet_of.create_of_algo("random:0")
rng = np.random.default_rng(0)
for (H, W) in ((384, 256), (768, 512)):
    a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for _ in range(3):
        algo.calc(a, b)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        flow, conf, logc = algo.calc(a, b)
    dt = (time.perf_counter() - t0) / n
    print(f"calc {W}x{H}: {dt * 1e3:.2f} ms per pair (flow + forward-backward confidence, numpy in / numpy out)")
