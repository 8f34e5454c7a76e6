#!/usr/bin/env python3
"""Timing of the mask kernel vs structuring-element size (B=64, 512x768).  OFX_MASK_VARIANT picks a workgroup shape of
the band kernel (read once per process): run as  `for v in 0 1 2 3 4 5; do OFX_MASK_VARIANT=$v python tools/mask_bench.py; done`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import ops
B, H, W = 64, 768, 512
conf = torch.rand((B, H, W), device="cuda")
ref = None
for ks in (1, 7, 15, 31):
    for _ in range(3):
        m = ops.generate_mask(conf, None, 0.95, ks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.generate_mask(conf, None, 0.95, ks)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"variant {os.environ.get('OFX_MASK_VARIANT', '0')} ksize {ks:2d}: {ms * 1e3:7.1f} us  ({B * H * W * 5 / ms / 1e9:.2f} TB/s algorithmic, "
          f"{B * H * W * 5 / ms / 1e9 / 80:.1f} % of 8 TB/s)  sum {int(m.sum())}")
