#!/usr/bin/env python3
"""Convex upsample with the warp inside (ofx_upsample_flow_warp) against upsample + warp as two kernels; B frames of 512x768."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h, w = 96, 64
g = torch.Generator(device="cuda").manual_seed(0)
ys, xs = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], -1)[None].repeat(B, 1, 1, 1) + (torch.rand((B, h, w, 2), device="cuda", generator=g) - 0.5) * 2).contiguous()
mask = torch.randn((B, h, w, 576), device="cuda", generator=g)
frame = torch.randint(0, 256, (8 * h, 8 * w, 3), device="cuda", dtype=torch.uint8, generator=g)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = h * w
H, W = 8 * h, 8 * w
byts = B * (n * 576 * 4.0 + n * 8.0 + H * W * 8.0 + H * W * 3.0) + H * W * 3.0
def two():
    f = ops.upsample_flow(coords, mask)
    return f, ops.warp(frame, f, mode="bilinear")
for name, fn, by in (("upsample + warp inside", lambda: ops.upsample_flow_warp(coords, mask, frame), byts),
                     ("same, flow_up not written", lambda: ops.upsample_flow_warp(coords, mask, frame, want_flow=False), byts - B * H * W * 8.0),
                     ("upsample, then warp", two, byts + B * H * W * 8.0)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:28s}: {ms * 1e3:7.1f} us  {by / ms / 1e9:.2f} TB/s of its algorithmic bytes ({by / ms / 1e9 / 8 * 100:.1f} % of 8 TB/s)")
