#!/usr/bin/env python3
"""Timing of the bilinear warp of one shared key frame along B = 64 flows at 512x768 (the bench's warp)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import ops
B, H, W = 64, 768, 512
g = torch.Generator(device="cuda").manual_seed(0)
frame = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
ys, xs = torch.meshgrid(torch.arange(H, device="cuda", dtype=torch.float32), torch.arange(W, device="cuda", dtype=torch.float32), indexing="ij")
flow = torch.stack([torch.stack([8 * torch.sin(2 * math.pi * ys / H + 0.37 * t) * torch.cos(2 * math.pi * xs / W),
                                 6 * torch.cos(2 * math.pi * xs / W + 0.2 * t)], -1) for t in range(B)]).contiguous()
for mode in ("bilinear", "cv2_cubic"):
    for _ in range(3):
        o = ops.warp(frame, flow, mode=mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.warp(frame, flow, mode=mode)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    byts = B * H * W * 11 + H * W * 3
    print(f"{mode}: {ms * 1e3:7.1f} us  ({byts / ms / 1e9:.2f} TB/s algorithmic, {byts / ms / 1e9 / 80:.1f} % of 8 TB/s)")
