#!/usr/bin/env python3
"""Timing of the correlation lookup (4 levels, radius 4) on the bench shape: B pairs of 96x64 feature maps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h, w, D = 96, 64, 256
g = torch.Generator(device="cuda").manual_seed(0)
f1 = torch.randn((B, h, w, D), device="cuda", generator=g)
f2 = torch.randn((B, h, w, D), device="cuda", generator=g)
pyr = ops.corr_volume(f1, f2)
ys, xs = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], -1)[None].repeat(B, 1, 1, 1) + (torch.rand((B, h, w, 2), device="cuda", generator=g) - 0.5) * 12).contiguous()
fn = lambda: ops.corr_lookup(pyr, coords, B, h, w)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
byts = B * h * w * (400 * 4 + 324 * 4 + 8)
print(f"lookup B={B}: {ms * 1e3:7.1f} us  {byts / ms / 1e9:.2f} TB/s algorithmic ({byts / ms / 1e9 / 80:.1f} % of 8 TB/s)")

# lookup, then convc1 (the motion encoder's 1x1 over the 324 correlation channels, update.py:86)
wt = (torch.rand((256, 324), device="cuda", generator=g) * 2 - 1) / 18
bias = torch.zeros((256,), device="cuda")
wc = ops.pack_conv_weight(wt.reshape(256, 324, 1, 1).cpu()).cuda()
two = lambda: ops.conv2d_nhwc(ops.corr_lookup(pyr, coords, B, h, w), wc, 1, 1, 256, shift=bias, act="relu")
for _ in range(3): two()
torch.cuda.synchronize()
e0.record()
for _ in range(20): two()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"lookup, then convc1: {ms * 1e3:7.1f} us  {2.0 * B * h * w * 324 * 256 / ms / 1e9:.1f} TFLOP/s (324-k GEMM)")
