#!/usr/bin/env python3
"""Rate of the full device-resident tail WITH the forward-backward confidence (two flow directions per frame)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import clip, pdcnet_of
algo = pdcnet_of.create_of_algo("random:0")
synth = clip.FrameSynthesizer(algo, warp_mode="bilinear", thres=0.95, ksize=7)
g = torch.Generator(device="cuda").manual_seed(0)
for B in (1, 16, 64):
    frames = torch.randint(0, 256, (B, 768, 512, 3), dtype=torch.uint8, device="cuda", generator=g)
    key = torch.randint(0, 256, (768, 512, 3), dtype=torch.uint8, device="cuda", generator=g)
    for _ in range(2):
        synth(frames, key, 255 - key)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        synth(frames, key, 255 - key)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {dt * 1e3:.1f} ms per batch, {B / dt:.1f} frames/s (flow both ways + confidence + warp + mask)")
