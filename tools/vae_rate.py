#!/usr/bin/env python3
"""Timing of the SD-inpaint hand-off at 1024x1024 (BASELINE config #5's frame size): Pillow-exact inputs + first-stage latent."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import handoff, ops
from sd_animation_optical_flow_amd.vae import VaeEncoder, random_vae_state_dict
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda").manual_seed(0)
frame = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
ref = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
mask = (torch.rand((B, H, W), device="cuda", generator=g) > 0.8).to(torch.uint8) * 255
vae = VaeEncoder(random_vae_state_dict(0))
def step():
    t = handoff.prepare_inpaint_inputs(frame, ref, mask, mask_blur=4)
    return vae.get_first_stage_encoding(t["image"])
for _ in range(2): step()
torch.cuda.synchronize()
ops.prof_enable(1)
t0 = time.perf_counter()
n = 5
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
k = ops.prof_collect(); ops.prof_enable(0)
print(f"hand-off + VAE encode {W}x{H} B={B}: {dt * 1e3:.2f} ms per call")
tot = sum(v["ms"] for v in k.values())
for name, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:8]:
    extra = f"  {v['flops'] / (v['ms'] * 1e-3) / 1e12:6.1f} TFLOP/s" if v.get("flops", 0) > 0 else ""
    print(f"  {name:<22} {v['calls'] // n:4d} launches  {v['ms'] / n:8.3f} ms  {100 * v['ms'] / tot:5.1f} %{extra}")
