"""Rate of the key-frame detector on the GPU box: `detect_edges` over batches of 512x768 frames already on the
device + the mean-pixel-distance decision of `frame_generator`, frames per second.
    python tools/keyframe_detect_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sd_animation_optical_flow_amd import keyframes

H, W, N = 512, 768, 256
rng = np.random.default_rng(0)
# smooth moving content (a random field has an edge at every pixel: the worst case of the hysteresis, not a video)
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
clip = np.stack([np.stack([(127 + 120 * np.sin((xx + 3 * t) / (17 + c)) * np.cos((yy - 2 * t) / (23 - c))) for c in range(3)], -1)
                 for t in range(16)]).astype(np.uint8)
noise = rng.integers(0, 256, (16, H, W, 3), dtype=np.uint8)
for name, data in (("smooth content", clip), ("random pixels", noise)):
    dev = torch.from_numpy(data).cuda()
    ks = keyframes.estimated_kernel_size(W, H)
    keyframes.detect_edges(dev, ks)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N // 16):
        e = keyframes.detect_edges(dev, ks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"detect_edges, {name:14s}: {N} frames of {W}x{H} in batches of 16: {dt * 1e3:7.1f} ms = {N / dt:7.0f} frames/s", flush=True)
fr = [data for data in clip] * 8
t = time.perf_counter()
keys = sum(1 for _, is_key, _ in keyframes.frame_generator(iter(fr), fps=30.0, batch=16) if is_key)
dt = time.perf_counter() - t
print(f"frame_generator over {len(fr)} host frames (upload + edges + decisions): {dt * 1e3:.1f} ms = {len(fr) / dt:.0f} frames/s, {keys} key frames")
