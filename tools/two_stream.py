#!/usr/bin/env python3
"""Experiment: one 64-pair batch on one stream vs two 32-pair batches on two streams (do the HBM-bound kernels of one half
hide behind the MFMA-bound convolutions of the other?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sd_animation_optical_flow_amd import ops
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
dev = torch.device("cuda")
frames, key, key_ai, conf = bench.make_clip(64, bench.H, bench.W, dev)
sd = random_state_dict(0)
e0, e1, e2 = RaftEngine(sd, dev), RaftEngine(sd, dev), RaftEngine(sd, dev)
def one():
    fl = e0.forward(frames, key, iters=20)
    return ops.warp_and_mask(key_ai, fl, conf, warp_mode="bilinear", thres=0.95, ksize=7)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
fa, fb = frames[:32].contiguous(), frames[32:].contiguous()
ca, cb = conf[:32].contiguous(), conf[32:].contiguous()
def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        f1 = e1.forward(fa, key, iters=20)
        ops.warp_and_mask(key_ai, f1, ca, warp_mode="bilinear", thres=0.95, ksize=7)
    with torch.cuda.stream(s2):
        f2 = e2.forward(fb, key, iters=20)
        ops.warp_and_mask(key_ai, f2, cb, warp_mode="bilinear", thres=0.95, ksize=7)
    cur.wait_stream(s1); cur.wait_stream(s2)
for name, fn in (("one stream, B=64", one), ("two streams, 2 x B=32", two), ("one stream, B=64", one)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt * 1e3:.1f} ms per 64 pairs = {64 / dt:.1f} pairs/s")
