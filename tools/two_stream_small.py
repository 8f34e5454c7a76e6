#!/usr/bin/env python3
"""Experiment (round 5): B pairs on one stream vs two half-batches on two streams, for the SMALL batches whose launches leave the chip
with ragged tails (B = 2 ... 32).  Does one half's tail fill with the other half's workgroups?   python tools/two_stream_small.py [B ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
dev = torch.device("cuda")
sd = random_state_dict(0)
e0, e1, e2 = RaftEngine(sd, dev), RaftEngine(sd, dev), RaftEngine(sd, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for B in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32]:
    frames, key, key_ai, conf = bench.make_clip(B, bench.H, bench.W, dev)
    h = B // 2
    fa, fb = frames[:h].contiguous(), frames[h:].contiguous()
    def one():
        return e0.forward(frames, key, iters=20, warp_frame=key_ai)
    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            e1.forward(fa, key, iters=20, warp_frame=key_ai)
        with torch.cuda.stream(s2):
            e2.forward(fb, key, iters=20, warp_frame=key_ai)
        cur.wait_stream(s1); cur.wait_stream(s2)
    res = []
    for fn in (one, two, one, two):
        fn(); fn(); torch.cuda.synchronize()
        n = max(3, 32 // B)
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e3)
    print(f"B={B:<3} one stream {min(res[0], res[2]):8.3f} ms ({B / min(res[0], res[2]) * 1e3:6.1f} pairs/s)   two streams of B/2 {min(res[1], res[3]):8.3f} ms "
          f"({B / min(res[1], res[3]) * 1e3:6.1f} pairs/s)", flush=True)
