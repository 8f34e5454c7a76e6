#!/usr/bin/env python3
"""CPU enqueue time vs GPU completion time of one single-pair forward (is the B=1 path launch-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
eng = RaftEngine(random_state_dict(0))
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 256, (1, 768, 512, 3), dtype=torch.uint8, device="cuda", generator=g)
k = torch.randint(0, 256, (768, 512, 3), dtype=torch.uint8, device="cuda", generator=g)
for serial in (False, True):
    for _ in range(3):
        eng.forward(a, k, serial=serial)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        eng.forward(a, k, serial=serial)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f"serial={serial}: enqueue {1e3 * min(enq):.2f} ms, complete {1e3 * min(tot):.2f} ms")
