#!/usr/bin/env python3
"""Does HIP-graph capture of a single-pair forward buy anything? (probe; prints eager vs replay latency)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
eng = RaftEngine(random_state_dict(0))
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 256, (1, 768, 512, 3), dtype=torch.uint8, device="cuda", generator=g)
k = torch.randint(0, 256, (768, 512, 3), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(3):
    ref = eng.forward(a, k)
torch.cuda.synchronize()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print(f"eager  : {timeit(lambda: eng.forward(a, k)):.2f} ms")
for serial in (True, False):
    try:
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                eng.forward(a, k, serial=serial)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(gr):
            out = eng.forward(a, k, serial=serial)
        torch.cuda.synchronize()
        print(f"graph (serial={serial}): {timeit(gr.replay):.2f} ms; max diff vs eager {(out - ref).abs().max().item():.2e}")
    except Exception as e:
        print(f"graph (serial={serial}) failed: {type(e).__name__}: {str(e)[:200]}")
