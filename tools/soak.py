"""Repeatability soak: the same inputs through the executor many times, single pair (small-grid kernels: split-K over patch
slabs, paired pipelines, side streams) and the 64-frame batch; every run must reproduce the first bit for bit.
    python tools/soak.py [single_runs] [batch_runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n64 = int(sys.argv[2]) if len(sys.argv) > 2 else 12
eng = RaftEngine(random_state_dict(0), "cuda")
frames, key, _, _ = bench.make_clip(64, bench.H, bench.W, torch.device("cuda"))
ref1 = eng.forward(frames[:1], key, iters=20).clone()
t = time.perf_counter()
for i in range(n1):
    out = eng.forward(frames[:1], key, iters=20)
    assert torch.equal(out, ref1), f"single pair run {i} differs"
torch.cuda.synchronize()
print(f"single pair: {n1} runs identical, {(time.perf_counter() - t) / n1 * 1e3:.2f} ms each")
ref = eng.forward(frames, key, iters=20).clone()
t = time.perf_counter()
for i in range(n64):
    out = eng.forward(frames, key, iters=20)
    assert torch.equal(out, ref), f"batch run {i} differs"
torch.cuda.synchronize()
print(f"64-frame batch: {n64} runs identical, {(time.perf_counter() - t) / n64 * 1e3:.1f} ms each")
assert torch.isfinite(ref).all()
