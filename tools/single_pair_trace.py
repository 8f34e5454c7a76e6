"""One 512x768 pair, 20 iterations, repeated: the workload of `single_pair` in bench.py, for a kernel trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/sp -o sp -- python tools/single_pair_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
eng = RaftEngine(random_state_dict(0), "cuda")
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 256, (1, 512, 768, 3), dtype=torch.uint8, device="cuda", generator=g)
b = torch.randint(0, 256, (512, 768, 3), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(3):
    eng.forward(a, b, iters=20)
torch.cuda.synchronize()
n = int(os.environ.get("SP_N", "20"))
t = time.perf_counter()
for _ in range(n):
    eng.forward(a, b, iters=20)
torch.cuda.synchronize()
print(f"single pair: {(time.perf_counter() - t) / n * 1e3:.3f} ms")
