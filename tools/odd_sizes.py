"""Throughput on maps that are not whole 8x16 patches: ms per forward and per 512x768-equivalent pair.  OFX_PATCH_MAX_WASTE selects when a
layer leaves the halo-patch kernel (conv.hip): python tools/odd_sizes.py [B H W ...]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
eng = RaftEngine(random_state_dict(0), "cuda")
g = torch.Generator(device="cuda").manual_seed(0)
args = [int(a) for a in sys.argv[1:]]
cases = list(zip(args[0::3], args[1::3], args[2::3])) or [(1, 768, 512), (1, 776, 520), (1, 544, 960), (16, 776, 520), (16, 800, 560), (16, 768, 512)]
for (B, H, W) in cases:
    a = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    k = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    for _ in range(3): eng.forward(a, k, iters=20)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 10 if B < 16 else 3
    for _ in range(n): eng.forward(a, k, iters=20)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
    hh, ww = H // 8, W // 8
    waste = ((hh + 7) // 8 * 8) * ((ww + 15) // 16 * 16) / (hh * ww)
    print(f"B={B} {H}x{W} (map {hh}x{ww}, 8x16-patch cover {waste:.2f}x): {ms:.2f} ms  ({ms / (B * H * W) * 768 * 512:.2f} ms per 512x768-equivalent pair)", flush=True)
