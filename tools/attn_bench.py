"""Time `ops.attention` on the UNet's self-/cross-attention shapes at 1024x1024 (latent 128x128) on the GPU box:
    python tools/attn_bench.py
Prints ms and fp32 TFLOP/s (4 * BH * Nq * Nk * D flops) per shape; the D = 48 rows run the unfused path for comparison."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_animation_optical_flow_amd import ops

SHAPES = [("level0 self", 8, 16384, 16384, 40), ("level0 self x2 (cfg)", 16, 16384, 16384, 40), ("level0 cross", 16, 16384, 77, 40),
          ("level1 self", 16, 4096, 4096, 80), ("level2 self", 16, 1024, 1024, 160), ("level0 kv-history x3", 8, 16384, 49152, 40),
          ("512^2 level0 self", 16, 4096, 4096, 40), ("unfused d=48", 8, 8192, 8192, 48), ("fused d=40 same", 8, 8192, 8192, 40)]
for name, BH, Nq, Nk, D in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn((BH, Nq, D), device="cuda", generator=g)
    k = torch.randn((BH, Nk, D), device="cuda", generator=g)
    v = torch.randn((BH, Nk, D), device="cuda", generator=g)
    ops.attention(q, k, v)
    torch.cuda.synchronize()
    n = 5
    t = time.perf_counter()
    for _ in range(n):
        o = ops.attention(q, k, v)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{name:28s} BH={BH:3d} Nq={Nq:6d} Nk={Nk:6d} D={D:4d}  {dt*1e3:9.3f} ms  {4.0*BH*Nq*Nk*D/dt/1e12:7.1f} TFLOP/s", flush=True)
    del q, k, v, o
