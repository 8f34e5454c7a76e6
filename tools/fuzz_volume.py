"""Randomised sweep of the A-stationary correlation-volume kernel (csrc/corr_split.hip) over every shape class it takes: maps of
8..72 x 16..112 feature pixels (row groups that are and are not whole, one to many quads), 1..9 pairs, shared and per-pair key frames.
The fp32 form must equal the generic batched GEMM bit for bit on all four pyramid levels; the bf16x6 form must stay within 4e-6 of the
row scale of it, bf16x3 within 6e-5.     python tools/fuzz_volume.py [cases=60] [seed=0]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_animation_optical_flow_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator(device="cuda").manual_seed(1)
bad = 0
for case in range(n_cases):
    h, w = 8 * rng.randint(1, 9), 16 * rng.randint(1, 7)
    B = rng.randint(1, 9)
    shared = rng.random() < 0.5
    f1 = torch.randn((B, h, w, 256), device="cuda", generator=g) * torch.exp(torch.randn((1, 1, 1, 256), device="cuda", generator=g))
    f2 = torch.randn((1 if shared else B, h, w, 256), device="cuda", generator=g)
    ref = ops.corr_volume(f1, f2.expand(B, h, w, 256).contiguous())
    scale = ref[0].abs().max().item()
    msg = []
    for prec, tol in (("fp32", 0.0), ("bf16x6", 4e-6), ("bf16x3", 6e-5)):
        got = ops.corr_volume_split(f1, f2, 4, prec)
        for l in range(4):
            if prec == "fp32":
                ok = torch.equal(got[l], ref[l])
            else:
                ok = (got[l] - ref[l]).abs().max().item() <= tol * scale
            if not ok:
                msg.append(f"{prec} level {l}: {(got[l] - ref[l]).abs().max().item() / scale:.2e}")
    if msg:
        bad += 1
        print("FAIL", (B, h, w, shared), msg)
print(f"{n_cases - bad} / {n_cases} cases clean")
sys.exit(1 if bad else 0)
