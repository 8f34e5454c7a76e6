#!/usr/bin/env python3
"""One-off sweep of the implicit-GEMM convolution over random shapes / tiles / options against torch (fp32)."""
import os, sys, random
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
random.seed(seed)
g = torch.Generator().manual_seed(seed)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
ws = torch.zeros((65536 + 1024 * 4 * 64 * 64 * 4,), dtype=torch.uint8, device="cuda")
TILES = [0, 0, 0, 16128128, 16128064, 32128032, 16064064, 32064064, 2032064064, 16128192, 16128096, 32128128, 32128064]
worst = 0.0
for case in range(n):
    B = random.choice([1, 1, 2, 3])
    H, W = random.randint(3, 70), random.randint(3, 70)
    if os.environ.get("FUZZ_MID"):      # grids between the small-grid tile and a full chip: the round-5 tile rule's branches
        B = random.choice([2, 4, 6, 9])
        H, W = random.randint(40, 130), random.randint(40, 130)
    ci = 4 * random.randint(1, 80)
    co = random.choice([2, 30, 64, 96, 126, 128, 192, 200, 256, 324])
    kh, kw = random.choice([(1, 1), (3, 3), (1, 5), (5, 1), (7, 7), (3, 3)])
    if kh == 7: ci = min(ci, 16)
    stride = random.choice([1, 1, 1, 2])
    act = random.choice([None, "relu", "tanh", "sigmoid"])
    tile = random.choice(TILES)
    use_res = random.random() < 0.3 and act in (None, "relu")
    use_sk = random.random() < 0.5 and tile == 0
    x = torch.randn((B, ci, H, W), generator=g)
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    b = torch.randn((co,), generator=g)
    sc = torch.rand((co,), generator=g) + 0.5 if random.random() < 0.3 else None
    y = F.conv2d(x, w, None, stride=stride, padding=(kh // 2, kw // 2))
    y = y * (sc.view(1, -1, 1, 1) if sc is not None else 1.0) + b.view(1, -1, 1, 1)
    y = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, None: lambda t: t}[act](y)
    res = torch.randn(y.shape, generator=g) if use_res else None
    if res is not None:
        y = torch.relu(y + res)
    out = ops.conv2d_nhwc(nhwc(x), ops.pack_conv_weight(w).cuda(), kh, kw, co, stride=stride, shift=b.cuda(),
                          scale=None if sc is None else sc.cuda(), act=act, res=None if res is None else nhwc(res), tile=tile,
                          splitk_ws=ws if use_sk else None)
    err = (out.permute(0, 3, 1, 2).cpu() - y).abs().max().item()
    worst = max(worst, err)
    print(f"case {case}: B{B} {H}x{W} {ci}->{co} {kh}x{kw} s{stride} {act} tile {tile} res {use_res} sk {use_sk}: {err:.1e}", flush=True)
    assert err < 5e-5, "mismatch"
assert int(ws[:65536].view(torch.int32).abs().max()) == 0
print(f"worst {worst:.2e}")
