"""The correlation volume in split-bf16 form (corr_split.hip) against the exact-fp32 GEMM, B = 64 at 512x768, one shared key frame:
accuracy of every pyramid level against a float64 product of sampled rows, and the kernel times (HIP events of ofx_prof).
    python tools/vol_split_bench.py [B] [shared]
    OFX_VOLSPLIT_VARIANT=nodb|xofs python tools/vol_split_bench.py     # diagnostic kernel variants"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_animation_optical_flow_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shared = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
h, w, D = 96, 64, 256
g = torch.Generator(device="cuda").manual_seed(0)
# feature maps of the real encoder are O(1) with a long tail; scale some channels so that the low planes matter
f1 = torch.randn((B, h, w, D), device="cuda", generator=g) * torch.exp(torch.randn((1, 1, 1, D), device="cuda", generator=g))
f2 = torch.randn((1 if shared else B, h, w, D), device="cuda", generator=g) * torch.exp(torch.randn((1, 1, 1, D), device="cuda", generator=g))
f2b = f2.expand(B, h, w, D).contiguous() if shared else f2


def f64_rows(b, rows):
    a = f1[b].reshape(-1, D)[rows].double()
    k = f2[0 if shared else b].reshape(-1, D).double()
    v = (a @ k.T / 16.0).reshape(len(rows), 1, h, w)
    out = [v]
    for _ in range(3):
        v = torch.nn.functional.avg_pool2d(v, 2, 2)
        out.append(v)
    return [o[:, 0] for o in out]


def check(pyr, tag):
    res = {}
    rows = torch.tensor([0, 1, 63, 64, 255, 256, 1000, 4097, h * w - 1], device="cuda")
    for b in sorted({0, B // 2, B - 1}):
        ref = f64_rows(b, rows)
        for l in range(4):
            got = ops.corr_unblock(pyr[l].reshape(B, h * w, -1)[b][rows], h >> l, w >> l).double()
            err = (got - ref[l]).abs().max().item()
            scale = ref[l].abs().max().item()
            res[f"l{l}"] = max(res.get(f"l{l}", 0.0), err / scale)
    print(tag, "max |err| / max |ref| per level:", {k: f"{v:.2e}" for k, v in res.items()})
    return res


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ops.prof_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    k = ops.prof_collect()
    ops.prof_enable(0)
    return {name: round(v["ms"] / n, 4) for name, v in k.items()}


out = {"B": B, "shared": shared, "variant": os.environ.get("OFX_VOLSPLIT_VARIANT", "default")}
p32 = ops.corr_volume(f1, f2b)
out["fp32_err"] = check(p32, "fp32 ")
out["fp32_ms"] = timed(lambda: ops.corr_volume(f1, f2b))
for prec in ("bf16x6", "bf16x3"):
    p = ops.corr_volume_split(f1, f2, 4, prec)
    out[prec + "_err"] = check(p, prec)
    # the split GEMM against the fp32 one, every element of two pairs
    for l in range(2):
        d = (p[l].reshape(B, h * w, -1)[[0, B - 1]] - p32[l].reshape(B, h * w, -1)[[0, B - 1]]).abs().max().item()
        out[f"{prec}_vs_fp32_l{l}"] = d
    del p
    out[prec + "_ms"] = timed(lambda: ops.corr_volume_split(f1, f2, 4, prec))
print(json.dumps(out))
