"""Timing only of the split volume GEMM (diagnostic switches: OFX_VOLSPLIT_DBG, OFX_VOLSPLIT_VARIANT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_animation_optical_flow_amd import ops
B, h, w, D = 64, 96, 64, 256
g = torch.Generator(device="cuda").manual_seed(0)
f1 = torch.randn((B, h, w, D), device="cuda", generator=g)
f2 = torch.randn((1, h, w, D), device="cuda", generator=g)
res = {}
for prec in ("bf16x6", "bf16x3", "fp32"):
    for _ in range(2):
        p = ops.corr_volume_split(f1, f2, 4, prec)
    torch.cuda.synchronize()
    ops.prof_enable(1)
    for _ in range(5):
        p = ops.corr_volume_split(f1, f2, 4, prec)
    torch.cuda.synchronize()
    k = ops.prof_collect()
    ops.prof_enable(0)
    res[prec] = round([v["ms"] / 5 for n, v in k.items() if n.startswith("corr_vol_")][0], 3)
    del p
print(os.environ.get("OFX_VOLSPLIT_DBG", "0"), os.environ.get("OFX_VOLSPLIT_VARIANT", "db"), res)

# the exact-fp32 form against the generic fp32 GEMM: bit for bit
f2b = f2.expand(B, h, w, D).contiguous()
a = ops.corr_volume(f1, f2b)
b = ops.corr_volume_split(f1, f2, 4, "fp32")
print("fp32 A-stationary == generic fp32 GEMM, bit for bit:", [bool(torch.equal(x, y)) for x, y in zip(a, b)])
ops.prof_enable(1)
for _ in range(3):
    ops.corr_volume(f1, f2b)
torch.cuda.synchronize()
k = ops.prof_collect()
ops.prof_enable(0)
print("generic:", {n: round(v["ms"] / 3, 3) for n, v in k.items()})
