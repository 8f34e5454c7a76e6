import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from sd_animation_optical_flow_amd import ops
B, h, w, D = 64, 64, 96, 256
g = torch.Generator(device="cuda").manual_seed(0)
f1 = torch.randn((B, h, w, D), device="cuda", generator=g)
f2 = torch.randn((B, h, w, D), device="cuda", generator=g)
for _ in range(2): pyr = ops.corr_volume(f1, f2)
torch.cuda.synchronize()
ops.prof_enable(1)
for _ in range(5): pyr = ops.corr_volume(f1, f2)
torch.cuda.synchronize()
k = ops.prof_collect(); ops.prof_enable(0)
for name, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:3]:
    print(os.environ.get("OFX_VOL_STAGGER"), name, round(v["ms"] / 5, 3), "ms")
