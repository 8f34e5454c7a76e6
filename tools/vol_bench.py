"""The correlation-volume GEMM (+ fused pyramid level 1) and the pooling of levels 2, 3 alone, B = 64 at 512x768:
    python tools/vol_bench.py [variant libofx.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_animation_optical_flow_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from sd_animation_optical_flow_amd import ops
B, h, w, D = 64, 96, 64, 256        # the bench geometry: 512x768 frames = 96 rows x 64 columns of features
g = torch.Generator(device="cuda").manual_seed(0)
f1 = torch.randn((B, h, w, D), device="cuda", generator=g)
f2 = torch.randn((B, h, w, D), device="cuda", generator=g)
for _ in range(2):
    pyr = ops.corr_volume(f1, f2)
torch.cuda.synchronize()
ops.prof_enable(1)
for _ in range(5):
    pyr = ops.corr_volume(f1, f2)
torch.cuda.synchronize()
k = ops.prof_collect()
ops.prof_enable(0)
for name, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:2]:
    print(os.path.basename(_lib.LIB_PATH), name, round(v["ms"] / 5, 3), "ms")
