"""The split volume GEMM INSIDE the step (real feature maps of the bench clip, not random planes): kernel times by HIP events.
    OFX_VOLSPLIT_VARIANT=r3|db|nodb python tools/vol_split_step.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sd_animation_optical_flow_amd import ops
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
dev = torch.device("cuda")
frames, key, key_ai, conf = bench.make_clip(64, bench.H, bench.W, dev)
res = {}
for mode in ("bf16x6", "bf16x3"):
    eng = RaftEngine(random_state_dict(0), dev, volume_precision=mode)
    step = bench.make_step(eng, frames, key, key_ai, conf)
    step(); step()
    torch.cuda.synchronize()
    ops.prof_enable(1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    k = ops.prof_collect()
    ops.prof_enable(0)
    res[mode] = {n: round(v["ms"] / 3, 3) for n, v in k.items() if n.startswith("corr_vol_split") or n in ("corr_split_planes", "corr_pyramid_pool")}
    del eng, step
print(os.environ.get("OFX_VOLSPLIT_VARIANT", "default"), res)
