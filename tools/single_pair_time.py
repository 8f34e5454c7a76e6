#!/usr/bin/env python3
"""Wall time of the bench step (flow + warp + mask, `bench.make_step`) for 1, 2 and 4 frames against one key frame at 512x768 --
BASELINE configs[1] and its neighbours.  `OFX_LIB_PATH` selects a variant build."""
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
dev = torch.device('cuda')
eng = RaftEngine(random_state_dict(0), dev)
frames, key, key_ai, conf = bench.make_clip(4, bench.H, bench.W, dev)
for B in (1, 2, 4):
    step = bench.make_step(eng, frames[:B].contiguous(), key, key_ai, conf[:B].contiguous())
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print(f"B={B}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms", flush=True)
