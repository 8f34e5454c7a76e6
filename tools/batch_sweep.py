#!/usr/bin/env python3
"""Wall time of the bench step (flow + warp + mask, `bench.make_step`) for B frames against one key frame at 512x768, B = 1 ... 32:
BASELINE configs[1] (B = 1), the batches `PDCNetAux` drives (16, ofgen_keyframe_inpaint.py:1128) and the road to configs[2] (64).
    python tools/batch_sweep.py [B ...]      -> one JSON line;  `OFX_LIB_PATH` selects a variant build."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict


def sweep(batches=(1, 2, 4, 8, 16, 32), reps=None):
    dev = torch.device('cuda')
    eng = RaftEngine(random_state_dict(0), dev)
    frames, key, key_ai, conf = bench.make_clip(max(batches), bench.H, bench.W, dev)
    out = []
    for B in batches:
        step = bench.make_step(eng, frames[:B].contiguous(), key, key_ai, conf[:B].contiguous())
        for _ in range(2):
            step()
        n = reps or max(3, min(20, 64 // B))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / n * 1e3
        out.append({"B": B, "ms": round(ms, 3), "pairs_per_s": round(B / ms * 1e3, 2)})
    return out


if __name__ == "__main__":
    bs = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 4, 8, 16, 32)
    print(json.dumps({"workload": "flow (RAFT 20 iters fp32) + warp + mask, B frames vs one key frame, 512x768", "batch_sweep": sweep(bs)}))
