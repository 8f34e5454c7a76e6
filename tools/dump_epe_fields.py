import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import bench
from oracle import raft_oracle as RO
from sd_animation_optical_flow_amd.raft import RaftEngine
sd = RO.init_state_dict(0); sd64 = RO.to_float64(sd)
eng = RaftEngine(sd)
frames, key, _, _ = bench.make_clip(64, bench.H, bench.W, torch.device("cuda"))
kf = key.cpu().permute(2, 0, 1)[None].float()
a = frames[0].cpu().permute(2, 0, 1)[None].float()
ITS = (1, 2, 3, 5, 10, 20)
tr32, tr64 = {"keep_iters": ITS}, {"keep_iters": ITS}
RO.raft_forward(sd, a, kf, iters=20, trace=tr32)
RO.raft_forward(sd64, a.double(), kf.double(), iters=20, trace=tr64)
out = {}
for n in ITS:
    up, lo = eng.forward(frames[0:1], key, iters=n, want_low=True)
    out[f"hip{n}"] = lo[0].cpu().numpy()
    out[f"c32_{n}"] = tr32["flow_low_at"][n][0].permute(1, 2, 0).numpy()
    out[f"c64_{n}"] = tr64["flow_low_at"][n][0].permute(1, 2, 0).numpy()
np.savez(os.path.join(ROOT, "gpurun_out", "r03_epe_fields.npz"), **out)
print("ok")
