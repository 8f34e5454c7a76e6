import os, sys
sys.path.insert(0, "/root/repo")
import torch
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
eng = RaftEngine(random_state_dict(0), "cuda")
g = torch.Generator(device="cuda").manual_seed(3)
outs = []
for kind in ("flat", "bright", "halfflat"):
    if kind == "flat":
        a = torch.full((2, 256, 384, 3), 200, dtype=torch.uint8, device="cuda"); b = torch.full((256, 384, 3), 200, dtype=torch.uint8, device="cuda")
    elif kind == "bright":
        a = (250 + torch.randint(0, 6, (2, 256, 384, 3), device="cuda", generator=g)).to(torch.uint8); b = (250 + torch.randint(0, 6, (256, 384, 3), device="cuda", generator=g)).to(torch.uint8)
    else:
        a = torch.randint(0, 256, (2, 256, 384, 3), dtype=torch.uint8, device="cuda", generator=g); a[:, :, :192] = 17
        b = torch.randint(0, 256, (256, 384, 3), dtype=torch.uint8, device="cuda", generator=g); b[:, :192] = 17
    outs.append(eng.forward(a, b, iters=12).cpu())
torch.save(outs, sys.argv[1])
