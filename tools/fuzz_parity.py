#!/usr/bin/env python3
"""One-off parity sweep: random frame sizes / batch sizes / iteration counts, engine vs CPU oracle (flow EPE).
Exercises the launcher's tile, paired-pipeline and split-K choices on shapes the fixed tests do not hit."""
import os, sys, random, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raft_oracle as RO
from sd_animation_optical_flow_amd.raft import RaftEngine

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
sd = RO.init_state_dict(0)
eng = RaftEngine(sd)
eng_b = RaftEngine(sd, cnet_norm="batch")     # RAFT_2 as the reference wrote it (train-mode BatchNorm on each image)
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    H, W = 8 * random.randint(8, 60), 8 * random.randint(8, 60)
    B = random.choice([1, 1, 2, 3, 5, 7])
    iters = random.choice([1, 2, 4, 7])
    shared = random.random() < 0.5
    alt = random.random() < 0.35 and not shared and min(H, W) >= 128      # the pooled fmap2 pyramid needs >= 16x16 features
    g = torch.Generator().manual_seed(case)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 20, W + 20), generator=g), 5, 1, 2)
    key = (base[0, :, 10:10 + H, 10:10 + W] * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous()
    frames = torch.stack([(base[0, :, 10 + (b % 5) - 2:10 + (b % 5) - 2 + H, 8 + b:8 + b + W] * 255).round().to(torch.uint8).permute(1, 2, 0)
                          for b in range(B)]).contiguous()
    img2 = key if shared else key[None].repeat(B, 1, 1, 1) if random.random() < 0.5 else frames.flip(0).contiguous()
    ref2 = (img2[None].repeat(B, 1, 1, 1) if img2.dim() == 3 else img2).permute(0, 3, 1, 2).float()
    norm = random.choice(["eval", "batch"])
    warp = random.random() < 0.5 and H >= 128 and W >= 128               # the warp inside the upsample (u8 frame of the same size)
    _, up_ref = RO.raft_forward(sd, frames.permute(0, 3, 1, 2).float(), ref2, iters=iters, alternate_corr=alt, cnet_norm=norm)
    e = eng_b if norm == "batch" else eng
    if warp:
        ai = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).cuda()
        up, wp = e.forward(frames.cuda(), img2.cuda(), iters=iters, alternate_corr=alt, warp_frame=ai)
        from oracle import warp_oracle as WO
        d = np.abs(wp[0].cpu().numpy().astype(np.int32) - WO.warp_frame(ai.cpu().numpy(), up[0].cpu().numpy(), mode="bilinear").astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, "warp inside the upsample"
    else:
        up = e.forward(frames.cuda(), img2.cuda(), iters=iters, alternate_corr=alt)
    epe = (up.cpu() - up_ref.permute(0, 2, 3, 1)).pow(2).sum(-1).sqrt().mean().item()
    worst = max(worst, epe)
    print(f"case {case:2d}: {W}x{H} B={B} iters={iters} shared={shared} alt={alt} norm={norm} warp={warp}: EPE {epe:.2e}", flush=True)
    assert epe < 1e-3, "parity bar exceeded"
print(f"worst EPE {worst:.2e}")
