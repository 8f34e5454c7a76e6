#!/usr/bin/env python3
"""Where does the headline-size flow error come from?  Per-iteration EPE of (a) the HIP engine and (b) the fp32 CPU oracle
against the SAME network evaluated in float64, on frames of the bench clip (512x768), at the 1/8-resolution flow after
1, 5, 10, 20 iterations and at the final upsampled flow.  Output: gpurun_out/r03_epe_curve.txt (copy to profiles/)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from oracle import raft_oracle as RO                    # noqa: E402
from sd_animation_optical_flow_amd.raft import RaftEngine   # noqa: E402

ITS = (1, 2, 5, 10, 20)
pairs = [int(x) for x in sys.argv[1:]] or [0, 31, 63]
sd = RO.init_state_dict(0)
sd64 = RO.to_float64(sd)
eng = RaftEngine(sd, precision=os.environ.get("OFX_EPE_PRECISION", "fp32"))   # OFX_LIB_PATH=tools/alt/libofx_exact_gates.so: libm gates
frames, key, _, _ = bench.make_clip(64, bench.H, bench.W, torch.device("cuda"))
kf = key.cpu().permute(2, 0, 1)[None].float()
epe = lambda a, b: (a.double() - b.double()).pow(2).sum(-1).sqrt().mean().item()
lines = ["# EPE in px against the float64 evaluation of the same network (oracle.raft_oracle.to_float64); 512x768, seeded weights",
         "# low = 1/8-resolution flow (coords1 - coords0) after n iterations; up = convex-upsampled flow after 20",
         f"{'pair':>4} {'iters':>5} {'HIP_vs_f64':>12} {'cpu32_vs_f64':>12} {'HIP_vs_cpu32':>12} {'|flow|_mean':>11}"]
t0 = time.time()
for b in pairs:
    a = frames[b].cpu().permute(2, 0, 1)[None].float()
    tr32, tr64 = {"keep_iters": ITS}, {"keep_iters": ITS}
    _, up32 = RO.raft_forward(sd, a, kf, iters=20, trace=tr32)
    _, up64 = RO.raft_forward(sd64, a.double(), kf.double(), iters=20, trace=tr64)
    for n in ITS:
        up, lo = eng.forward(frames[b:b + 1], key, iters=n, want_low=True)
        l64 = tr64["flow_low_at"][n][0].permute(1, 2, 0)
        l32 = tr32["flow_low_at"][n][0].permute(1, 2, 0)
        lines.append(f"{b:>4} {n:>5} {epe(lo[0].cpu(), l64):>12.3e} {epe(l32, l64):>12.3e} {epe(lo[0].cpu(), l32):>12.3e} {l64.norm(dim=-1).mean().item():>11.3f}")
    u64, u32 = up64[0].permute(1, 2, 0), up32[0].permute(1, 2, 0)
    lines.append(f"{b:>4} {'up20':>5} {epe(up[0].cpu(), u64):>12.3e} {epe(u32, u64):>12.3e} {epe(up[0].cpu(), u32):>12.3e} {u64.norm(dim=-1).mean().item():>11.3f}")
    # the batch the bench times (64 frames, large-grid tile schedules) for the same frame
    # which stage drifts?  the same pair with the instance-norm statistics taken by their own f64 pass (not out of the conv epilogues)
    up_s, lo_s = eng.forward(frames[b:b + 1], key, iters=20, want_low=True, separate_stats=True)
    l64 = tr64["flow_low_at"][20][0].permute(1, 2, 0)
    lines.append(f"{b:>4} {'sep20':>5} {epe(lo_s[0].cpu(), l64):>12.3e} {'':>12} {'':>12}   # statistics by the separate f64 pass, low-res flow after 20")
    lines.append(f"{b:>4} {'sepup':>5} {epe(up_s[0].cpu(), u64):>12.3e}")
    # stage errors after ONE iteration against f64: feature maps, context, first hidden state
    eng.forward(frames[b:b + 1], key, iters=1)
    h8, w8 = bench.H // 8, bench.W // 8
    fm1 = eng.buffer("fmap1").cpu().reshape(1, h8, w8, 256).permute(0, 3, 1, 2)
    hx = eng.buffer("hx").cpu().reshape(1, h8, w8, 384).permute(0, 3, 1, 2)
    rel = lambda x, r: ((x.double() - r).abs().mean() / r.abs().mean()).item()
    lines.append(f"#   pair {b} stage errors (mean |err| / mean |ref|, against f64): fmap1 HIP {rel(fm1, tr64['fmap1']):.2e} cpu32 {rel(tr32['fmap1'], tr64['fmap1']):.2e}; "
                 f"inp HIP {rel(hx[:, 256:], tr64['inp']):.2e} cpu32 {rel(tr32['inp'], tr64['inp']):.2e}; "
                 f"net(it 1) HIP {rel(hx[:, :128], tr64['net_it0']):.2e} cpu32 {rel(tr32['net_it0'], tr64['net_it0']):.2e}")
    eng.forward(frames[b:b + 1], key, iters=1, separate_stats=True)
    fm1s = eng.buffer("fmap1").cpu().reshape(1, h8, w8, 256).permute(0, 3, 1, 2)
    lines.append(f"#   pair {b} fmap1 with separate statistics: HIP {rel(fm1s, tr64['fmap1']):.2e}")
lines.append(f"# batch of 64 (the timed schedule), same frames:")
fl = eng.forward(frames, key, iters=20)
for b in pairs:
    a = frames[b].cpu().permute(2, 0, 1)[None].float()
    _, up64 = RO.raft_forward(sd64, a.double(), kf.double(), iters=20)
    lines.append(f"{b:>4} {'b64':>5} {epe(fl[b].cpu(), up64[0].permute(1, 2, 0)):>12.3e}")
lines.append(f"# {time.time() - t0:.0f} s")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", os.environ.get("OFX_EPE_OUT", "r03_epe_curve.txt")), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
