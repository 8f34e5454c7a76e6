#!/usr/bin/env python3
"""Per-layer time / executed TFLOP/s of one RAFT forward (library profiler in per-layer mode).
usage: python tools/layer_profile.py [--batch 64] [--H 768 --W 512] [--iters 20] [--precision fp32|bf16x3] [--out file.json]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import ops
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--H", type=int, default=768)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    eng = RaftEngine(random_state_dict(0), precision=a.precision)
    g = torch.Generator(device="cuda").manual_seed(0)
    frames = torch.randint(0, 256, (a.batch, a.H, a.W, 3), dtype=torch.uint8, device="cuda", generator=g)
    key = torch.randint(0, 256, (a.H, a.W, 3), dtype=torch.uint8, device="cuda", generator=g)
    eng.forward(frames, key, iters=a.iters)
    torch.cuda.synchronize()
    ops.prof_enable(2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.forward(frames, key, iters=a.iters)
    e1.record()
    rec = ops.prof_collect()
    ops.prof_enable(0)
    total = sum(v["ms"] for v in rec.values())
    rows = sorted(rec.items(), key=lambda kv: -kv[1]["ms"])
    print(f"forward {e0.elapsed_time(e1):.2f} ms wall, {total:.2f} ms in kernels; B={a.batch} {a.W}x{a.H} iters={a.iters} {a.precision}")
    print(f"{'kernel:layer':<44}{'calls':>6}{'ms':>10}{'share':>8}{'avg us':>10}{'TFLOP/s':>9}")
    for k, v in rows:
        tf = v["flops"] / v["ms"] / 1e9 if v["flops"] else float("nan")
        print(f"{k:<44}{v['calls']:>6}{v['ms']:>10.3f}{v['ms'] / total:>8.1%}{1e3 * v['ms'] / v['calls']:>10.1f}{tf:>9.1f}")
    if a.out:
        json.dump({"config": vars(a), "kernel_ms": total, "layers": rec}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
