#!/usr/bin/env python3
"""Benchmark of the hot path: frame-pairs/sec for flow + warp + mask at 512x768 on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input per GPU: B frames
(default 64 = BASELINE.json configs[2]/[3], "64-frame 512x768 clip ... HBM-resident cost volumes")
against one shared key frame: RAFT dense flow (fp32, 20 iterations, frame -> key frame) -> bilinear
backward warp of the AI key frame -> low-confidence inpaint mask (conf < 0.95, 7x7 ellipse dilate).
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU (torchrun), frames
sharded frame-parallel (weak scaling, B per GPU), one RCCL broadcast of the key frame pair per step,
no other collective.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     the dominant kernel family (fp32-MFMA implicit-GEMM convolutions) measured live
                  with HIP events on the launch stream inside the timed region,
  "kernels":      the same measurement for the HBM-bound kernels (correlation volume, lookup,
                  upsample, warp, mask) against the 8 TB/s HBM peak,
  "cpu_baseline": the CPU oracle (a port; the reference's Python cannot travel to the GPU box) timed
                  on the host cores on a bounded sample of the same workload,
  "single_pair":  BASELINE.json configs[1] (one 512x768 pair) latency/throughput.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 768, 512                 # "512x768" = W x H (SURVEY conventions)
ITERS = 20
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32 dense peak


# ----------------------------------------------------------------------------------------------
# algorithmic work per pair, used as roofline numerators
# ----------------------------------------------------------------------------------------------
def conv_flops(hw, cin, cout, kh, kw):
    return 2.0 * hw * cin * cout * kh * kw


def encoder_flops(H, W):
    f = conv_flops((H // 2) * (W // 2), 3, 64, 7, 7)
    hw2, hw4, hw8 = (H // 2) * (W // 2), (H // 4) * (W // 4), (H // 8) * (W // 8)
    f += 4 * conv_flops(hw2, 64, 64, 3, 3)
    f += conv_flops(hw4, 64, 96, 3, 3) + 3 * conv_flops(hw4, 96, 96, 3, 3) + conv_flops(hw4, 64, 96, 1, 1)
    f += conv_flops(hw8, 96, 128, 3, 3) + 3 * conv_flops(hw8, 128, 128, 3, 3) + conv_flops(hw8, 96, 128, 1, 1)
    f += conv_flops(hw8, 128, 256, 1, 1)
    return f


def update_flops(H, W, with_mask):
    n = (H // 8) * (W // 8)
    f = conv_flops(n, 324, 256, 1, 1) + conv_flops(n, 256, 192, 3, 3) + conv_flops(n, 2, 128, 7, 7)
    f += conv_flops(n, 128, 64, 3, 3) + conv_flops(n, 256, 126, 3, 3)
    f += 6 * conv_flops(n, 384, 128, 1, 5)
    f += conv_flops(n, 128, 256, 3, 3) + conv_flops(n, 256, 2, 3, 3)
    if with_mask:
        f += conv_flops(n, 128, 256, 3, 3) + conv_flops(n, 256, 576, 1, 1)
    return f


def algorithmic_work(H, W, B, shared_key=True):
    """FLOPs of all convolutions and bytes of the HBM-bound kernels for one step of B pairs."""
    n = (H // 8) * (W // 8)
    n_enc = (B + 1) if shared_key else 2 * B          # fnet images
    conv = (n_enc + B) * encoder_flops(H, W)          # + cnet on every frame
    conv += B * (ITERS * update_flops(H, W, False) + (update_flops(H, W, True) - update_flops(H, W, False)))
    pyr = sum((n * ((H // 8) >> l) * ((W // 8) >> l)) for l in range(4)) * 4.0
    return {
        "conv_flops": conv,
        # what the kernels actually execute: the context (`inp`) third of the six GRU convolutions is
        # loop-invariant and evaluated once per pair instead of once per iteration
        "conv_flops_executed": conv - B * (ITERS - 1) * 6 * conv_flops(n, 128, 128, 1, 5),
        "volume_flops": B * 2.0 * n * n * 256,
        "volume_bytes": B * (n * 256 * 4.0 + n * n * 4.0) + (1 if shared_key else B) * n * 256 * 4.0,   # GEMM: read fmaps, write level 0
        "pool_bytes": B * (n * n * 4.0 + (pyr - n * n * 4.0)),                                          # read level 0, write levels 1-3
        "lookup_bytes": B * n * (400 * 4.0 + 324 * 4.0 + 8.0),                                          # per launch
        "upsample_bytes": B * (n * 576 * 4.0 + n * 8.0 + H * W * 8.0),
        "warp_bytes": B * (H * W * 8.0 + H * W * 3.0) + H * W * 3.0,
        "mask_bytes": B * (H * W * 4.0 + H * W * 1.0),
    }


# ----------------------------------------------------------------------------------------------
# synthetic clip
# ----------------------------------------------------------------------------------------------
def make_clip(B, H, W, device, rank=0):
    g = torch.Generator(device="cpu").manual_seed(1234)
    noise = torch.rand((1, 3, H, W), generator=g)
    key = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(noise, (4, 4, 4, 4), mode="reflect"), 9, 1)
    key = (key - key.min()) / (key.max() - key.min())
    key = key.to(device)
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    frames = []
    for t in range(B):
        ph = 0.37 * (t + 1 + 64 * rank)
        fx = 8.0 * torch.sin(2 * math.pi * ys / H + ph) * torch.cos(2 * math.pi * xs / W)
        fy = 6.0 * torch.cos(2 * math.pi * xs / W + 0.5 * ph)
        gx = 2 * (xs + fx) / (W - 1) - 1
        gy = 2 * (ys + fy) / (H - 1) - 1
        fr = torch.nn.functional.grid_sample(key, torch.stack([gx, gy], -1)[None], mode="bilinear", padding_mode="border",
                                             align_corners=True)
        fr = fr * 255 + 2.0 * torch.randn(fr.shape, device=device)
        frames.append(fr.clamp(0, 255).round().to(torch.uint8)[0].permute(1, 2, 0))
    frames = torch.stack(frames).contiguous()
    key_u8 = (key[0] * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous()
    key_ai = (255 - key_u8).contiguous()
    gc = torch.Generator(device="cpu").manual_seed(4321 + rank)
    cn = torch.randn((B, 1, H // 16, W // 16), generator=gc)
    conf = torch.sigmoid(torch.nn.functional.interpolate(cn, size=(H, W), mode="bilinear", align_corners=False)[:, 0] * 2.0 + 3.6)
    conf[:, ::37, ::29] = 0.95          # planted exact-threshold values
    conf[:, 5::41, 3::31] = 0.9
    return frames, key_u8, key_ai, conf.to(device).contiguous()


# ----------------------------------------------------------------------------------------------
def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box can
    report hundreds of CPUs while the container is limited to a few -- oversubscribing torch's thread pool
    there is catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(math.ceil(int(q) / int(p)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(math.ceil(q / p))))
        except Exception:
            pass
    return max(1, n)


def pick_cpu_threads() -> int:
    """Micro-benchmark a representative convolution at a few thread counts and keep the fastest."""
    cap = usable_cores()
    cands = sorted({min(cap, c) for c in (cap, 64, 32, 16, 8)})
    x = torch.randn(1, 128, 96, 64)
    w = torch.randn(128, 128, 3, 3)
    best, best_t = cands[0], float("inf")
    for c in cands:
        torch.set_num_threads(c)
        torch.nn.functional.conv2d(x, w, padding=1)
        t0 = time.perf_counter()
        for _ in range(5):
            torch.nn.functional.conv2d(x, w, padding=1)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = c, dt
        if dt > 2.0:          # pathological: stop probing larger pools
            break
    return best


def cpu_baseline(budget_s=15.0, max_pairs=24):   # a bounded sample: ~15-20 s of CPU work
    """The CPU oracle (port of the reference's path) on the host cores: flow + warp + mask per pair."""
    from oracle import mask_oracle, raft_oracle, warp_oracle
    threads = pick_cpu_threads()
    torch.set_num_threads(threads)
    sd = raft_oracle.init_state_dict(0)
    frames, key, key_ai, conf = make_clip(max_pairs, H, W, "cpu")
    done, t0 = 0, time.time()
    while done < max_pairs and (done == 0 or time.time() - t0 < budget_s):
        a = frames[done].permute(2, 0, 1)[None].float()
        b = key.permute(2, 0, 1)[None].float()
        _, up = raft_oracle.raft_forward(sd, a, b, iters=ITERS)
        flow = up[0].permute(1, 2, 0).contiguous().numpy()
        warp_oracle.warp_frame(key_ai.numpy(), flow, mode="bilinear")
        mask_oracle.generate_mask(conf[done].numpy(), conf[done].numpy().copy(), 0.95, 7)
        done += 1
    dt = time.time() - t0
    return {"value": done / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{done} pair(s) 512x768, RAFT {ITERS} iters fp32 + bilinear warp + mask, torch-CPU oracle, {dt:.1f} s"}


def flush_c_stdio():
    """RCCL / the HIP runtime write banner lines through C stdio; when stdout is a pipe they sit in a buffer until the
    process exits and would land AFTER rank 0's result line (on any rank: torchrun merges the streams)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--warp-mode", default="bilinear", choices=["bilinear", "bicubic", "cv2_cubic"])
    ap.add_argument("--no-prof", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="matrix-core arithmetic of the timed run (fp32 = the reference's; bf16x3 = opt-in split-bf16 fast mode)")
    ap.add_argument("--no-fast", action="store_true", help="skip the secondary bf16x3 measurement")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # OFX_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, broadcast, barrier, all-reduce) on one rank
    use_dist = world > 1 or (os.environ.get("OFX_BENCH_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}", file=sys.stderr)

    from sd_animation_optical_flow_amd import clip, ops
    from sd_animation_optical_flow_amd.raft import RaftEngine
    from sd_animation_optical_flow_amd.weights import random_state_dict

    B = args.batch
    eng = RaftEngine(random_state_dict(0), dev, precision=args.precision)
    frames, key, key_ai, conf = make_clip(B, H, W, dev, rank)

    def step():
        clip.broadcast_keyframe([key, key_ai], src=0)                 # the path's only exchange
        flow = eng.forward(frames, key, iters=ITERS)                  # frame -> key frame, shared image2
        warped, mask = ops.warp_and_mask(key_ai, flow, conf, warp_mode=args.warp_mode, thres=0.95, ksize=7)
        return flow, warped, mask

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = not args.no_prof
    barrier()
    flush_c_stdio()                          # every rank: library banners out before anybody prints a result
    if prof:
        ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern = {}
    if prof:
        kern = ops.prof_collect()
        ops.prof_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        flush_c_stdio()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = n_gpus * B * args.steps / elapsed
    work = algorithmic_work(H, W, B)
    out = {
        "metric": "frame-pairs/sec (flow+warp+mask) at 512x768",
        "value": round(value, 3), "unit": "pairs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16x3 (fp32 operands split into two bf16, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: {B}-frame 512x768 clip per GPU vs one shared key frame, RAFT {ITERS} iters "
                               f"fp32 + {args.warp_mode} warp + mask(conf<0.95, 7x7); configs[3] sharding at N>1",
                   "frames_per_gpu": B, "H": H, "W": W, "iters": ITERS, "parallelism": f"frame-parallel x{n_gpus}"},
    }

    def per_launch(names):
        ms = sum(kern[n]["ms"] for n in names if n in kern)
        calls = sum(kern[n]["calls"] for n in names if n in kern)
        return ms, calls

    if kern:
        conv_names = ["igemm_conv", "igemm_conv_gru_zr", "igemm_conv_gru_q", "igemm_conv_flow"]
        ms, calls = per_launch(conv_names)
        steps = args.steps
        if ms > 0:
            tf = work["conv_flops"] * steps / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "igemm_kernel (fp32 MFMA implicit-GEMM conv, all epilogues)",
                               "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                               "launches_per_step": calls // steps, "avg_launch_ms": round(ms / calls, 4),
                               "flops_per_step": work["conv_flops"], "share_of_step": round(ms / steps / ms_per_step, 4),
                               "executed_tflops": round(work["conv_flops_executed"] * steps / (ms * 1e-3) / 1e12, 2),
                               "executed_frac": round(work["conv_flops_executed"] * steps / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                               "note": "achieved = algorithmic FLOPs of the reference's convolutions / measured kernel time; "
                                       "executed_* discounts the GRU context term hoisted out of the 20-iteration loop"}
        ks = {}

        def hbm(name, key_bytes, label):
            m, c = per_launch([name])
            if c:
                per = work[key_bytes] / (c / steps)
                gbs = per / (m / c * 1e-3) / 1e9
                ks[label] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_launch": per, "avg_launch_ms": round(m / c, 4),
                             "launches_per_step": c // steps}
        hbm("igemm_corr_volume", "volume_bytes", "corr_volume_gemm")
        hbm("corr_pyramid_pool", "pool_bytes", "corr_pyramid_pool")
        work["lookup_total"] = work["lookup_bytes"] * ITERS
        hbm("corr_lookup", "lookup_total", "corr_lookup")
        hbm("upsample_flow", "upsample_bytes", "upsample_flow")
        hbm("warp_u8", "warp_bytes", "warp")
        hbm("generate_mask", "mask_bytes", "mask")
        m, c = per_launch(["igemm_corr_volume"])
        if c:
            tfv = work["volume_flops"] * steps / (m * 1e-3) / 1e12
            ks["corr_volume_gemm"]["tflops"] = round(tfv, 2)
            ks["corr_volume_gemm"]["mfma_frac"] = round(tfv / MFMA_F32_PEAK_TFLOPS, 4)
        # HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs; FETCH_SIZE doubled per the gfx950 correction, calibrated on the pooling
        # kernel: corrected 12.83 GB = its exact algorithmic 12.83 GB).  PMC cannot be collected inside this run.
        import glob
        tps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_b64.json")))   # newest round last
        if B == 64 and tps:
            tp = tps[-1]
            tr = json.load(open(tp))
            if "roofline" in out and "igemm_conv_all" in tr:
                out["roofline"]["traffic"] = tr["igemm_conv_all"]["hbm_bytes_per_launch_corrected"]
                out["roofline"]["traffic_source"] = f"profiles/{os.path.basename(tp)} (avg over all convolution launches)"
            for label, tkey in (("corr_volume_gemm", "corr_volume_gemm"), ("corr_lookup", "corr_lookup"),
                               ("corr_pyramid_pool", "pyramid_pool"), ("upsample_flow", "upsample"),
                               ("warp", "warp"), ("mask", "mask")):
                if label in ks and tkey in tr:
                    ks[label]["traffic"] = tr[tkey]["hbm_bytes_per_launch_corrected"]
                    # bytes the memory system actually moved (PMC) over the measured launch time: how hard the kernel
                    # drives HBM, as opposed to `frac` (algorithmic bytes: how much of that traffic was necessary)
                    if ks[label].get("avg_launch_ms"):
                        gbs = ks[label]["traffic"] / (ks[label]["avg_launch_ms"] * 1e-3) / 1e9
                        ks[label]["traffic_gbs"] = round(gbs, 1)
                        ks[label]["traffic_frac"] = round(gbs / HBM_PEAK_GBS, 4)
        out["kernels"] = ks
        tot = sum(v["ms"] for v in kern.values())
        out["kernel_time_share"] = {k: round(v["ms"] / tot, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:8]}

    if not args.no_single and world == 1:
        f1 = frames[:1].contiguous()
        c1 = conf[:1].contiguous()
        for _ in range(2):
            fl = eng.forward(f1, key, iters=ITERS)
            ops.warp_and_mask(key_ai, fl, c1, warp_mode=args.warp_mode, thres=0.95, ksize=7)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            fl = eng.forward(f1, key, iters=ITERS)
            ops.warp_and_mask(key_ai, fl, c1, warp_mode=args.warp_mode, thres=0.95, ksize=7)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        out["single_pair"] = {"workload": "BASELINE configs[1]: one 512x768 pair", "ms": round(dt * 1e3, 3), "pairs_per_s": round(1 / dt, 2)}

    if not args.no_fast and world == 1 and args.precision == "fp32":
        # secondary measurement: the opt-in split-bf16 mode on the same clip, with its flow error against the fp32 run
        ref_flow = eng.forward(frames, key, iters=ITERS)
        fast = RaftEngine(random_state_dict(0), dev, precision="bf16x3")

        def fstep():
            fl = fast.forward(frames, key, iters=ITERS)
            ops.warp_and_mask(key_ai, fl, conf, warp_mode=args.warp_mode, thres=0.95, ksize=7)
            return fl
        fl = fstep()
        epe = float((fl - ref_flow).pow(2).sum(-1).sqrt().mean())
        emax = float((fl - ref_flow).pow(2).sum(-1).sqrt().max())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            fstep()
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t1) / args.steps
        out["fast_mode"] = {"precision": "bf16x3", "value": round(B / dtf, 3), "unit": "pairs/s", "ms_per_step": round(dtf * 1e3, 3),
                            "flow_epe_vs_fp32_px": epe, "flow_max_err_vs_fp32_px": emax,
                            "note": "opt-in; not the headline value (the reference computes in fp32)"}
        del fast, ref_flow

    if not args.no_cpu_baseline and world == 1:
        # child process with a hard time limit: the baseline must never take the GPU number down with it
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                               timeout=240, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": usable_cores(), "kind": "port", "error": repr(e)[:200]}
    if dist is not None:
        dist.destroy_process_group()       # before the result line: RCCL prints its own banner lines on teardown
    flush_c_stdio()
    sys.stderr.flush()
    print(json.dumps(out), flush=True)      # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
