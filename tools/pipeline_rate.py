#!/usr/bin/env python3
"""End-to-end rate of `pipeline.ClipPipeline.run` over a workspace on ONE GPU, with the device-only rate beside it.

    python tools/pipeline_rate.py [frames=256] [io_threads=default_io_threads()]

A 512x768 workspace of `frames` PNGs (the bench clip's frames: 4 key-frame segments) is written to a temporary directory, then

  end_to_end     `ClipPipeline.run`: PNG decode -> pinned staging -> H2D -> flow both ways + forward-backward confidence -> warp
                 (inside the convex upsample) + mask -> SD-inpaint inputs -> render hook -> D2H -> PNG encode, every file on disk;
  inline_io      the same with `io_threads=0` (decode / upload / download / encode on the thread that enqueues the kernels: the
                 reference's arrangement, ofgen_keyframe_inpaint.py:415-432,585-600);
  device_only    the same `process_batch` + render calls on frames already resident in HBM, nothing written.

SURVEY 8(e) names input decode / H2D per rank as the limit of the 8-GPU scaling; `ratio` = end_to_end / device_only is the share of
the device rate the host side sustains (target >= 0.85), `cores` the host cores this process may use.  `host_cpu_s_per_frame` is
the process's CPU time (every thread: decode, encode, the enqueueing thread) per frame of the end-to-end run, and
`cores_for_8_ranks` = 8 ranks x that rank's frames/s x CPU-seconds per frame: the host cores one 8-GPU node needs for all of its
ranks to sustain this rate (no 8-GPU node was reachable from the build sessions: a projection, not a measurement).
"""
import json
import os
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def measure(n_frames: int = 256, io_threads=None, seg: int = 64, batch: int = 64, keep: bool = False, reps: int = 2, inline: bool = True,
            edge_batch: int = 16, share_pool: bool = True) -> dict:
    import bench
    from sd_animation_optical_flow_amd import pdcnet_of, pipeline
    from sd_animation_optical_flow_amd.workspace import VideoData
    dev = torch.device("cuda")
    H, W = bench.H, bench.W
    if io_threads is None:
        io_threads = pipeline.default_io_threads()
    n_seg = max(1, n_frames // seg)
    frames, key, _, _ = bench.make_clip(seg - 1, H, W, dev)
    clip_frames = []
    for s in range(n_seg):                                    # segment s: its key frame, then seg - 1 frames that move against it
        k = torch.roll(key, shifts=17 * s, dims=1)
        clip_frames.append(k.cpu().numpy())
        clip_frames += [f for f in torch.roll(frames, shifts=17 * s, dims=2).cpu().numpy()]
    n = len(clip_frames)
    flags = [i % seg == 0 for i in range(n)]
    root = tempfile.mkdtemp(prefix="ofx_ws_")
    try:
        t0 = time.perf_counter()
        video = VideoData(clip_frames, (W, H), root)
        t_extract = time.perf_counter() - t0
        png_mb = sum(os.path.getsize(os.path.join(root, "raw-frames", f)) for f in os.listdir(os.path.join(root, "raw-frames"))) / n / 1e6
        algo = pdcnet_of.PDCNetPlus("random:0", device=dev)

        def run(threads):
            pipe = pipeline.ClipPipeline(algo, batch=batch, warp_mode="bilinear", thres=0.95, ksize=7, io_threads=threads,
                                         edge_batch=edge_batch, share_pool=share_pool)
            shutil.rmtree(os.path.join(root, "ai-frames"))
            os.makedirs(os.path.join(root, "ai-frames"))
            torch.cuda.synchronize()
            c = time.process_time()
            t = time.perf_counter()
            pipe.run(video, flags)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            cpu = time.process_time() - c
            assert all(video.generated(i) for i in range(n))
            return n / dt, cpu / n

        # device only: the same compute calls over resident frames
        pipe = pipeline.ClipPipeline(algo, batch=batch, warp_mode="bilinear", thres=0.95, ksize=7, io_threads=0, edge_batch=0)
        dev_frames = torch.from_numpy(__import__("numpy").stack(clip_frames)).to(dev)

        def device_only():
            torch.cuda.synchronize()
            t = time.perf_counter()
            for s in range(n_seg):
                k_raw = dev_frames[s * seg]
                k_ai = pipe.render_key(k_raw).contiguous()
                ids = list(range(s * seg + 1, (s + 1) * seg))
                for b0 in range(0, len(ids), batch):
                    sub = ids[b0:b0 + batch]
                    raws = dev_frames[sub[0]:sub[-1] + 1]
                    for pkt, raw in zip(pipe.process_batch(k_raw, k_ai, raws, sub, s * seg), raws):
                        pipe.render(pkt, raw)
            torch.cuda.synchronize()
            return n / (time.perf_counter() - t)

        device_only()                                         # warm-up: workspaces, kernels
        d = max(device_only() for _ in range(max(1, reps)))
        run(io_threads)                                       # warm-up: thread pools, pinned buffers, page cache
        e, cpu_s = max(run(io_threads) for _ in range(max(1, reps)))
        out = {"workload": f"{n}-frame 512x768 workspace, {n_seg} key-frame segments, ClipPipeline.run (flow both ways + forward-backward "
                           f"confidence, warp + mask, SD-inpaint inputs, render, PNG in / PNG out), 1 GPU",
               "frames": n, "png_mb_per_frame": round(png_mb, 3), "io_threads": io_threads, "edge_batch": edge_batch, "share_pool": share_pool,
               "cores": bench.usable_cores(),
               "end_to_end_fps": round(e, 2), "device_only_fps": round(d, 2), "ratio": round(e / d, 4),
               "host_cpu_s_per_frame": round(cpu_s, 5), "cores_for_8_ranks": round(8 * e * cpu_s, 1), "extract_s": round(t_extract, 2)}
        if inline:
            i0, _ = run(0)
            out["inline_io_fps"] = round(i0, 2)
            out["inline_ratio"] = round(i0 / d, 4)
        return out
    finally:
        if not keep:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    th = int(sys.argv[2]) if len(sys.argv) > 2 else None
    eb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    sp = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
    print(json.dumps(measure(nf, th, edge_batch=eb, share_pool=sp)))
