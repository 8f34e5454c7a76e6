#!/usr/bin/env python3
"""Where a rank's host CPU time goes during `ClipPipeline.run`: per-thread user+system seconds (from /proc/self/task) around one
end-to-end run of a 64-frame 512x768 workspace.    python tools/host_cpu_profile.py [frames=64] [io_threads]"""
import os, sys, time, tempfile, shutil, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sd_animation_optical_flow_amd import pdcnet_of, pipeline
from sd_animation_optical_flow_amd.workspace import VideoData


def threads():
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (name, (int(rest[11]) + int(rest[12])) / tick)
        except Exception:
            pass
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
th = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda")
frames, key, _, _ = bench.make_clip(n - 1, bench.H, bench.W, dev)
clip_frames = [key.cpu().numpy()] + [f for f in frames.cpu().numpy()]
root = tempfile.mkdtemp(prefix="ofx_ws_")
video = VideoData(clip_frames, (bench.W, bench.H), root)
flags = [i == 0 for i in range(n)]
algo = pdcnet_of.PDCNetPlus("random:0", device=dev)
import threading
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
lock = threading.Lock()
def wrap(fn, label):
    def w(*a, **k):
        c, t = time.thread_time(), time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            with lock:
                e = acc[label]; e[0] += 1; e[1] += time.thread_time() - c; e[2] += time.perf_counter() - t
    return w
video.get_raw_frame = wrap(video.get_raw_frame, "decode (get_raw_frame)")
video.put_ai_frame = wrap(video.put_ai_frame, "encode+write (put_ai_frame)")
seen, names, stop = {}, {}, threading.Event()
def monitor():
    while not stop.is_set():
        for tid, (name, sec) in threads().items():
            seen[tid] = sec
        for t in threading.enumerate():
            if t.native_id:
                names[t.native_id] = t.name
        time.sleep(0.01)
for rep in range(2):
    pipe = pipeline.ClipPipeline(algo, batch=64, warp_mode="bilinear", thres=0.95, ksize=7, io_threads=th)
    shutil.rmtree(os.path.join(root, "ai-frames")); os.makedirs(os.path.join(root, "ai-frames"))
    torch.cuda.synchronize()
    acc.clear(); m0 = time.thread_time(); seen.clear(); stop.clear()
    mon = threading.Thread(target=monitor, name="monitor"); mon.start()
    before = threads(); t0 = time.perf_counter(); c0 = time.process_time()
    pipe.run(video, flags)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0; cpu = time.process_time() - c0
    after = threads(); stop.set(); mon.join()
    agg = collections.defaultdict(float)
    for tid, (name, s) in after.items():
        agg[name] += s - before.get(tid, (name, 0.0))[1]
    print(f"run {rep}: {n / dt:.1f} frames/s, wall {dt:.3f} s, process CPU {cpu:.2f} s = {cpu / n * 1e3:.1f} ms per frame")
    print(f"    main thread CPU {time.thread_time() - m0:.2f} s")
    for label, (cnt, c, w) in acc.items():
        print(f"    {label:<30} {cnt:4d} calls, CPU {c:6.2f} s ({c / cnt * 1e3:6.1f} ms each), wall {w / cnt * 1e3:6.1f} ms each")
    per = sorted(((s - before.get(tid, (name, 0.0))[1], tid) for tid, (name, s) in after.items()), reverse=True)[:10]
    rows = sorted(((sec - before.get(tid, ("", 0.0))[1], names.get(tid, "(native)"), tid) for tid, sec in seen.items()), reverse=True)
    by = collections.defaultdict(lambda: [0, 0.0])
    for sec, nm, tid in rows:
        key = nm.rsplit("_", 1)[0] if nm.startswith("ofx-") else nm
        by[key][0] += 1; by[key][1] += sec
    for key, (cnt, sec) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"    threads {key:<16} x{cnt:<3} CPU {sec:6.2f} s")
    print("    native top:", " ".join(f"{sec:.2f}" for sec, nm, _ in rows if nm == "(native)")[:120])
    print("    top threads alive at the end (s):", " ".join(f"{s:.2f}" for s, _ in per))
shutil.rmtree(root, ignore_errors=True)
