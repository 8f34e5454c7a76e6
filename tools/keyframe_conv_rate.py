"""Rate of the KeyframeConv scoring sweep on the GPU box: one window of N frames at 512x768 = N*(N-1) ordered pairs
through `PDCNetAux.keyframe_scores_device` (shared per-frame features, on-device confidence sums).
    python tools/keyframe_conv_rate.py [N ...]        (default 9 15: the reference's kernel 17 / dilation 2 and a 15-frame window)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sd_animation_optical_flow_amd import ofgen, pdcnet_of
from sd_animation_optical_flow_amd.weights import random_state_dict
from sd_animation_optical_flow_amd.workspace import VideoFrameIndices

H, W = 512, 768


class Clip:                                   # the two members PDCNetAux needs of a VideoData
    def __init__(self, n):
        rng = np.random.default_rng(0)
        base = rng.integers(0, 256, (H + 64, W + 64, 3), dtype=np.uint8)
        self.frames = [np.ascontiguousarray(base[2 * i:2 * i + H, 3 * i:3 * i + W]) for i in range(n)]
        self.size_hw = (H, W)

    def get_raw_frame(self, i):
        return self.frames[i]


algo = pdcnet_of.create_of_algo(random_state_dict(0))
for n in [int(a) for a in sys.argv[1:]] or [9, 15]:
    clip = Clip(n)
    with tempfile.TemporaryDirectory() as ws:
        aux = ofgen.PDCNetAux(algo, ws)
        win = VideoFrameIndices.from_n(n)
        aux.keyframe_scores_device(clip, win)                       # warm-up (uploads the frames, sizes the workspace)
        torch.cuda.synchronize()
        reps = 3
        t = time.perf_counter()
        for _ in range(reps):
            s = aux.keyframe_scores_device(clip, win)
        best = int(torch.argmax(s).item())
        dt = (time.perf_counter() - t) / reps
        print(f"window of {n:2d} frames: {n * (n - 1):3d} ordered pairs in {dt * 1e3:8.1f} ms = {n * (n - 1) / dt:6.1f} pairs/s  (winner {best})", flush=True)
