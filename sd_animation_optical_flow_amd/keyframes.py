"""Key-frame detection on the device ("next" row f4).

Mirrors the detector of the reference's drivers (ofgen_keyframe_inpaint.py:143-192,327-370): a frame becomes a key
frame when the mean absolute difference between its dilated Canny edge map and the current key frame's exceeds
`th * (max_gap - gap) / max_gap`.  Video decoding and the INTER_AREA resize of `frame_generator` stay with the
caller (they are I/O); everything from the decoded BGR frame on runs through `ofx_detect_edges` /
`ofx_abs_diff_sum_u8`.  Parity: OpenCV's Canny / dilate restated (oracle/keyframe_oracle.py), unpinned -- cv2 is not
available to compare with.
"""
from __future__ import annotations

import math
from typing import Iterable, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import ops


def estimated_kernel_size(frame_width: int, frame_height: int) -> int:
    """ofgen_keyframe_inpaint.py:153-158."""
    size = 4 + round(math.sqrt(frame_width * frame_height) / 192)
    if size % 2 == 0:
        size += 1
    return size


def _dev(frames, device) -> torch.Tensor:
    t = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames))
    if t.dtype != torch.uint8:
        raise RuntimeError("frames must be uint8")
    return t.to(device).contiguous()


def detect_edges(frame_bgr, ksize: Optional[int] = None, device="cuda") -> torch.Tensor:
    """`detect_edges(frame)` (:190-192) for one [H,W,3] frame or a batch [B,H,W,3]; returns uint8 [H,W] / [B,H,W] on
    the device.  ksize defaults to estimated_kernel_size(W, H) (the reference fixes it from its first frame)."""
    f = _dev(frame_bgr, device)
    single = f.dim() == 3
    if single:
        f = f[None]
    if ksize is None:
        ksize = estimated_kernel_size(f.shape[2], f.shape[1])
    e = ops.detect_edges(f, ksize)
    return e[0] if single else e


def mean_pixel_distance(left: torch.Tensor, right: torch.Tensor) -> float:
    """:143-150 for two uint8 [H,W] edge maps on the device (one 8-byte read-back)."""
    if left.dim() != 2 or tuple(left.shape) != tuple(right.shape):
        raise AssertionError("two 2-D images of the same shape expected")
    s = ops.abs_diff_sum_u8(left[None].contiguous(), right.contiguous())
    return float(int(s.item()) / float(left.shape[0] * left.shape[1]))


def gaps(fps: float, min_gap: int = -1, max_gap: int = -1) -> Tuple[int, int]:
    """:330-341."""
    mn = int(10 * fps / 30) if min_gap == -1 else int(max(1, min_gap) * fps / 30)
    mx = int(300 * fps / 30) if max_gap == -1 else int(max(10, max_gap) * fps / 30)
    return mn, mx


def frame_generator(frames: Iterable, fps: float = 30.0, th: float = 8.5, min_gap: int = -1, max_gap: int = -1,
                    batch: int = 16, device="cuda", keep_every: int = 1,
                    max_decoded: int = 60 * 60) -> Iterator[Tuple[object, bool, int]]:
    """The decision loop of the reference's `frame_generator` (:342-368).  `frames` iterates over EVERY decoded frame
    (numpy or tensors, BGR uint8 [H,W,3], already resized); only every `keep_every`-th one is examined and yielded as
    (frame, is_keyframe, index-among-kept-frames), but -- as in the reference, which does `gap += 1` before its
    `ctr % keep_every` filter (:346-352) -- the gap that relaxes the threshold counts decoded frames, dropped ones
    included.  Stops after the frame with decoded index `max_decoded` (:366-367).  Edge maps are computed `batch`
    kept frames at a time on the device; the sequential part is one small reduction per frame."""
    _, mx = gaps(fps, min_gap, max_gap)
    keep_every = max(1, int(keep_every))
    key_edges, gap, ksize, idx = None, 0, None, -1
    buf: List = []          # (frame, decoded frames since the previous kept frame)

    def flush(buf):
        nonlocal key_edges, gap, ksize, idx
        dev = torch.stack([_dev(f, device) for f, _ in buf])
        if ksize is None:
            ksize = estimated_kernel_size(dev.shape[2], dev.shape[1])
        edges = ops.detect_edges(dev, ksize)
        for j, (frame, inc) in enumerate(buf):
            idx += 1
            gap += inc
            if key_edges is None:
                key_edges = edges[j]
                yield frame, True, idx
                continue
            delta = mean_pixel_distance(edges[j], key_edges)
            if th * (mx - gap) / mx < delta:
                key_edges, gap = edges[j], 0
                yield frame, True, idx
            else:
                yield frame, False, idx

    pending = 0
    for ctr, frame in enumerate(frames):
        pending += 1
        if ctr % keep_every == 0:
            buf.append((frame, pending))
            pending = 0
            if len(buf) == batch:
                yield from flush(buf)
                buf = []
        if ctr >= max_decoded:
            break
    if buf:
        yield from flush(buf)
