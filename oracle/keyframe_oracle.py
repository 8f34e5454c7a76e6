"""TEST INFRASTRUCTURE -- CPU oracle of the key-frame detector ("next" row f4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product never does.

PARITY UNPINNED: the reference's detector (ofgen_keyframe_inpaint.py:143-192,327-370) is made of OpenCV calls
(cv2.cvtColor BGR2HSV, cv2.Canny, cv2.dilate) and cv2 is neither in the reference tree nor in this image, so no
output of the real library could be captured.  The functions below restate OpenCV 4.x's published algorithms
(imgproc/src/canny.cpp, color_hsv.simd.hpp, morph) on integers; they are the definition the HIP kernels are held
to, bit for bit.

    detect_edges(frame)   = dilate(Canny(V, low, high), ones(k, k))            :161-192
        V       = max(B, G, R)                                                 (HSV value channel of an 8-bit image)
        median  = np.median(V); low = int(max(0, (1 - 1/3) * median)); high = int(min(255, (1 + 1/3) * median))
        k       = 4 + round(sqrt(W * H) / 192), made odd                       :153-158  (from the FIRST frame)
    mean_pixel_distance   = sum(|a - b|) / (H * W)                             :143-150
    frame_generator       : a frame is a key frame when mean_pixel_distance(edges, key_edges) exceeds
                            th * (max_gap - gap) / max_gap                     :327-370
"""
from __future__ import annotations

import math

import numpy as np

TG22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)     # canny.cpp: tan(22.5 deg) in 15-bit fixed point


def hsv_value(frame_bgr: np.ndarray) -> np.ndarray:
    """V channel of cv2.cvtColor(frame, COLOR_BGR2HSV) for uint8 input: max of the three channels."""
    return frame_bgr.max(axis=2).astype(np.uint8)


def estimated_kernel_size(frame_width: int, frame_height: int) -> int:
    size = 4 + round(math.sqrt(frame_width * frame_height) / 192)
    if size % 2 == 0:
        size += 1
    return size


def canny_thresholds(lum: np.ndarray) -> tuple:
    sigma = 1.0 / 3.0
    median = float(np.median(lum))
    return int(max(0, (1.0 - sigma) * median)), int(min(255, (1.0 + sigma) * median))


def _sobel16(lum: np.ndarray):
    """cv2.Sobel(src, CV_16S, ., ., 3, 1, 0, BORDER_REPLICATE) for dx and dy."""
    p = np.pad(lum.astype(np.int32), 1, mode="edge")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return dx, dy


def canny_map(lum: np.ndarray, low: int, high: int) -> np.ndarray:
    """Non-maximum suppression stage of cv2.Canny (aperture 3, L1 gradient): 2 = strong edge, 0 = candidate that
    survives only if connected to a strong edge, 1 = not an edge."""
    if low > high:
        low, high = high, low
    H, W = lum.shape
    dx, dy = _sobel16(lum)
    mag = np.zeros((H + 2, W + 2), dtype=np.int64)            # one zero pixel all around, like the row buffers
    mag[1:-1, 1:-1] = np.abs(dx) + np.abs(dy)
    m = mag[1:-1, 1:-1]
    x = np.abs(dx).astype(np.int64)
    y = np.abs(dy).astype(np.int64) << 15
    tg22x = x * TG22
    tg67x = tg22x + (x << 16)
    left, right = mag[1:-1, :-2], mag[1:-1, 2:]
    up, down = mag[:-2, 1:-1], mag[2:, 1:-1]
    s_neg = (dx ^ dy) < 0                                      # s = -1 where the signs differ
    # diagonal neighbours: (prev row, j - s) and (next row, j + s)
    up_d = np.where(s_neg, mag[:-2, 2:], mag[:-2, :-2])
    down_d = np.where(s_neg, mag[2:, :-2], mag[2:, 2:])
    horiz = y < tg22x
    vert = (~horiz) & (y > tg67x)
    diag = (~horiz) & (~vert)
    is_max = (horiz & (m > left) & (m >= right)) | (vert & (m > up) & (m >= down)) | (diag & (m > up_d) & (m > down_d))
    cand = (m > low) & is_max
    out = np.ones((H, W), dtype=np.uint8)
    out[cand] = 0
    out[cand & (m > high)] = 2
    return out


def canny(lum: np.ndarray, low: int, high: int) -> np.ndarray:
    """cv2.Canny(lum, low, high): hysteresis = every candidate 8-connected (through candidates) to a strong edge."""
    from scipy import ndimage
    mp = canny_map(lum, int(low), int(high))
    lab, n = ndimage.label(mp != 1, structure=np.ones((3, 3), dtype=int))
    keep = np.zeros(n + 1, dtype=bool)
    keep[np.unique(lab[mp == 2])] = True
    keep[0] = False
    return np.where(keep[lab], 255, 0).astype(np.uint8)


def dilate_square(a: np.ndarray, k: int) -> np.ndarray:
    """cv2.dilate(a, np.ones((k, k), uint8)), k odd: window maximum, pixels outside the image are ignored."""
    r = k // 2
    H, W = a.shape
    p = np.zeros((H + 2 * r, W + 2 * r), dtype=np.uint8)
    p[r:r + H, r:r + W] = a
    out = np.zeros_like(a)
    for dy in range(k):
        for dx in range(k):
            out = np.maximum(out, p[dy:dy + H, dx:dx + W])
    return out


def detect_edges(frame_bgr: np.ndarray, ksize: int = None) -> np.ndarray:
    lum = hsv_value(frame_bgr)
    if ksize is None:
        ksize = estimated_kernel_size(lum.shape[1], lum.shape[0])
    low, high = canny_thresholds(lum)
    return dilate_square(canny(lum, low, high), ksize)


def mean_pixel_distance(left: np.ndarray, right: np.ndarray) -> float:
    assert left.ndim == 2 and left.shape == right.shape
    return float(np.sum(np.abs(left.astype(np.int32) - right.astype(np.int32))) / float(left.shape[0] * left.shape[1]))


def gaps(fps: float, min_gap: int = -1, max_gap: int = -1) -> tuple:
    """frame_generator :330-341."""
    mn = int(10 * fps / 30) if min_gap == -1 else int(max(1, min_gap) * fps / 30)
    mx = int(300 * fps / 30) if max_gap == -1 else int(max(10, max_gap) * fps / 30)
    return mn, mx


def keyframe_flags(frames, fps: float = 30.0, th: float = 8.5, min_gap: int = -1, max_gap: int = -1,
                   keep_every: int = 1) -> list:
    """The decisions of frame_generator (:342-368) for an already decoded / resized frame sequence: one flag per KEPT
    frame (`ctr % keep_every == 0`, :351).  `gap` counts every decoded frame: the reference increments it (:347)
    before the keep_every filter (:351-352)."""
    _, mx = gaps(fps, min_gap, max_gap)
    flags, key_edges, gap, ksize = [], None, 0, None
    for ctr, frame in enumerate(frames):
        gap += 1
        if ctr % keep_every != 0:
            continue
        if ksize is None:
            ksize = estimated_kernel_size(frame.shape[1], frame.shape[0])
        edges = detect_edges(frame, ksize)
        if key_edges is None:
            key_edges = edges
            flags.append(True)
            continue
        delta = mean_pixel_distance(edges, key_edges)
        if th * (mx - gap) / mx < delta:
            key_edges, gap = edges, 0
            flags.append(True)
        else:
            flags.append(False)
    return flags
