"""CPU oracle for the confidence -> inpaint-mask part of the hot path.  TEST INFRASTRUCTURE ONLY.

numpy restatement of the integer / byte steps around the warp.  The reference
implements them with numpy + OpenCV calls (OpenCV absent here, unpinned upstream: exact-cv2 parity
is unpinned; the OpenCV pieces follow the library's published algorithms, cited inline).

  of_calc                ofgen_keyframe_inpaint.py:113-133 ; ofgen_pixel_inpaint.py:105-118
  generate_mask          ofgen_keyframe_inpaint.py:317-322 ; ofgen_pixel_inpaint.py:262-267
  confidence_to_mask     ofgen_keyframe_inpaint.py:237-248
  expand_mask            ofgen_keyframe_inpaint.py:968-973
  merge_images           ofgen_keyframe_inpaint.py:676-688
  mix_propagated_ai_frame ofgen_keyframe_inpaint.py:306-315
  compose (greedy multi-reference warp+mask)   ofgen_keyframe_inpaint.py:995-1027
  keyframe score         ofgen_keyframe_inpaint.py:665-668
"""
from __future__ import annotations

import numpy as np

from . import warp_oracle


# --------------------------------------------------------------------------------------
# morphology
# --------------------------------------------------------------------------------------
def ellipse_kernel(k: int) -> np.ndarray:
    """cv2.getStructuringElement(MORPH_ELLIPSE, (k,k)) (imgproc/src/morph.dispatch.cpp):
    r = c = k//2; row i: dy = i - r; if |dy| <= r: dx = cvRound(c*sqrt((r^2-dy^2)/r^2)),
    ones on [c-dx, c+dx].  7x7 -> row widths 1,5,7,7,7,5,1."""
    r = k // 2
    c = k // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    out = np.zeros((k, k), np.uint8)
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            j1 = max(c - dx, 0)
            j2 = min(c + dx + 1, k)
            out[i, j1:j2] = 1
    return out


def dilate(mask: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """cv2.dilate, anchor at the centre, default border (constant = lowest value, i.e. pixels
    outside the image never win the max)."""
    kh, kw = kern.shape
    ay, ax = kh // 2, kw // 2
    h, w = mask.shape
    pad = np.zeros((h + kh - 1, w + kw - 1), mask.dtype)
    pad[ay:ay + h, ax:ax + w] = mask
    out = np.zeros_like(mask)
    for i in range(kh):
        for j in range(kw):
            if kern[i, j]:
                np.maximum(out, pad[i:i + h, j:j + w], out=out)
    return out


# --------------------------------------------------------------------------------------
# a15 of_calc distance map
# --------------------------------------------------------------------------------------
def travel_distance(flow: np.ndarray, confidence: np.ndarray, conf_floor: float = 0.9) -> np.ndarray:
    """v = |map - grid| with the f64 -> f32 -> subtract round trip of
    ofgen_keyframe_inpaint.py:118-126, zeroed where confidence < 0.9 (strict, f32 compare)."""
    h, w = flow.shape[:2]
    xs, ys = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    mx = (xs + flow[:, :, 0]).astype(np.float32)
    my = (ys + flow[:, :, 1]).astype(np.float32)
    mx = mx - np.arange(w).astype(np.float32)
    my = my - np.arange(h).astype(np.float32)[:, None]
    v = np.sqrt(mx * mx + my * my).astype(np.float32)
    v[confidence < np.float32(conf_floor)] = 0
    return v


# --------------------------------------------------------------------------------------
# a16 / a17 masks
# --------------------------------------------------------------------------------------
def generate_mask(confidence: np.ndarray, log_confidence: np.ndarray, thres: float = 0.8, ksize: int = 7):
    """Returns (mask uint8 {0,255}, log_confidence with low-confidence pixels reset to 0)."""
    low = confidence < np.float32(thres)          # numpy compares in the array dtype (f32)
    mask = np.where(low, np.uint8(255), np.uint8(0))
    lc = log_confidence.copy()
    lc[low] = 0
    return dilate(mask, ellipse_kernel(ksize)), lc


def confidence_to_mask(confidence, flow, dist, travel, thres, warp_mode="cv2_cubic", ksize: int = 15):
    """Stateful travel-distance mask.  Returns (mask, new_travel)."""
    low = confidence < np.float32(0.9)
    mask = np.where(low, np.uint8(255), np.uint8(0))
    travel = warp_oracle.warp_frame(travel, flow, mode=warp_mode) + dist
    travel = travel.astype(np.float32)
    travel[low] = 0
    far = travel > np.float32(thres)
    mask[far] = 255
    travel[far] = 0
    return dilate(mask, ellipse_kernel(ksize)), travel


# --------------------------------------------------------------------------------------
# a19 expand_mask
# --------------------------------------------------------------------------------------
# cv::cvtColor RGB2GRAY for 8u (imgproc/src/color_rgb.simd.hpp): 15-bit fixed point
R2Y, G2Y, B2Y, GRAY_SHIFT = 9798, 19235, 3735, 15


def laplacian_edges(image_bgr: np.ndarray, edge_thres: int = 20) -> np.ndarray:
    """abs(Laplacian(img, CV_64F)) (ksize=1 -> [[0,1,0],[1,-4,1],[0,1,0]], BORDER_REFLECT_101),
    .astype(uint8) (wraps modulo 256 -- values reach 1020), cvtColor(COLOR_RGB2GRAY) applied to a
    BGR image (channel 0 gets the 'R' weight), > 20 -> 255."""
    img = image_bgr.astype(np.int64)
    p = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="reflect")
    lap = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4 * img
    a = (np.abs(lap) & 255)
    gray = (a[:, :, 0] * R2Y + a[:, :, 1] * G2Y + a[:, :, 2] * B2Y + (1 << (GRAY_SHIFT - 1))) >> GRAY_SHIFT
    return np.where(gray > edge_thres, np.uint8(255), np.uint8(0))


def expand_mask(mask: np.ndarray, image_bgr: np.ndarray, ksize: int = 7) -> np.ndarray:
    return mask | dilate(laplacian_edges(image_bgr), ellipse_kernel(ksize))


# --------------------------------------------------------------------------------------
# a20 merges
# --------------------------------------------------------------------------------------
def merge_images(base: np.ndarray, second: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """'naive': mask2 = uint8(mask/255) (1 only where mask == 255); uint8 arithmetic."""
    m = (mask / 255).astype(np.uint8)[:, :, None]
    return base * (1 - m) + second * m


def mix_propagated_ai_frame(raw, warped, mask, ppw: float = 1.0):
    if ppw < 0.001:
        return raw
    wts = np.zeros(raw.shape[:2], np.float32)
    wts[mask <= 127] = ppw
    wts[mask > 127] = 1 - ppw
    wts = wts[:, :, None]
    out = raw.astype(np.float32) * (1 - wts) + warped.astype(np.float32) * wts
    return np.clip(out, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# a18 greedy multi-reference compose
# --------------------------------------------------------------------------------------
def compose(flow_mat: np.ndarray, ai_frames, original_bgr, thres: float, warp_mode: str = "cv2_cubic",
            expand: bool = True):
    """flow_mat f32[N,1,H,W,3] (fx, fy, conf); ai_frames: list of N uint8[H,W,3].
    Returns (ret_frame uint8[H,W,3], mask2 uint8[H,W], order list)."""
    fm = flow_mat.copy()
    fm[..., 2] = (fm[..., 2] > np.float32(thres)).astype(np.float32)
    n, _, h, w, _ = fm.shape
    mask = np.zeros((h, w), np.uint8)
    ret = None
    order = []
    for _ in range(n):
        score = fm[:, :, :, :, 2].sum(axis=(1, 2, 3))
        s = int(np.argmax(score))
        order.append(s)
        warped = warp_oracle.warp_frame(ai_frames[s], fm[s, 0, :, :, 0:2], mode=warp_mode)
        last = fm[s, 0, :, :, 2].copy()
        cur = (last * 255).astype(np.uint8)
        mask = mask | cur
        ret = warped.copy() if ret is None else merge_images(ret, warped, cur)
        fm[:, 0, :, :, 2] -= last[None]
        fm[:, 0, :, :, 2] = np.clip(fm[:, 0, :, :, 2], 0, 1)
    mask2 = 255 - mask
    if expand:
        mask2 = expand_mask(mask2, original_bgr)
    return ret, mask2, order


def keyframe_scores(flow_mat: np.ndarray) -> np.ndarray:
    """einops.reduce(flow_mat[..., 2], 's t h w -> s', 'sum') (ofgen_keyframe_inpaint.py:666)."""
    return flow_mat[:, :, :, :, 2].astype(np.float64).sum(axis=(1, 2, 3))
