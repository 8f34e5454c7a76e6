"""CPU oracle for the backward warp.  TEST INFRASTRUCTURE ONLY (see oracle/raft_oracle.py header).

Restates `warp_frame` / `warp_frame_latent` of the reference:
  pdcnet_of.py:34-42      warp_frame (PDCNet convention: out(y,x) = frame(y+fy, x+fx))
  pdcnet_of.py:19-32      warp_frame_latent
  ofgen_keyframe_inpaint.py:92-111   RAFT-convention twins (flow negated, default border)

The reference calls `cv2.remap(..., INTER_CUBIC, BORDER_CONSTANT)`.  OpenCV is a third-party
dependency that is NOT in /root/reference nor in this image and is not version-pinned by the
reference (no requirements file): **exact-cv2 parity is unpinned**.  Three modes are restated:

  'bilinear'   north_star's primary mode.  Pinned against `torch.nn.functional.grid_sample(
               mode='bilinear', padding_mode='zeros', align_corners=True)` in tests.
  'bicubic'    float Keys cubic, A = -0.75, zero border.  Pinned against grid_sample(mode='bicubic').
  'cv2_cubic'  OpenCV's published remap algorithm for INTER_CUBIC (imgproc/src/imgwarp.cpp,
               4.x): coordinates rounded to 1/32 px (INTER_BITS=5), a 32x32 table of 4x4 weights;
               for uint8 sources the weights are 15-bit fixed point (INTER_REMAP_COEF_BITS), the
               table's rounding residual is folded into one of the 4 central taps, and the result is
               (sum + 2^14) >> 15 saturated; for float sources the same quantised fractions with
               float weights.  Border = constant 0, applied per tap.

uint8 outputs of the float modes are `rint` (round-half-even, = cv::saturate_cast<uchar>) clamped
to [0,255].
"""
from __future__ import annotations

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
COEF_BITS = 15
COEF_SCALE = 1 << COEF_BITS
CUBIC_A = np.float32(-0.75)


# --------------------------------------------------------------------------------------
# shared helpers
# --------------------------------------------------------------------------------------
def _maps(flow: np.ndarray, sign: float = 1.0):
    """pdcnet_of.py:35-40: X,Y from linspace (f64), + displacement, cast to f32."""
    h, w = flow.shape[:2]
    xs, ys = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    mx = (xs + sign * flow[:, :, 0].astype(np.float64)).astype(np.float32)
    my = (ys + sign * flow[:, :, 1].astype(np.float64)).astype(np.float32)
    return mx, my


def _as3(frame: np.ndarray):
    f = frame if frame.ndim == 3 else frame[:, :, None]
    return np.ascontiguousarray(f), frame.ndim == 2


def _finish(acc: np.ndarray, like: np.ndarray, squeeze: bool):
    if like.dtype == np.uint8:
        acc = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    else:
        acc = acc.astype(like.dtype)
    return acc[:, :, 0] if squeeze else acc


def _gather(src: np.ndarray, yi: np.ndarray, xi: np.ndarray) -> np.ndarray:
    """src[yi, xi, :] as float32 with zeros outside."""
    h, w = src.shape[:2]
    ok = (yi >= 0) & (yi < h) & (xi >= 0) & (xi < w)
    v = src[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)].astype(np.float32)
    return np.where(ok[:, :, None], v, np.float32(0))


# --------------------------------------------------------------------------------------
# float modes
# --------------------------------------------------------------------------------------
def warp_bilinear(frame: np.ndarray, mx: np.ndarray, my: np.ndarray) -> np.ndarray:
    src, sq = _as3(frame)
    x0 = np.floor(mx)
    y0 = np.floor(my)
    fx = (mx - x0).astype(np.float32)[:, :, None]
    fy = (my - y0).astype(np.float32)[:, :, None]
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    one = np.float32(1)
    acc = _gather(src, y0, x0) * ((one - fx) * (one - fy))
    acc = acc + _gather(src, y0, x0 + 1) * (fx * (one - fy))
    acc = acc + _gather(src, y0 + 1, x0) * ((one - fx) * fy)
    acc = acc + _gather(src, y0 + 1, x0 + 1) * (fx * fy)
    return _finish(acc, frame, sq)


def cubic_coeffs(t: np.ndarray):
    """Keys cubic convolution weights, A=-0.75, for taps at -1,0,+1,+2 (fraction t in [0,1))."""
    a = CUBIC_A
    t = t.astype(np.float32)
    one = np.float32(1)
    w0 = ((a * (t + one) - np.float32(5) * a) * (t + one) + np.float32(8) * a) * (t + one) - np.float32(4) * a
    w1 = ((a + np.float32(2)) * t - (a + np.float32(3))) * t * t + one
    u = one - t
    w2 = ((a + np.float32(2)) * u - (a + np.float32(3))) * u * u + one
    w3 = one - w0 - w1 - w2
    return w0, w1, w2, w3


def warp_bicubic(frame: np.ndarray, mx: np.ndarray, my: np.ndarray) -> np.ndarray:
    src, sq = _as3(frame)
    x0 = np.floor(mx)
    y0 = np.floor(my)
    wx = cubic_coeffs(mx - x0)
    wy = cubic_coeffs(my - y0)
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    acc = np.zeros(src.shape, np.float32)
    for k1 in range(4):
        row = np.zeros(src.shape, np.float32)
        for k2 in range(4):
            row = row + _gather(src, y0 - 1 + k1, x0 - 1 + k2) * wx[k2][:, :, None]
        acc = acc + row * wy[k1][:, :, None]
    return _finish(acc, frame, sq)


# --------------------------------------------------------------------------------------
# OpenCV fixed-point cubic remap
# --------------------------------------------------------------------------------------
_TAB_CACHE = {}


def cv2_cubic_tables():
    """(tab_f32[1024,16], tab_i16[1024,16]); row index = (fy_idx*32 + fx_idx), column = k1*4+k2
    (k1 = y tap, k2 = x tap).  Restates initInterTab1D/initInterTab2D for INTER_CUBIC."""
    if "t" in _TAB_CACHE:
        return _TAB_CACHE["t"]
    frac = (np.arange(INTER_TAB_SIZE, dtype=np.float32) * np.float32(1.0 / INTER_TAB_SIZE)).astype(np.float32)
    c = np.stack(cubic_coeffs(frac), axis=1).astype(np.float32)        # [32,4]
    tabf = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 16), np.float32)
    tabi = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 16), np.int32)
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            v = (c[i][:, None] * c[j][None, :]).astype(np.float32).reshape(16)
            tabf[i * INTER_TAB_SIZE + j] = v
            # saturate_cast<short>(v*SCALE): round-half-even then clamp to int16
            it = np.clip(np.rint(v * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
            isum = int(it.sum())
            if isum != COEF_SCALE:
                diff = isum - COEF_SCALE
                # search only the central 2x2 taps (k1,k2 in {2,3}) exactly as OpenCV does
                mk = Mk = 2 * 4 + 2
                for k1 in (2, 3):
                    for k2 in (2, 3):
                        k = k1 * 4 + k2
                        if it[k] < it[mk]:
                            mk = k
                        elif it[k] > it[Mk]:
                            Mk = k
                if diff < 0:
                    it[Mk] = np.int16(it[Mk] - diff)
                else:
                    it[mk] = np.int16(it[mk] - diff)
            tabi[i * INTER_TAB_SIZE + j] = it
    _TAB_CACHE["t"] = (tabf, tabi.astype(np.int16))
    return _TAB_CACHE["t"]


def _fixed_coords(mx: np.ndarray, my: np.ndarray):
    """sx = cvRound(map*32); integer part = sx>>5 saturated to int16, fraction = sx&31."""
    sx = np.rint(mx.astype(np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    sy = np.rint(my.astype(np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)
    tab = (sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1))
    return ix, iy, tab


def warp_cv2_cubic(frame: np.ndarray, mx: np.ndarray, my: np.ndarray) -> np.ndarray:
    src, sq = _as3(frame)
    h, w = src.shape[:2]
    tabf, tabi = cv2_cubic_tables()
    ix, iy, tab = _fixed_coords(mx, my)
    x0 = ix - 1
    y0 = iy - 1
    if src.dtype == np.uint8:
        acc = np.zeros(src.shape, np.int64)
        wt = tabi[tab].astype(np.int64)                         # [H,W,16]
        for k1 in range(4):
            for k2 in range(4):
                yi = y0 + k1
                xi = x0 + k2
                ok = (yi >= 0) & (yi < h) & (xi >= 0) & (xi < w)
                v = src[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)].astype(np.int64)
                v = np.where(ok[:, :, None], v, 0)
                acc += v * wt[:, :, k1 * 4 + k2][:, :, None]
        out = np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)
    else:
        acc = np.zeros(src.shape, np.float32)
        wt = tabf[tab]
        for k1 in range(4):
            for k2 in range(4):
                v = _gather(src, y0 + k1, x0 + k2)
                # separate multiply and add (no fused multiply-add), tap order k1-major
                acc = (acc + (v * wt[:, :, k1 * 4 + k2][:, :, None]).astype(np.float32)).astype(np.float32)
        out = acc.astype(src.dtype)
    return out[:, :, 0] if sq else out


_MODES = {"bilinear": warp_bilinear, "bicubic": warp_bicubic, "cv2_cubic": warp_cv2_cubic}


def warp_frame(frame: np.ndarray, flow: np.ndarray, mode: str = "bilinear", convention: str = "pdcnet"):
    """convention 'pdcnet': sample at p + flow (pdcnet_of.py:34-42);
    convention 'raft': sample at p - flow (ofgen_keyframe_inpaint.py:92-98)."""
    mx, my = _maps(flow, 1.0 if convention == "pdcnet" else -1.0)
    return _MODES[mode](frame, mx, my)


# --------------------------------------------------------------------------------------
# cubic resize (cv2.resize INTER_CUBIC, float images) for warp_frame_latent
# --------------------------------------------------------------------------------------
def resize_cubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(..., INTER_CUBIC) for float32 HxWxC images: source coordinate
    sx = (dx+0.5)*scale-0.5, 4 taps with Keys A=-0.75, border taps clamped (replicated)."""
    h, w = img.shape[:2]
    src = img.astype(np.float32)

    def axis_tables(n_out, n_in):
        scale = np.float64(n_in) / np.float64(n_out)
        d = np.arange(n_out, dtype=np.float64)
        f = (d + 0.5) * scale - 0.5
        s = np.floor(f)
        t = (f - s).astype(np.float32)
        s = s.astype(np.int64)
        ws = np.stack(cubic_coeffs(t), axis=1).astype(np.float32)       # [n_out,4]
        idx = np.clip(s[:, None] - 1 + np.arange(4)[None, :], 0, n_in - 1)
        return idx, ws

    ix, wx = axis_tables(out_w, w)
    iy, wy = axis_tables(out_h, h)
    # horizontal then vertical (OpenCV's resize is separable: rows first via hresize, then vresize)
    tmp = np.zeros((h, out_w, src.shape[2]), np.float32)
    for k in range(4):
        tmp += src[:, ix[:, k], :] * wx[:, k][None, :, None]
    out = np.zeros((out_h, out_w, src.shape[2]), np.float32)
    for k in range(4):
        out += tmp[iy[:, k], :, :] * wy[:, k][:, None, None]
    return out


def warp_frame_latent(latent: np.ndarray, flow: np.ndarray, mode: str = "cv2_cubic",
                      convention: str = "pdcnet") -> np.ndarray:
    """pdcnet_of.py:19-32.  latent [1,C,lh,lw] float32 -> [1,C,lh,lw]."""
    lat = np.transpose(latent[0], (1, 2, 0))
    lh, lw = lat.shape[:2]
    h, w = flow.shape[:2]
    up = resize_cubic(lat, h, w)
    wp = warp_frame(up, flow, mode=mode, convention=convention)
    dn = resize_cubic(wp, lh, lw)
    return np.transpose(dn, (2, 0, 1))[None]
