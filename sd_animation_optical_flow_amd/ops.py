"""Device operators: thin torch-tensor wrappers over the C ABI of libofx.so.

PyTorch is plumbing here (device memory + the current HIP stream); every byte of arithmetic happens
in the hand-written HIP kernels.  Preconditions mirror the reference's extension
(`RAFT/alt_cuda_corr/correlation.cpp:19-21`): tensors must be on the GPU and contiguous, otherwise a
RuntimeError is raised -- nothing silently falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, check

WARP_MODES = {"bilinear": 0, "bicubic": 1, "cv2_cubic": 2}
ACTS = {None: 0, "none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}
EPI_PLAIN, EPI_GRU_ZR, EPI_GRU_Q, EPI_FLOW = 0, 1, 2, 3


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")       # correlation.cpp:19
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")          # correlation.cpp:20
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# --------------------------------------------------------------------------------------
# profiling
# --------------------------------------------------------------------------------------
def prof_enable(on) -> None:
    """False/0: off; True/1: per kernel family; 2: per layer of the RAFT executor ("family:layer")."""
    check(_lib.lib().ofx_prof_enable(int(on)), "ofx_prof_enable")


def prof_collect() -> dict:
    buf = C.create_string_buffer(1 << 18)
    check(_lib.lib().ofx_prof_collect(buf, len(buf)), "ofx_prof_collect")
    return json.loads(buf.value.decode())


# --------------------------------------------------------------------------------------
# warp
# --------------------------------------------------------------------------------------
def warp(frame: torch.Tensor, flow: torch.Tensor, mode: str = "bilinear", sign: float = 1.0) -> torch.Tensor:
    """frame [H,W,C] (one frame shared by all flows) or [B,H,W,C], uint8 or float32;
    flow f32 [B,H,W,2] or [H,W,2].  Returns the warped frames, [B,H,W,C] (or [H,W,C])."""
    squeeze = flow.dim() == 3
    fl = _chk(flow if not squeeze else flow[None], "flow", torch.float32)
    B, H, W, _ = fl.shape
    if frame.dim() == 2:
        raise RuntimeError("frame must be HWC; add a channel axis")
    shared = frame.dim() == 3
    fr = _chk(frame, "frame")
    Cn = fr.shape[-1]
    if tuple(fr.shape[-3:-1]) != (H, W) or (not shared and fr.shape[0] != B):
        raise RuntimeError(f"frame {tuple(fr.shape)} does not match flow {tuple(fl.shape)}")
    out = torch.empty((B, H, W, Cn), dtype=fr.dtype, device=fr.device)
    stride = 0 if shared else H * W * Cn
    L = _lib.lib()
    if fr.dtype == torch.uint8:
        fn, nm = L.ofx_warp_u8, "ofx_warp_u8"
    elif fr.dtype == torch.float32:
        fn, nm = L.ofx_warp_f32, "ofx_warp_f32"
    else:
        raise RuntimeError(f"warp: unsupported dtype {fr.dtype}")
    check(fn(_ptr(fr), stride, _ptr(fl), _ptr(out), B, H, W, Cn, WARP_MODES[mode], float(sign), _stream()), nm)
    return out[0] if (squeeze and shared) else out


def resize_cubic(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """f32 [B,H,W,C] -> [B,out_h,out_w,C], cv2.resize(INTER_CUBIC) semantics."""
    x = _chk(img, "img", torch.float32)
    B, H, W, Cn = x.shape
    out = torch.empty((B, out_h, out_w, Cn), dtype=torch.float32, device=x.device)
    check(_lib.lib().ofx_resize_cubic_f32(_ptr(x), _ptr(out), B, H, W, out_h, out_w, Cn, _stream()), "ofx_resize_cubic_f32")
    return out


# --------------------------------------------------------------------------------------
# masks
# --------------------------------------------------------------------------------------
def generate_mask(conf: torch.Tensor, log_conf: Optional[torch.Tensor] = None, thres: float = 0.8,
                  ksize: int = 7, cmp_gt: bool = False) -> torch.Tensor:
    """conf f32 [B,H,W]; log_conf (optional, modified IN PLACE like the reference). -> uint8 [B,H,W]."""
    c = _chk(conf, "confidence", torch.float32)
    B, H, W = c.shape
    if log_conf is not None:
        _chk(log_conf, "log_confidence", torch.float32)
    out = torch.empty((B, H, W), dtype=torch.uint8, device=c.device)
    check(_lib.lib().ofx_generate_mask(_ptr(c), _ptr(log_conf), _ptr(out), B, H, W, float(thres), int(ksize),
                                       1 if cmp_gt else 0, _stream()), "ofx_generate_mask")
    return out


def dilate(mask: torch.Tensor, ksize: int) -> torch.Tensor:
    m = _chk(mask, "mask", torch.uint8)
    B, H, W = m.shape
    out = torch.empty_like(m)
    check(_lib.lib().ofx_dilate_u8(_ptr(m), _ptr(out), B, H, W, int(ksize), _stream()), "ofx_dilate_u8")
    return out


def expand_mask(mask: torch.Tensor, image_bgr: torch.Tensor, edge_thres: int = 20, ksize: int = 7) -> torch.Tensor:
    m = _chk(mask, "mask", torch.uint8)
    img = _chk(image_bgr, "image", torch.uint8)
    B, H, W = m.shape
    if tuple(img.shape) != (B, H, W, 3):
        raise RuntimeError("image must be uint8 [B,H,W,3]")
    out = torch.empty_like(m)
    check(_lib.lib().ofx_expand_mask(_ptr(m), _ptr(img), _ptr(out), None, B, H, W, int(edge_thres), int(ksize), _stream()),
          "ofx_expand_mask")
    return out


def travel_distance(flow: torch.Tensor, conf: torch.Tensor, conf_floor: float = 0.9) -> torch.Tensor:
    fl = _chk(flow, "flow", torch.float32)
    c = _chk(conf, "confidence", torch.float32)
    B, H, W, _ = fl.shape
    out = torch.empty((B, H, W), dtype=torch.float32, device=fl.device)
    check(_lib.lib().ofx_travel_distance(_ptr(fl), _ptr(c), _ptr(out), B, H, W, float(conf_floor), _stream()),
          "ofx_travel_distance")
    return out


def flow_magnitude(flow: torch.Tensor) -> torch.Tensor:
    """|flow| per pixel: f32 [...,2] -> f32 [...] (the RAFT-variant of_calc's `v`, reference ofgen.py:45-49)."""
    fl = _chk(flow, "flow", torch.float32)
    if fl.shape[-1] != 2:
        raise RuntimeError("flow must be [...,2]")
    out = torch.empty(tuple(fl.shape[:-1]), dtype=torch.float32, device=fl.device)
    check(_lib.lib().ofx_flow_magnitude(_ptr(fl), _ptr(out), out.numel(), _stream()), "ofx_flow_magnitude")
    return out


def travel_mask(conf, flow, dist, travel, thres: float, warp_mode: str = "cv2_cubic") -> Tuple[torch.Tensor, torch.Tensor]:
    """confidence_to_mask core (before the 15x15 dilation): returns (raw mask, new travel)."""
    c = _chk(conf, "confidence", torch.float32)
    fl = _chk(flow, "flow", torch.float32)
    d = _chk(dist, "dist", torch.float32)
    t = _chk(travel, "travel", torch.float32)
    B, H, W = c.shape
    tout = torch.empty_like(t)
    raw = torch.empty((B, H, W), dtype=torch.uint8, device=c.device)
    check(_lib.lib().ofx_travel_mask(_ptr(c), _ptr(fl), _ptr(d), _ptr(t), _ptr(tout), _ptr(raw), B, H, W, float(thres),
                                     WARP_MODES[warp_mode], _stream()), "ofx_travel_mask")
    return raw, tout


def merge_images(base, second, mask) -> torch.Tensor:
    b = _chk(base, "base", torch.uint8)
    s = _chk(second, "second", torch.uint8)
    m = _chk(mask, "mask", torch.uint8)
    B, H, W, Cn = b.shape
    out = torch.empty_like(b)
    check(_lib.lib().ofx_merge_images(_ptr(b), _ptr(s), _ptr(m), _ptr(out), B, H, W, Cn, _stream()), "ofx_merge_images")
    return out


def mix_frames(raw, warped, mask, ppw: float) -> torch.Tensor:
    r = _chk(raw, "raw", torch.uint8)
    w = _chk(warped, "warped", torch.uint8)
    m = _chk(mask, "mask", torch.uint8)
    B, H, W, Cn = r.shape
    out = torch.empty_like(r)
    check(_lib.lib().ofx_mix_frames(_ptr(r), _ptr(w), _ptr(m), _ptr(out), B, H, W, Cn, float(ppw), _stream()), "ofx_mix_frames")
    return out


def conf_sum(x: torch.Tensor, chan: int) -> torch.Tensor:
    """x f32 [N,H,W,nchan] -> f64 [N] sums of channel `chan` over H,W."""
    t = _chk(x, "x", torch.float32)
    N, H, W, nc = t.shape
    out = torch.empty((N,), dtype=torch.float64, device=t.device)
    check(_lib.lib().ofx_conf_sum(_ptr(t), _ptr(out), N, H * W, nc, int(chan), _stream()), "ofx_conf_sum")
    return out


def fb_confidence(flow_fw: torch.Tensor, flow_bw: torch.Tensor, sigma: float = 3.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Extension: forward-backward consistency (confidence, log_confidence), each f32 [B,H,W]."""
    a = _chk(flow_fw, "flow_fw", torch.float32)
    b = _chk(flow_bw, "flow_bw", torch.float32)
    B, H, W, _ = a.shape
    conf = torch.empty((B, H, W), dtype=torch.float32, device=a.device)
    logc = torch.empty_like(conf)
    check(_lib.lib().ofx_fb_confidence(_ptr(a), _ptr(b), _ptr(conf), _ptr(logc), B, H, W, float(sigma), _stream()),
          "ofx_fb_confidence")
    return conf, logc


def warp_and_mask(frame, flow, conf, warp_mode="bilinear", sign=1.0, thres=0.95, ksize=7, cmp_gt=False):
    """Fused tail of the hot path for a batch: (warped uint8 [B,H,W,C], mask uint8 [B,H,W])."""
    fl = _chk(flow, "flow", torch.float32)
    c = _chk(conf, "confidence", torch.float32)
    fr = _chk(frame, "frame", torch.uint8)
    B, H, W, _ = fl.shape
    shared = fr.dim() == 3
    Cn = fr.shape[-1]
    warped = torch.empty((B, H, W, Cn), dtype=torch.uint8, device=fr.device)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=fr.device)
    check(_lib.lib().ofx_warp_and_mask(_ptr(fr), 0 if shared else H * W * Cn, _ptr(fl), _ptr(c), _ptr(warped), _ptr(mask),
                                       B, H, W, Cn, WARP_MODES[warp_mode], float(sign), float(thres), int(ksize),
                                       1 if cmp_gt else 0, _stream()), "ofx_warp_and_mask")
    return warped, mask


# --------------------------------------------------------------------------------------
# network building blocks (exposed for stage-level parity tests and custom pipelines)
# --------------------------------------------------------------------------------------
def pack_conv_weight(w_oihw: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    """OIHW fp32 (CPU) -> packed [Cout, Kpad] fp32 (CPU)."""
    w = w_oihw.detach().to(torch.float32).contiguous().cpu()
    co, ci, kh, kw = w.shape
    cp = cin_pad if cin_pad else ((ci + 3) // 4) * 4
    L = _lib.lib()
    kpad = L.ofx_pack_conv_weight(None, co, ci, kh, kw, cp, None)
    if kpad < 0:
        raise _lib.OfxError(int(kpad), "ofx_pack_conv_weight")
    out = torch.empty((co, kpad), dtype=torch.float32)
    L.ofx_pack_conv_weight(C.c_void_p(w.data_ptr()), co, ci, kh, kw, cp, C.c_void_p(out.data_ptr()))
    return out


def split_conv_weight(w_packed: torch.Tensor) -> torch.Tensor:
    """Packed fp32 weights (CPU) -> the pre-split bf16x3 operand format (same shape, fp32 container); feed it to
    conv2d_nhwc(..., precision="bf16x3_w")."""
    w = w_packed.detach().to(torch.float32).contiguous().cpu()
    out = torch.empty_like(w)
    check(_lib.lib().ofx_split_conv_weight(C.c_void_p(w.data_ptr()), w.numel(), C.c_void_p(out.data_ptr())), "ofx_split_conv_weight")
    return out


def split_conv_weight3(w_packed: torch.Tensor) -> torch.Tensor:
    """Packed fp32 weights (CPU, the whole [Cout, Kpad] matrix) -> the pre-split bf16x6 operand format: a flat fp32 container of
    1.5 x the size ([hi x4 | mid x4] groups, then the [lo x4] groups); feed it to conv2d_nhwc(..., precision="bf16x6_w")."""
    w = w_packed.detach().to(torch.float32).contiguous().cpu()
    out = torch.empty((w.numel() * 3 // 2,), dtype=torch.float32)
    check(_lib.lib().ofx_split_conv_weight3(C.c_void_p(w.data_ptr()), w.numel(), C.c_void_p(out.data_ptr())), "ofx_split_conv_weight3")
    return out


def conv2d_nhwc(x: torch.Tensor, w_packed: torch.Tensor, kh: int, kw: int, cout: int, *, stride: int = 1,
                shift: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None, act: Optional[str] = None,
                x2: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                nmean: Optional[torch.Tensor] = None, nrstd: Optional[torch.Tensor] = None, tile: int = 0,
                precision: str = "fp32", splitk_ws: Optional[torch.Tensor] = None, pad: Optional[Tuple[int, int]] = None,
                out_hw: Optional[Tuple[int, int]] = None, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Plain-epilogue convolution: x [B,H,W,C0] (+ optional second channel segment x2 [B,H,W,C1]),
    'same' padding (k//2) unless `pad` = (top, left) is given; `out_hw` overrides the output size (taps beyond the
    input read zeros: pad (0, 0) with out_hw = (H/2, W/2) is the VAE's F.pad(x, (0,1,0,1)) + stride-2 convolution).
    `addend` [B,Hout,Wout,cout] is added before the activation (`res` adds after it and applies ReLU).
    Returns [B,Hout,Wout,cout]."""
    x = _chk(x, "x", torch.float32)
    B, H, W, c0 = x.shape
    d = ConvDesc()
    d.in0, d.ld0, d.c0 = x.data_ptr(), c0, c0
    if x2 is not None:
        x2 = _chk(x2, "x2", torch.float32)
        d.in1, d.ld1, d.c1 = x2.data_ptr(), x2.shape[-1], x2.shape[-1]
    wp = _chk(w_packed, "w_packed", torch.float32)
    d.w = wp.data_ptr()
    d.scale = 0 if scale is None else _chk(scale, "scale", torch.float32).data_ptr()
    d.shift = 0 if shift is None else _chk(shift, "shift", torch.float32).data_ptr()
    ph, pw = (kh // 2, kw // 2) if pad is None else pad
    Ho, Wo = ((H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1) if out_hw is None else out_hw
    out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    d.out, d.ldo = out.data_ptr(), cout
    if addend is not None:
        addend = _chk(addend, "addend", torch.float32)
        if tuple(addend.shape) != (B, Ho, Wo, cout):
            raise RuntimeError(f"addend must be {(B, Ho, Wo, cout)}, got {tuple(addend.shape)}")
        d.addend, d.ldadd = addend.data_ptr(), cout
    if res is not None:
        res = _chk(res, "res", torch.float32)
        d.res, d.ldres = res.data_ptr(), res.shape[-1]
    if nmean is not None:
        d.nmean = _chk(nmean, "nmean", torch.float32).data_ptr()
        d.nrstd = _chk(nrstd, "nrstd", torch.float32).data_ptr()
    d.B, d.Hin, d.Win, d.Hout, d.Wout, d.Cout = B, H, W, Ho, Wo, cout
    d.KH, d.KW, d.stride, d.padH, d.padW = kh, kw, stride, ph, pw
    d.act, d.epi, d.tile = ACTS[act], EPI_PLAIN, tile
    d.precision = {"fp32": 0, "bf16x3": 1, "bf16x3_w": 2, "bf16x6": 3, "bf16x6_w": 4}[precision]   # *_w: weight from split_conv_weight(3)
    if splitk_ws is not None:       # uint8 scratch whose first 64 KiB are zero (see ofx_conv_desc.splitk_ws): allows split-K
        ws = _chk(splitk_ws, "splitk_ws", torch.uint8)
        d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
    check(_lib.lib().ofx_conv2d(C.byref(d), _stream()), "ofx_conv2d")
    return out


def conv2d_desc(d: ConvDesc) -> None:
    check(_lib.lib().ofx_conv2d(C.byref(d), _stream()), "ofx_conv2d")


def inorm_stats(x: torch.Tensor, eps: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    x = _chk(x, "x", torch.float32)
    B, H, W, Cn = x.shape
    mean = torch.empty((B, Cn), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    scratch = torch.empty((max(B * 64, min(B, 7) * 256) * Cn * 2,), dtype=torch.float64, device=x.device)
    check(_lib.lib().ofx_inorm_stats(_ptr(x), Cn, _ptr(mean), _ptr(rstd), _ptr(scratch), B, H * W, Cn, float(eps), _stream()),
          "ofx_inorm_stats")
    return mean, rstd


def inorm_apply(x, mean, rstd, res=None, res_mean=None, res_rstd=None, relu=True) -> torch.Tensor:
    x = _chk(x, "x", torch.float32)
    B, H, W, Cn = x.shape
    out = torch.empty_like(x)
    check(_lib.lib().ofx_inorm_apply(_ptr(x), _ptr(mean), _ptr(rstd), _ptr(res), _ptr(res_mean), _ptr(res_rstd), _ptr(out),
                                     B, H * W, Cn, 1 if relu else 0, _stream()), "ofx_inorm_apply")
    return out


def preprocess_u8(img: torch.Tensor, bgr: bool = False) -> torch.Tensor:
    x = _chk(img, "img", torch.uint8)
    if x.shape[-1] != 3:
        raise RuntimeError("img must be [...,3]")
    out = torch.empty(tuple(x.shape[:-1]) + (4,), dtype=torch.float32, device=x.device)
    check(_lib.lib().ofx_preprocess_u8(_ptr(x), _ptr(out), x.numel() // 3, 1 if bgr else 0, _stream()), "ofx_preprocess_u8")
    return out


def corr_slice_floats(hl: int, wl: int) -> int:
    """Floats per pixel slice of an hl x wl pyramid level in the blocked layout (4 x 8 blocks, see include/ofx.h)."""
    return ((hl + 3) // 4) * ((wl + 7) // 8) * 32


def corr_volume(f1: torch.Tensor, f2: torch.Tensor, levels: int = 4) -> List[torch.Tensor]:
    """f1, f2 f32 [B,h,w,D] (NHWC) -> pyramid list, level l: [B*h*w, corr_slice_floats(h>>l, w>>l)] in the blocked
    layout `corr_lookup` reads; `corr_unblock` gives the [B*h*w, h>>l, w>>l] view of CorrBlock.corr_pyramid."""
    a = _chk(f1, "fmap1", torch.float32)
    b = _chk(f2, "fmap2", torch.float32)
    B, h, w, D = a.shape
    pyr = [torch.empty((B * h * w, corr_slice_floats(h >> l, w >> l)), dtype=torch.float32, device=a.device) for l in range(levels)]
    arr = (C.c_void_p * levels)(*[p.data_ptr() for p in pyr])
    check(_lib.lib().ofx_corr_volume(_ptr(a), _ptr(b), arr, B, h, w, D, levels, _stream()), "ofx_corr_volume")
    return pyr


_VOLUME_PLANES = {"fp32": 1, "bf16x3": 2, "bf16x6": 3}


def corr_volume_split(f1: torch.Tensor, f2: torch.Tensor, levels: int = 4, precision: str = "bf16x6") -> List[torch.Tensor]:
    """`corr_volume` with the GEMM in split-bf16 arithmetic (`ofx_corr_volume_split`; opt-in, RAFT/core/corr.py:52-60 is fp32).
    f1 f32 [B,h,w,256]; f2 f32 [B,h,w,256] or [1,h,w,256] (one key frame shared by the B pairs).  precision 'bf16x3' | 'bf16x6', or
    'fp32': the same A-stationary kernel on the fp32 matrix cores -- bit-identical to `corr_volume`."""
    a = _chk(f1, "fmap1", torch.float32)
    b = _chk(f2, "fmap2", torch.float32)
    if precision not in _VOLUME_PLANES:
        raise ValueError("precision: 'fp32', 'bf16x3' or 'bf16x6'")
    B, h, w, D = a.shape
    if b.shape[0] not in (1, B) or tuple(b.shape[1:]) != (h, w, D):
        raise RuntimeError("fmap2: [B,h,w,D] or [1,h,w,D]")
    pyr = [torch.empty((B * h * w, corr_slice_floats(h >> l, w >> l)), dtype=torch.float32, device=a.device) for l in range(levels)]
    arr = (C.c_void_p * levels)(*[p.data_ptr() for p in pyr])
    check(_lib.lib().ofx_corr_volume_split(_ptr(a), _ptr(b), arr, B, h, w, D, levels, _VOLUME_PLANES[precision],
                                           1 if (b.shape[0] == 1 and B > 1) else 0, _stream()), "ofx_corr_volume_split")
    return pyr


def corr_unblock(p: torch.Tensor, hl: int, wl: int) -> torch.Tensor:
    """Blocked slices [M, corr_slice_floats(hl, wl)] (or a flat buffer) -> row-major [M, hl, wl] (pure indexing: for
    tests and inspection, not on any hot path)."""
    hb, wb = (hl + 3) // 4, (wl + 7) // 8
    q = p.reshape(-1, hb, wb, 4, 8).permute(0, 1, 3, 2, 4).reshape(-1, hb * 4, wb * 8)
    return q[:, :hl, :wl].contiguous()


def upsample_flow_warp(coords1: torch.Tensor, mask: torch.Tensor, frame: torch.Tensor, sign: float = 1.0, want_flow: bool = True):
    """RAFT.upsample_flow (raft.py:72-83) + the bilinear backward warp of one shared uint8 frame [8h,8w,3] in ONE kernel.
    coords1 f32 [B,h,w,2], mask f32 [B,h,w,576] -> (flow_up f32 [B,8h,8w,2] or None, warped u8 [B,8h,8w,3])."""
    c = _chk(coords1, "coords1", torch.float32)
    m = _chk(mask, "mask", torch.float32)
    fr = _chk(frame, "frame", torch.uint8)
    B, h, w, _ = c.shape
    if tuple(fr.shape) != (8 * h, 8 * w, 3) or tuple(m.shape) != (B, h, w, 576):
        raise RuntimeError("shapes: coords1 [B,h,w,2], mask [B,h,w,576], frame [8h,8w,3]")
    flow = torch.empty((B, 8 * h, 8 * w, 2), dtype=torch.float32, device=c.device) if want_flow else None
    warped = torch.empty((B, 8 * h, 8 * w, 3), dtype=torch.uint8, device=c.device)
    check(_lib.lib().ofx_upsample_flow_warp(_ptr(c), _ptr(m), _ptr(flow) if want_flow else C.c_void_p(0), _ptr(fr), _ptr(warped), B, h, w,
                                            float(sign), _stream()), "ofx_upsample_flow_warp")
    return flow, warped


def corr_lookup(pyr: Sequence[torch.Tensor], coords: torch.Tensor, B: int, h: int, w: int, radius: int = 4) -> torch.Tensor:
    """coords f32 [B,h,w,2] (x,y) -> [B,h,w,L*(2r+1)^2] (channels-last version of CorrBlock.__call__)."""
    c = _chk(coords, "coords", torch.float32)
    for i, p in enumerate(pyr):
        _chk(p, f"pyr[{i}]", torch.float32)
    levels = len(pyr)
    nch = levels * (2 * radius + 1) ** 2
    out = torch.empty((B, h, w, nch), dtype=torch.float32, device=c.device)
    arr = (C.c_void_p * levels)(*[p.data_ptr() for p in pyr])
    check(_lib.lib().ofx_corr_lookup(arr, _ptr(c), _ptr(out), nch, B, h, w, levels, radius, _stream()), "ofx_corr_lookup")
    return out


def local_corr(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, radius: int) -> torch.Tensor:
    """`alt_cuda_corr.forward` semantics; see alt_cuda_corr.py."""
    a = _chk(fmap1, "fmap1", torch.float32)
    b = _chk(fmap2, "fmap2", torch.float32)
    c = _chk(coords, "coords", torch.float32)
    B, H1, W1, Cn = a.shape
    _, H2, W2, _ = b.shape
    N = c.shape[1]
    rd = 2 * radius + 1
    out = torch.empty((B, N, rd * rd, H1, W1), dtype=torch.float32, device=a.device)
    check(_lib.lib().ofx_local_corr_fwd(_ptr(a), _ptr(b), _ptr(c), _ptr(out), B, H1, W1, H2, W2, Cn, N, radius, _stream()),
          "ofx_local_corr_fwd")
    return out


def local_corr_backward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, corr_grad: torch.Tensor, radius: int):
    """`alt_cuda_corr.backward` semantics: (fmap1_grad, fmap2_grad) for corr_grad f32[B,N,(2r+1)^2,H1,W1]."""
    a = _chk(fmap1, "fmap1", torch.float32)
    b = _chk(fmap2, "fmap2", torch.float32)
    c = _chk(coords, "coords", torch.float32)
    g = _chk(corr_grad, "corr_grad", torch.float32)
    B, H1, W1, Cn = a.shape
    _, H2, W2, _ = b.shape
    N = c.shape[1]
    rd = 2 * radius + 1
    if tuple(g.shape) != (B, N, rd * rd, H1, W1):
        raise RuntimeError(f"corr_grad must be {(B, N, rd * rd, H1, W1)}, got {tuple(g.shape)}")
    g1 = torch.empty_like(a)
    g2 = torch.empty_like(b)
    check(_lib.lib().ofx_local_corr_bwd(_ptr(a), _ptr(b), _ptr(c), _ptr(g), _ptr(g1), _ptr(g2), B, H1, W1, H2, W2, Cn, N, radius,
                                        _stream()), "ofx_local_corr_bwd")
    return g1, g2


def avgpool2_nhwc(x: torch.Tensor) -> torch.Tensor:
    x = _chk(x, "x", torch.float32)
    B, H, W, Cn = x.shape
    out = torch.empty((B, H // 2, W // 2, Cn), dtype=torch.float32, device=x.device)
    check(_lib.lib().ofx_avgpool2_nhwc(_ptr(x), _ptr(out), B, H, W, Cn, _stream()), "ofx_avgpool2_nhwc")
    return out


def upsample_flow(coords1: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """coords1 f32 [B,h,w,2], mask f32 [B,h,w,576] -> flow_up f32 [B,8h,8w,2]."""
    c = _chk(coords1, "coords1", torch.float32)
    m = _chk(mask, "mask", torch.float32)
    B, h, w, _ = c.shape
    out = torch.empty((B, 8 * h, 8 * w, 2), dtype=torch.float32, device=c.device)
    check(_lib.lib().ofx_upsample_flow(_ptr(c), _ptr(m), _ptr(out), B, h, w, _stream()), "ofx_upsample_flow")
    return out


# --------------------------------------------------------------------------------------
# SD-inpaint hand-off: Pillow-exact mask blur / resize, composite + conditioning tensors
# --------------------------------------------------------------------------------------
def gaussian_blur_u8(mask: torch.Tensor, radius: float) -> torch.Tensor:
    """uint8 [B,H,W] -> PIL.ImageFilter.GaussianBlur(radius) of every [H,W] plane."""
    m = _chk(mask, "mask", torch.uint8)
    B, H, W = m.shape
    out = torch.empty_like(m)
    scratch = torch.empty_like(m)
    check(_lib.lib().ofx_gaussian_blur_u8(_ptr(m), _ptr(out), _ptr(scratch), B, H, W, float(radius), _stream()), "ofx_gaussian_blur_u8")
    return out


def resize_bicubic_u8(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """uint8 [B,H,W] -> PIL Image.resize((out_w, out_h)) (default BICUBIC resample) of every plane."""
    m = _chk(img, "img", torch.uint8)
    B, H, W = m.shape
    out = torch.empty((B, out_h, out_w), dtype=torch.uint8, device=m.device)
    scratch = torch.empty((B, H, out_w), dtype=torch.uint8, device=m.device)
    check(_lib.lib().ofx_resize_bicubic_u8(_ptr(m), _ptr(out), _ptr(scratch), B, H, W, int(out_h), int(out_w), _stream()),
          "ofx_resize_bicubic_u8")
    return out


def sd_handoff(image_bgr: torch.Tensor, reference_bgr: torch.Tensor, image_mask: torch.Tensor, mask_latent: torch.Tensor):
    """See ofx_sd_handoff in include/ofx.h.  Returns (image, cond_image, cond_mask, latmask, cond_mask_latent)."""
    a = _chk(image_bgr, "image_bgr", torch.uint8)
    r = _chk(reference_bgr, "reference_bgr", torch.uint8)
    m = _chk(image_mask, "image_mask", torch.uint8)
    ml = _chk(mask_latent, "mask_latent", torch.uint8)
    B, H, W, c3 = a.shape
    if c3 != 3 or tuple(r.shape) != tuple(a.shape) or tuple(m.shape) != (B, H, W) or ml.dim() != 3 or ml.shape[0] != B:
        raise RuntimeError("sd_handoff: shapes must be image/reference [B,H,W,3], image_mask [B,H,W], mask_latent [B,h,w]")
    h, w = ml.shape[1:]
    dev = a.device
    image = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    cond_image = torch.empty_like(image)
    cond_mask = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    latmask = torch.empty((B, 4, h, w), dtype=torch.float32, device=dev)
    cml = torch.empty((B, h, w), dtype=torch.float32, device=dev)
    check(_lib.lib().ofx_sd_handoff(_ptr(a), _ptr(r), _ptr(m), _ptr(ml), _ptr(image), _ptr(cond_image), _ptr(cond_mask), _ptr(latmask),
                                    _ptr(cml), B, H, W, h, w, _stream()), "ofx_sd_handoff")
    return image, cond_image, cond_mask, latmask, cml


def groupnorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], groups: int = 32, eps: float = 1e-6,
              silu: bool = False) -> torch.Tensor:
    """GroupNorm(groups, C, eps, affine) of an NHWC tensor [B,H,W,C] (+ x * sigmoid(x) when silu): `Normalize` /
    `nonlinearity` of ldm/modules/diffusionmodules/model.py:35-41."""
    t = _chk(x, "x", torch.float32)
    B, H, W, Cn = t.shape
    L = _lib.lib()
    need = L.ofx_groupnorm_scratch_bytes(B, Cn)
    scratch = torch.empty((need,), dtype=torch.uint8, device=t.device)
    out = torch.empty_like(t)
    g = None if gamma is None else _chk(gamma, "gamma", torch.float32)
    b = None if beta is None else _chk(beta, "beta", torch.float32)
    check(L.ofx_groupnorm(_ptr(t), _ptr(g), _ptr(b), _ptr(out), _ptr(scratch), need, B, H * W, Cn, int(groups), float(eps),
                          1 if silu else 0, _stream()), "ofx_groupnorm")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: Optional[torch.Tensor] = None,
              scale: Optional[float] = None, max_workspace_bytes: int = 8 << 30) -> torch.Tensor:
    """softmax(q k^T * scale + bias) v for fp32 [BH,Nq,D] / [BH,Nk,D] tensors; bias [Nq,Nk] (shared) or [BH,Nq,Nk].
    scale defaults to D^-0.5.  Batch-heads are processed in slices that keep the score matrix under
    `max_workspace_bytes`."""
    q = _chk(q, "q", torch.float32)
    k = _chk(k, "k", torch.float32)
    v = _chk(v, "v", torch.float32)
    BH, Nq, D = q.shape
    Nk = k.shape[1]
    if tuple(k.shape) != (BH, Nk, D) or tuple(v.shape) != (BH, Nk, D):
        raise RuntimeError("attention: q [BH,Nq,D], k / v [BH,Nk,D] expected")
    per_bh = bias is not None and bias.dim() == 3
    if bias is not None:
        bias = _chk(bias, "bias", torch.float32)
        if tuple(bias.shape) not in ((Nq, Nk), (BH, Nq, Nk)):
            raise RuntimeError("attention: bias must be [Nq,Nk] or [BH,Nq,Nk]")
    scale = float(D) ** -0.5 if scale is None else float(scale)
    L = _lib.lib()
    one = L.ofx_attention_workspace_bytes(1, Nq, Nk, D)          # 0: the fused kernel takes this head size, no scores in HBM
    step = BH if one == 0 else max(1, min(BH, int(max_workspace_bytes // one)))
    ws = torch.empty((max(16, L.ofx_attention_workspace_bytes(step, Nq, Nk, D)),), dtype=torch.uint8, device=q.device)
    out = torch.empty_like(q)
    for z0 in range(0, BH, step):
        n = min(step, BH - z0)
        bz = None if bias is None else (bias[z0:z0 + n] if per_bh else bias)
        check(L.ofx_attention_f32(_ptr(q[z0:z0 + n]), _ptr(k[z0:z0 + n]), _ptr(v[z0:z0 + n]), _ptr(bz), Nq * Nk if per_bh else 0,
                                  _ptr(out[z0:z0 + n]), n, Nq, Nk, D, scale, _ptr(ws), ws.numel(), _stream()), "ofx_attention_f32")
    return out


# --------------------------------------------------------------------------------------
# key-frame detector
# --------------------------------------------------------------------------------------
def detect_edges(frames_bgr: torch.Tensor, ksize: int) -> torch.Tensor:
    """uint8 [B,H,W,3] (cv2 channel order) -> uint8 [B,H,W] dilated Canny edge maps (_detect_edges of the reference)."""
    f = _chk(frames_bgr, "frames_bgr", torch.uint8)
    B, H, W, c3 = f.shape
    if c3 != 3:
        raise RuntimeError("frames_bgr must be [B,H,W,3]")
    L = _lib.lib()
    need = L.ofx_detect_edges_scratch_bytes(B, H, W)
    scratch = torch.empty((need,), dtype=torch.uint8, device=f.device)
    out = torch.empty((B, H, W), dtype=torch.uint8, device=f.device)
    check(L.ofx_detect_edges(_ptr(f), _ptr(out), _ptr(scratch), need, B, H, W, int(ksize), _stream()), "ofx_detect_edges")
    return out


def abs_diff_sum_u8(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a: uint8 [B,...]; b: same shape or one image [...] shared by the batch -> int64 [B] sums of |a - b|."""
    x = _chk(a, "a", torch.uint8)
    y = _chk(b, "b", torch.uint8)
    B = x.shape[0]
    n = x[0].numel()
    shared = y.dim() == x.dim() - 1
    if (shared and y.numel() != n) or (not shared and tuple(y.shape) != tuple(x.shape)):
        raise RuntimeError("abs_diff_sum_u8: shape mismatch")
    sums = torch.empty((B,), dtype=torch.int64, device=x.device)
    check(_lib.lib().ofx_abs_diff_sum_u8(_ptr(x), n, _ptr(y), 0 if shared else n, _ptr(sums), B, n, _stream()), "ofx_abs_diff_sum_u8")
    return sums
