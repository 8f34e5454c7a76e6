"""Drop-in for the reference's `pdcnet_of.py` call surface (pdcnet_of.py:19-79), MI355X-native.

Exports exactly the names the reference's drivers import
(`ofgen_pixel_inpaint.py:17`, `ofgen_keyframe_inpaint.py:26`):

    create_of_algo(ckpt) -> algo
    algo.calc(frame1_bgr, frame2_bgr) -> (flow f32[H,W,2], confidence f32[H,W], log_confidence f32[H,W])  numpy
    algo.calc_batch(src_rgb_u8[B,H,W,3], tgt_rgb_u8[B,H,W,3]) -> (flow_est[B,H,W,2], confidence[B,H,W])   indexable
    algo.to(device) -> algo
    warp_frame(frame, flow) -> ndarray          (backward warp: out(y,x) = frame(y+fy, x+fx))
    warp_frame_latent(latent, flow) -> CPU tensor [1,C,lh,lw]

Differences from the reference, all forced by what is (not) in its tree:
  * The reference wraps PDCNet+ from the un-vendored DenseMatching checkout (pdcnet_of.py:6-13).  That
    network is not available, so the flow network here is the RAFT that IS vendored (RAFT/core), run in
    the same orientation PDCNetPlus.calc uses: flow is defined on frame2 (target) and points into
    frame1 (source), i.e. RAFT(image1=frame2, image2=frame1).
  * RAFT emits no confidence; `confidence` / `log_confidence` come from a forward-backward
    consistency check (ofx_fb_confidence) -- a labelled extension with PDCNetPlus.calc's output shape.
  * device-resident fast paths are added (`calc_batch_device`, `synthesize`) so flow, warp and mask
    never leave HBM; the numpy-returning methods keep the reference's host-side contract.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from .raft import RaftEngine
from .weights import load_checkpoint

DEFAULT_WARP_MODE = "cv2_cubic"      # the reference warps with cv2.remap(INTER_CUBIC), pdcnet_of.py:41


class PDCNetPlus:
    """Flow + confidence estimator with the duck type of the reference's `PDCNetPlus` (pdcnet_of.py:45-75)."""

    def __init__(self, ckpt_path="pre_trained_models/PDCNet_plus_m.pth.tar", device=None, iters: int = 20,
                 confidence_sigma: float = 3.0, precision: str = "fp32", volume_precision: Optional[str] = None):
        self.state_dict = load_checkpoint(ckpt_path)
        self.precision = precision
        self.volume_precision = volume_precision       # extension: 'bf16x6' / 'bf16x3' = ONLY the correlation volume in split-bf16 form
        self.iters = int(iters)
        self.sigma = float(confidence_sigma)
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.network = RaftEngine(self.state_dict, self.device, precision=precision, volume_precision=volume_precision)   # like `.cuda()` at pdcnet_of.py:61

    def to(self, device):
        """`pdcnet_model.to(device)` (ofgen_keyframe_inpaint.py:555).  Moving re-uploads the weights."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("this flow algorithm only runs on a HIP device (no CPU fallback)")
        cur = self.device.index if self.device.index is not None else torch.cuda.current_device()
        new = device.index if device.index is not None else torch.cuda.current_device()
        if cur != new:
            self.device = device
            self.network = RaftEngine(self.state_dict, device, precision=self.precision, volume_precision=self.volume_precision)
        return self

    # ---- device-resident core ------------------------------------------------------------------
    @torch.no_grad()
    def calc_batch_device(self, source: torch.Tensor, target: torch.Tensor, bgr: bool = False,
                          want_confidence: bool = True, warp_frame: Optional[torch.Tensor] = None):
        """source/target: uint8 [B,H,W,3] on the device (`source` may be a single [H,W,3] key frame shared
        by the batch).  Returns (flow f32[B,H,W,2] on the target grid pointing into source, confidence
        f32[B,H,W], log_confidence f32[B,H,W]) -- all on the device.
        warp_frame: uint8 [H,W,3] on the device (the rendered AI key frame, shared by the batch): a fourth result is appended,
        `warp_frame` warped backward along each flow with bilinear taps (`warp_frame(ai, flow)` of pdcnet_of.py:34-42 in the
        north star's bilinear mode), u8 [B,H,W,3].  On frames whose sides are multiples of 8 it comes out of the convex
        upsample itself (`ofx_raft_forward_warp` / `ofx_raft_forward_pairs_warp`: the flow is never re-read), bit-identical to
        `ops.warp(warp_frame, flow, 'bilinear')`, which padded frames fall back to."""
        net = self.network
        H0, W0 = target.shape[-3], target.shape[-2]
        fuse = warp_frame is not None and H0 % 8 == 0 and W0 % 8 == 0
        if warp_frame is not None and (not warp_frame.is_cuda or warp_frame.dtype != torch.uint8 or tuple(warp_frame.shape) != (H0, W0, 3)):
            raise RuntimeError(f"warp_frame must be a CUDA uint8 tensor [{H0},{W0},3]")

        def done(flow, conf, logc, warped=None):
            if warp_frame is None:
                return flow, conf, logc
            if warped is None:                                 # padded frame: the warp runs on the cropped flow, as the reference's would
                warped = ops.warp(warp_frame.contiguous(), flow.contiguous(), mode="bilinear", sign=1.0)
            return flow, conf, logc, warped
        if not want_confidence:
            if fuse:
                flow, warped = net.forward(target, source, iters=self.iters, bgr=bgr, warp_frame=warp_frame)   # target -> source
                return done(flow, None, None, warped)
            flow = net.forward(target, source, iters=self.iters, bgr=bgr)                      # target -> source
            return done(_unpad(flow, H0, W0), None, None)
        # Both directions in ONE indexed-pairs call: every image is encoded once (two separate forwards encode each
        # image twice) and the forward / backward refinements share their launches (2B pairs per batch).
        shared = source.dim() == 3
        src = net.pad_to_8((source[None] if shared else source).contiguous())
        tgt = net.pad_to_8((target[None] if target.dim() == 3 else target).contiguous())
        B, ns = tgt.shape[0], src.shape[0]
        if not shared and ns != B:
            raise RuntimeError("source / target batch sizes differ")
        step = max(1, net.max_pairs_now(tgt.shape[1], tgt.shape[2], 2 * B) // 2)      # 32-bit offsets AND the memory free right now
        fts, fss, wps = [], [], []
        for b0 in range(0, B, step):
            t = tgt[b0:b0 + step]
            s = src if shared else src[b0:b0 + step]
            n, m = t.shape[0], s.shape[0]
            it = [m + i for i in range(n)]                     # image index of target i in cat([s, t])
            isrc = [0] * n if shared else list(range(n))
            if fuse:                                           # the first n pairs (target -> source) are the ones the AI frame rides on
                out, wp = net.forward_pairs(torch.cat([s, t]), it + isrc, isrc + it, iters=self.iters, bgr=bgr, warp_frame=warp_frame,
                                            n_warp=n)
                wps.append(wp)
            else:
                out = net.forward_pairs(torch.cat([s, t]), it + isrc, isrc + it, iters=self.iters, bgr=bgr)
            fts.append(out[:n])                                # on the target grid, pointing into the source
            fss.append(out[n:])                                # on the source grid, pointing into the target
        flow_t = fts[0] if len(fts) == 1 else torch.cat(fts)
        flow_s = fss[0] if len(fss) == 1 else torch.cat(fss)
        # the consistency check runs on the padded grids (both flows live there); the reference's `calc` promises
        # [H,W] outputs (pdcnet_of.py:72-75), so the replicate padding is cropped away afterwards
        conf, logc = ops.fb_confidence(flow_t.contiguous(), flow_s.contiguous(), self.sigma)
        warped = (wps[0] if len(wps) == 1 else torch.cat(wps)) if fuse else None
        return done(_unpad(flow_t, H0, W0), _unpad(conf, H0, W0), _unpad(logc, H0, W0), warped)

    @torch.no_grad()
    def calc_pairs(self, frames: torch.Tensor, pairs, bgr: bool = False, max_flows: int = 64):
        """Many (source, target) pairs over a small set of frames (KeyframeConv / calculate_pairwise,
        ofgen_keyframe_inpaint.py:627-668).  frames: uint8 [n,H,W,3] on the device; pairs: list of (s, t)
        indices into `frames`.  Returns device tensors (flow f32[P,H,W,2] on t's grid pointing into s,
        confidence f32[P,H,W]).  Each frame is encoded once per chunk, and the backward flow needed by the
        confidence of (s, t) is shared with pair (t, s) when both are requested."""
        pairs = [(int(s), int(t)) for s, t in pairs]
        H0, W0 = frames.shape[1], frames.shape[2]
        frames = self.network.pad_to_8(frames.contiguous())
        # one executor call addresses its operands with 32-bit offsets: 113 pairs at 512x768 but 21 at 1920x1080
        max_flows = self.network.max_pairs_now(frames.shape[1], frames.shape[2], max_flows)   # ... and what fits the free memory
        need = {}                                    # directed flow (image1, image2) -> slot
        for s, t in pairs:
            need.setdefault((t, s), len(need))       # flow on t's grid into s
            need.setdefault((s, t), len(need))       # its backward companion
        keys = list(need.keys())
        flows = [None] * len(keys)
        for c0 in range(0, len(keys), max_flows):
            chunk = keys[c0:c0 + max_flows]
            used = sorted({i for k in chunk for i in k})
            remap = {g: l for l, g in enumerate(used)}
            sub = frames[torch.tensor(used, device=frames.device)]
            out = self.network.forward_pairs(sub, [remap[a] for a, _ in chunk], [remap[b] for _, b in chunk],
                                             iters=self.iters, bgr=bgr)
            for j in range(len(chunk)):
                flows[c0 + j] = out[j]
        fw = torch.stack([flows[need[(t, s)]] for s, t in pairs])
        bw = torch.stack([flows[need[(s, t)]] for s, t in pairs])
        conf, _ = ops.fb_confidence(fw.contiguous(), bw.contiguous(), self.sigma)
        return _unpad(fw, H0, W0), _unpad(conf, H0, W0)

    # ---- reference-compatible host API ---------------------------------------------------------
    @torch.no_grad()
    def calc(self, frame1: np.ndarray, frame2: np.ndarray):
        """pdcnet_of.py:65-75: BGR uint8 numpy frames in, numpy (flow, confidence, log_confidence) out."""
        src = torch.from_numpy(np.ascontiguousarray(frame1)).to(self.device)[None]
        tgt = torch.from_numpy(np.ascontiguousarray(frame2)).to(self.device)[None]
        flow, conf, logc = self.calc_batch_device(src, tgt, bgr=True)
        return flow[0].cpu().numpy(), conf[0].cpu().numpy(), logc[0].cpu().numpy()

    @torch.no_grad()
    def calc_batch(self, source: torch.Tensor, target: torch.Tensor):
        """`calc_batch` as called at ofgen_keyframe_inpaint.py:594: RGB uint8 tensors [B,H,W,3] on the
        device; returns (flow_est, confidence) whose items assign into numpy slots (:598-599)."""
        flow, conf, _ = self.calc_batch_device(source.contiguous(), target.contiguous(), bgr=False)
        return flow.cpu().numpy(), conf.cpu().numpy()


def _unpad(t: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """Undo `RaftEngine.pad_to_8` (InputPadder 'sintel': centred replicate padding, utils.py:9-16) on a [B,Hp,Wp(,C)]
    tensor.  No-op (no copy) when nothing was padded."""
    Hp, Wp = t.shape[1], t.shape[2]
    if Hp == H and Wp == W:
        return t
    y0, x0 = (Hp - H) // 2, (Wp - W) // 2
    return t[:, y0:y0 + H, x0:x0 + W].contiguous()


def create_of_algo(ckpt, precision: str = "fp32", volume_precision: Optional[str] = None) -> PDCNetPlus:
    """pdcnet_of.py:77-79.  `precision` / `volume_precision` are extensions (default: the reference's fp32 arithmetic everywhere)."""
    return PDCNetPlus(ckpt, precision=precision, volume_precision=volume_precision)


# --------------------------------------------------------------------------------------------------
def _to_dev(a, device) -> torch.Tensor:
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device).contiguous()


def warp_frame(frame, flow, mode: Optional[str] = None, device="cuda"):
    """pdcnet_of.py:34-42.  frame: ndarray [H,W] or [H,W,C], uint8 or float32; flow f32[H,W,2].
    Returns an ndarray of the same shape/dtype.  `mode`: 'cv2_cubic' (reference-faithful default),
    'bicubic' (float Keys cubic) or 'bilinear'."""
    mode = mode or DEFAULT_WARP_MODE
    fr = np.asarray(frame)
    squeeze = fr.ndim == 2
    if fr.dtype not in (np.uint8, np.float32):
        fr = fr.astype(np.float32)
    f = _to_dev(fr[:, :, None] if squeeze else fr, device)
    out = ops.warp(f, _to_dev(np.asarray(flow, dtype=np.float32), device), mode=mode, sign=1.0)
    out = out.cpu().numpy()
    return out[:, :, 0] if squeeze else out


def warp_frame_latent(latent: torch.Tensor, flow, mode: Optional[str] = None, device="cuda") -> torch.Tensor:
    """pdcnet_of.py:19-32: cubic-resize the latent to the flow's size, warp, resize back."""
    mode = mode or DEFAULT_WARP_MODE
    lat = latent.detach().to(torch.float32).to(device)
    _, _, lh, lw = lat.shape
    fl = _to_dev(np.asarray(flow, dtype=np.float32) if not torch.is_tensor(flow) else flow, device)
    h, w = fl.shape[:2]
    x = lat.permute(0, 2, 3, 1).contiguous()
    up = ops.resize_cubic(x, h, w)
    wp = ops.warp(up, fl[None].contiguous(), mode=mode, sign=1.0)
    dn = ops.resize_cubic(wp, lh, lw)
    return dn.permute(0, 3, 1, 2).contiguous().cpu()
