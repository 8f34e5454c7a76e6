"""Python handle of the native RAFT engine (csrc/raft_engine.cpp).

`RaftEngine(state_dict)` packs and uploads the weights once (state_dict keys are the reference
checkpoint's, `raft-things.pth` loads unchanged -- `module.` prefixes are accepted);
`forward(image1, image2)` runs `RAFT.forward(iters=20, test_mode=True)` (RAFT/core/raft.py:86-144)
for a batch of uint8 frame pairs resident on the GPU and returns flow f32[B,H,W,2] on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check

FLAG_BGR, FLAG_SHARED_IMG2, FLAG_SHARED_IMG1, FLAG_ALT_CORR, FLAG_BF16X3, FLAG_SERIAL, FLAG_BF16X6, FLAG_BN_BATCH, FLAG_SEPARATE_STATS = 1, 2, 4, 8, 16, 32, 64, 128, 256
CNET_NORMS = {"eval": 0, "batch": FLAG_BN_BATCH}
PRECISIONS = {"fp32": 0, "bf16x3": FLAG_BF16X3, "bf16x6": FLAG_BF16X6}
FLAG_VOL_BF16X3, FLAG_VOL_BF16X6 = 512, 1024
VOLUME_PRECISIONS = {None: 0, "fp32": 0, "bf16x3": FLAG_VOL_BF16X3, "bf16x6": FLAG_VOL_BF16X6}


class RaftEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device: Optional[torch.device] = None, precision: str = "fp32",
                 cnet_norm: str = "eval", volume_precision: Optional[str] = None):
        """NOTE THE DEFAULT SPLIT: `RaftEngine` (and the `pdcnet_of` surface built on it) defaults to cnet_norm='eval'; `ofgen.RAFT_2`
        -- the reference's RAFT wrapper AS WRITTEN -- passes cnet_norm='batch'.  The two networks differ by more than 1 px of flow:
        build the engine with cnet_norm='batch' to mirror `RAFT_2`.

        cnet_norm: how the context encoder's BatchNorm layers (RAFT/core/raft.py:55) are evaluated.  'eval' (default):
        with their running statistics, folded into the convolutions -- a model in `.eval()`, what PDCNet's own code does
        (pdcnet_of.py:62) and canonical RAFT inference.  'batch': with the statistics of the image itself -- the
        reference's `RAFT_2` AS WRITTEN, which never calls `.eval()` (ofgen_keyframe_inpaint.py:47-60) and feeds one image
        per call; every image of a batch is normalised by itself, so a batch equals as many reference calls.
        precision: 'fp32' (default; exact fp32 matrix-core arithmetic, the reference's), 'bf16x3' (opt-in fast
        mode: operands split into two bf16 values, three bf16 MFMAs per product, fp32 accumulate; flow EPE
        ~1e-4 px against the fp32 path) or 'bf16x6' (opt-in: three bf16 pieces = the fp32 value exactly, six
        products, fp32 accumulate: fp32-level accuracy on the bf16 matrix cores).
        volume_precision: None / 'fp32' (default), 'bf16x3' or 'bf16x6' -- the arithmetic of the all-pairs correlation volume ALONE
        (RAFT/core/corr.py:52-60), every convolution stays exact fp32: the GEMM runs on the bf16 matrix cores from operands pre-split
        into bf16 planes (csrc/corr_split.hip; needs H % 64 == 0 and W % 128 == 0, other shapes silently take the fp32 GEMM)."""
        if volume_precision not in VOLUME_PRECISIONS:
            raise ValueError("volume_precision must be None, 'fp32', 'bf16x3' or 'bf16x6'")
        self.volume_precision = volume_precision
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        if cnet_norm not in CNET_NORMS:
            raise ValueError(f"cnet_norm must be one of {sorted(CNET_NORMS)}")
        self.precision = precision
        self.cnet_norm = cnet_norm
        L = _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("RaftEngine needs a HIP device (no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is not None:
            torch.cuda.set_device(self.device)
        keep = []
        arr = (_lib.Tensor * len(state_dict))()
        n = 0
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            t = v.detach().to(torch.float32).contiguous().cpu()
            if t.dim() > 4:
                continue
            keep.append(t)
            arr[n].name = k.encode()
            arr[n].data = t.data_ptr()
            arr[n].ndim = t.dim()
            for j in range(4):
                arr[n].shape[j] = t.shape[j] if j < t.dim() else 1
            n += 1
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(L.ofx_raft_create(arr, n, C.byref(h)), "ofx_raft_create")
        self._h = h
        self._ws: Optional[torch.Tensor] = None
        self._ws_key: Optional[Tuple[int, int, int]] = None
        self.ws_budget_bytes: Optional[int] = None     # see pairs_that_fit

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().ofx_raft_destroy(h)
            except Exception:
                pass
            self._h = None

    def _workspace(self, B: int, H: int, W: int) -> torch.Tensor:
        need = _lib.lib().ofx_raft_workspace_bytes(self._h, B, H, W)
        if need == 0:
            raise RuntimeError(f"unsupported shape B={B} H={H} W={W} (H, W must be multiples of 8)")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        return self._ws

    def pairs_that_fit(self, B: int, H: int, W: int) -> int:
        """Largest batch <= B whose executor workspace (dominated by the all-pairs correlation pyramid: ~1.33 x (H/8 x W/8)^2 floats
        per pair -- 201 MB at 512x768, 5.6 GB at 1080x1920, 89 GB at 2160x3840) fits the device memory that is free right now
        (plus this engine's cached workspace and the caching allocator's idle blocks), so that a large batch of large frames is
        processed in slices instead of failing in the allocator.  `ws_budget_bytes` (attribute) overrides the measured budget."""
        L = _lib.lib()
        budget = self.ws_budget_bytes
        if budget is None and self._ws is not None and L.ofx_raft_workspace_bytes(self._h, B, H, W) <= self._ws.numel():
            return B                       # the cached workspace already holds this batch: nothing to measure
        if budget is None:
            free, _total = torch.cuda.mem_get_info(self.device)
            idle = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
            have = self._ws.numel() if self._ws is not None else 0
            budget = int(0.92 * (free + idle + have))
        if L.ofx_raft_workspace_bytes(self._h, B, H, W) <= budget:
            return B
        lo, hi = 1, B                      # need(b) grows with b: largest b that fits (at least 1: let the allocator speak for itself)
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if L.ofx_raft_workspace_bytes(self._h, mid, H, W) <= budget:
                lo = mid
            else:
                hi = mid - 1
        return lo

    def max_pairs_now(self, H: int, W: int, limit: Optional[int] = None) -> int:
        """`max_pairs` further bounded by the device memory that is free right now, for the indexed-pairs calls (`forward_pairs`:
        every pair may bring two images of its own -- the bound assumes so): what `pdcnet_of` slices its batches by."""
        L = _lib.lib()
        B = self.max_pairs(H, W) if limit is None else max(1, min(int(limit), self.max_pairs(H, W)))
        Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
        need = lambda b: L.ofx_raft_workspace_bytes_pairs(self._h, 2 * b, b, Hp, Wp)
        budget = self.ws_budget_bytes
        if budget is None and self._ws is not None and need(B) <= self._ws.numel():
            return B
        if budget is None:
            free, _total = torch.cuda.mem_get_info(self.device)
            idle = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
            budget = int(0.92 * (free + idle + (self._ws.numel() if self._ws is not None else 0)))
        lo, hi = 1, B
        if need(B) <= budget:
            return B
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if need(mid) <= budget:
                lo = mid
            else:
                hi = mid - 1
        return lo

    @staticmethod
    def max_pairs(H: int, W: int) -> int:
        """Pairs one executor call can take at this frame size: the convolution kernels address their operands with
        32-bit byte offsets (buffer descriptors), so the widest array (the 768-float row of the hoisted GRU context
        term) must stay below 2 GiB.  Larger batches are processed in slices (pairs are independent)."""
        return max(1, ((1 << 31) - 4096) // (((H + 7) // 8) * ((W + 7) // 8) * 768 * 4))

    @staticmethod
    def pad_to_8(img: torch.Tensor) -> torch.Tensor:
        """InputPadder('sintel') of RAFT/core/utils/utils.py:7-19 for uint8 [B,H,W,3] tensors:
        replicate padding, centred.  No-op when H and W are multiples of 8."""
        H, W = img.shape[1:3]
        ph = (((H // 8) + 1) * 8 - H) % 8
        pw = (((W // 8) + 1) * 8 - W) % 8
        if ph == 0 and pw == 0:
            return img
        x = img.permute(0, 3, 1, 2).float()
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), mode="replicate")
        return x.permute(0, 2, 3, 1).to(torch.uint8).contiguous()

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor, iters: int = 20, bgr: bool = False,
                alternate_corr: bool = False, want_low: bool = False, serial: bool = False, separate_stats: bool = False,
                warp_frame: Optional[torch.Tensor] = None, warp_sign: float = 1.0,
                want_flow: bool = True):
        """image1: uint8 [B,H,W,3] or [H,W,3] (shared by the batch); image2 likewise.  Flow is defined
        on image1's grid and points into image2.  Returns flow_up f32[B,H,W,2] (and flow_low).
        serial=True keeps every launch on the current stream (small batches otherwise overlap their
        independent chains on the engine's side streams; results are identical).  separate_stats=True (diagnostic) takes the
        instance-norm statistics with their own f64 pass instead of out of the convolution epilogues.
        warp_frame: uint8 [H,W,3] (one frame shared by the batch -- the rendered AI key frame): its bilinear backward warp along the
        final flow is produced INSIDE the convex upsample (`ofx_raft_forward_warp`; warp_sign +1 = pdcnet_of.warp_frame's convention,
        -1 = ofgen.warp_frame's) and returned after the flow: (flow_up[, flow_low], warped u8 [B,H,W,3]) -- bit-identical to
        `ops.warp(warp_frame, flow_up, mode="bilinear")`.  want_flow=False skips writing the full-resolution flow (None is returned in
        its place)."""
        for nm, t in (("image1", image1), ("image2", image2)):
            if not t.is_cuda:
                raise RuntimeError(f"{nm} must be a CUDA tensor")
            if t.dtype != torch.uint8:
                raise RuntimeError(f"{nm} must be uint8")
        flags = (FLAG_BGR if bgr else 0) | PRECISIONS[self.precision] | (FLAG_SERIAL if serial else 0) | CNET_NORMS[self.cnet_norm] | (FLAG_SEPARATE_STATS if separate_stats else 0)
        flags |= VOLUME_PRECISIONS[self.volume_precision]
        sh1, sh2 = image1.dim() == 3, image2.dim() == 3
        if sh1 and sh2:
            image1, sh1 = image1[None], False
            image2, sh2 = image2[None], False
        i1 = self.pad_to_8(image1[None] if sh1 else image1).contiguous()
        i2 = self.pad_to_8(image2[None] if sh2 else image2).contiguous()
        B = i2.shape[0] if sh1 else i1.shape[0]
        H, W = i1.shape[1:3]
        if tuple(i2.shape[1:]) != (H, W, 3) or i1.shape[3] != 3:
            raise RuntimeError(f"image shapes differ: {tuple(i1.shape)} vs {tuple(i2.shape)}")
        if not sh1 and not sh2 and i1.shape[0] != i2.shape[0]:
            raise RuntimeError("batch sizes differ")
        if sh1:
            flags |= FLAG_SHARED_IMG1
        if sh2:
            flags |= FLAG_SHARED_IMG2
        if alternate_corr:
            flags |= FLAG_ALT_CORR
        if warp_frame is not None:
            if not warp_frame.is_cuda or warp_frame.dtype != torch.uint8 or tuple(warp_frame.shape) != (H, W, 3):
                raise RuntimeError(f"warp_frame must be a CUDA uint8 tensor [{H},{W},3] (the padded frame size)")
            if warp_sign not in (1.0, -1.0):
                raise ValueError("warp_sign must be +1 or -1")
            warp_frame = warp_frame.contiguous()
        elif not want_flow:
            raise ValueError("want_flow=False only makes sense together with warp_frame")
        max_pairs = self.max_pairs(H, W)
        if B > 1:
            max_pairs = min(max_pairs, self.pairs_that_fit(min(B, max_pairs), H, W))
        if B > max_pairs:
            outs = []
            for b0 in range(0, B, max_pairs):
                a = image1 if sh1 else image1[b0:b0 + max_pairs]
                c = image2 if sh2 else image2[b0:b0 + max_pairs]
                r = self.forward(a, c, iters=iters, bgr=bgr, alternate_corr=alternate_corr, want_low=want_low, serial=serial, separate_stats=separate_stats,
                                 warp_frame=warp_frame, warp_sign=warp_sign, want_flow=want_flow)
                outs.append(r if isinstance(r, tuple) else (r,))
            cat = tuple(None if parts[0] is None else torch.cat(parts) for parts in zip(*outs))
            return cat if len(cat) > 1 else cat[0]
        ws = self._workspace(B, H, W)
        flow_up = torch.empty((B, H, W, 2), dtype=torch.float32, device=self.device) if want_flow else None
        flow_low = torch.empty((B, H // 8, W // 8, 2), dtype=torch.float32, device=self.device) if want_low else None
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        fu = C.c_void_p(flow_up.data_ptr() if want_flow else 0)
        fl = C.c_void_p(flow_low.data_ptr() if want_low else 0)
        if warp_frame is None:
            check(_lib.lib().ofx_raft_forward(self._h, C.c_void_p(i1.data_ptr()), C.c_void_p(i2.data_ptr()), B, H, W, int(iters),
                                              flags, fu, fl, C.c_void_p(ws.data_ptr()), ws.numel(), stream), "ofx_raft_forward")
            return (flow_up, flow_low) if want_low else flow_up
        warped = torch.empty((B, H, W, 3), dtype=torch.uint8, device=self.device)
        check(_lib.lib().ofx_raft_forward_warp(self._h, C.c_void_p(i1.data_ptr()), C.c_void_p(i2.data_ptr()), B, H, W, int(iters), flags,
                                               fu, fl, C.c_void_p(warp_frame.data_ptr()), float(warp_sign), C.c_void_p(warped.data_ptr()),
                                               C.c_void_p(ws.data_ptr()), ws.numel(), stream), "ofx_raft_forward_warp")
        return (flow_up, flow_low, warped) if want_low else (flow_up, warped)

    @torch.no_grad()
    def forward_pairs(self, images: torch.Tensor, idx1, idx2, iters: int = 20, bgr: bool = False,
                      warp_frame: Optional[torch.Tensor] = None, warp_sign: float = 1.0, n_warp: int = 0):
        """images: uint8 [n,H,W,3] on the device (H, W multiples of 8); pair b = (idx1[b], idx2[b]):
        flow b lives on image idx1[b] and points into image idx2[b].  Every image is encoded once however
        many pairs use it (KeyframeConv's N x N sweep).  Returns f32 [B,H,W,2] on the device.
        warp_frame (uint8 [H,W,3], one frame shared by the batch) + n_warp: the first n_warp pairs also get the bilinear
        backward warp of warp_frame along their final flow, produced inside the convex upsample
        (`ofx_raft_forward_pairs_warp`); returns (flow, warped u8 [n_warp,H,W,3])."""
        if not images.is_cuda or images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
            raise RuntimeError("images must be a CUDA uint8 tensor [n,H,W,3]")
        imgs = images.contiguous()
        n, H, W, _ = imgs.shape
        if H % 8 or W % 8:
            raise RuntimeError("forward_pairs needs H and W to be multiples of 8 (pad the frames first)")
        B = len(idx1)
        if B == 0 or len(idx2) != B:
            raise RuntimeError("idx1 / idx2 must be non-empty and of equal length")
        a1 = (C.c_int * B)(*[int(i) for i in idx1])
        a2 = (C.c_int * B)(*[int(i) for i in idx2])
        L = _lib.lib()
        need = L.ofx_raft_workspace_bytes_pairs(self._h, n, B, H, W)
        if need == 0:
            raise RuntimeError(f"unsupported shape n={n} B={B} H={H} W={W}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        flow_up = torch.empty((B, H, W, 2), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        flags = (FLAG_BGR if bgr else 0) | PRECISIONS[self.precision] | CNET_NORMS[self.cnet_norm] | VOLUME_PRECISIONS[self.volume_precision]
        if warp_frame is None:
            check(L.ofx_raft_forward_pairs(self._h, C.c_void_p(imgs.data_ptr()), n, a1, a2, B, H, W, int(iters), flags,
                                           C.c_void_p(flow_up.data_ptr()), None,
                                           C.c_void_p(self._ws.data_ptr()), self._ws.numel(), stream), "ofx_raft_forward_pairs")
            return flow_up
        if not warp_frame.is_cuda or warp_frame.dtype != torch.uint8 or tuple(warp_frame.shape) != (H, W, 3):
            raise RuntimeError(f"warp_frame must be a CUDA uint8 tensor [{H},{W},3]")
        if warp_sign not in (1.0, -1.0):
            raise ValueError("warp_sign must be +1 or -1")
        n_warp = int(n_warp)
        if not 0 < n_warp <= B:
            raise ValueError("n_warp must be in 1..B")
        wf = warp_frame.contiguous()
        warped = torch.empty((n_warp, H, W, 3), dtype=torch.uint8, device=self.device)
        check(L.ofx_raft_forward_pairs_warp(self._h, C.c_void_p(imgs.data_ptr()), n, a1, a2, B, H, W, int(iters), flags,
                                            C.c_void_p(flow_up.data_ptr()), None, C.c_void_p(wf.data_ptr()), float(warp_sign), n_warp,
                                            C.c_void_p(warped.data_ptr()), C.c_void_p(self._ws.data_ptr()), self._ws.numel(), stream),
              "ofx_raft_forward_pairs_warp")
        return flow_up, warped

    def buffer(self, name: str) -> torch.Tensor:
        """Copy of a named intermediate of the last forward (flat f32) -- for stage-level parity tests."""
        p, n = C.c_void_p(), C.c_size_t()
        check(_lib.lib().ofx_raft_buffer(self._h, name.encode(), C.byref(p), C.byref(n)), "ofx_raft_buffer")
        ws = self._ws
        off = p.value - ws.data_ptr()
        assert 0 <= off and off + 4 * n.value <= ws.numel()
        return ws[off:off + 4 * n.value].view(torch.float32).clone()
