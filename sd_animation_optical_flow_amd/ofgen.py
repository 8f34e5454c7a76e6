"""Drop-in for the flow / warp / mask helpers of the reference's `ofgen*.py` drivers, MI355X-native.

Same names, argument meaning and mutation behaviour as the reference functions they replace
(file:line in the reference repo):

    RAFT_2().calc(img1_bgr, img2_bgr) -> flow        ofgen_keyframe_inpaint.py:47-71 ; ofgen.py:55-79
    warp_frame(frame, flow)          (RAFT convention) ofgen_keyframe_inpaint.py:92-98
    warp_frame_latent(latent, flow)                  ofgen_keyframe_inpaint.py:100-111
    of_calc(frame1, frame2, algo)                    ofgen_keyframe_inpaint.py:113-133 ; ofgen.py:45-49 (bare-flow algo -> (flow, v))
    generate_mask(conf, log_conf, thres)             ofgen_keyframe_inpaint.py:317-322  (mutates log_conf)
    create_mask_aux / confidence_to_mask             ofgen_keyframe_inpaint.py:237-248, 292-304
    mix_propagated_ai_frame                          ofgen_keyframe_inpaint.py:306-315
    merge_images(base, second, mask, 'naive')        ofgen_keyframe_inpaint.py:676-681
    expand_mask(mask, ori_image)                     ofgen_keyframe_inpaint.py:968-973
    PDCNetAux                                        ofgen_keyframe_inpaint.py:549-653  (+ .npy pair cache)
    compose_warp_and_mask                            ofgen_keyframe_inpaint.py:995-1027 (greedy multi-reference)

numpy in / numpy out like the reference; every computation is a HIP kernel call (ops.py).  The
`*_device` variants keep tensors in HBM for callers that chain steps.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from . import pdcnet_of as _pd
from .raft import RaftEngine
from .weights import load_checkpoint

warp_frame_pdcnet = _pd.warp_frame          # the reference's import alias (ofgen_keyframe_inpaint.py:26)


def _dev(a, device="cuda") -> torch.Tensor:
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device).contiguous()


class namespace:
    """`args` holder of the reference (ofgen_keyframe_inpaint.py:43-45)."""
    def __contains__(self, m):
        return hasattr(self, m)


class RAFT_2:
    """ofgen_keyframe_inpaint.py:47-71.  `model` = checkpoint path / state_dict / 'random:<seed>'.

    `cnet_norm='batch'` (the default here) is the reference AS WRITTEN: `RAFT_2.__init__` (:47-60) never calls `.eval()`,
    so the context encoder's BatchNorm2d layers (RAFT/core/raft.py:55) normalise each call's single image with its own
    statistics (then gamma / beta) and the checkpoint's running statistics are never used.  `cnet_norm='eval'` is
    canonical RAFT inference (running statistics) for callers who want that instead.  Pinned by
    tests/golden/raft_ref_trainbn_128x160.npz (the reference module left in train mode) -- DESIGN.md section 2."""

    def __init__(self, model="../RAFT/models/raft-things.pth", device="cuda", iters: int = 20, alternate_corr: bool = False,
                 cnet_norm: str = "batch"):
        self.device = torch.device(device)
        self.iters = iters
        self.alternate_corr = alternate_corr
        self.model = RaftEngine(load_checkpoint(model), self.device, cnet_norm=cnet_norm)

    @torch.no_grad()
    def calc(self, img1: np.ndarray, img2: np.ndarray) -> np.ndarray:
        """BGR uint8 frames -> flow f32[H',W',2] on img1's grid (H',W' = padded to /8: the reference does
        not un-pad, :70)."""
        a = _dev(img1, self.device)[None]
        b = _dev(img2, self.device)[None]
        flo = self.model.forward(a, b, iters=self.iters, bgr=True, alternate_corr=self.alternate_corr)
        return flo[0].cpu().numpy()


def create_of_algo(ckpt="../DenseMatching/pre_trained_models/PDCNet_plus_m.pth.tar"):
    """ofgen_keyframe_inpaint.py:73-77."""
    return _pd.create_of_algo(ckpt)


def warp_frame(frame, flow, mode: Optional[str] = None, device="cuda") -> np.ndarray:
    """RAFT-convention warp (ofgen_keyframe_inpaint.py:92-98): out(y,x) = frame(y - fy, x - fx)."""
    mode = mode or _pd.DEFAULT_WARP_MODE
    fr = np.asarray(frame)
    squeeze = fr.ndim == 2
    if fr.dtype not in (np.uint8, np.float32):
        fr = fr.astype(np.float32)
    out = ops.warp(_dev(fr[:, :, None] if squeeze else fr, device), _dev(np.asarray(flow, np.float32), device), mode=mode, sign=-1.0)
    out = out.cpu().numpy()
    return out[:, :, 0] if squeeze else out


def warp_frame_latent(latent: torch.Tensor, flow, mode: Optional[str] = None, device="cuda") -> torch.Tensor:
    """ofgen_keyframe_inpaint.py:100-111 (RAFT convention)."""
    mode = mode or _pd.DEFAULT_WARP_MODE
    lat = latent.detach().to(torch.float32).to(device)
    _, _, lh, lw = lat.shape
    fl = _dev(np.asarray(flow, np.float32) if not torch.is_tensor(flow) else flow, device)
    h, w = fl.shape[:2]
    up = ops.resize_cubic(lat.permute(0, 2, 3, 1).contiguous(), h, w)
    wp = ops.warp(up, fl[None].contiguous(), mode=mode, sign=-1.0)
    return ops.resize_cubic(wp, lh, lw).permute(0, 3, 1, 2).contiguous().cpu()


def of_calc(frame1, frame2, algo, verbose: bool = False):
    """Both of the reference's `of_calc`s, told apart by what `algo.calc` returns:

    * a PDCNet-style algo (`calc -> (flow, confidence, log_confidence)`): ofgen_keyframe_inpaint.py:113-133 --
      `(flow, confidence, v, log_confidence)`, v = |flow| with v[confidence < 0.9] = 0;
    * a RAFT-style algo (`calc -> flow`, e.g. this module's `RAFT_2`): ofgen.py:45-49, the live caller at ofgen.py:137 --
      `(flow, v)`, v = sqrt(fx*fx + fy*fy) everywhere."""
    res = algo.calc(frame1, frame2)
    if isinstance(res, (tuple, list)):
        flow, confidence, log_confidence = res
        v = ops.travel_distance(_dev(flow)[None], _dev(confidence)[None], 0.9)[0].cpu().numpy()
        if verbose:
            print("v.max()", v.max(), "v.min()", v.min())
        return flow, confidence, v, log_confidence
    v = ops.flow_magnitude(_dev(np.asarray(res, np.float32))).cpu().numpy()   # no confidence floor in this variant
    return res, v


def generate_mask(cum_confidence: np.ndarray, log_confidence: np.ndarray, thres: float = 0.8, ksize: int = 7):
    """ofgen_keyframe_inpaint.py:317-322.  Like the reference, `log_confidence` is modified in place
    (pixels that will be inpainted are reset to 0) and also returned."""
    conf = _dev(np.asarray(cum_confidence, np.float32))[None]
    lc = _dev(np.asarray(log_confidence, np.float32))[None]
    mask = ops.generate_mask(conf, lc, thres, ksize)
    np.copyto(log_confidence, lc[0].cpu().numpy().astype(log_confidence.dtype, copy=False))
    return mask[0].cpu().numpy(), log_confidence


def create_mask_aux(h: int, w: int, pixel_dist_thres: float):
    """ofgen_keyframe_inpaint.py:292-304."""
    aux = namespace()
    aux.pixel_travel_dist = np.zeros((h, w), dtype=np.float32)
    aux.thres = pixel_dist_thres
    return aux


def confidence_to_mask(confidence, flow, dist, mask_aux, warp_mode: Optional[str] = None) -> np.ndarray:
    """ofgen_keyframe_inpaint.py:237-248 (stateful: updates mask_aux.pixel_travel_dist)."""
    mode = warp_mode or _pd.DEFAULT_WARP_MODE
    raw, travel = ops.travel_mask(_dev(np.asarray(confidence, np.float32))[None], _dev(np.asarray(flow, np.float32))[None],
                                  _dev(np.asarray(dist, np.float32))[None], _dev(mask_aux.pixel_travel_dist)[None],
                                  float(mask_aux.thres), mode)
    mask_aux.pixel_travel_dist = travel[0].cpu().numpy()
    return ops.dilate(raw, 15)[0].cpu().numpy()


def mix_propagated_ai_frame(raw_ai_frame, warped_propagated_ai_frame, mask, propagated_pixel_weight: float = 1.0):
    """ofgen_keyframe_inpaint.py:306-315."""
    if propagated_pixel_weight < 0.001:
        return raw_ai_frame
    out = ops.mix_frames(_dev(raw_ai_frame)[None], _dev(warped_propagated_ai_frame)[None], _dev(mask)[None],
                         propagated_pixel_weight)
    return out[0].cpu().numpy()


def merge_images(base_image, second_image, mask, method: str = "naive"):
    """ofgen_keyframe_inpaint.py:676-688 ('poisson' needs cv2.seamlessClone and is unused by the drivers)."""
    if method != "naive":
        raise NotImplementedError("only the 'naive' merge of the reference's live path is provided")
    return ops.merge_images(_dev(base_image)[None], _dev(second_image)[None], _dev(mask)[None])[0].cpu().numpy()


def expand_mask(mask: np.ndarray, ori_image: np.ndarray) -> np.ndarray:
    """ofgen_keyframe_inpaint.py:968-973."""
    return ops.expand_mask(_dev(mask)[None], _dev(ori_image)[None], 20, 7)[0].cpu().numpy()


# --------------------------------------------------------------------------------------------------
# greedy multi-reference warp + mask composition (a18)
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def compose_warp_and_mask(flow_mat: np.ndarray, ai_frames: Sequence[np.ndarray], original_frame: np.ndarray,
                          thres: float = 0.5, warp_mode: Optional[str] = None, expand: bool = True):
    """The warp/mask block of generate_ai_frame_with_ref_warp_and_inpaint_crossattn
    (ofgen_keyframe_inpaint.py:995-1027).  flow_mat f32[N,1,H,W,3] = (fx, fy, confidence) is updated
    in place exactly like the reference (:995, :1023-1024).  Returns (ret_frame, mask2, order)."""
    mode = warp_mode or _pd.DEFAULT_WARP_MODE
    dev = "cuda"
    fm = _dev(flow_mat, dev)                                   # [N,1,H,W,3]
    n, _, h, w, _ = fm.shape
    fm[..., 2] = (fm[..., 2] > thres).to(torch.float32)        # :995 (plumbing: dtype/select only)
    mask = torch.zeros((1, h, w), dtype=torch.uint8, device=dev)
    ret = None
    order: List[int] = []
    for _ in range(n):
        scores = ops.conf_sum(fm.reshape(n, h, w, 3), 2)       # einops.reduce(..., 'sum'), :1000
        s = int(torch.argmax(scores).item())
        order.append(s)
        flow = fm[s, 0, :, :, 0:2].contiguous()[None]
        warped = ops.warp(_dev(ai_frames[s], dev), flow, mode=mode, sign=1.0)            # [1,H,W,3]
        last = fm[s, 0, :, :, 2].clone()
        cur = (last * 255).to(torch.uint8)[None]
        mask = mask | cur                                      # cv2.bitwise_or, :1009
        ret = warped.clone() if ret is None else ops.merge_images(ret, warped, cur)
        fm[:, 0, :, :, 2] -= last[None]
        fm[:, 0, :, :, 2].clamp_(0, 1)
    mask2 = 255 - mask
    if expand:
        mask2 = ops.expand_mask(mask2.contiguous(), _dev(original_frame, dev)[None], 20, 7)
    np.copyto(flow_mat, fm.cpu().numpy())
    return ret[0].cpu().numpy(), mask2[0].cpu().numpy(), order


# --------------------------------------------------------------------------------------------------
# PDCNetAux: pair flow over a workspace, device-resident, with the on-disk .npy pair cache
# --------------------------------------------------------------------------------------------------
def chunks(lst, n):
    for i in range(0, len(lst), n):
        yield lst[i:i + n]


class PDCNetAux:
    """Pair flow + confidence over the frames of a workspace (ofgen_keyframe_inpaint.py:549-653).

    The unit of work is a LIST OF PAIRS over a small set of frames (`pair_fields`): every distinct frame is decoded and
    uploaded once (a bounded device-side frame cache survives across calls -- KeyframeConv's windows overlap), every
    distinct frame is encoded once by the flow engine however many pairs use it (`PDCNetPlus.calc_pairs`), and flow and
    confidence stay in HBM.  The reference's matrix builders (`calculate_multiple_to_one` -> f32[N,1,H,W,3],
    `calculate_pairwise` -> f32[N,N,H,W,3], identity pairs = zero flow / confidence 1, the `pdcnet/{s:05d}-{t:05d}.npy`
    cache with the same dtype and layout) are views assembled from that list; `keyframe_scores` / `ofgen.keyframe_conv`
    reduce on the device instead and never build the matrix.

    `video` needs `.size_hw` and `.get_raw_frame(i) -> BGR uint8`; index containers need `.indices` and `__len__`
    (`workspace.VideoData` / `workspace.VideoFrameIndices`, or the reference's own classes)."""

    def __init__(self, pdcnet_model, workspace_dir: str, batch_size: int = 16, device=torch.device("cuda:0"),
                 async_save: bool = False, frame_cache: int = 64) -> None:
        """async_save=True (extension): the 4.7 MB-per-pair `.npy` dumps are written by a background
        thread from a private copy, so they leave the caller's critical path; `flush()` (also called by
        `load_cached`, `purge` and on deletion) waits for them.  Default: written before the call returns, like
        the reference (:600)."""
        self._pool = None
        self._pending = []
        if async_save:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix="ofx-npy")
        self.workspace_dir = workspace_dir
        self.batch_size = batch_size
        self.device = device
        self.pdcnet_model = pdcnet_model.to(device)
        self.pair_dir = os.path.join(workspace_dir, "pdcnet")
        os.makedirs(self.pair_dir, exist_ok=True)
        self.cached_pair = set()
        for f in glob.glob(os.path.join(self.pair_dir, "*.npy")):
            s, t = os.path.basename(f)[:-4].split("-")
            self.cached_pair.add((int(s), int(t)))
        self._frames: "Dict[int, torch.Tensor]" = {}          # frame index -> uint8 RGB [H,W,3] on the device (LRU)
        self._frame_cap = max(2, int(frame_cache))
        self._frames_of = None                                # the video the cache belongs to

    # ---- on-disk pair cache --------------------------------------------------------------------------
    def _pair_path(self, s: int, t: int) -> str:
        return os.path.join(self.pair_dir, f"{s:05d}-{t:05d}.npy")

    def _save(self, s: int, t: int, arr: np.ndarray) -> None:
        if self._pool is None:
            np.save(self._pair_path(s, t), arr)
        else:
            self._pending.append(self._pool.submit(np.save, self._pair_path(s, t), np.array(arr, copy=True)))   # callers mutate `ret` (:995)

    def flush(self) -> None:
        """Wait for queued `.npy` writes (no-op in the default synchronous mode); re-raises a failed write."""
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()

    def __del__(self):
        try:
            self.flush()
            if self._pool is not None:
                self._pool.shutdown(wait=True)
        except Exception:
            pass

    def purge(self):
        self.flush()
        self.cached_pair = set()
        for f in glob.glob(os.path.join(self.pair_dir, "*.npy")):
            os.remove(f)

    def load_cached(self, s, t):
        assert (s, t) in self.cached_pair
        self.flush()
        return np.load(self._pair_path(s, t))

    # ---- device-resident core ------------------------------------------------------------------------
    def _device_frames(self, video, ids: Sequence[int]) -> torch.Tensor:
        """uint8 RGB [n,H,W,3] on the device for the frame indices `ids`; each frame crosses PCIe once while it stays in
        the cache."""
        if self._frames_of is not video:
            self._frames, self._frames_of = {}, video
        missing = [i for i in ids if i not in self._frames]
        if missing:
            host = np.stack([np.ascontiguousarray(video.get_raw_frame(i)[:, :, ::-1]) for i in missing])   # BGR -> RGB (:591-592)
            dev = torch.from_numpy(host).to(self.device)
            for k, i in enumerate(missing):
                self._frames[i] = dev[k]
        for i in ids:                                         # refresh recency (dicts keep insertion order)
            self._frames[i] = self._frames.pop(i)
        out = torch.stack([self._frames[i] for i in ids])
        while len(self._frames) > self._frame_cap:
            self._frames.pop(next(iter(self._frames)))
        return out

    @torch.no_grad()
    def pair_fields(self, video, pairs: Sequence[Tuple[int, int]]):
        """(flow f32[P,H,W,2], confidence f32[P,H,W]) ON THE DEVICE for the (source, target) pairs, in order; flow p
        lives on target's grid and points into source (PDCNetPlus.calc's orientation).  Pairs are processed in slices of
        whole frames so that one slice's frame set fits the engine's per-call limit."""
        pairs = [(int(s), int(t)) for s, t in pairs]
        if not pairs:
            h, w = video.size_hw
            return (torch.zeros((0, h, w, 2), device=self.device), torch.zeros((0, h, w), device=self.device))
        model = self.pdcnet_model
        if not hasattr(model, "calc_pairs"):                  # a foreign of-algo with the reference's duck type only
            flows, confs = [], []
            for batch in chunks(pairs, self.batch_size):
                src = self._device_frames(video, [s for s, _ in batch])
                tgt = self._device_frames(video, [t for _, t in batch])
                fl, cf = model.calc_batch(src, tgt)
                flows.append(torch.as_tensor(fl, device=self.device))
                confs.append(torch.as_tensor(cf, device=self.device))
            return torch.cat(flows), torch.cat(confs)
        ids = sorted({i for p in pairs for i in p})
        local = {g: l for l, g in enumerate(ids)}
        frames = self._device_frames(video, ids)
        return model.calc_pairs(frames, [(local[s], local[t]) for s, t in pairs])

    def calculate_given_pairs(self, video, to_calculate_pairs: List[Tuple[int, int]], s2i_map: Dict[int, int],
                              t2i_map: Dict[int, int], ret: np.ndarray):
        """:585-600: fills `ret[s2i[s], t2i[t]] = (fx, fy, confidence)` for the listed pairs and writes each to the
        pair cache."""
        if not to_calculate_pairs:
            return
        flow, conf = self.pair_fields(video, to_calculate_pairs)
        packed = torch.cat([flow, conf[..., None]], dim=-1).cpu().numpy()      # one D2H copy, [P,H,W,3]
        for k, (s, t) in enumerate(to_calculate_pairs):
            slot = ret[s2i_map[s], t2i_map[t]]
            slot[...] = packed[k]
            self._save(s, t, slot)

    def calcualte_single(self, video, s, t):          # (sic) the reference's spelling, :576
        if (s, t) in self.cached_pair:
            return self.load_cached(s, t)
        ret = np.zeros((1, 1, *video.size_hw, 3), dtype=np.float32)
        self.calculate_given_pairs(video, [(s, t)], {s: 0}, {t: 0}, ret)
        self.cached_pair.add((s, t))
        return ret[0, 0]

    def _matrix(self, video, sources: Sequence[int], targets: Sequence[int]) -> np.ndarray:
        """f32[len(sources), len(targets), H, W, 3]: cached pairs from disk, missing ones computed (and cached), identity
        pairs = zero flow with confidence 1 (:621-623, :649-651)."""
        s2i = {s: i for i, s in enumerate(sources)}
        t2i = {t: j for j, t in enumerate(targets)}
        todo = [(s, t) for s in sources for t in targets if s != t and (s, t) not in self.cached_pair]
        ret = np.zeros((len(sources), len(targets), *video.size_hw, 3), dtype=np.float32)
        fresh = set(todo)
        self.calculate_given_pairs(video, todo, s2i, t2i, ret)
        for s in sources:
            for t in targets:
                if s == t:
                    ret[s2i[s], t2i[t], :, :, 2] = 1
                elif (s, t) not in fresh:
                    ret[s2i[s], t2i[t]] = self.load_cached(s, t)
        self.cached_pair.update(todo)
        return ret

    def calculate_multiple_to_one(self, video, source_indices, target_index: int) -> np.ndarray:
        """-> f32[N, 1, H, W, 3] (:602-625)."""
        return self._matrix(video, list(source_indices.indices), [target_index])

    def calculate_pairwise(self, video, indices) -> np.ndarray:
        """-> f32[N, N, H, W, 3] (:627-653).  Compatibility view (1.06 GB for a 15-frame window at 512x768):
        `keyframe_scores_device` answers KeyframeConv's question without it."""
        return self._matrix(video, list(indices.indices), list(indices.indices))

    # ---- KeyframeConv's reduction, on the device -------------------------------------------------------
    @torch.no_grad()
    def keyframe_scores_device(self, video, indices, save_pairs: bool = False) -> torch.Tensor:
        """`einops.reduce(flow_mat[:, :, :, :, 2], 's t h w -> s', 'sum')` of KeyframeConv (:666) for the window
        `indices`, as a float64 tensor [N] on the device: the N*(N-1) confidence maps are reduced where they are
        produced (`ofx_conf_sum`, f64 accumulation); the identity pair contributes H*W.  Pairs already in the cache are
        read back from disk; with save_pairs=True the fresh ones are also written (keeps the workspace interchangeable
        with the reference at 4.7 MB per pair)."""
        ids = list(indices.indices)
        pos = {s: i for i, s in enumerate(ids)}
        h, w = video.size_hw
        scores = torch.full((len(ids),), float(h * w), dtype=torch.float64, device=self.device)
        fresh = [(s, t) for s in ids for t in ids if s != t and (s, t) not in self.cached_pair]
        if fresh:
            flow, conf = self.pair_fields(video, fresh)
            sums = ops.conf_sum(conf[..., None].contiguous(), 0)             # f64 [P]
            owner = torch.tensor([pos[s] for s, _ in fresh], device=self.device)
            scores.index_add_(0, owner, sums)
            if save_pairs:
                packed = torch.cat([flow, conf[..., None]], dim=-1).cpu().numpy()
                for k, (s, t) in enumerate(fresh):
                    self._save(s, t, packed[k])
                self.cached_pair.update(fresh)
        fresh_set = set(fresh)
        old = [(s, t) for s in ids for t in ids if s != t and (s, t) not in fresh_set]
        for batch in chunks(old, 16):
            conf = torch.from_numpy(np.stack([self.load_cached(s, t)[:, :, 2] for s, t in batch])).to(self.device)
            sums = ops.conf_sum(conf[..., None].contiguous(), 0)
            scores.index_add_(0, torch.tensor([pos[s] for s, _ in batch], device=self.device), sums)
        return scores

    def keyframe_scores(self, flow_mat: np.ndarray) -> np.ndarray:
        """The same reduction for a matrix the caller already holds on the host (f32[N,M,H,W,3])."""
        n, m, h, w, _ = flow_mat.shape
        per = ops.conf_sum(_dev(flow_mat, self.device).reshape(n * m, h, w, 3), 2).reshape(n, m).sum(1)
        return per.cpu().numpy()


def keyframe_conv(pdcnet: PDCNetAux, workspace: str, video, frames, kernel_size: int = 17, stride: int = 8, dilation: int = 2,
                  save_pairs: bool = False):
    """`KeyframeConv` (ofgen_keyframe_inpaint.py:655-674): slide a window over `frames`, and from every window keep the
    frame whose confidence towards all the others is largest.  `workspace` caches the result as `{idx:05d}.png` files
    exactly like the reference (a non-empty directory short-circuits the computation, :656-660).

    Device-resident: per window the frames are uploaded once (and stay cached across the overlapping windows), flows for
    all ordered pairs share one encoder pass per frame, the `s t h w -> s` sum and the arg-max run on the device; the
    only thing that crosses PCIe per window is the winner's position.  Ties go to the earliest frame (np.argmax)."""
    from .workspace import VideoFrameIndices, _write_png_bgr
    if os.path.exists(workspace):
        found = [int(os.path.basename(f).split(".")[0]) for f in glob.glob(os.path.join(workspace, "*.png"))]
        if found:
            return VideoFrameIndices(found)
    else:
        os.makedirs(workspace)
    winners = set()
    for window in frames.conv_indices(kernel_size, stride, dilation):
        scores = pdcnet.keyframe_scores_device(video, window, save_pairs=save_pairs)
        winners.add(window.indices[int(torch.argmax(scores).item())])       # first maximum, like np.argmax (:667)
    for idx in winners:
        _write_png_bgr(os.path.join(workspace, f"{idx:05d}.png"), video.get_raw_frame(idx))
    return VideoFrameIndices(winners)


def KeyframeConv(pdcnet: PDCNetAux, workspace: str, video, frames, kernel_size: int = 17, stride: int = 8, dilation: int = 2,
                 save_pairs: bool = True):
    """The reference's spelling (ofgen_keyframe_inpaint.py:655-674) with the reference's side effect: every pair it computes is
    left in the `pdcnet/{s:05d}-{t:05d}.npy` cache (:600), so later `calculate_multiple_to_one` calls find it.  `keyframe_conv`
    is the same computation without the 4.7 MB-per-pair dump."""
    return keyframe_conv(pdcnet, workspace, video, frames, kernel_size, stride, dilation, save_pairs=save_pairs)
