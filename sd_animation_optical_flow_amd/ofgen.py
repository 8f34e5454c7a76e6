"""Drop-in for the flow / warp / mask helpers of the reference's `ofgen*.py` drivers, MI355X-native.

Same names, argument meaning and mutation behaviour as the reference functions they replace
(file:line in the reference repo):

    RAFT_2().calc(img1_bgr, img2_bgr) -> flow        ofgen_keyframe_inpaint.py:47-71 ; ofgen.py:55-79
    warp_frame(frame, flow)          (RAFT convention) ofgen_keyframe_inpaint.py:92-98
    warp_frame_latent(latent, flow)                  ofgen_keyframe_inpaint.py:100-111
    of_calc(frame1, frame2, algo)                    ofgen_keyframe_inpaint.py:113-133
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
    """ofgen_keyframe_inpaint.py:47-71.  `model` = checkpoint path / state_dict / 'random:<seed>'."""

    def __init__(self, model="../RAFT/models/raft-things.pth", device="cuda", iters: int = 20, alternate_corr: bool = False):
        self.device = torch.device(device)
        self.iters = iters
        self.alternate_corr = alternate_corr
        self.model = RaftEngine(load_checkpoint(model), self.device)

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
    """ofgen_keyframe_inpaint.py:113-133: (flow, confidence, v, log_confidence); v = |flow| with
    v[confidence < 0.9] = 0."""
    flow, confidence, log_confidence = algo.calc(frame1, frame2)
    v = ops.travel_distance(_dev(flow)[None], _dev(confidence)[None], 0.9)[0].cpu().numpy()
    if verbose:
        print("v.max()", v.max(), "v.min()", v.min())
    return flow, confidence, v, log_confidence


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
# PDCNetAux: batched pair flow with the on-disk .npy pair cache
# --------------------------------------------------------------------------------------------------
def chunks(lst, n):
    for i in range(0, len(lst), n):
        yield lst[i:i + n]


class PDCNetAux:
    """ofgen_keyframe_inpaint.py:549-653.  `video` needs `.size_hw` and `.get_raw_frame(i) -> BGR uint8`;
    index containers need `.indices` and `__len__` (the reference's VideoData / VideoFrameIndices)."""

    def __init__(self, pdcnet_model, workspace_dir: str, batch_size: int = 16, device=torch.device("cuda:0"),
                 async_save: bool = False) -> None:
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
        self.cached_pair = set()
        self.batch_size = batch_size
        self.device = device
        self.pdcnet_model = pdcnet_model.to(device)
        self.pair_dir = os.path.join(workspace_dir, "pdcnet")
        if os.path.exists(self.pair_dir):
            for f in glob.glob(os.path.join(self.pair_dir, "*.npy")):
                name = os.path.split(f)[-1].split(".")[0]
                s, t = name.split("-")
                self.cached_pair.add((int(s), int(t)))
        else:
            os.makedirs(self.pair_dir, exist_ok=True)

    def _save(self, s: int, t: int, arr: np.ndarray) -> None:
        path = os.path.join(self.pair_dir, f"{s:05d}-{t:05d}.npy")
        if self._pool is None:
            np.save(path, arr)
        else:
            self._pending.append(self._pool.submit(np.save, path, np.array(arr, copy=True)))   # callers mutate `ret` (:995)

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
        return np.load(os.path.join(self.pair_dir, f"{s:05d}-{t:05d}.npy"))

    def calcualte_single(self, video, s, t):          # (sic) the reference's spelling, :576
        if (s, t) in self.cached_pair:
            return self.load_cached(s, t)
        ret = np.zeros((1, 1, *video.size_hw, 3), dtype=np.float32)
        self.calculate_given_pairs(video, [(s, t)], {s: 0}, {t: 0}, ret)
        self.cached_pair.add((s, t))
        return ret[0, 0]

    def calculate_given_pairs(self, video, to_calculate_pairs: List[Tuple[int, int]], s2i_map: Dict[int, int],
                              t2i_map: Dict[int, int], ret: np.ndarray):
        if len(to_calculate_pairs) > 1 and hasattr(self.pdcnet_model, "calc_pairs"):
            # fast path ("next" row f1): every distinct frame is decoded, uploaded and encoded once; the
            # N*(N-1) ordered pairs of a KeyframeConv window share N feature / context maps
            ids = sorted({i for p in to_calculate_pairs for i in p})
            lm = {g: l for l, g in enumerate(ids)}
            frames = np.stack([np.ascontiguousarray(video.get_raw_frame(i)[:, :, ::-1]) for i in ids])     # RGB
            flow, conf = self.pdcnet_model.calc_pairs(torch.from_numpy(frames).to(self.device),
                                                      [(lm[s], lm[t]) for s, t in to_calculate_pairs])
            flow, conf = flow.cpu().numpy(), conf.cpu().numpy()
            for i, (s, t) in enumerate(to_calculate_pairs):
                si, ti = s2i_map[s], t2i_map[t]
                ret[si, ti, :, :, 0:2] = flow[i]
                ret[si, ti, :, :, 2] = conf[i]
                self._save(s, t, ret[si, ti])
            return
        for pair_batch in chunks(to_calculate_pairs, self.batch_size):
            bs = len(pair_batch)
            inp_source = np.zeros((bs, *video.size_hw, 3), dtype=np.uint8)
            inp_target = np.zeros((bs, *video.size_hw, 3), dtype=np.uint8)
            for i, (s, t) in enumerate(pair_batch):
                inp_source[i] = video.get_raw_frame(s)[:, :, ::-1]     # BGR -> RGB (:591-592)
                inp_target[i] = video.get_raw_frame(t)[:, :, ::-1]
            flow_est, confidence = self.pdcnet_model.calc_batch(torch.from_numpy(inp_source).to(self.device),
                                                                torch.from_numpy(inp_target).to(self.device))
            for i, (s, t) in enumerate(pair_batch):
                si, ti = s2i_map[s], t2i_map[t]
                ret[si, ti, :, :, 0:2] = flow_est[i]
                ret[si, ti, :, :, 2] = confidence[i]
                self._save(s, t, ret[si, ti])

    def calculate_multiple_to_one(self, video, source_indices, target_index: int) -> np.ndarray:
        """-> f32[N, 1, H, W, 3] (:602-625)."""
        to_calc: List[Tuple[int, int]] = []
        n = len(source_indices)
        s2i, t2i = {}, {target_index: 0}
        for i, s in enumerate(source_indices.indices):
            s2i[s] = i
            if s != target_index and (s, target_index) not in self.cached_pair:
                to_calc.append((s, target_index))
        ret = np.zeros((n, 1, *video.size_hw, 3), dtype=np.float32)
        self.calculate_given_pairs(video, to_calc, s2i, t2i, ret)
        for i, s in enumerate(source_indices.indices):
            if s != target_index:
                if (s, target_index) in self.cached_pair:
                    ret[i, 0] = self.load_cached(s, target_index)
            else:
                ret[i, 0, :, :, 0:2] = 0
                ret[i, 0, :, :, 2] = 1
        self.cached_pair.update(to_calc)
        return ret

    def calculate_pairwise(self, video, indices) -> np.ndarray:
        """-> f32[N, N, H, W, 3] (:627-653)."""
        n = len(indices)
        to_calc: List[Tuple[int, int]] = []
        s2i, t2i = {}, {}
        for i, s in enumerate(indices.indices):
            s2i[s] = i
            for j, t in enumerate(indices.indices):
                t2i[t] = j
                if s != t and (s, t) not in self.cached_pair:
                    to_calc.append((s, t))
        ret = np.zeros((n, n, *video.size_hw, 3), dtype=np.float32)
        self.calculate_given_pairs(video, to_calc, s2i, t2i, ret)
        for i, s in enumerate(indices.indices):
            for j, t in enumerate(indices.indices):
                if s != t:
                    if (s, t) in self.cached_pair:
                        ret[i, j] = self.load_cached(s, t)
                else:
                    ret[i, j, :, :, 0:2] = 0
                    ret[i, j, :, :, 2] = 1
        self.cached_pair.update(to_calc)
        return ret

    def keyframe_scores(self, flow_mat: np.ndarray) -> np.ndarray:
        """`einops.reduce(flow_mat[..., 2], 's t h w -> s', 'sum')` of KeyframeConv (:666) on the device."""
        n, m, h, w, _ = flow_mat.shape
        per = ops.conf_sum(_dev(flow_mat, self.device).reshape(n * m, h, w, 3), 2).reshape(n, m).sum(1)
        return per.cpu().numpy()
