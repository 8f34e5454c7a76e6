"""Frame-parallel clip processing: the multi-GPU layer of the hot path.

The reference has no distributed code at all (one process, one GPU).  Frame<->keyframe
pairs are independent units (`calculate_given_pairs` treats them as an unordered list,
ofgen_keyframe_inpaint.py:585-600), so a clip shards by frame index with no data-path collective:

  * one process per GPU (`torchrun`), rank r owns a contiguous block of the clip's frames;
  * the only exchange is ONE broadcast per key frame of the rendered AI key frame (uint8[H,W,3],
    1.18 MB at 512x768) -- and of the raw key frame when ranks do not share storage -- from the rank
    that rendered it.  `torch.distributed.broadcast` on the `nccl` backend is RCCL over xGMI; the
    payload is latency-bound, so no bucketing / ring tuning applies;
  * every rank then runs flow -> warp -> mask for its frames against the key frame, entirely in HBM;
    results stay on the owning rank.

The compute step is injectable so that the sharding/broadcast logic is covered by world_size-2
`gloo` tests on CPU (tests/test_clip_dist.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(num_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced block partition: the first (num_items % world) ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(num_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def dist_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_keyframe(tensors: Sequence[torch.Tensor], src: int = 0, group=None) -> None:
    """In-place broadcast of the key-frame tensors from `src` to every rank (no-op on one rank).
    The tensors must already be allocated with identical shapes/dtypes on every rank."""
    rank, world = dist_info()
    if world == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


@dataclass
class SegmentPlan:
    """One key frame and the frames that follow it up to the next key frame, spread over the ranks."""
    key: int                           # workspace index of the key frame
    owner: int                         # rank that renders the key frame (and broadcasts it when the segment is split)
    frames: List[List[int]]            # per rank: the segment's frame indices that rank processes (possibly empty)

    @property
    def ranks(self) -> List[int]:
        return [r for r, f in enumerate(self.frames) if f]

    @property
    def needs_broadcast(self) -> bool:
        return any(r != self.owner for r in self.ranks)


def plan_segments(flags: Sequence[bool], world: int) -> List[SegmentPlan]:
    """Key-frame segments -> ranks.  Segment lengths are content-dependent (the detector's gaps,
    ofgen_keyframe_inpaint.py:443-469), so contiguous frame blocks would leave ranks idle: segments are taken longest first
    and poured into the least-loaded rank up to ceil(frames / world) -- whole segments stay on one rank (no exchange at all)
    wherever the balance allows, a segment is cut only when it must be, and then the rank holding its largest piece
    renders the key frame and broadcasts it.  Deterministic: every rank computes the same plan from the same flags.
    Returned in key-frame order."""
    if world <= 0:
        raise ValueError(f"bad world size {world}")
    n = len(flags)
    if n and not flags[0]:
        raise ValueError("the first frame of a clip is a key frame")
    segs = []
    i = 0
    while i < n:
        j = i + 1
        while j < n and not flags[j]:
            j += 1
        segs.append((i, list(range(i + 1, j))))
        i = j
    total = sum(len(f) for _, f in segs)
    target = max(1, -(-total // world))
    load = [0] * world
    plans = []
    for key, frames in sorted(segs, key=lambda kf: (-len(kf[1]), kf[0])):
        per = [[] for _ in range(world)]
        rest = frames
        while rest:
            r = min(range(world), key=lambda q: (load[q], q))
            take = max(1, min(len(rest), target - load[r]))
            per[r] = per[r] + rest[:take]
            load[r] += take
            rest = rest[take:]
        # a key frame nobody warps from (two key frames in a row) still has to be rendered by someone
        owner = max(range(world), key=lambda q: (len(per[q]), -q)) if frames else min(range(world), key=lambda q: (load[q], q))
        plans.append(SegmentPlan(key, owner, per))
    plans.sort(key=lambda p_: p_.key)
    return plans


@dataclass
class SynthesisResult:
    frame_indices: List[int]
    flow: List[torch.Tensor]          # per batch: f32[b,H,W,2]
    warped: List[torch.Tensor]        # per batch: uint8[b,H,W,3]
    mask: List[torch.Tensor]          # per batch: uint8[b,H,W]


class FrameSynthesizer:
    """flow -> warp -> mask for a batch of frames against one key frame, device-resident: THE step of the hot path
    (`bench.py` times exactly this object, `pipeline.ClipPipeline` and `process_clip` run it).

    Three ways to get flow + confidence, in this order:
      * `engine` (a `raft.RaftEngine`) with a `confidence` passed to the call -- the flow network alone, one flow per pair;
        the confidence comes from the caller (a PDCNet-style confidence head, or `bench.py`'s synthetic one);
      * `algo` (a `pdcnet_of.PDCNetPlus`): `calc_batch_device` -- flow in both directions + forward-backward confidence;
      * `flow_fn(frames_u8[B,H,W,3], keyframe_u8[H,W,3]) -> (flow f32[B,H,W,2], confidence f32[B,H,W])`: anything else
        (the CPU tests of the rank logic).
    With `warp_mode='bilinear'` (the north star's warp) and one shared AI key frame the warp happens INSIDE the convex upsample of
    the flow network (`ofx_raft_forward_warp` / `ofx_raft_forward_pairs_warp`): the full-resolution flow is never re-read.  The
    cubic modes and `flow_fn` run upsample, then `ops.warp_and_mask`."""

    def __init__(self, algo=None, flow_fn: Optional[Callable] = None, warp_mode: str = "bilinear", thres: float = 0.95,
                 ksize: int = 7, cmp_gt: bool = False, engine=None, iters: int = 20, bgr: bool = False, fuse_warp: bool = True):
        if flow_fn is None and algo is None and engine is None:
            raise ValueError("need an engine, an algo or a flow_fn")
        self.algo, self.engine, self.flow_fn = algo, engine, flow_fn
        self.iters, self.bgr, self.fuse_warp = int(iters), bool(bgr), bool(fuse_warp)
        self.warp_mode, self.thres, self.ksize, self.cmp_gt = warp_mode, thres, ksize, cmp_gt

    def __call__(self, frames: torch.Tensor, key_raw: torch.Tensor, key_ai: torch.Tensor, confidence: Optional[torch.Tensor] = None):
        """-> (flow f32[B,H,W,2] on each frame's grid pointing into the key frame, warped u8 [B,H,W,3], mask u8 [B,H,W])."""
        flow, _, warped, mask = self.synthesize(frames, key_raw, key_ai, confidence)
        return flow, warped, mask

    def synthesize(self, frames: torch.Tensor, key_raw: torch.Tensor, key_ai: torch.Tensor, confidence: Optional[torch.Tensor] = None):
        """-> (flow, confidence f32[B,H,W], warped, mask)."""
        from . import ops
        H, W = frames.shape[-3], frames.shape[-2]
        fuse = (self.fuse_warp and self.warp_mode == "bilinear" and self.flow_fn is None and key_ai.dim() == 3 and H % 8 == 0
                and W % 8 == 0)
        if self.flow_fn is not None:
            flow, conf = self.flow_fn(frames, key_raw)
            warped = None
        elif confidence is not None and self.engine is not None:
            conf = confidence
            if fuse:
                flow, warped = self.engine.forward(frames, key_raw, iters=self.iters, bgr=self.bgr, warp_frame=key_ai)
            else:
                from .pdcnet_of import _unpad
                flow, warped = _unpad(self.engine.forward(frames, key_raw, iters=self.iters, bgr=self.bgr), H, W), None   # (padded to /8 inside)
        elif self.algo is not None:
            if fuse:
                flow, conf, _, warped = self.algo.calc_batch_device(key_raw, frames, bgr=self.bgr, warp_frame=key_ai)
            else:
                (flow, conf, _), warped = self.algo.calc_batch_device(key_raw, frames, bgr=self.bgr), None
            if confidence is not None:
                conf = confidence
        else:
            raise ValueError("an engine-only FrameSynthesizer needs the confidence passed to the call")
        if warped is not None:
            mask = ops.generate_mask(conf.contiguous(), None, self.thres, self.ksize, cmp_gt=self.cmp_gt)
        else:
            warped, mask = ops.warp_and_mask(key_ai.contiguous(), flow.contiguous(), conf.contiguous(), warp_mode=self.warp_mode,
                                             thres=self.thres, ksize=self.ksize, cmp_gt=self.cmp_gt)
        return flow, conf, warped, mask


def process_clip(frames: torch.Tensor, key_raw: torch.Tensor, key_ai: torch.Tensor, step: Callable, batch_size: int = 64,
                 key_src: int = 0, sharded_input: bool = False, group=None) -> SynthesisResult:
    """Run `step(frames_batch, key_raw, key_ai) -> (flow, warped, mask)` over this rank's share of a clip.

    frames: uint8 [T,H,W,3].  With `sharded_input=False` every rank holds the whole clip and takes
    `shard_range(T, rank, world)`; with True, `frames` already is this rank's block.
    key_raw / key_ai: allocated on every rank; their contents on rank `key_src` are broadcast."""
    rank, world = dist_info()
    broadcast_keyframe([key_raw, key_ai], src=key_src, group=group)
    total = frames.shape[0]
    if sharded_input:
        idx = list(range(total))
        local = frames
        base = 0
    else:
        r = shard_range(total, rank, world)
        idx = list(r)
        local = frames[r.start:r.stop]
        base = r.start
    out = SynthesisResult(frame_indices=[base + i for i in range(len(idx))] if sharded_input else idx, flow=[], warped=[], mask=[])
    for i0 in range(0, local.shape[0], batch_size):
        fl, wp, mk = step(local[i0:i0 + batch_size].contiguous(), key_raw, key_ai)
        out.flow.append(fl)
        out.warped.append(wp)
        out.mask.append(mk)
    return out
