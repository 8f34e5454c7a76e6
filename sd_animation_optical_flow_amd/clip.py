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
class SynthesisResult:
    frame_indices: List[int]
    flow: List[torch.Tensor]          # per batch: f32[b,H,W,2]
    warped: List[torch.Tensor]        # per batch: uint8[b,H,W,3]
    mask: List[torch.Tensor]          # per batch: uint8[b,H,W]


class FrameSynthesizer:
    """flow -> warp -> mask for a batch of frames against one key frame, device-resident.

    `flow_fn(frames_u8[B,H,W,3], keyframe_u8[H,W,3]) -> (flow f32[B,H,W,2], confidence f32[B,H,W])`;
    the default wraps a `pdcnet_of.PDCNetPlus` (`calc_batch_device`)."""

    def __init__(self, algo=None, flow_fn: Optional[Callable] = None, warp_mode: str = "bilinear", thres: float = 0.95,
                 ksize: int = 7, cmp_gt: bool = False):
        if flow_fn is None:
            if algo is None:
                raise ValueError("need an algo or a flow_fn")

            def flow_fn(frames, key):
                flow, conf, _ = algo.calc_batch_device(key, frames)
                return flow, conf
        self.flow_fn = flow_fn
        self.warp_mode, self.thres, self.ksize, self.cmp_gt = warp_mode, thres, ksize, cmp_gt

    def __call__(self, frames: torch.Tensor, key_raw: torch.Tensor, key_ai: torch.Tensor):
        from . import ops
        flow, conf = self.flow_fn(frames, key_raw)
        warped, mask = ops.warp_and_mask(key_ai, flow, conf, warp_mode=self.warp_mode, thres=self.thres, ksize=self.ksize,
                                         cmp_gt=self.cmp_gt)
        return flow, warped, mask


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
