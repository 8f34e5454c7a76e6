"""The hot path end to end over a workspace, device-resident: the non-generative half of the reference's
`run_exp` / `generate_ai_frame_with_ref_warp_and_inpaint*` loop (ofgen_keyframe_inpaint.py:861-1100, 1123-1262).

For a clip stored as a `workspace.VideoData`:

  1. key frames             `keyframes.frame_generator` decisions (`ofx_detect_edges`, `ofx_abs_diff_sum_u8`)
  2. for every other frame  flow against its key frame + forward-backward confidence (`PDCNetPlus.calc_batch_device`),
                            backward warp of the AI key frame, low-confidence inpaint mask (`ofx_warp_and_mask`)
  3. SD-inpaint inputs      Pillow-exact mask blur / composite / latent mask (`handoff.prepare_inpaint_inputs`) and,
                            when a VAE is given, the first-stage latent (`vae.VaeEncoder`)

Everything between reading a PNG and handing tensors to the diffusion model stays in HBM.  What the diffusion model does
with them (UNet, sampler, ControlNet) is out of scope: `render` is the caller's hook for it -- by
default the "rendered" frame is the warped AI key frame with the raw frame pasted where the mask says repaint.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import clip, handoff, keyframes, ops


@dataclass
class FramePacket:
    """What the SD stage receives for one frame (all tensors on the device)."""
    index: int
    key_index: int
    flow: torch.Tensor                 # f32 [H,W,2], on the frame's grid, pointing into the key frame
    confidence: torch.Tensor           # f32 [H,W]
    warped: torch.Tensor               # u8 [H,W,3] BGR: AI key frame warped onto this frame
    mask: torch.Tensor                 # u8 [H,W], 255 = repaint
    inpaint: Dict[str, torch.Tensor] = field(default_factory=dict)   # handoff.prepare_inpaint_inputs (+ "init_latent")


def _paste_raw(pkt: FramePacket, raw_bgr: torch.Tensor) -> torch.Tensor:
    """Default stand-in for the inpainting model: raw pixels where the mask asks for a repaint (merge_images 'naive')."""
    return ops.merge_images(pkt.warped[None], raw_bgr[None], pkt.mask[None])[0]


def default_io_threads(cap: int = 6) -> int:
    """Decode (and, separately, encode) threads per rank: the cores this process may run on, shared by the ranks of this host
    (`LOCAL_WORLD_SIZE`, as torchrun sets it) and by the two pools -- min(cap, cores // local ranks // 2), at least 1.  One host
    with 8 ranks and 128 usable cores gets 6 + 6 per rank; the 16-core single-GPU boxes of this pool get 6 + 6 too; 8 ranks
    squeezed onto 16 cores get 1 + 1 instead of 96 threads fighting over them."""
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    try:
        local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local = 1
    return max(1, min(int(cap), cores // local // 2))


def split_edge_batches(units: list, edge: int) -> list:
    """`units`: a rank's load units in consumption order, `(segment, 'key' | None | [frame ids])`.  The first and the last list
    of frame ids are cut into a piece of `edge` frames (first: in front; last: behind) and the rest; see `ClipPipeline`."""
    idx = [i for i, (_, what) in enumerate(units) if isinstance(what, list)]
    if edge <= 0 or not idx:
        return units
    first, last = idx[0], idx[-1]
    out = []
    for i, (seg, what) in enumerate(units):
        pieces = [what]
        if isinstance(what, list) and len(what) > edge:
            if i == first and i == last:
                pieces = [what[:edge], what[edge:-edge], what[-edge:]] if len(what) > 2 * edge else [what[:edge], what[edge:]]
            elif i == first:
                pieces = [what[:edge], what[edge:]]
            elif i == last:
                pieces = [what[:-edge], what[-edge:]]
        out += [(seg, p) for p in pieces]
    return out


class ClipPipeline:
    """`algo`: a `pdcnet_of.PDCNetPlus`; `vae`: optional `vae.VaeEncoder`; `render(packet, raw_bgr) -> u8 [H,W,3]` turns a
    packet into the AI frame (default: `_paste_raw`); key frames are rendered by `render_key(raw_bgr) -> u8 [H,W,3]`
    (default: identity).  `batch` frames share one executor call per key frame.  Rank-aware: under an initialised
    `torch.distributed` group every rank runs the same `run(video)` and takes its share (`packets`) -- `run` and `packets` are
    COLLECTIVE calls: every rank of `group` must make them, with the same flags.  `io_threads` / `prefetch`: the asynchronous host
    side (`hostio.FrameLoader` / `FrameWriter`); the output is byte-identical with it off (`io_threads=0`).  `io_threads=None`
    sizes the pools for the ranks that share this host: `default_io_threads()`.
    `edge_batch`: the first and the last batch of a rank's share are cut into a piece of `edge_batch` frames and the rest, so that
    the kernels start after `edge_batch` decoded frames instead of a whole batch and only `edge_batch` frames are left to encode
    behind the last kernel (a rank whose share is one 64-frame batch -- BASELINE configs[3] -- otherwise pays a batch of decode in
    front of and a batch of encode behind its 0.6 s of kernels); 0 switches it off.  Part of the plan, not of the host side: it
    applies with `io_threads=0` too, so the two stay byte-identical."""

    def __init__(self, algo, vae=None, render: Optional[Callable] = None, render_key: Optional[Callable] = None, batch: int = 64,
                 warp_mode: str = "bilinear", thres: float = 0.95, ksize: int = 7, mask_blur: float = 4.0, device=None,
                 io_threads: Optional[int] = None, prefetch: int = 2, edge_batch: int = 16, group=None, share_pool: bool = True):
        self.algo, self.vae = algo, vae
        self.render = render or _paste_raw
        self.render_key = render_key or (lambda raw: raw)
        self.batch, self.warp_mode, self.thres, self.ksize, self.mask_blur = int(batch), warp_mode, float(thres), int(ksize), float(mask_blur)
        self.device = torch.device(device if device is not None else algo.device)
        self.group = group
        self.share_pool = bool(share_pool)
        # host side (hostio.py): PNG decode / encode on `io_threads` workers each, `prefetch` batches decoded ahead of the GPU;
        # io_threads = 0 runs all of it inline on the calling thread (the reference's behaviour)
        self.io_threads = default_io_threads() if io_threads is None else max(0, int(io_threads))
        self.prefetch, self.edge_batch = max(1, int(prefetch)), max(0, int(edge_batch))
        # the compute step is the same object bench.py times: with the bilinear warp the AI key frame is warped inside the flow
        # network's convex upsample (one kernel), the cubic modes take upsample -> ofx_warp_and_mask
        self.synth = clip.FrameSynthesizer(algo, warp_mode=warp_mode, thres=self.thres, ksize=self.ksize, bgr=True) if algo is not None else None

    def _rank_world(self):
        """(rank, world) of THIS pipeline's process group: the plan is sized by it and owners are ranks inside it."""
        if self.group is None or not (dist.is_available() and dist.is_initialized()):
            return clip.dist_info()
        r = dist.get_rank(self.group)
        if r < 0:
            raise RuntimeError("ClipPipeline: this process is not a member of the pipeline's process group")
        return r, dist.get_world_size(self.group)

    def _global_rank(self, group_rank: int) -> int:
        """`src` of a collective is a GLOBAL rank; plan owners are ranks of `group`."""
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    def key_frame_flags(self, video, th: float = 8.5) -> List[bool]:
        """One flag per WORKSPACE frame (flags[i] belongs to `video.get_raw_frame(i)`).  The workspace is already decimated
        (`keep_every` was applied when it was extracted, ofgen_keyframe_inpaint.py:342-368), so every frame is examined here and
        the reference's 3601-frame cap of the extraction loop does not apply."""
        n = video.num_frames
        frames = (video.get_raw_frame(i) for i in range(n))
        flags = [k for _, k, _ in keyframes.frame_generator(frames, fps=getattr(video, "fps", 30.0), th=th, keep_every=1,
                                                           device=self.device, max_decoded=n)]
        assert len(flags) == n, (len(flags), n)
        return flags

    @torch.no_grad()
    def process_batch(self, key_raw: torch.Tensor, key_ai: torch.Tensor, raws: torch.Tensor, ids: List[int], key_index: int):
        """flow + confidence -> warp + mask -> SD-inpaint inputs for `raws` (u8 [b,H,W,3] BGR, device) against one key frame;
        returns the packets.  The one compute step of the pipeline (tests of the rank logic substitute it on the CPU)."""
        # source = key frame, target = frames: flow on each frame's grid pointing into the key frame (BGR in)
        flow, conf, warped, mask = self.synth.synthesize(raws, key_raw, key_ai.contiguous())
        inp = handoff.prepare_inpaint_inputs(warped, raws, mask, mask_blur=self.mask_blur, device=self.device)
        if self.vae is not None:
            inp["init_latent"] = self.vae.get_first_stage_encoding(inp["image"])
        return [FramePacket(t, key_index, flow[k], conf[k], warped[k], mask[k], {name: v[k] for name, v in inp.items()})
                for k, t in enumerate(ids)]

    @torch.no_grad()
    def packets(self, video, flags: List[bool], writer=None):
        """Yields (FramePacket | None, raw_bgr tensor, index) for THIS rank's share of the clip: None for the key frames this
        rank renders.  One process: everything.  Under `torch.distributed` (one process per GPU): `clip.plan_segments` spreads
        the key-frame segments over the ranks; the rank that owns a segment renders its key frame, and only when a segment had
        to be cut does the rendered key frame travel -- one `broadcast_keyframe` (RCCL on the `nccl` backend), the single
        collective of the path.  Raw frames come from the workspace, which the ranks of a node share.
        Host side (`hostio.FrameLoader`): the PNGs of the next `prefetch` batches are decoded on a thread pool into pinned
        buffers while the current batch's kernels run, and uploaded on a copy stream.  `writer`: where rendered key frames go
        (`run` passes its asynchronous `hostio.FrameWriter`; default: synchronous `video.put_ai_frame`)."""
        from . import hostio
        rank, world = self._rank_world()
        # this rank's load units, in the order they are consumed: (segment, 'key' | list of frame ids)
        units = []
        plans = clip.plan_segments(flags, world)
        for seg in plans:
            mine = seg.frames[rank]
            owner = rank == seg.owner
            if not (owner or mine or seg.needs_broadcast):
                continue
            units.append((seg, "key" if (owner or mine) else None))
            for b0 in range(0, len(mine), self.batch):
                units.append((seg, mine[b0:b0 + self.batch]))
        units = split_edge_batches(units, self.edge_batch)
        loader = hostio.FrameLoader(video, self.device, threads=self.io_threads, slots=self.prefetch + 1, batch=self.batch,
                                    pool=getattr(writer, "pool", None) if self.share_pool else None)
        tickets = {}

        def ahead(k: int) -> None:
            for j in range(k, min(len(units), k + 1 + self.prefetch)):
                if j not in tickets and units[j][1] is not None:
                    what = units[j][1]
                    tickets[j] = loader.request([units[j][0].key] if what == "key" else what)
        try:
            key_raw = key_ai = None
            for k, (seg, what) in enumerate(units):
                ahead(k)
                if what is None or what == "key":
                    owner = rank == seg.owner
                    shape = (*video.size_hw, 3)
                    key_raw = loader.fetch(tickets.pop(k))[0] if what == "key" else None
                    if owner:
                        key_ai = self.render_key(key_raw).contiguous()
                        if writer is not None:
                            writer.put(seg.key, key_ai)
                        else:
                            video.put_ai_frame(seg.key, key_ai.cpu().numpy())
                        yield None, key_raw, seg.key
                    else:
                        key_ai = torch.empty(shape, dtype=torch.uint8, device=self.device)
                    if seg.needs_broadcast:
                        # the one collective of the path; every rank of the group takes part, also those with no frame of this segment
                        clip.broadcast_keyframe([key_ai], src=self._global_rank(seg.owner), group=self.group)
                    continue
                ids = what
                raws = loader.fetch(tickets.pop(k))
                for pkt, raw in zip(self.process_batch(key_raw, key_ai, raws, ids, seg.key), raws):
                    yield pkt, raw, pkt.index
        finally:
            loader.close()

    def _check_collective(self) -> None:
        """`run` / `shared_flags` under more than one rank are collective: refuse set-ups that would hang instead of failing."""
        import torch.distributed as dist
        backend = dist.get_backend(self.group)
        if backend == "nccl" and self.device.type != "cuda":
            raise RuntimeError(f"ClipPipeline on device '{self.device}' under the '{backend}' backend: RCCL broadcasts need the rank's "
                               "GPU tensors -- construct the pipeline with device='cuda:<local rank>'")

    def shared_flags(self, video, th: float = 8.5) -> List[bool]:
        """Key-frame flags every rank agrees on: rank 0 runs the detector and broadcasts its decisions (one byte per frame).  Each
        rank deriving them for itself would repeat the detection world-size times and -- should two ranks ever disagree on one
        frame (another GPU, driver or library build) -- issue different broadcast sequences and hang without a diagnostic.
        COLLECTIVE: every rank of the group calls it; the other ranks wait in the broadcast while rank 0 detects (4 900 frames/s
        from host memory, DESIGN.md: a 10-minute collective timeout covers 2.9 M frames -- pass explicit `flags` to `run`, or a
        group created with a longer timeout, for anything beyond that)."""
        rank, world = self._rank_world()
        if world == 1:
            return self.key_frame_flags(video, th)
        import torch.distributed as dist
        self._check_collective()
        n = video.num_frames
        t = torch.zeros((n,), dtype=torch.uint8, device=self.device)
        if rank == 0:
            t.copy_(torch.tensor(self.key_frame_flags(video, th), dtype=torch.uint8))
        dist.broadcast(t, src=self._global_rank(0), group=self.group)
        return [bool(v) for v in t.cpu().tolist()]

    def run(self, video, flags: Optional[List[bool]] = None) -> List[int]:
        """Processes this rank's share of the workspace (all of it in one process); writes `ai-frames/{n:05d}.png` for the frames
        it owns -- results stay with the owning rank; returns the key-frame indices this rank rendered.  Rendered frames leave
        through `hostio.FrameWriter`: D2H into pinned memory on a side stream, PNG encoding on worker threads, one `flush()` at
        the end (every file is on disk when `run` returns).
        COLLECTIVE under `torch.distributed`: every rank of the group calls `run` on the same workspace (with `flags=None` the
        first thing it does is the broadcast of `shared_flags`); a rank that stays away leaves the others waiting for the
        group's collective timeout."""
        from . import hostio
        if self._rank_world()[1] > 1:
            self._check_collective()
        flags = flags if flags is not None else self.shared_flags(video)
        # staging slots for three batches: `put` must never wait for an encoder while the next batch's kernels are still to be enqueued
        # one pool of 2 x io_threads workers for decode AND encode (round 6): a rank's decoders are idle while its last frames are being
        # encoded and its encoders while the first are being decoded -- with a single 64-frame batch per rank (BASELINE configs[3])
        # those two ends are a seventh of the run
        pool = hostio.shared_pool(self.io_threads) if self.share_pool else None
        writer = hostio.FrameWriter(video, self.device, threads=self.io_threads, slots=3 * self.batch, pool=pool)
        keys = []
        try:
            for pkt, raw, idx in self.packets(video, flags, writer=writer):
                if pkt is None:
                    keys.append(idx)
                    continue
                writer.put(idx, self.render(pkt, raw))
        except BaseException:
            try:
                writer.close()                               # stop the encoders; what they have to say must not replace the failure above
            except Exception:
                pass
            if pool is not None:
                pool.shutdown(wait=True, cancel_futures=True)
            raise
        try:
            writer.close()                                   # flush: joins the encoders, re-raises their first failure
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
        return keys
