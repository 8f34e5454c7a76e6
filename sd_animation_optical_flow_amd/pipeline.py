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


class ClipPipeline:
    """`algo`: a `pdcnet_of.PDCNetPlus`; `vae`: optional `vae.VaeEncoder`; `render(packet, raw_bgr) -> u8 [H,W,3]` turns a
    packet into the AI frame (default: `_paste_raw`); key frames are rendered by `render_key(raw_bgr) -> u8 [H,W,3]`
    (default: identity).  `batch` frames share one executor call per key frame.  Rank-aware: under an initialised
    `torch.distributed` group every rank runs the same `run(video)` and takes its share (`packets`)."""

    def __init__(self, algo, vae=None, render: Optional[Callable] = None, render_key: Optional[Callable] = None, batch: int = 64,
                 warp_mode: str = "bilinear", thres: float = 0.95, ksize: int = 7, mask_blur: float = 4.0, device=None):
        self.algo, self.vae = algo, vae
        self.render = render or _paste_raw
        self.render_key = render_key or (lambda raw: raw)
        self.batch, self.warp_mode, self.thres, self.ksize, self.mask_blur = int(batch), warp_mode, float(thres), int(ksize), float(mask_blur)
        self.device = device if device is not None else algo.device

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
        flow, conf, _ = self.algo.calc_batch_device(key_raw, raws, bgr=True)
        warped, mask = ops.warp_and_mask(key_ai.contiguous(), flow.contiguous(), conf.contiguous(), warp_mode=self.warp_mode,
                                         thres=self.thres, ksize=self.ksize)
        inp = handoff.prepare_inpaint_inputs(warped, raws, mask, mask_blur=self.mask_blur, device=self.device)
        if self.vae is not None:
            inp["init_latent"] = self.vae.get_first_stage_encoding(inp["image"])
        return [FramePacket(t, key_index, flow[k], conf[k], warped[k], mask[k], {name: v[k] for name, v in inp.items()})
                for k, t in enumerate(ids)]

    @torch.no_grad()
    def packets(self, video, flags: List[bool]):
        """Yields (FramePacket | None, raw_bgr tensor, index) for THIS rank's share of the clip: None for the key frames this
        rank renders.  One process: everything.  Under `torch.distributed` (one process per GPU): `clip.plan_segments` spreads
        the key-frame segments over the ranks; the rank that owns a segment renders its key frame, and only when a segment had
        to be cut does the rendered key frame travel -- one `broadcast_keyframe` (RCCL on the `nccl` backend), the single
        collective of the path.  Raw frames come from the workspace, which the ranks of a node share."""
        rank, world = clip.dist_info()
        for seg in clip.plan_segments(flags, world):
            mine = seg.frames[rank]
            owner = rank == seg.owner
            if not (owner or mine or seg.needs_broadcast):
                continue
            shape = (*video.size_hw, 3)
            key_raw = torch.from_numpy(video.get_raw_frame(seg.key)).to(self.device) if (owner or mine) else None
            if owner:
                key_ai = self.render_key(key_raw).contiguous()
                video.put_ai_frame(seg.key, key_ai.cpu().numpy())
                yield None, key_raw, seg.key
            else:
                key_ai = torch.empty(shape, dtype=torch.uint8, device=self.device)
            if seg.needs_broadcast:
                # the one collective of the path; every rank of the group takes part, also those with no frame of this segment
                clip.broadcast_keyframe([key_ai], src=seg.owner)
            for b0 in range(0, len(mine), self.batch):
                ids = mine[b0:b0 + self.batch]
                raws = torch.from_numpy(np.stack([video.get_raw_frame(t) for t in ids])).to(self.device)
                for pkt, raw in zip(self.process_batch(key_raw, key_ai, raws, ids, seg.key), raws):
                    yield pkt, raw, pkt.index

    def run(self, video, flags: Optional[List[bool]] = None) -> List[int]:
        """Processes this rank's share of the workspace (all of it in one process); writes `ai-frames/{n:05d}.png` for the frames
        it owns -- results stay with the owning rank; returns the key-frame indices this rank rendered."""
        flags = flags if flags is not None else self.key_frame_flags(video)
        keys = []
        for pkt, raw, idx in self.packets(video, flags):
            if pkt is None:
                keys.append(idx)
                continue
            video.put_ai_frame(idx, self.render(pkt, raw).cpu().numpy())
        return keys
