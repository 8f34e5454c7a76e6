#!/usr/bin/env python3
"""Generates the committed golden vectors.  Run ONLY in the build container, where the reference is
mounted read-only at /root/reference:

    python tests/golden/make_golden.py

What it pins
  raft_ref_128x160.npz   outputs of the REAL reference RAFT (`/root/reference/RAFT/core`, imported, not
                         copied) loaded with `oracle.raft_oracle.init_state_dict(0)`: feature maps,
                         context split, correlation pyramid (level 3 in full + per-level checksums),
                         CorrBlock lookups at integer / fractional / out-of-range coordinates, one update
                         step, the convex upsample, and the final 20-iteration flow.
                         Also asserts, at generation time, that the oracle restatement reproduces every
                         one of them (so a stale oracle cannot ship).
  raft_ref_trainbn_128x160.npz
                         the same reference module driven EXACTLY as `RAFT_2` drives it (ofgen_keyframe_inpaint.py:47-71):
                         `DataParallel(RAFT(args))`, `load_state_dict`, NO `.eval()` -- so the context encoder's BatchNorm
                         layers run in train mode on the call's one image -- `InputPadder`, `iters=20, test_mode=True`,
                         `flow_up[0].permute(1,2,0)` without un-padding.  Two BGR frame pairs: 128x160 and 132x156 (padded
                         to 136x160; the reference itself returns NaN once a pyramid level is one pixel wide --
                         `bilinear_sampler` divides by W-1, utils.py:62-63 -- i.e. for frames under 128 px).  Asserts that `oracle.raft_oracle.raft2_calc(cnet_norm="batch")` reproduces both.
  warp_grid_sample.npz   torch.nn.functional.grid_sample (bilinear / bicubic, zeros, align_corners=True)
                         outputs for the warp modes that have an importable third-party implementation.
  cv2_semantics.npz      cv2 is not installed anywhere we can reach and is un-pinned upstream ("exact-cv2
                         parity unpinned"): these vectors are produced by the oracle's restatement of
                         OpenCV's published algorithms and freeze it against regressions.

Nothing of the reference's source text is stored -- only inputs and outputs.
"""
import hashlib
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/RAFT/core"

from oracle import mask_oracle as MO   # noqa: E402
from oracle import raft_oracle as RO   # noqa: E402
from oracle import warp_oracle as WO   # noqa: E402


def sd_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def make_raft():
    warnings.filterwarnings("ignore")
    sys.path.insert(0, REF)
    from corr import CorrBlock          # reference module, imported from where it lies
    from raft import RAFT

    class NS:
        def __contains__(self, m):
            return hasattr(self, m)
    a = NS()
    a.small, a.mixed_precision, a.alternate_corr = False, False, False
    model = RAFT(a).eval()
    sd = RO.init_state_dict(0)
    model.load_state_dict(sd, strict=True)

    H, W = 128, 160
    g = torch.Generator().manual_seed(7)
    base = torch.rand((1, 3, H + 16, W + 16), generator=g)
    base = F.avg_pool2d(base, 5, 1, 2)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round()
    img1 = base[:, :, 8:8 + H, 8:8 + W].contiguous()
    img2 = base[:, :, 6:6 + H, 11:11 + W].contiguous()
    out = {"image1": img1.to(torch.uint8).numpy(), "image2": img2.to(torch.uint8).numpy(), "state_dict_sha256": sd_digest(sd)}
    with torch.no_grad():
        i1 = 2 * (img1 / 255.0) - 1.0
        i2 = 2 * (img2 / 255.0) - 1.0
        fmap1, fmap2 = model.fnet([i1, i2])
        cnet = model.cnet(i1)
        net, inp = torch.split(cnet, [128, 128], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        cb = CorrBlock(fmap1, fmap2, radius=4)
        h, w = H // 8, W // 8
        coords0 = RO.coords_grid(1, h, w)
        gen = torch.Generator().manual_seed(8)
        jitter = (torch.rand((1, 2, h, w), generator=gen) - 0.5)
        lookups = {"int": cb(coords0), "frac": cb(coords0 + jitter * 5.0), "far": cb(coords0 + jitter * 60.0)}
        corr0 = lookups["int"]
        net1, mask1, delta1 = model.update_block(net, inp, corr0, coords0 - coords0)
        up1 = model.upsample_flow(delta1, mask1)
        flow_low, flow_up = model(img1, img2, iters=20, test_mode=True)

        # the oracle must reproduce all of it
        tr = {}
        lo_o, up_o = RO.raft_forward(sd, img1, img2, 20, trace=tr)
        chk = lambda a_, b_, tol, nm: (float((a_ - b_).abs().max()) < tol) or (_ for _ in ()).throw(AssertionError(nm))
        chk(tr["fmap1"], fmap1, 1e-4, "fmap1")
        chk(tr["fmap2"], fmap2, 1e-4, "fmap2")
        chk(tr["net0"], net, 1e-4, "net")
        chk(tr["inp"], inp, 1e-4, "inp")
        for l in range(4):
            chk(tr["pyramid"][l], cb.corr_pyramid[l], 1e-4, f"pyr{l}")
        pyr_o = RO.corr_pyramid(fmap1, fmap2)
        chk(RO.corr_lookup(pyr_o, coords0 + jitter * 5.0), lookups["frac"], 2e-4, "lookup frac")
        chk(RO.corr_lookup(pyr_o, coords0 + jitter * 60.0), lookups["far"], 2e-4, "lookup far")
        chk(RO.alternate_corr_lookup(fmap1, fmap2, coords0 + jitter * 5.0), lookups["frac"], 2e-4, "alt lookup")
        n1, m1, d1 = RO.update_block(sd, net, inp, corr0, coords0 - coords0)
        chk(n1, net1, 1e-4, "update net")
        chk(m1, mask1, 1e-4, "update mask")
        chk(d1, delta1, 1e-4, "update delta")
        chk(RO.upsample_flow(delta1, mask1), up1, 1e-4, "upsample")
        chk(lo_o, flow_low, 1e-3, "flow_low")
        chk(up_o, flow_up, 1e-3, "flow_up")
        epe = float((up_o - flow_up).pow(2).sum(1).sqrt().mean())
        print(f"oracle vs reference RAFT: final flow EPE {epe:.3e} px (|flow| mean {float(flow_up.abs().mean()):.2f})")

    f16 = lambda t: t.numpy().astype(np.float16)
    out.update(
        fmap1_f16=f16(fmap1), fmap2_f16=f16(fmap2),         # half precision keeps the fixture small;
        net_f16=f16(net), inp_f16=f16(inp),                  # compared with a matching tolerance
        fmap1_stats=np.array([float(fmap1.sum()), float(fmap1.abs().sum()), float(fmap2.sum()), float(fmap2.abs().sum())]),
        pyr3=cb.corr_pyramid[3].numpy(),
        pyr_stats=np.array([[float(p.sum()), float(p.abs().sum()), float(p.pow(2).sum())] for p in cb.corr_pyramid]),
        lookup_int=lookups["int"].numpy()[:, :, ::3, ::3], lookup_frac=lookups["frac"].numpy()[:, :, ::3, ::3],
        lookup_far=lookups["far"].numpy()[:, :, ::3, ::3], lookup_jitter=jitter.numpy(),
        update_net1_f16=f16(net1), update_delta1=delta1.numpy(), update_mask1_stats=np.array([float(mask1.sum()), float(mask1.abs().sum())]),
        upsample1=up1.numpy()[:, :, ::2, ::2],
        flow_low=flow_low.numpy(), flow_up=flow_up.numpy(),
    )
    np.savez_compressed(os.path.join(HERE, "raft_ref_128x160.npz"), **out)


def make_raft_trainbn():
    warnings.filterwarnings("ignore")
    sys.path.insert(0, REF)
    from raft import RAFT
    from utils.utils import InputPadder

    class NS:
        def __contains__(self, m):
            return hasattr(self, m)
    a = NS()
    a.small, a.mixed_precision, a.alternate_corr = False, False, False
    model = torch.nn.DataParallel(RAFT(a))                       # RAFT_2.__init__: wrapped, loaded, never .eval()'d
    sd = RO.init_state_dict(0)
    model.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)
    assert model.training and model.module.cnet.norm1.training

    out = {"state_dict_sha256": sd_digest(sd)}
    for tag, (H, W), seed in (("a", (128, 160), 21), ("b", (132, 156), 22)):
        g = torch.Generator().manual_seed(seed)
        base = F.avg_pool2d(torch.rand((1, 3, H + 16, W + 16), generator=g), 5, 1, 2)
        base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)
        f1 = base[0, :, 8:8 + H, 8:8 + W].permute(1, 2, 0).contiguous().numpy()      # "BGR" frames as cv2.imread hands them over
        f2 = base[0, :, 11:11 + H, 6:6 + W].permute(1, 2, 0).contiguous().numpy()
        with torch.no_grad():                                    # RAFT_2.calc, cv2.cvtColor(BGR2RGB) = channel reversal
            i1 = torch.from_numpy(np.ascontiguousarray(f1[:, :, ::-1])).permute(2, 0, 1).float()[None]
            i2 = torch.from_numpy(np.ascontiguousarray(f2[:, :, ::-1])).permute(2, 0, 1).float()[None]
            padder = InputPadder(i1.shape)
            p1, p2 = padder.pad(i1, i2)
            flow_low, flow_up = model(p1, p2, iters=20, test_mode=True)
            flo = flow_up[0].permute(1, 2, 0).cpu().numpy()
            cn = model.module.cnet(2 * (p1 / 255.0) - 1.0)       # train-mode BatchNorm on this one image
            net, inp = torch.tanh(cn[:, :128]), torch.relu(cn[:, 128:])
        mine = RO.raft2_calc(sd, f1, f2, iters=20, cnet_norm="batch")
        assert mine.shape == flo.shape
        epe = float(np.sqrt(((mine - flo) ** 2).sum(-1)).mean())
        ev = RO.raft2_calc(sd, f1, f2, iters=20, cnet_norm="eval")
        gap = float(np.sqrt(((ev - flo) ** 2).sum(-1)).mean())
        print(f"train-mode BN [{tag}] {H}x{W}: oracle(batch) vs reference RAFT_2 EPE {epe:.3e} px; eval-mode oracle is {gap:.3e} px away")
        assert epe < 1e-4 and gap > 10 * epe
        out.update({f"frame1_{tag}": f1, f"frame2_{tag}": f2, f"flow_{tag}": flo, f"flow_low_{tag}": flow_low.numpy(),
                    f"net_f16_{tag}": net.numpy().astype(np.float16), f"inp_f16_{tag}": inp.numpy().astype(np.float16),
                    f"eval_gap_{tag}": np.float32(gap)})
    np.savez_compressed(os.path.join(HERE, "raft_ref_trainbn_128x160.npz"), **out)


def make_warp():
    rng = np.random.default_rng(3)
    H, W = 40, 56
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    flow = (rng.standard_normal((H, W, 2)) * 6).astype(np.float32)
    flow[0, 0] = (300, -300)
    mx, my = WO._maps(flow)
    grid = torch.from_numpy(np.stack([2 * mx / (W - 1) - 1, 2 * my / (H - 1) - 1], -1))[None]
    src = torch.from_numpy(frame.astype(np.float32)).permute(2, 0, 1)[None]
    outs = {}
    for mode in ("bilinear", "bicubic"):
        gs = F.grid_sample(src, grid, mode=mode, padding_mode="zeros", align_corners=True)[0].permute(1, 2, 0).numpy()
        outs[mode] = gs
        mine = WO.warp_frame(frame.astype(np.float32), flow, mode)
        sane = (np.abs(flow) < 100).all(-1)
        assert np.abs(mine - gs)[sane].max() < 5e-3, mode
    np.savez_compressed(os.path.join(HERE, "warp_grid_sample.npz"), frame=frame, flow=flow, **outs)

    # oracle-frozen OpenCV semantics
    conf = rng.random((H, W)).astype(np.float32)
    conf[::5, ::7] = np.float32(0.95)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img[5:20, 5:20] = 128
    tabf, tabi = WO.cv2_cubic_tables()
    mask95, _ = MO.generate_mask(conf, conf.copy(), 0.95, 7)
    np.savez_compressed(
        os.path.join(HERE, "cv2_semantics.npz"), frame=frame, flow=flow, conf=conf, img=img,
        warp_cv2_u8=WO.warp_frame(frame, flow, "cv2_cubic"),
        warp_cv2_u8_raft=WO.warp_frame(frame, flow, "cv2_cubic", convention="raft"),
        tab_i16_rows=tabi[[0, 1, 33, 528, 1023]], tab_sum=tabi.astype(np.int64).sum(1),
        ellipse7=MO.ellipse_kernel(7), ellipse15=MO.ellipse_kernel(15),
        mask95=mask95, expand=MO.expand_mask(mask95, img), edges=MO.laplacian_edges(img),
        travel=MO.travel_distance(flow, conf),
    )


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("the reference is not mounted here; golden vectors can only be regenerated in the build container")
    only = sys.argv[1:]
    if not only or "raft" in only:
        make_raft()
    if not only or "trainbn" in only:
        make_raft_trainbn()
    if not only or "warp" in only:
        make_warp()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
