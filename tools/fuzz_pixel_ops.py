#!/usr/bin/env python3
"""One-off sweep of the bit-exact per-pixel operators over random (often awkward) shapes against their oracles."""
import os, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import handoff_oracle as HO, keyframe_oracle as KO, mask_oracle as MO, warp_oracle as WO
from sd_animation_optical_flow_amd import keyframes, ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
random.seed(seed)
rng = np.random.default_rng(seed)
for case in range(n):
    H, W = random.randint(2, 150), random.randint(4, 170)
    if random.random() < 0.3:
        W = 4 * random.randint(1, 40)
    conf = rng.random((H, W)).astype(np.float32)
    conf[rng.random((H, W)) < 0.05] = np.float32(0.95)
    ks = random.choice([1, 3, 5, 7, 9, 11, 15, 31])
    thres = random.choice([0.95, 0.5, 0.9])
    logc = np.log(np.maximum(conf, 1e-6)).astype(np.float32)
    for cmp_gt in (False, True):
        if cmp_gt:
            ref = MO.dilate(np.where(conf > np.float32(thres), 0, 255).astype(np.uint8), MO.ellipse_kernel(ks))
            out = ops.generate_mask(torch.from_numpy(conf).cuda()[None], None, thres, ks, cmp_gt=True)[0].cpu().numpy()
            assert np.array_equal(out, ref), ("mask gt", H, W, ks)
        else:
            ref, rlog = MO.generate_mask(conf, logc.copy(), thres, ks)
            lg = torch.from_numpy(logc.copy()).cuda()[None]
            out = ops.generate_mask(torch.from_numpy(conf).cuda()[None], lg, thres, ks)[0].cpu().numpy()
            assert np.array_equal(out, ref) and np.array_equal(lg[0].cpu().numpy(), rlog), ("mask", H, W, ks)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    m0 = (rng.random((H, W)) < 0.1).astype(np.uint8) * 255
    if H >= 2 and W >= 2:
        ke = random.choice([3, 7, 9])
        ref = MO.expand_mask(m0, img, ke)
        out = ops.expand_mask(torch.from_numpy(m0).cuda()[None], torch.from_numpy(img).cuda()[None], 20, ke)[0].cpu().numpy()
        assert np.array_equal(out, ref), ("expand", H, W, ke)
    grey = rng.integers(0, 256, (H, W), dtype=np.uint8)
    assert np.array_equal(ops.dilate(torch.from_numpy(grey).cuda()[None], ks)[0].cpu().numpy(), MO.dilate(grey, MO.ellipse_kernel(ks))), ("dilate", H, W, ks)
    flow = (rng.standard_normal((H, W, 2)) * random.choice([0.5, 3, 20])).astype(np.float32)
    if H >= 2 and W >= 4:
        for mode in ("cv2_cubic",):
            ref = WO.warp_frame(img, flow, mode=mode)
            out = ops.warp(torch.from_numpy(img).cuda(), torch.from_numpy(flow).cuda(), mode=mode).cpu().numpy()
            assert np.array_equal(out, ref), ("warp", mode, H, W)
        f4 = np.concatenate([img, img[:, :, :1]], -1)
        o4 = ops.warp(torch.from_numpy(f4).cuda(), torch.from_numpy(flow).cuda(), mode="bilinear").cpu().numpy()
        o3 = ops.warp(torch.from_numpy(img).cuda(), torch.from_numpy(flow).cuda(), mode="bilinear").cpu().numpy()
        assert np.array_equal(o4[:, :, :3], o3), ("warp bilinear fast vs generic", H, W)
    rad = random.choice([0.0, 1.0, 2.5, 4.0, 9.0])
    assert np.array_equal(ops.gaussian_blur_u8(torch.from_numpy(grey).cuda()[None], rad)[0].cpu().numpy(), HO.gaussian_blur_u8(grey, rad)), ("blur", H, W, rad)
    oh, ow = random.randint(1, 40), random.randint(1, 40)
    assert np.array_equal(ops.resize_bicubic_u8(torch.from_numpy(grey).cuda()[None], oh, ow)[0].cpu().numpy(), HO.resize_bicubic_u8(grey, oh, ow)), ("resize", H, W, oh, ow)
    if H >= 3 and W >= 3:
        k = random.choice([1, 3, 5, 7])
        e = keyframes.detect_edges(img, ksize=k).cpu().numpy()
        assert np.array_equal(e, KO.detect_edges(img, k)), ("edges", H, W, k)
    print(f"case {case}: {W}x{H} ksize {ks} ok", flush=True)
print("all ok")
