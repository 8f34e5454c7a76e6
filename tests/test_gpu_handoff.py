"""-m gpu: the SD-inpaint hand-off kernels against the Pillow-pinned oracle (bit-exact) and the golden Pillow vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import handoff_oracle as HO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sd_handoff_pil.npz")


def _ops():
    from sd_animation_optical_flow_amd import ops
    return ops


def test_blur_and_resize_reproduce_the_pillow_vectors(cuda):
    ops = _ops()
    g = np.load(G)
    for name in "abcd":
        mask, blur = g[f"{name}_mask"], float(g[f"{name}_blur"])
        H, W = mask.shape
        b = ops.gaussian_blur_u8(torch.from_numpy(mask).cuda()[None], blur)
        assert np.array_equal(b[0].cpu().numpy(), g[f"{name}_pil_blur"]), name
        lat = ops.resize_bicubic_u8(b, H // 8, W // 8)
        assert np.array_equal(lat[0].cpu().numpy(), g[f"{name}_pil_latent"]), name


@pytest.mark.parametrize("H,W,radius", [(768, 512, 4.0), (45, 37, 4.0), (64, 200, 1.0), (33, 9, 12.0), (128, 96, 0.0)])
def test_gaussian_blur_matches_oracle(cuda, H, W, radius):
    """Full frame size, ragged sizes, a radius larger than the image width, radius 0 (identity), batch of 2."""
    ops = _ops()
    rng = np.random.default_rng(H * 1000 + W)
    m = (rng.random((2, H, W)) < 0.35).astype(np.uint8) * 255
    m[1] = rng.integers(0, 256, (H, W), dtype=np.uint8)          # grey values too, not only 0/255
    out = ops.gaussian_blur_u8(torch.from_numpy(m).cuda(), radius).cpu().numpy()
    for b in range(2):
        assert np.array_equal(out[b], HO.gaussian_blur_u8(m[b], radius)), b


@pytest.mark.parametrize("H,W,h,w", [(768, 512, 96, 64), (50, 70, 6, 8), (40, 40, 40, 40), (16, 24, 40, 36)])
def test_resize_bicubic_matches_oracle(cuda, H, W, h, w):
    """Downscale by 8 (the latent mask), ragged ratio, identity size, upscale."""
    ops = _ops()
    rng = np.random.default_rng(H + 7 * W)
    a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    out = ops.resize_bicubic_u8(torch.from_numpy(a).cuda()[None], h, w)[0].cpu().numpy()
    assert np.array_equal(out, HO.resize_bicubic_u8(a, h, w))


def test_prepare_inpaint_inputs_matches_oracle_and_reference_shapes(cuda):
    from sd_animation_optical_flow_amd import handoff
    g = np.load(G)
    image_bgr = np.ascontiguousarray(g["a_image_rgb"][..., ::-1])
    ref_bgr = np.ascontiguousarray(g["a_reference_rgb"][..., ::-1])
    mask = g["a_mask"]
    H, W = mask.shape
    r = HO.sd_handoff(image_bgr, ref_bgr, mask, 4.0)
    out = handoff.prepare_inpaint_inputs(image_bgr, ref_bgr, mask, mask_blur=4.0)
    assert tuple(out["image"].shape) == (1, 3, H, W) and tuple(out["latmask"].shape) == (1, 4, H // 8, W // 8)
    assert tuple(out["conditioning_mask"].shape) == (1, 1, H, W) and tuple(out["conditioning_mask_latent"].shape) == (1, 1, H // 8, W // 8)
    assert np.array_equal(out["image_mask"][0].cpu().numpy(), r["image_mask"])
    assert np.array_equal(out["image"][0].cpu().numpy(), r["image"])                       # f32, bit for bit
    assert np.array_equal(out["latmask"][0].cpu().numpy(), r["latmask"])
    assert np.array_equal(out["conditioning_mask"][0, 0].cpu().numpy(), r["cond_mask"])
    assert np.array_equal(out["conditioning_image"][0].cpu().numpy(), r["cond_image"])
    assert np.array_equal(out["conditioning_mask_latent"][0, 0].cpu().numpy(), r["cond_mask_latent"])
    # the composite is Pillow's (golden), not just the oracle's
    rgb = ((out["image"][0].permute(1, 2, 0) + 1.0) * 127.5).round().to(torch.uint8).cpu().numpy()
    assert np.array_equal(rgb, g["a_pil_composite"])


def test_handoff_chains_from_the_flow_path_on_device(cuda, raft_sd):
    """flow -> warp -> mask -> hand-off without leaving the device: batch of 3 frames, outputs finite and consistent
    with running the hand-off frame by frame."""
    from sd_animation_optical_flow_amd import clip, handoff
    from sd_animation_optical_flow_amd.raft import RaftEngine
    eng = RaftEngine(raft_sd)
    gc = torch.Generator(device="cuda").manual_seed(6)

    def flow_fn(frames, key):          # RAFT flow + a synthetic confidence map (RAFT emits none)
        flow = eng.forward(frames, key, iters=4)
        return flow, torch.rand(flow.shape[:3], device="cuda", generator=gc)

    synth = clip.FrameSynthesizer(flow_fn=flow_fn, warp_mode="bilinear", thres=0.3, ksize=7)
    g = torch.Generator(device="cuda").manual_seed(5)
    frames = torch.randint(0, 256, (3, 128, 160, 3), dtype=torch.uint8, device="cuda", generator=g)
    key = torch.randint(0, 256, (128, 160, 3), dtype=torch.uint8, device="cuda", generator=g)
    key_ai = 255 - key
    flow, warped, mask = synth(frames, key, key_ai)
    out = handoff.prepare_inpaint_inputs(warped, frames, mask)
    assert all(torch.isfinite(v.float()).all() for v in out.values())
    one = handoff.prepare_inpaint_inputs(warped[1], frames[1], mask[1])
    for k in out:
        assert torch.equal(out[k][1:2], one[k]), k
