"""-m gpu: first-stage (VAE) encoding and the attention primitive of the SD-inpaint hand-off on the HIP
kernels, against the oracle and -- directly -- against the vectors of the real reference encoder."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as VO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_ref_64x48.npz")


def test_groupnorm_and_silu(cuda):
    from sd_animation_optical_flow_amd import ops
    g = torch.Generator().manual_seed(1)
    for (B, H, W, C) in ((2, 9, 7, 128), (1, 16, 12, 512), (3, 5, 5, 64)):
        x = torch.randn((B, C, H, W), generator=g) * 3 + 1.5
        gamma, beta = torch.randn((C,), generator=g), torch.randn((C,), generator=g)
        ref = torch.nn.functional.group_norm(x, 32, gamma, beta, eps=1e-6)
        nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
        out = ops.groupnorm(nhwc, gamma.cuda(), beta.cuda(), 32, 1e-6, False).permute(0, 3, 1, 2).cpu()
        assert (out - ref).abs().max().item() < 2e-5
        out_s = ops.groupnorm(nhwc, gamma.cuda(), beta.cuda(), 32, 1e-6, True).permute(0, 3, 1, 2).cpu()
        assert (out_s - ref * torch.sigmoid(ref)).abs().max().item() < 2e-5


@pytest.mark.parametrize("BH,Nq,Nk,D", [(3, 50, 77, 40), (2, 96, 96, 80), (1, 130, 130, 512), (5, 33, 7, 160), (2, 70, 45, 48),
                                        (8, 300, 333, 64), (16, 257, 515, 128), (8, 1000, 1100, 40), (8, 129, 64, 160)])
def test_attention_matches_the_oracle(cuda, BH, Nq, Nk, D):
    """memory_efficient_attention's shapes: cross attention against 77 text tokens (Nk not a multiple of 4), SD's head
    sizes 40 / 80 / 160, the VAE mid block's single 512-wide head; without bias, with a shared [Nq,Nk] bias and with a
    per-head one."""
    from sd_animation_optical_flow_amd import ops
    g = torch.Generator().manual_seed(BH * 100 + D)
    q, k, v = torch.randn((BH, Nq, D), generator=g), torch.randn((BH, Nk, D), generator=g), torch.randn((BH, Nk, D), generator=g)
    for bias in (None, torch.randn((Nq, Nk), generator=g) * 2, torch.randn((BH, Nq, Nk), generator=g) * 2):
        ref = VO.attention(q, k, v, bias)
        out = ops.attention(q.cuda(), k.cuda(), v.cuda(), None if bias is None else bias.cuda()).cpu()
        assert (out - ref).abs().max().item() < 2e-5, (None if bias is None else tuple(bias.shape))
    # slicing the batch-heads to bound the workspace does not change the result
    out_sliced = ops.attention(q.cuda(), k.cuda(), v.cuda(), None, max_workspace_bytes=1).cpu()
    assert (out_sliced - VO.attention(q, k, v)).abs().max().item() < 2e-5


def test_fused_attention_masked_keys_and_large_logits(cuda):
    """The fused kernel's online softmax: -inf bias entries (masked keys, including whole leading key blocks), logits
    far from zero (the running maximum must carry the scale), and a row whose every key is masked (NaN, like softmax)."""
    from sd_animation_optical_flow_amd import ops
    g = torch.Generator().manual_seed(3)
    BH, Nq, Nk, D = 8, 200, 150, 40
    q, k, v = torch.randn((BH, Nq, D), generator=g) * 6, torch.randn((BH, Nk, D), generator=g) * 6, torch.randn((BH, Nk, D), generator=g)
    bias = torch.zeros((Nq, Nk))
    bias[:, :70] = float("-inf")                         # the first two 32-key blocks and part of the third
    bias[torch.rand((Nq, Nk), generator=g) < 0.3] = float("-inf")
    bias[5, :] = float("-inf")
    ref = VO.attention(q, k, v, bias)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), bias.cuda()).cpu()
    assert torch.isnan(out[:, 5]).all() and torch.isnan(ref[:, 5]).all()
    keep = torch.ones(Nq, dtype=torch.bool)
    keep[5] = False
    keep &= ~torch.isinf(bias).all(1)
    assert (out[:, keep] - ref[:, keep]).abs().max().item() < 5e-5


def test_fused_attention_randomised_sweep(cuda):
    """Seeded sweep over the fused kernel's head sizes with ragged token counts (tails of the 128-query tile, of the 32-key
    block and of the 64-key tile), batch-head counts on and off the XCD-grouped mapping, and the three bias forms."""
    from sd_animation_optical_flow_amd import ops
    rng = np.random.default_rng(77)
    for it in range(24):
        D = int(rng.choice([40, 64, 80, 128, 160]))
        BH = int(rng.choice([1, 3, 8, 16]))
        Nq, Nk = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        g = torch.Generator().manual_seed(500 + it)
        q, k, v = torch.randn((BH, Nq, D), generator=g), torch.randn((BH, Nk, D), generator=g), torch.randn((BH, Nk, D), generator=g)
        mode = it % 3
        bias = None if mode == 0 else torch.randn((Nq, Nk), generator=g) if mode == 1 else torch.randn((BH, Nq, Nk), generator=g)
        ref = VO.attention(q, k, v, bias)
        out = ops.attention(q.cuda(), k.cuda(), v.cuda(), None if bias is None else bias.cuda()).cpu()
        assert (out - ref).abs().max().item() < 3e-5, (it, BH, Nq, Nk, D, mode)


def test_xformers_shim_on_the_device(cuda):
    """The call exactly as ldm/modules/attention.py:314 makes it: [b*heads, n, dim_head] half tensors and a 2-D bias."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shims"))
    import xformers.ops
    g = torch.Generator().manual_seed(9)
    q, k, v = (torch.randn((8, 64, 40), generator=g) for _ in range(3))
    bias = torch.randn((64, 64), generator=g)
    out = xformers.ops.memory_efficient_attention(q.cuda().half(), k.cuda().half(), v.cuda().half(), attn_bias=bias.cuda().half(), op=None)
    assert out.dtype == torch.float16 and tuple(out.shape) == (8, 64, 40)
    ref = VO.attention(q.half().float(), k.half().float(), v.half().float(), bias.half().float())
    assert (out.float().cpu() - ref).abs().max().item() < 2e-3                   # fp16 output rounding
    out4 = xformers.ops.memory_efficient_attention(q.cuda().reshape(2, 4, 64, 40).permute(0, 2, 1, 3), k.cuda().reshape(2, 4, 64, 40).permute(0, 2, 1, 3),
                                                   v.cuda().reshape(2, 4, 64, 40).permute(0, 2, 1, 3))
    assert (out4.permute(0, 2, 1, 3).reshape(8, 64, 40).cpu() - VO.attention(q, k, v)).abs().max().item() < 2e-5


@pytest.fixture(scope="module")
def vae_enc(cuda):
    from sd_animation_optical_flow_amd.vae import VaeEncoder, random_vae_state_dict
    return VaeEncoder(random_vae_state_dict(0))


def test_vae_encoder_against_the_reference_vectors_directly(vae_enc):
    """One hop: the HIP encoder on the golden input against what the REAL reference `Encoder` + `quant_conv` +
    `DiagonalGaussianDistribution.sample` + scale_factor produced (tests/golden/make_golden_vae.py)."""
    g = np.load(GOLD)
    image = torch.from_numpy(g["image"]).cuda()
    mo = vae_enc.encode_moments(image).cpu()
    ref = torch.from_numpy(g["moments"])
    assert tuple(mo.shape) == tuple(ref.shape)
    assert (mo - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    z = vae_enc.get_first_stage_encoding(image, torch.from_numpy(g["noise"]).cuda()).cpu()
    assert (z - torch.from_numpy(g["latent"])).abs().max().item() < 2e-4


def test_vae_encoder_batch_and_other_sizes_against_the_oracle(vae_enc):
    sd = VO.init_vae_state_dict(0)
    g = torch.Generator().manual_seed(21)
    for (B, H, W) in ((2, 40, 72), (1, 128, 96)):
        image = torch.rand((B, 3, H, W), generator=g) * 2 - 1
        ref = VO.encode_moments(sd, image)
        mo = vae_enc.encode_moments(image.cuda()).cpu()
        assert (mo - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item()), (B, H, W)
    with pytest.raises(RuntimeError):
        vae_enc.encode_moments(torch.zeros((1, 3, 36, 40), device="cuda"))        # not a multiple of 8
    with pytest.raises(RuntimeError):
        vae_enc.encode_moments(torch.zeros((1, 3, 40, 40)))                       # CPU tensor


def test_vae_encoder_1024x1024_hand_off(vae_enc):
    """BASELINE config #5's frame size: the device-resident hand-off (Pillow-exact blur / composite, then the first-stage
    latent) runs end to end at 1024x1024; finiteness, shapes and the mid-block attention over 16384 tokens."""
    from sd_animation_optical_flow_amd import handoff
    g = torch.Generator(device="cuda").manual_seed(5)
    H = W = 1024
    frame = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    ref = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    mask = (torch.rand((1, H, W), device="cuda", generator=g) > 0.8).to(torch.uint8) * 255
    t = handoff.prepare_inpaint_inputs(frame, ref, mask, mask_blur=4)
    z = vae_enc.get_first_stage_encoding(t["image"])
    assert tuple(z.shape) == (1, 4, H // 8, W // 8) and bool(torch.isfinite(z).all())
    assert tuple(t["latmask"].shape) == (1, 4, H // 8, W // 8)


def test_vae_encoder_slices_batches_beyond_the_2gib_activation_limit(vae_enc):
    """ClipPipeline hands the encoder up to 64 frames at once; conv_in's [B,H,W,128] output crosses the convolution kernel's
    32-bit byte offsets at 11 frames of 512x768.  `encode_moments` slices (images are independent): 12 frames at 512x768
    come back, and equal the frames encoded one by one."""
    assert vae_enc.max_batch(768, 512) == 10 and vae_enc.max_batch(1024, 1024) == 3
    g = torch.Generator(device="cuda").manual_seed(9)
    image = torch.rand((12, 3, 768, 512), device="cuda", generator=g) * 2 - 1
    mo = vae_enc.encode_moments(image)
    assert tuple(mo.shape) == (12, 8, 96, 64) and bool(torch.isfinite(mo).all())
    for b in (0, 10, 11):
        one = vae_enc.encode_moments(image[b:b + 1])
        assert (mo[b:b + 1] - one).abs().max().item() < 1e-4 * max(1.0, one.abs().max().item())
