"""CPU: the first-stage (VAE) oracle against the vectors the REAL reference encoder produced (tests/golden/
make_golden_vae.py), the product's seeded weights against the oracle's, the attention oracle against torch's SDPA, and
the `shims/xformers` shim the reference's `ldm/` imports."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as VO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_ref_64x48.npz")


def test_vae_oracle_reproduces_the_reference_encoder():
    g = np.load(GOLD)
    sd = VO.init_vae_state_dict(0)
    image = torch.from_numpy(g["image"])
    mo = VO.encode_moments(sd, image)
    assert tuple(mo.shape) == (1, 8, 8, 6)
    assert (mo - torch.from_numpy(g["moments"])).abs().max().item() < 1e-4
    z = VO.get_first_stage_encoding(sd, image, torch.from_numpy(g["noise"]))
    assert (z - torch.from_numpy(g["latent"])).abs().max().item() < 1e-4


def test_product_vae_weights_equal_the_oracles_and_cover_the_reference_keys():
    from sd_animation_optical_flow_amd import vae
    a, b = vae.random_vae_state_dict(0), VO.init_vae_state_dict(0)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert vae.encoder_tensors() == VO.encoder_tensors()
    n = sum(v.numel() for v in a.values())
    assert n == 34163664                                    # SD-v1 encoder + quant_conv
    assert "encoder.down.1.block.0.nin_shortcut.weight" in a and "encoder.down.3.downsample.conv.weight" not in a


def test_attention_oracle_matches_torch_sdpa():
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn((4, n, 40), generator=g) for n in (50, 77, 77))
    bias = torch.randn((50, 77), generator=g)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=bias)
    assert (VO.attention(q, k, v, bias) - ref).abs().max().item() < 1e-5
    ref2 = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    assert (VO.attention(q, k, v) - ref2).abs().max().item() < 1e-5


def test_xformers_shim_resolves_to_the_hip_attention():
    """ldm/modules/attention.py:12-18 does `import xformers; import xformers.ops`: with <repo>/shims on sys.path
    those imports resolve to this repository's shim (no compute here: no GPU)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shims"))
    import xformers
    import xformers.ops
    from sd_animation_optical_flow_amd import attention
    assert xformers.ops.memory_efficient_attention is attention.memory_efficient_attention
    with pytest.raises(RuntimeError):
        xformers.ops.memory_efficient_attention(torch.zeros(2, 4, 8), torch.zeros(2, 4, 8), torch.zeros(2, 4, 8))   # CPU tensors
    with pytest.raises(TypeError):
        xformers.ops.memory_efficient_attention(torch.zeros(2, 4, 8), torch.zeros(2, 4, 8), torch.zeros(2, 4, 8), attn_bias=object())
