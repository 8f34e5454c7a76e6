#!/usr/bin/env python3
"""Golden vectors of the first-stage encoder, produced by the REAL reference code in the build container:

    python tests/golden/make_golden_vae.py

Imports `ldm.modules.diffusionmodules.model.Encoder` and `DiagonalGaussianDistribution` from /root/reference (imported
from where they lie, nothing copied), builds the SD-v1 encoder of guided_ldm_inpaint4_v15.yaml:41-55, loads
`oracle.vae_oracle.init_vae_state_dict(0)` into it with strict key matching (so keys and shapes are the reference's),
applies `quant_conv` like AutoencoderKL.encode (autoencoder.py:350-352) and stores input, moments and the scaled sample
in tests/golden/vae_ref_64x48.npz.  Asserts at generation time that the oracle reproduces them.  xformers is not
installed here, so the reference takes its `AttnBlock` (vanilla attention) -- the arithmetic xformers approximates.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import vae_oracle as VO   # noqa: E402


def main():
    from ldm.modules.diffusionmodules.model import Encoder
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    enc = Encoder(**dd).eval()
    sd = VO.init_vae_state_dict(0)
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc.load_state_dict(enc_sd, strict=True)
    g = torch.Generator().manual_seed(11)
    H, W = 64, 48
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 8, W + 8), generator=g), 5, 1, 2)[:, :, 4:4 + H, 4:4 + W]
    image = (base - base.min()) / (base.max() - base.min()) * 2 - 1
    with torch.no_grad():
        h = enc(image)
        moments = torch.nn.functional.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
        post = DiagonalGaussianDistribution(moments)
        noise = torch.randn(post.mean.shape, generator=torch.Generator().manual_seed(12))
        z = 0.18215 * (post.mean + post.std * noise)
        mo = VO.encode_moments(sd, image)
        err = float((mo - moments).abs().max())
        assert err < 1e-4 * max(1.0, float(moments.abs().max())), err
        assert float((VO.get_first_stage_encoding(sd, image, noise) - z).abs().max()) < 1e-4
        print(f"oracle vs reference encoder: max |d moments| = {err:.2e} (|moments| max {float(moments.abs().max()):.2f})")
    np.savez_compressed(os.path.join(HERE, "vae_ref_64x48.npz"), image=image.numpy(), moments=moments.numpy(), noise=noise.numpy(),
                        latent=z.numpy(), enc_out_stats=np.array([float(h.mean()), float(h.std())]))


if __name__ == "__main__":
    main()
