"""First-stage (VAE) encoding of the SD-inpaint hand-off on the device.

`img2img_inpaint` turns the composited frame into `init_latent = get_first_stage_encoding(encode_first_stage(image))`
(guided_ldm_inpainting.py:302) before the first denoising step.  `encode_first_stage` is `AutoencoderKL.encode`
(ldm/models/autoencoder.py:350-354): `Encoder.forward` (ldm/modules/diffusionmodules/model.py:518-543) -> `quant_conv`
-> a diagonal Gaussian whose sample, times `scale_factor` (0.18215, guided_ldm_inpaint4_v15.yaml:16), is the latent
(ldm/models/diffusion/ddpm.py:655-662, ldm/modules/distributions/distributions.py:24-37).

Everything runs through the C ABI of libofx.so on NHWC fp32 tensors: the 3x3 / 1x1 convolutions on the fp32 matrix
cores (`ofx_conv2d`, residual sums as pre-activation addends of the epilogue), GroupNorm(32, eps 1e-6) + x*sigmoid(x)
(`ofx_groupnorm`), the mid-block self-attention (`ofx_attention_f32`; the reference picks xformers there when it is
installed, model.py:282-290).  State-dict keys are the reference's (`encoder.*`, `quant_conv.*`; the `first_stage_model.`
prefix of full SD checkpoints is accepted).  No checkpoint ships with the reference tree: parity is pinned with seeded
weights loaded into the reference's own `Encoder` (tests/golden/make_golden_vae.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

SCALE_FACTOR = 0.18215
SD_V1_CONFIG = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4, double_z=True, embed_dim=4)


def encoder_tensors(cfg: dict = SD_V1_CONFIG) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of every tensor of `encoder.*` + `quant_conv.*` for a ddconfig, in module order."""
    ch, mult, nres = cfg["ch"], tuple(cfg["ch_mult"]), cfg["num_res_blocks"]
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, co, ci, k):
        out.append((f"{name}.weight", (co, ci, k, k)))
        out.append((f"{name}.bias", (co,)))

    def norm(name, c):
        out.append((f"{name}.weight", (c,)))
        out.append((f"{name}.bias", (c,)))

    def resblock(name, ci, co):
        norm(f"{name}.norm1", ci)
        conv(f"{name}.conv1", co, ci, 3)
        norm(f"{name}.norm2", co)
        conv(f"{name}.conv2", co, co, 3)
        if ci != co:
            conv(f"{name}.nin_shortcut", co, ci, 1)

    conv("encoder.conv_in", ch, cfg["in_channels"], 3)
    block_in = ch
    for lvl, m in enumerate(mult):
        block_out = ch * m
        for j in range(nres):
            resblock(f"encoder.down.{lvl}.block.{j}", block_in, block_out)
            block_in = block_out
        if lvl != len(mult) - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", block_in, block_in, 3)
    resblock("encoder.mid.block_1", block_in, block_in)
    norm("encoder.mid.attn_1.norm", block_in)
    for nm in ("q", "k", "v", "proj_out"):
        conv(f"encoder.mid.attn_1.{nm}", block_in, block_in, 1)
    resblock("encoder.mid.block_2", block_in, block_in)
    norm("encoder.norm_out", block_in)
    zc = cfg["z_channels"] * (2 if cfg["double_z"] else 1)
    conv("encoder.conv_out", zc, block_in, 3)
    conv("quant_conv", 2 * cfg["embed_dim"], zc, 1)
    return out


def random_vae_state_dict(seed: int = 0, cfg: dict = SD_V1_CONFIG) -> Dict[str, torch.Tensor]:
    """Seeded stand-in for the absent checkpoint: fan-in-scaled normal convolution weights, norm scales around 1."""
    g = torch.Generator().manual_seed(int(seed) + 7919)
    sd = {}
    for key, shape in encoder_tensors(cfg):
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[key] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        elif ".norm" in key and key.endswith(".weight"):
            sd[key] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[key] = 0.05 * torch.randn(shape, generator=g)
    return sd


class VaeEncoder:
    """`AutoencoderKL.encode` + `get_first_stage_encoding` on a HIP device."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", cfg: dict = SD_V1_CONFIG, scale_factor: float = SCALE_FACTOR):
        if not torch.cuda.is_available():
            raise RuntimeError("VaeEncoder needs a HIP device (no CPU fallback)")
        self.cfg, self.device, self.scale_factor = dict(cfg), torch.device(device), float(scale_factor)
        sd = {}
        for k, v in state_dict.items():
            k = k[len("first_stage_model."):] if k.startswith("first_stage_model.") else k
            sd[k] = v
        self.w: Dict[str, torch.Tensor] = {}
        for key, shape in encoder_tensors(self.cfg):
            if key not in sd:
                raise KeyError(f"VAE checkpoint lacks {key}")
            t = sd[key].detach().to(torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{key}: shape {tuple(t.shape)} != {shape}")
            if len(shape) == 4:
                self.w[key] = ops.pack_conv_weight(t).to(self.device)           # [Cout, Kpad], Cin padded to a multiple of 4
            else:
                self.w[key] = t.contiguous().to(self.device)

    # ---- building blocks (model.py line numbers) -----------------------------------------------------------
    def _conv(self, name: str, x: torch.Tensor, k: int, stride: int = 1, addend: Optional[torch.Tensor] = None,
              pad=None, out_hw=None) -> torch.Tensor:
        w = self.w[f"{name}.weight"]
        return ops.conv2d_nhwc(x, w, k, k, w.shape[0], stride=stride, shift=self.w[f"{name}.bias"], addend=addend, pad=pad, out_hw=out_hw)

    def _norm(self, name: str, x: torch.Tensor, silu: bool) -> torch.Tensor:
        return ops.groupnorm(x, self.w[f"{name}.weight"], self.w[f"{name}.bias"], 32, 1e-6, silu)

    def _resblock(self, name: str, x: torch.Tensor) -> torch.Tensor:
        """ResnetBlock.forward (:129-149), temb = None, dropout 0."""
        h = self._conv(f"{name}.conv1", self._norm(f"{name}.norm1", x, True), 3)
        h = self._norm(f"{name}.norm2", h, True)
        skip = self._conv(f"{name}.nin_shortcut", x, 1) if f"{name}.nin_shortcut.weight" in self.w else x
        return self._conv(f"{name}.conv2", h, 3, addend=skip)                 # x + h in the epilogue

    def _attn(self, name: str, x: torch.Tensor) -> torch.Tensor:
        """AttnBlock.forward (:179-203): one head over all H*W positions, d = channels."""
        B, H, W, Cn = x.shape
        hn = self._norm(f"{name}.norm", x, False)
        q = self._conv(f"{name}.q", hn, 1).reshape(B, H * W, Cn)
        k = self._conv(f"{name}.k", hn, 1).reshape(B, H * W, Cn)
        v = self._conv(f"{name}.v", hn, 1).reshape(B, H * W, Cn)
        a = ops.attention(q, k, v, None, float(Cn) ** -0.5).reshape(B, H, W, Cn)
        return self._conv(f"{name}.proj_out", a, 1, addend=x)

    def max_batch(self, H: int, W: int) -> int:
        """Images one pass of the encoder can take at this size: the convolution kernel addresses its operands with 32-bit byte
        offsets, so the widest activation -- the full-resolution `ch * ch_mult[0]`-channel maps of level 0 -- must stay under
        2 GiB (10 frames at 512x768, 4 at 1024x1024).  `encode_moments` slices larger batches (images are independent)."""
        widest = H * W * self.cfg["ch"] * max(1, self.cfg["ch_mult"][0]) * 4
        return max(1, ((1 << 31) - 4096) // widest)

    @torch.no_grad()
    def encode_moments(self, image: torch.Tensor) -> torch.Tensor:
        """image f32 [B,3,H,W] in [-1,1] (what img2img_inpaint builds, guided_ldm_inpainting.py:299-301), H and W
        multiples of 8 -> moments f32 [B, 2*z, H/8, W/8] = quant_conv(encoder(image)) (autoencoder.py:350-352)."""
        if not image.is_cuda or image.dtype != torch.float32 or image.dim() != 4 or image.shape[1] != self.cfg["in_channels"]:
            raise RuntimeError("image must be a CUDA float32 tensor [B,3,H,W]")
        B, _, H, W = image.shape
        n_down = len(self.cfg["ch_mult"]) - 1
        if H % (1 << n_down) or W % (1 << n_down):
            raise RuntimeError(f"H and W must be multiples of {1 << n_down}")
        mb = self.max_batch(H, W)
        if B > mb:
            return torch.cat([self.encode_moments(image[b0:b0 + mb]) for b0 in range(0, B, mb)])
        x = torch.zeros((B, H, W, 4), dtype=torch.float32, device=image.device)        # NHWC, channel 3 = 0 (Cin padded to 4)
        x[..., :3] = image.permute(0, 2, 3, 1)
        h = self._conv("encoder.conv_in", x, 3)
        for lvl in range(len(self.cfg["ch_mult"])):
            for j in range(self.cfg["num_res_blocks"]):
                h = self._resblock(f"encoder.down.{lvl}.block.{j}", h)
            if lvl != n_down:
                # Downsample (:80-84): F.pad(x, (0,1,0,1)) + 3x3 stride-2 conv without padding = taps past the bottom /
                # right edge read zeros
                h = self._conv(f"encoder.down.{lvl}.downsample.conv", h, 3, stride=2, pad=(0, 0), out_hw=(h.shape[1] // 2, h.shape[2] // 2))
        h = self._resblock("encoder.mid.block_1", h)
        h = self._attn("encoder.mid.attn_1", h)
        h = self._resblock("encoder.mid.block_2", h)
        h = self._conv("encoder.conv_out", self._norm("encoder.norm_out", h, True), 3)
        m = self._conv("quant_conv", h, 1)
        return m.permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def sample(moments: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """DiagonalGaussianDistribution(moments).sample() (distributions.py:24-37); noise defaults to torch.randn."""
        mean, logvar = torch.chunk(moments, 2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        if noise is None:
            noise = torch.randn(mean.shape, device=moments.device)
        return mean + std * noise

    @torch.no_grad()
    def get_first_stage_encoding(self, image: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`self.get_first_stage_encoding(self.encode_first_stage(image))` (guided_ldm_inpainting.py:302)."""
        return self.scale_factor * self.sample(self.encode_moments(image), noise)
