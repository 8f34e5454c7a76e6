"""TEST INFRASTRUCTURE -- CPU oracle of the first-stage (VAE) encoding and of the attention primitive.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product never does.

A functional fp32 restatement (torch on CPU) of
    Encoder.forward            ldm/modules/diffusionmodules/model.py:518-543
      ResnetBlock.forward      :129-149      Downsample.forward :80-87      AttnBlock.forward :179-203
      Normalize / nonlinearity :35-41
    AutoencoderKL.encode       ldm/models/autoencoder.py:350-354  (quant_conv on the encoder output)
    DiagonalGaussianDistribution.sample   ldm/modules/distributions/distributions.py:24-37
    get_first_stage_encoding   ldm/models/diffusion/ddpm.py:655-662
    xformers.ops.memory_efficient_attention(q, k, v, attn_bias)   as used at ldm/modules/attention.py:314,426

PINNED: tests/golden/make_golden_vae.py loads `init_vae_state_dict(0)` into the reference's own `Encoder` (strict key
match) in the build container and stores its output; `attention` is pinned against
torch.nn.functional.scaled_dot_product_attention (xformers itself is not installed anywhere we can reach).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD_V1_CONFIG = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4, double_z=True, embed_dim=4)
SCALE_FACTOR = 0.18215


def encoder_tensors(cfg: dict = SD_V1_CONFIG):
    ch, mult, nres = cfg["ch"], tuple(cfg["ch_mult"]), cfg["num_res_blocks"]
    out = []
    conv = lambda name, co, ci, k: out.extend([(f"{name}.weight", (co, ci, k, k)), (f"{name}.bias", (co,))])
    norm = lambda name, c: out.extend([(f"{name}.weight", (c,)), (f"{name}.bias", (c,))])

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
        for j in range(nres):
            resblock(f"encoder.down.{lvl}.block.{j}", block_in, ch * m)
            block_in = ch * m
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


def init_vae_state_dict(seed: int = 0, cfg: dict = SD_V1_CONFIG) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(int(seed) + 7919)
    sd = {}
    for key, shape in encoder_tensors(cfg):
        if len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(shape[1] * shape[2] * shape[3]))
        elif ".norm" in key and key.endswith(".weight"):
            sd[key] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[key] = 0.05 * torch.randn(shape, generator=g)
    return sd


def _norm(sd, name, x):                                   # Normalize, model.py:40-41
    return F.group_norm(x, 32, sd[f"{name}.weight"], sd[f"{name}.bias"], eps=1e-6)


def _silu(x):                                             # nonlinearity, model.py:35-37
    return x * torch.sigmoid(x)


def _conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[f"{name}.weight"], sd[f"{name}.bias"], stride=stride, padding=padding)


def _resblock(sd, name, x):                               # model.py:129-149 with temb = None, dropout 0
    h = _conv(sd, f"{name}.conv1", _silu(_norm(sd, f"{name}.norm1", x)), padding=1)
    h = _conv(sd, f"{name}.conv2", _silu(_norm(sd, f"{name}.norm2", h)), padding=1)
    if f"{name}.nin_shortcut.weight" in sd:
        x = _conv(sd, f"{name}.nin_shortcut", x)
    return x + h


def _attn(sd, name, x):                                   # model.py:179-203
    hn = _norm(sd, f"{name}.norm", x)
    q, k, v = (_conv(sd, f"{name}.{n}", hn) for n in ("q", "k", "v"))
    b, c, h, w = q.shape
    w_ = torch.bmm(q.reshape(b, c, h * w).permute(0, 2, 1), k.reshape(b, c, h * w)) * (int(c) ** (-0.5))
    w_ = torch.softmax(w_, dim=2)
    h_ = torch.bmm(v.reshape(b, c, h * w), w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(sd, f"{name}.proj_out", h_)


@torch.no_grad()
def encode_moments(sd: Dict[str, torch.Tensor], image: torch.Tensor, cfg: dict = SD_V1_CONFIG, trace: Optional[dict] = None) -> torch.Tensor:
    """image f32 [B,3,H,W] in [-1,1] -> moments [B, 2z, H/8, W/8]."""
    h = _conv(sd, "encoder.conv_in", image, padding=1)
    n_lvl = len(cfg["ch_mult"])
    for lvl in range(n_lvl):
        for j in range(cfg["num_res_blocks"]):
            h = _resblock(sd, f"encoder.down.{lvl}.block.{j}", h)
        if lvl != n_lvl - 1:
            h = _conv(sd, f"encoder.down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2)     # model.py:80-84
    if trace is not None:
        trace["down"] = h
    h = _resblock(sd, "encoder.mid.block_1", h)
    h = _attn(sd, "encoder.mid.attn_1", h)
    h = _resblock(sd, "encoder.mid.block_2", h)
    h = _conv(sd, "encoder.conv_out", _silu(_norm(sd, "encoder.norm_out", h)), padding=1)
    return _conv(sd, "quant_conv", h)


def sample(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:       # distributions.py:24-37
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def get_first_stage_encoding(sd, image, noise, cfg: dict = SD_V1_CONFIG) -> torch.Tensor:      # ddpm.py:655-662
    return SCALE_FACTOR * sample(encode_moments(sd, image, cfg), noise)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: Optional[torch.Tensor] = None, scale: Optional[float] = None):
    """softmax(q k^T * scale + bias) v on [BH, N, D] tensors (f64 accumulation: the checker is stricter than either side)."""
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    s = torch.bmm(q.double(), k.double().transpose(1, 2)) * scale
    if bias is not None:
        s = s + bias.double()
    return torch.bmm(torch.softmax(s, dim=-1), v.double()).float()
