"""CPU oracle for the RAFT dense-flow part of the hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, with plain fp32 PyTorch-CPU ops, the algorithm of the reference's vendored
RAFT (`/root/reference/RAFT/core`).  It is the checker for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product package
(`sd_animation_optical_flow_amd`) never does.

Pinning: `tests/golden/make_golden.py` (run in the build container, where `/root/reference` is
mounted) loads `init_state_dict(seed)` into the *real* reference `RAFT` module and checks every
stage of this restatement against it; the resulting vectors are committed under `tests/golden/`
and re-checked by `tests/test_oracle_golden.py` on every run.

Everything is functional: weights live in a flat dict whose keys are exactly the reference
checkpoint's `state_dict()` keys (without the `module.` DataParallel prefix), so a real
`raft-things.pth` can be dropped in.

Reference citations (file:line relative to /root/reference):
  encoder            RAFT/core/extractor.py:6-56 (ResidualBlock), :118-192 (BasicEncoder)
  all-pairs volume   RAFT/core/corr.py:13-27, :52-60
  pyramid lookup     RAFT/core/corr.py:29-50 ; RAFT/core/utils/utils.py:57-71
  local correlation  RAFT/core/corr.py:63-91 ; RAFT/alt_cuda_corr/correlation_kernel.cu:18-119
  update block       RAFT/core/update.py:6-14, :33-60, :79-97, :114-136
  convex upsample    RAFT/core/raft.py:72-83
  forward            RAFT/core/raft.py:86-144
  padding            RAFT/core/utils/utils.py:7-24
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
HDIM = 128          # raft.py:38
CDIM = 128          # raft.py:39
CORR_LEVELS = 4     # raft.py:40
CORR_RADIUS = 4     # raft.py:41


# --------------------------------------------------------------------------------------
# deterministic weights with the reference's state_dict key set
# --------------------------------------------------------------------------------------
def _encoder_conv_shapes() -> List[Tuple[str, Tuple[int, int, int, int]]]:
    """(name, (cout, cin, kh, kw)) for BasicEncoder; extractor.py:134-146, :159-166."""
    out = [("conv1", (64, 3, 7, 7))]
    cin = 64
    for li, (dim, stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
        out.append((f"layer{li}.0.conv1", (dim, cin, 3, 3)))
        out.append((f"layer{li}.0.conv2", (dim, dim, 3, 3)))
        if stride != 1:
            out.append((f"layer{li}.0.downsample.0", (dim, cin, 1, 1)))
        out.append((f"layer{li}.1.conv1", (dim, dim, 3, 3)))
        out.append((f"layer{li}.1.conv2", (dim, dim, 3, 3)))
        cin = dim
    return out


def _encoder_bn_names() -> List[Tuple[str, int]]:
    """BatchNorm modules of the 'batch' encoder (cnet) with their channel counts.
    `norm3` of a strided block is registered twice (as `.norm3` and as `.downsample.1`,
    extractor.py:43-44) so both key sets appear in the reference state_dict."""
    names = [("norm1", 64)]
    for li, (dim, stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
        for bi in (0, 1):
            names.append((f"layer{li}.{bi}.norm1", dim))
            names.append((f"layer{li}.{bi}.norm2", dim))
        if stride != 1:
            names.append((f"layer{li}.0.norm3", dim))
    return names


UPDATE_CONV_SHAPES: List[Tuple[str, Tuple[int, int, int, int]]] = [
    # update.py:79-86
    ("encoder.convc1", (256, CORR_LEVELS * (2 * CORR_RADIUS + 1) ** 2, 1, 1)),
    ("encoder.convc2", (192, 256, 3, 3)),
    ("encoder.convf1", (128, 2, 7, 7)),
    ("encoder.convf2", (64, 128, 3, 3)),
    ("encoder.conv", (128 - 2, 64 + 192, 3, 3)),
    # update.py:33-42
    ("gru.convz1", (128, 384, 1, 5)),
    ("gru.convr1", (128, 384, 1, 5)),
    ("gru.convq1", (128, 384, 1, 5)),
    ("gru.convz2", (128, 384, 5, 1)),
    ("gru.convr2", (128, 384, 5, 1)),
    ("gru.convq2", (128, 384, 5, 1)),
    # update.py:6-11
    ("flow_head.conv1", (256, 128, 3, 3)),
    ("flow_head.conv2", (2, 256, 3, 3)),
    # update.py:122-125
    ("mask.0", (256, 128, 3, 3)),
    ("mask.2", (64 * 9, 256, 1, 1)),
]


def init_state_dict(seed: int = 0) -> Dict[str, Tensor]:
    """Seeded random weights carrying every key of the reference `RAFT(args).state_dict()`
    (args.small=False).  The distribution is ours (the reference checkpoint is absent):
    encoder convs N(0, 2/fan_out) like extractor.py:150-152, everything else U(+-1/sqrt(fan_in));
    cnet BatchNorm gets non-trivial running stats / affine so that eval-mode BN is exercised."""
    g = torch.Generator().manual_seed(int(seed))
    sd: Dict[str, Tensor] = {}

    def uniform(shape, bound):
        return (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound

    for enc in ("fnet", "cnet"):
        for name, (co, ci, kh, kw) in _encoder_conv_shapes():
            fan_out = co * kh * kw
            fan_in = ci * kh * kw
            sd[f"{enc}.{name}.weight"] = torch.randn((co, ci, kh, kw), generator=g) * math.sqrt(2.0 / fan_out)
            sd[f"{enc}.{name}.bias"] = uniform((co,), 1.0 / math.sqrt(fan_in))
        # output projection conv2 (extractor.py:142): 128 -> 256
        sd[f"{enc}.conv2.weight"] = torch.randn((256, 128, 1, 1), generator=g) * math.sqrt(2.0 / 256)
        sd[f"{enc}.conv2.bias"] = uniform((256,), 1.0 / math.sqrt(128))
    for name, ch in _encoder_bn_names():
        w = 0.8 + 0.4 * torch.rand((ch,), generator=g)
        b = 0.1 * torch.randn((ch,), generator=g)
        rm = 0.1 * torch.randn((ch,), generator=g)
        rv = 0.5 + torch.rand((ch,), generator=g)
        keys = [f"cnet.{name}"]
        if name.endswith("norm3"):
            keys.append(f"cnet.{name[:-len('norm3')]}downsample.1")
        for k in keys:
            sd[f"{k}.weight"] = w.clone()
            sd[f"{k}.bias"] = b.clone()
            sd[f"{k}.running_mean"] = rm.clone()
            sd[f"{k}.running_var"] = rv.clone()
            sd[f"{k}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    for name, (co, ci, kh, kw) in UPDATE_CONV_SHAPES:
        bound = 1.0 / math.sqrt(ci * kh * kw)
        sd[f"update_block.{name}.weight"] = uniform((co, ci, kh, kw), bound)
        sd[f"update_block.{name}.bias"] = uniform((co,), bound)
    return sd


def to_float64(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """The same weights as float64 tensors: `raft_forward(to_float64(sd), image1.double(), image2.double())` evaluates the network
    in double precision -- the yardstick that says how far the fp32 CPU oracle and the fp32 HIP path each sit from the exact
    result of the same arithmetic (tests/test_gpu_raft.py, bench.py `verified`)."""
    return {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}


def strip_module_prefix(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """`raft-things.pth` is saved from `DataParallel(RAFT)` (ofgen_keyframe_inpaint.py:59-60)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------
def _norm(sd, key: str, x: Tensor, kind: str) -> Tensor:
    if kind == "instance":
        # nn.InstanceNorm2d defaults: affine=False, no running stats, eps=1e-5 (extractor.py:27-31)
        return F.instance_norm(x, eps=1e-5)
    if kind == "batch":
        return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                            sd[key + ".weight"], sd[key + ".bias"], training=False, eps=1e-5)
    if kind == "batch_train":
        # nn.BatchNorm2d in TRAIN mode: the reference's RAFT_2 never calls .eval() (ofgen_keyframe_inpaint.py:47-60), so
        # raft.py:55's BatchNorm layers normalise with the statistics of the call's batch -- always ONE image there.  A batch
        # here stands for as many reference calls: every image is normalised by itself.
        return torch.cat([F.batch_norm(x[i:i + 1], None, None, sd[key + ".weight"], sd[key + ".bias"], training=True, eps=1e-5)
                          for i in range(x.shape[0])], 0)
    raise ValueError(kind)


def _conv(sd, key: str, x: Tensor, stride=1, padding=0) -> Tensor:
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=padding)


def residual_block(sd, p: str, x: Tensor, kind: str, stride: int) -> Tensor:
    """extractor.py:46-56."""
    y = torch.relu(_norm(sd, p + ".norm1", _conv(sd, p + ".conv1", x, stride, 1), kind))
    y = torch.relu(_norm(sd, p + ".norm2", _conv(sd, p + ".conv2", y, 1, 1), kind))
    if stride != 1:
        x = _norm(sd, p + ".norm3", _conv(sd, p + ".downsample.0", x, stride, 0), kind)
    return torch.relu(x + y)


def encoder(sd, enc: str, x: Tensor, kind: str) -> Tensor:
    """BasicEncoder.forward, extractor.py:168-192.  x: [B,3,H,W] already scaled to [-1,1]."""
    x = torch.relu(_norm(sd, f"{enc}.norm1", _conv(sd, f"{enc}.conv1", x, 2, 3), kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = residual_block(sd, f"{enc}.layer{li}.0", x, kind, stride)
        x = residual_block(sd, f"{enc}.layer{li}.1", x, kind, 1)
    return _conv(sd, f"{enc}.conv2", x, 1, 0)


# --------------------------------------------------------------------------------------
# correlation: volume + pyramid + lookup   (CorrBlock)
# --------------------------------------------------------------------------------------
def corr_volume(fmap1: Tensor, fmap2: Tensor) -> Tensor:
    """corr.py:52-60 -> [B, h, w, h, w];  value = <f1[:, i], f2[:, j]> / sqrt(D)."""
    b, d, h, w = fmap1.shape
    a = fmap1.reshape(b, d, h * w).transpose(1, 2)
    c = torch.matmul(a, fmap2.reshape(b, d, h * w))
    return (c / math.sqrt(d)).reshape(b, h, w, h, w)


def corr_pyramid(fmap1: Tensor, fmap2: Tensor, levels: int = CORR_LEVELS) -> List[Tensor]:
    """corr.py:19-27 -> list of [B*h*w, 1, h/2^l, w/2^l]."""
    b, d, h, w = fmap1.shape
    vol = corr_volume(fmap1, fmap2).reshape(b * h * w, 1, h, w)
    pyr = [vol]
    for _ in range(levels - 1):
        vol = F.avg_pool2d(vol, 2, stride=2)
        pyr.append(vol)
    return pyr


def _sample_zero_pad(img: Tensor, x: Tensor, y: Tensor) -> Tensor:
    """Bilinear sample of img[N,1,H,W] at pixel coordinates (x, y) [N, ...], zeros outside;
    the direct statement of grid_sample(align_corners=True, padding_mode='zeros') that
    utils.py:57-71 reaches through a normalise/un-normalise round trip."""
    n, _, hh, ww = img.shape
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    fx = x - x0
    fy = y - y0
    x0 = x0.long()
    y0 = y0.long()
    flat = img.reshape(n, hh * ww)
    shp = x.shape
    out = torch.zeros(shp, dtype=img.dtype)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi = x0 + dx
            yi = y0 + dy
            ok = (xi >= 0) & (xi < ww) & (yi >= 0) & (yi < hh)
            idx = (yi.clamp(0, hh - 1) * ww + xi.clamp(0, ww - 1)).reshape(n, -1)
            v = torch.gather(flat, 1, idx).reshape(shp)
            out = out + torch.where(ok, v, torch.zeros_like(v)) * (wx * wy)
    return out


def corr_lookup(pyr: List[Tensor], coords: Tensor, radius: int = CORR_RADIUS) -> Tensor:
    """CorrBlock.__call__, corr.py:29-50.  coords: [B,2,h,w] (channel 0 = x, 1 = y).
    Returns [B, L*(2r+1)^2, h, w]; channel k = l*(2r+1)^2 + i*(2r+1) + j samples level l at
    (x/2^l + (i-r), y/2^l + (j-r)) -- the x offset is the slow index."""
    b, _, h, w = coords.shape
    n = b * h * w
    rd = 2 * radius + 1
    c = coords.permute(0, 2, 3, 1).reshape(n, 2)
    off = torch.arange(-radius, radius + 1, dtype=coords.dtype)
    outs = []
    for lvl, vol in enumerate(pyr):
        cx = (c[:, 0] / 2 ** lvl).reshape(n, 1, 1) + off.reshape(1, rd, 1)   # varies with i
        cy = (c[:, 1] / 2 ** lvl).reshape(n, 1, 1) + off.reshape(1, 1, rd)   # varies with j
        xs = cx.expand(n, rd, rd)
        ys = cy.expand(n, rd, rd)
        outs.append(_sample_zero_pad(vol, xs, ys).reshape(b, h, w, rd * rd))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------
# local (on-the-fly) correlation   (AlternateCorrBlock + alt_cuda_corr.forward)
# --------------------------------------------------------------------------------------
def local_corr_level(fmap1: Tensor, fmap2: Tensor, coords: Tensor, radius: int) -> Tensor:
    """Semantics of `alt_cuda_corr.forward` (correlation_kernel.cu:18-119, :260-286).

    fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] channels-last; coords [B,N,H1,W1,2] (x,y).
    Returns corr [B,N,(2r+1)^2,H1,W1] *without* the 1/sqrt(C) factor (applied by the caller,
    corr.py:91).  For every integer tap (iy,ix) in [0,2r+1]^2 at (floor(y)-r+iy, floor(x)-r+ix) the
    dot product s is splatted with bilinear weights into up to four output channels
    (cu:92-114); channel index = iy + (2r+1)*ix (x-major).  Out-of-range taps contribute 0."""
    b, h1, w1, c = fmap1.shape
    _, h2, w2, _ = fmap2.shape
    nn = coords.shape[1]
    rd = 2 * radius + 1
    out = torch.zeros((b, nn, rd * rd, h1, w1), dtype=fmap1.dtype)
    for n in range(nn):
        x = coords[:, n, :, :, 0]
        y = coords[:, n, :, :, 1]
        x0 = torch.floor(x)
        y0 = torch.floor(y)
        dx = x - x0
        dy = y - y0
        x0 = x0.long()
        y0 = y0.long()
        # dot products with the (rd+1)^2 integer neighbours
        s = torch.zeros((b, rd + 1, rd + 1, h1, w1), dtype=fmap1.dtype)
        bi = torch.arange(b).reshape(b, 1, 1).expand(b, h1, w1)
        for iy in range(rd + 1):
            for ix in range(rd + 1):
                yy = y0 - radius + iy
                xx = x0 - radius + ix
                ok = (yy >= 0) & (yy < h2) & (xx >= 0) & (xx < w2)
                g = fmap2[bi, yy.clamp(0, h2 - 1), xx.clamp(0, w2 - 1)]      # [b,h1,w1,c]
                d = (fmap1 * g).sum(-1)
                s[:, iy, ix] = torch.where(ok, d, torch.zeros_like(d))
        o = out[:, n].reshape(b, rd, rd, h1, w1)       # [b, ix, iy, h, w] since channel = iy + rd*ix
        for iy in range(rd + 1):
            for ix in range(rd + 1):
                v = s[:, iy, ix]
                if iy > 0 and ix > 0:
                    o[:, ix - 1, iy - 1] += v * dy * dx
                if iy > 0 and ix < rd:
                    o[:, ix, iy - 1] += v * dy * (1 - dx)
                if iy < rd and ix > 0:
                    o[:, ix - 1, iy] += v * (1 - dy) * dx
                if iy < rd and ix < rd:
                    o[:, ix, iy] += v * (1 - dy) * (1 - dx)
    return out


def alternate_corr_lookup(fmap1: Tensor, fmap2: Tensor, coords: Tensor,
                          levels: int = CORR_LEVELS, radius: int = CORR_RADIUS) -> Tensor:
    """AlternateCorrBlock, corr.py:63-91: fmap2 is average-pooled per level, fmap1 stays full-res
    (`pyramid[0][0]`, corr.py:82), coords are divided by 2^l.  Returns [B, L*(2r+1)^2, h, w]."""
    b, d, h, w = fmap1.shape
    f1 = fmap1.permute(0, 2, 3, 1).contiguous()
    outs = []
    f2 = fmap2
    for lvl in range(levels):
        f2l = f2.permute(0, 2, 3, 1).contiguous()
        c = (coords.permute(0, 2, 3, 1) / 2 ** lvl).reshape(b, 1, h, w, 2)
        outs.append(local_corr_level(f1, f2l, c, radius).squeeze(1))
        f2 = F.avg_pool2d(f2, 2, stride=2)
    return torch.stack(outs, dim=1).reshape(b, -1, h, w) / math.sqrt(d)


# --------------------------------------------------------------------------------------
# update block
# --------------------------------------------------------------------------------------
def motion_encoder(sd, flow: Tensor, corr: Tensor) -> Tensor:
    """BasicMotionEncoder.forward, update.py:88-97."""
    p = "update_block.encoder."
    cor = torch.relu(_conv(sd, p + "convc1", corr, 1, 0))
    cor = torch.relu(_conv(sd, p + "convc2", cor, 1, 1))
    flo = torch.relu(_conv(sd, p + "convf1", flow, 1, 3))
    flo = torch.relu(_conv(sd, p + "convf2", flo, 1, 1))
    out = torch.relu(_conv(sd, p + "conv", torch.cat([cor, flo], 1), 1, 1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(sd, h: Tensor, x: Tensor) -> Tensor:
    """SepConvGRU.forward, update.py:44-60: a (1x5) pass then a (5x1) pass."""
    p = "update_block.gru."
    for tag, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(sd, p + "convz" + tag, hx, 1, pad))
        r = torch.sigmoid(_conv(sd, p + "convr" + tag, hx, 1, pad))
        q = torch.tanh(_conv(sd, p + "convq" + tag, torch.cat([r * h, x], 1), 1, pad))
        h = (1 - z) * h + z * q
    return h


def update_block(sd, net: Tensor, inp: Tensor, corr: Tensor, flow: Tensor,
                 want_mask: bool = True) -> Tuple[Tensor, Optional[Tensor], Tensor]:
    """BasicUpdateBlock.forward, update.py:127-136."""
    x = torch.cat([inp, motion_encoder(sd, flow, corr)], 1)
    net = sep_conv_gru(sd, net, x)
    p = "update_block.flow_head."
    delta = _conv(sd, p + "conv2", torch.relu(_conv(sd, p + "conv1", net, 1, 1)), 1, 1)
    mask = None
    if want_mask:
        m = torch.relu(_conv(sd, "update_block.mask.0", net, 1, 1))
        mask = 0.25 * _conv(sd, "update_block.mask.2", m, 1, 0)
    return net, mask, delta


# --------------------------------------------------------------------------------------
# convex upsample, grids, padding
# --------------------------------------------------------------------------------------
def coords_grid(b: int, h: int, w: int, dtype=torch.float32) -> Tensor:
    """utils.py:74-77: channel 0 = x (column), channel 1 = y (row)."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=dtype),
                            torch.arange(w, dtype=dtype), indexing="ij")
    return torch.stack([xs, ys], 0)[None].repeat(b, 1, 1, 1)


def upsample_flow(flow: Tensor, mask: Tensor) -> Tensor:
    """raft.py:72-83.  flow [B,2,h,w], mask [B,576,h,w] (channel = k*64 + i*8 + j,
    k = 3x3 neighbour in unfold order, (i,j) = sub-pixel) -> [B,2,8h,8w]."""
    b, _, h, w = flow.shape
    m = torch.softmax(mask.reshape(b, 1, 9, 8, 8, h, w), dim=2)
    nb = F.unfold(8 * flow, [3, 3], padding=1).reshape(b, 2, 9, 1, 1, h, w)
    up = (m * nb).sum(2)                       # [b,2,8,8,h,w]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(b, 2, 8 * h, 8 * w)


def pad_to_8(img: Tensor) -> Tuple[Tensor, Tuple[int, int, int, int]]:
    """InputPadder 'sintel' mode (utils.py:9-19): replicate-pad, centred."""
    hh, ww = img.shape[-2:]
    ph = (((hh // 8) + 1) * 8 - hh) % 8
    pw = (((ww // 8) + 1) * 8 - ww) % 8
    pad = (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
    if ph == 0 and pw == 0:
        return img, pad
    return F.pad(img, pad, mode="replicate"), pad


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
@torch.no_grad()
def raft_forward(sd: Dict[str, Tensor], image1: Tensor, image2: Tensor, iters: int = 20,
                 alternate_corr: bool = False, trace: Optional[dict] = None, cnet_norm: str = "eval"
                 ) -> Tuple[Tensor, Tensor]:
    """RAFT.forward(test_mode=True), raft.py:86-144.  image1/2: [B,3,H,W] float in [0,255] (RGB),
    H and W multiples of 8.  Returns (flow_low [B,2,H/8,W/8], flow_up [B,2,H,W]).
    `trace`, if a dict, receives intermediate tensors for stage-level parity tests.
    cnet_norm: 'eval' = BatchNorm with running statistics (model.eval()); 'batch' = BatchNorm in train mode, one image
    per call (RAFT_2 as the reference wrote it, see `_norm`)."""
    i1 = 2 * (image1 / 255.0) - 1.0
    i2 = 2 * (image2 / 255.0) - 1.0
    b = i1.shape[0]
    fm = encoder(sd, "fnet", torch.cat([i1, i2], 0), "instance")
    fmap1, fmap2 = fm[:b], fm[b:]
    cn = encoder(sd, "cnet", i1, {"eval": "batch", "batch": "batch_train"}[cnet_norm])
    net = torch.tanh(cn[:, :HDIM])
    inp = torch.relu(cn[:, HDIM:HDIM + CDIM])
    _, _, h, w = fmap1.shape
    pyr = None if alternate_corr else corr_pyramid(fmap1, fmap2)
    coords0 = coords_grid(b, h, w, fmap1.dtype)
    coords1 = coords_grid(b, h, w, fmap1.dtype)
    if trace is not None:
        trace.update(fmap1=fmap1, fmap2=fmap2, net0=net, inp=inp, pyramid=pyr)
    mask = None
    for it in range(iters):
        if alternate_corr:
            corr = alternate_corr_lookup(fmap1, fmap2, coords1)
        else:
            corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        # the reference evaluates the mask head and the upsample on every iteration
        # (raft.py:128-139) and keeps only the last; evaluating it once is identical.
        net, mask, delta = update_block(sd, net, inp, corr, flow, want_mask=(it == iters - 1))
        coords1 = coords1 + delta
        if trace is not None and it == 0:
            trace.update(corr_it0=corr, net_it0=net, delta_it0=delta)
        if trace is not None and (it + 1) in trace.get("keep_iters", ()):
            trace.setdefault("flow_low_at", {})[it + 1] = coords1 - coords0    # 1/8-resolution flow after it + 1 iterations
    flow_low = coords1 - coords0
    flow_up = upsample_flow(flow_low, mask)
    if trace is not None:
        trace.update(mask=mask)
    return flow_low, flow_up


@torch.no_grad()
def raft2_calc(sd, img1_bgr, img2_bgr, iters: int = 20, cnet_norm: str = "batch"):
    """`RAFT_2.calc` (ofgen_keyframe_inpaint.py:62-71): BGR uint8 HxWx3 numpy in -> flow
    f32[H',W',2] numpy out, H',W' = padded size (the reference does not un-pad, :70)."""
    import numpy as np
    a = torch.from_numpy(np.ascontiguousarray(img1_bgr[:, :, ::-1])).permute(2, 0, 1).float()[None]
    c = torch.from_numpy(np.ascontiguousarray(img2_bgr[:, :, ::-1])).permute(2, 0, 1).float()[None]
    a, _ = pad_to_8(a)
    c, _ = pad_to_8(c)
    _, up = raft_forward(sd, a, c, iters=iters, cnet_norm=cnet_norm)
    return up[0].permute(1, 2, 0).contiguous().numpy()


def local_corr_backward(fmap1: Tensor, fmap2: Tensor, coords: Tensor, corr_grad: Tensor, radius: int):
    """Semantics of `alt_cuda_corr.backward` (correlation_kernel.cu:122-256, :288-324): for every pixel, n and
    integer tap (iy, ix) the adjoint of the bilinear splat gives one weight g (cu:207-222); then
    fmap1_grad[pixel] += g * fmap2[tap] and fmap2_grad[tap] += g * fmap1[pixel] (cu:224-237); out-of-range taps
    contribute nothing; coords_grad is returned as zeros (cu:305).  Returns (fmap1_grad, fmap2_grad, coords_grad)."""
    b, h1, w1, c = fmap1.shape
    _, h2, w2, _ = fmap2.shape
    nn = coords.shape[1]
    rd = 2 * radius + 1
    g1 = torch.zeros_like(fmap1)
    g2 = torch.zeros_like(fmap2)
    bi = torch.arange(b).reshape(b, 1, 1).expand(b, h1, w1)
    for n in range(nn):
        x = coords[:, n, :, :, 0]
        y = coords[:, n, :, :, 1]
        x0 = torch.floor(x)
        y0 = torch.floor(y)
        dx = x - x0
        dy = y - y0
        x0 = x0.long()
        y0 = y0.long()
        go = corr_grad[:, n].reshape(b, rd, rd, h1, w1)          # [b, ix, iy, h, w]: channel = iy + rd*ix
        for iy in range(rd + 1):
            for ix in range(rd + 1):
                g = torch.zeros((b, h1, w1), dtype=fmap1.dtype)
                if iy > 0 and ix > 0:
                    g = g + go[:, ix - 1, iy - 1] * dy * dx
                if iy > 0 and ix < rd:
                    g = g + go[:, ix, iy - 1] * dy * (1 - dx)
                if iy < rd and ix > 0:
                    g = g + go[:, ix - 1, iy] * (1 - dy) * dx
                if iy < rd and ix < rd:
                    g = g + go[:, ix, iy] * (1 - dy) * (1 - dx)
                yy = y0 - radius + iy
                xx = x0 - radius + ix
                ok = (yy >= 0) & (yy < h2) & (xx >= 0) & (xx < w2)
                g = torch.where(ok, g, torch.zeros_like(g))
                yc, xc = yy.clamp(0, h2 - 1), xx.clamp(0, w2 - 1)
                g1 += g[..., None] * fmap2[bi, yc, xc]
                g2.index_put_((bi, yc, xc), g[..., None] * fmap1, accumulate=True)
    return g1, g2, torch.zeros_like(coords)
