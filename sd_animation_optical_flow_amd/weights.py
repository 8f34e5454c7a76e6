"""Checkpoint loading for the flow network.

The reference loads `RAFT/models/raft-things.pth` into `DataParallel(RAFT)` (ofgen.py:63-68); that file
is not shipped with the reference, so besides `load_checkpoint` this module offers
`random_state_dict(seed)`: seeded weights with the reference's exact `state_dict()` key set, used by
bench.py and the demos when no checkpoint is available (timing is weight-independent).
"""
from __future__ import annotations

import math
import os
from typing import Dict

import torch

_ENC_CONVS = [("conv1", (64, 3, 7, 7))]
_cin = 64
for _li, (_dim, _stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
    _ENC_CONVS.append((f"layer{_li}.0.conv1", (_dim, _cin, 3, 3)))
    _ENC_CONVS.append((f"layer{_li}.0.conv2", (_dim, _dim, 3, 3)))
    if _stride != 1:
        _ENC_CONVS.append((f"layer{_li}.0.downsample.0", (_dim, _cin, 1, 1)))
    _ENC_CONVS.append((f"layer{_li}.1.conv1", (_dim, _dim, 3, 3)))
    _ENC_CONVS.append((f"layer{_li}.1.conv2", (_dim, _dim, 3, 3)))
    _cin = _dim

_BN = [("norm1", 64)]
for _li, (_dim, _stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
    for _bi in (0, 1):
        _BN += [(f"layer{_li}.{_bi}.norm1", _dim), (f"layer{_li}.{_bi}.norm2", _dim)]
    if _stride != 1:
        _BN.append((f"layer{_li}.0.norm3", _dim))

_UPDATE = [
    ("encoder.convc1", (256, 324, 1, 1)), ("encoder.convc2", (192, 256, 3, 3)),
    ("encoder.convf1", (128, 2, 7, 7)), ("encoder.convf2", (64, 128, 3, 3)),
    ("encoder.conv", (126, 256, 3, 3)),
    ("gru.convz1", (128, 384, 1, 5)), ("gru.convr1", (128, 384, 1, 5)), ("gru.convq1", (128, 384, 1, 5)),
    ("gru.convz2", (128, 384, 5, 1)), ("gru.convr2", (128, 384, 5, 1)), ("gru.convq2", (128, 384, 5, 1)),
    ("flow_head.conv1", (256, 128, 3, 3)), ("flow_head.conv2", (2, 256, 3, 3)),
    ("mask.0", (256, 128, 3, 3)), ("mask.2", (576, 256, 1, 1)),
]


def random_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights for the basic (non-small) RAFT: Kaiming-normal(fan_out) encoder convolutions
    (RAFT/core/extractor.py:150-152), uniform(+-1/sqrt(fan_in)) elsewhere, non-trivial BatchNorm
    running statistics for the context encoder."""
    g = torch.Generator().manual_seed(int(seed))
    sd: Dict[str, torch.Tensor] = {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound

    for enc in ("fnet", "cnet"):
        for name, (co, ci, kh, kw) in _ENC_CONVS:
            sd[f"{enc}.{name}.weight"] = torch.randn((co, ci, kh, kw), generator=g) * math.sqrt(2.0 / (co * kh * kw))
            sd[f"{enc}.{name}.bias"] = uni((co,), 1.0 / math.sqrt(ci * kh * kw))
        sd[f"{enc}.conv2.weight"] = torch.randn((256, 128, 1, 1), generator=g) * math.sqrt(2.0 / 256)
        sd[f"{enc}.conv2.bias"] = uni((256,), 1.0 / math.sqrt(128))
    for name, ch in _BN:
        w = 0.8 + 0.4 * torch.rand((ch,), generator=g)
        b = 0.1 * torch.randn((ch,), generator=g)
        rm = 0.1 * torch.randn((ch,), generator=g)
        rv = 0.5 + torch.rand((ch,), generator=g)
        keys = [f"cnet.{name}"]
        if name.endswith("norm3"):
            keys.append(f"cnet.{name[:-5]}downsample.1")
        for k in keys:
            sd[k + ".weight"], sd[k + ".bias"] = w.clone(), b.clone()
            sd[k + ".running_mean"], sd[k + ".running_var"] = rm.clone(), rv.clone()
            sd[k + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    for name, (co, ci, kh, kw) in _UPDATE:
        bound = 1.0 / math.sqrt(ci * kh * kw)
        sd[f"update_block.{name}.weight"] = uni((co, ci, kh, kw), bound)
        sd[f"update_block.{name}.bias"] = uni((co,), bound)
    return sd


def load_checkpoint(ckpt) -> Dict[str, torch.Tensor]:
    """`ckpt`: a state_dict, a path to a `torch.save`d state_dict (optionally under a 'state_dict'
    key, optionally with `module.` prefixes), or the string 'random:<seed>'."""
    if isinstance(ckpt, dict):
        sd = ckpt
    elif isinstance(ckpt, str) and ckpt.startswith("random:"):
        return random_state_dict(int(ckpt.split(":", 1)[1]))
    elif isinstance(ckpt, (str, os.PathLike)):
        if not os.path.exists(ckpt):
            raise FileNotFoundError(
                f"flow checkpoint {ckpt!r} not found (pass a RAFT state_dict path such as raft-things.pth, "
                "a state_dict, or 'random:<seed>')")
        # a state_dict of tensors needs nothing but tensors: never unpickle arbitrary objects from a user-supplied
        # path (OFX_UNSAFE_CHECKPOINT=1 opts back in for legacy .pth.tar files that pickle their args namespace)
        sd = torch.load(ckpt, map_location="cpu", weights_only=os.environ.get("OFX_UNSAFE_CHECKPOINT") != "1")
        if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
    else:
        raise TypeError(f"unsupported checkpoint spec {type(ckpt)}")
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
