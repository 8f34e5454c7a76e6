"""Drop-in for the reference's `alt_cuda_corr` torch extension (RAFT/alt_cuda_corr/correlation.cpp:51-54).

    import sd_animation_optical_flow_amd.alt_cuda_corr as alt_cuda_corr
    corr, = alt_cuda_corr.forward(fmap1, fmap2, coords, radius)      # RAFT/core/corr.py:86

`forward(fmap1 f32[B,H1,W1,C], fmap2 f32[B,H2,W2,C], coords f32[B,N,H1,W1,2], radius) -> [corr]`
with `corr f32[B,N,(2r+1)^2,H1,W1]`; all tensors must be CUDA and contiguous (RuntimeError otherwise,
correlation.cpp:19-21).  Unlike the reference, the launch goes to the *current* torch stream (the CUDA
original uses the legacy default stream) and C only needs to be a multiple of 4 (the original silently
requires a multiple of 32).  `backward(fmap1, fmap2, coords, corr_grad, radius) -> [fmap1_grad, fmap2_grad,
coords_grad]` (correlation.cpp:35-49) is provided for completeness of the native inventory; the reference's
inference callers run under no_grad and never reach it.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops


def forward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, radius: int) -> List[torch.Tensor]:
    return [ops.local_corr(fmap1, fmap2, coords, int(radius))]


def backward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, corr_grad: torch.Tensor, radius: int) -> List[torch.Tensor]:
    g1, g2 = ops.local_corr_backward(fmap1, fmap2, coords, corr_grad, int(radius))
    return [g1, g2, torch.zeros_like(coords)]                 # coords_grad is all zeros in the reference too (cu:305)
