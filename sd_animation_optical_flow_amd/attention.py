"""`xformers.ops.memory_efficient_attention` for the reference's UNet on MI355X.

The reference cannot run `ldm/` without xformers: both attention modules call
`xformers.ops.memory_efficient_attention(q, k, v, attn_bias=attn_bias, op=self.attention_op)`
(ldm/modules/attention.py:314 and :426) on `[batch * heads, tokens, dim_head]` tensors, with an optional additive bias
built from the flow-guided neighbourhood (:283-311, :392-423).  `memory_efficient_attention` below has that signature
and those semantics -- softmax(q k^T / sqrt(dim_head) + attn_bias) v -- and runs on the HIP kernels of libofx.so
(`ofx_attention_f32`: both products on the fp32 matrix cores, exact fp32 softmax; the UNet's head sizes 40 / 64 / 80 /
128 / 160 take the fused online-softmax kernel of csrc/attn_flash.hip, which keeps the scores on the CU).  The `xformers` package under `shims/`
re-exports it: with `<repo>/shims` on `sys.path`, `import xformers.ops` in the reference resolves here unchanged.

Half / bfloat16 inputs are computed in fp32 and cast back (xformers computes them at reduced precision; fp32 is the
stricter arithmetic).  A 4-D bias [batch, heads, Nq, Nk] is accepted like xformers'.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def memory_efficient_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn_bias: Optional[torch.Tensor] = None,
                               p: float = 0.0, scale: Optional[float] = None, op=None) -> torch.Tensor:
    if p != 0.0:
        raise NotImplementedError("attention dropout is a training feature; the reference's inference calls use p = 0")
    if attn_bias is not None and not torch.is_tensor(attn_bias):
        # xformers' AttentionBias objects (LowerTriangularMask, BlockDiagonalMask ...): the reference never passes one
        raise TypeError(f"attn_bias must be a tensor (additive bias); {type(attn_bias).__name__} objects are not supported")
    if not query.is_cuda:
        raise RuntimeError("memory_efficient_attention needs CUDA tensors (no CPU fallback)")
    dt = query.dtype
    four_d = query.dim() == 4                      # xformers' [B, M, H, K] layout
    if four_d:
        B, M, Hh, K = query.shape
        to3 = lambda t: t.permute(0, 2, 1, 3).reshape(B * Hh, t.shape[1], K)
        q, k, v = to3(query), to3(key), to3(value)
    else:
        q, k, v = query, key, value
    q, k, v = (t.to(torch.float32).contiguous() for t in (q, k, v))
    bias = None
    if attn_bias is not None:
        bias = attn_bias.to(torch.float32)
        if bias.dim() == 4:
            bias = bias.reshape(-1, bias.shape[-2], bias.shape[-1])
        if bias.dim() == 3 and bias.shape[0] == 1:
            bias = bias[0]
        if bias.dim() == 3 and bias.shape[0] != q.shape[0]:
            bias = bias.expand(q.shape[0], -1, -1)
        bias = bias.contiguous()
    out = ops.attention(q, k, v, bias, scale)
    if four_d:
        out = out.reshape(B, Hh, M, K).permute(0, 2, 1, 3)
    return out.to(dt)
