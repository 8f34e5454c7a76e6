"""Top-level `xformers` shim: the reference's `ldm/` imports `xformers` / `xformers.ops` (ldm/modules/attention.py:12-18,
ldm/modules/diffusionmodules/model.py:11-16) and calls exactly one function of it.  With the repository root on
`sys.path`, `xformers.ops.memory_efficient_attention` is the MI355X-native attention of
`sd_animation_optical_flow_amd.attention` (HIP kernels; no CPU fallback)."""
from . import ops  # noqa: F401

__version__ = "0.0.0+ofx"
