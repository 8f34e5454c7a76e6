"""Top-level `xformers` shim: the reference's `ldm/` imports `xformers` / `xformers.ops` (ldm/modules/attention.py:12-18,
ldm/modules/diffusionmodules/model.py:11-16) and calls exactly one function of it.  It lives under `shims/` -- NOT on the
import path by default, so it can never shadow a real xformers install: add `<repo>/shims` to `sys.path` (INTEGRATION.md
section 1d) and `xformers.ops.memory_efficient_attention` is the MI355X-native attention of
`sd_animation_optical_flow_amd.attention` (HIP kernels; no CPU fallback)."""
from . import ops  # noqa: F401

__version__ = "0.0.0+ofx"
