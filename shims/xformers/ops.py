"""`xformers.ops` surface used by the reference (ldm/modules/attention.py:314,426; diffusionmodules/model.py:234)."""
from sd_animation_optical_flow_amd.attention import memory_efficient_attention  # noqa: F401


class MemoryEfficientAttentionFlashAttentionOp:      # named in the reference as an optional `op=` choice; ignored here
    pass


__all__ = ["memory_efficient_attention", "MemoryEfficientAttentionFlashAttentionOp"]
