"""MI355X-native optical-flow-guided frame synthesis hot path (flow -> warp -> inpaint mask).

Drop-in for the `pdcnet_of.py` / `ofgen_*` call surface of zyddnys/sd_animation_optical_flow; all
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of `libofx.so`
(include/ofx.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"
