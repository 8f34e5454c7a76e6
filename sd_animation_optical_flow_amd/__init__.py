"""MI355X-native optical-flow-guided frame synthesis hot path (flow -> warp -> inpaint mask).

Drop-in for the `pdcnet_of.py` / `ofgen_*` call surface of zyddnys/sd_animation_optical_flow; all
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of `libofx.so`
(include/ofx.h).  There is no CPU fallback.

Modules: `pdcnet_of`, `ofgen`, `alt_cuda_corr` (the reference's names; `ofgen.keyframe_conv` = `KeyframeConv`),
`workspace` (`VideoData` / `VideoFrameIndices`: the reference's on-disk workspace), `raft` (the native executor's handle),
`clip` (frame-parallel sharding + key-frame broadcast), `handoff` + `vae` + `attention` (SD-inpaint inputs, first-stage
latent and `memory_efficient_attention` on the device), `keyframes` (Canny-based key-frame detector), `ops` (one wrapper per
C-ABI entry point), `_lib` (ctypes binding).  The repository root carries two shims named as the reference imports them:
`alt_cuda_corr` and `xformers`.
"""
__version__ = "0.2.0"
