"""Top-level `alt_cuda_corr` module: the name the reference imports (RAFT/core/corr.py:5-9, used at :86).

With the repository root on `sys.path`, the reference's `AlternateCorrBlock` works unchanged:

    import alt_cuda_corr
    corr, = alt_cuda_corr.forward(fmap1, fmap2, coords, radius)

It re-exports `forward` / `backward` of `sd_animation_optical_flow_amd.alt_cuda_corr` (same signatures as the
pybind11 module of RAFT/alt_cuda_corr/correlation.cpp:51-54), which launch the HIP kernels of libofx.so
(`ofx_local_corr_fwd` / `ofx_local_corr_bwd`).  There is no CPU path: CPU or non-contiguous tensors raise
RuntimeError exactly where the reference's TORCH_CHECKs do (correlation.cpp:19-21).
"""
from sd_animation_optical_flow_amd.alt_cuda_corr import backward, forward  # noqa: F401

__all__ = ["forward", "backward"]
