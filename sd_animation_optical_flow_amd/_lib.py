"""ctypes binding of libofx.so (the C ABI declared in include/ofx.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, importing
anything that computes raises immediately (`OfxLibraryError`).  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C sd_animation_optical_flow_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OFX_LIB_PATH: a diagnostic build of the same library (e.g. tools/alt/libofx_exact_gates.so); the default is the in-tree one
LIB_PATH = os.environ.get("OFX_LIB_PATH") or os.path.join(_HERE, "libofx.so")


class OfxLibraryError(RuntimeError):
    pass


class OfxError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        super().__init__(f"{where}: {error_string(code)} (code {code})")


class ConvDesc(C.Structure):
    """Mirror of `ofx_conv_desc` (include/ofx.h)."""
    _fields_ = [
        ("in0", C.c_void_p), ("ld0", C.c_int), ("c0", C.c_int),
        ("in1", C.c_void_p), ("ld1", C.c_int), ("c1", C.c_int),
        ("w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int),
        ("res", C.c_void_p), ("ldres", C.c_int),
        ("nmean", C.c_void_p), ("nrstd", C.c_void_p),
        ("addend", C.c_void_p), ("ldadd", C.c_int),
        ("aux_z", C.c_void_p), ("aux_rh", C.c_void_p), ("aux_h", C.c_void_p), ("ldh", C.c_int),
        ("aux_coords", C.c_void_p), ("aux_flow4", C.c_void_p),
        ("a_zs", C.c_long), ("w_zs", C.c_long), ("o_zs", C.c_long), ("nz", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("Cout", C.c_int), ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int),
        ("padH", C.c_int), ("padW", C.c_int),
        ("act", C.c_int), ("epi", C.c_int), ("tile", C.c_int), ("precision", C.c_int),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_size_t),
    ]


class Tensor(C.Structure):
    """Mirror of `ofx_tensor`."""
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_long * 4)]


_p, _i, _l, _f, _z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# every symbol include/ofx.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ofx_version": (_i, []),
    "ofx_error_string": (C.c_char_p, [_i]),
    "ofx_prof_enable": (_i, [_i]),
    "ofx_prof_collect": (_i, [C.c_char_p, _z]),
    "ofx_warp_u8": (_i, [_p, _l, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "ofx_warp_f32": (_i, [_p, _l, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "ofx_resize_cubic_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ofx_generate_mask": (_i, [_p, _p, _p, _i, _i, _i, _f, _i, _i, _p]),
    "ofx_dilate_u8": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "ofx_expand_mask": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ofx_travel_distance": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "ofx_flow_magnitude": (_i, [_p, _p, _l, _p]),
    "ofx_travel_mask": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "ofx_merge_images": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ofx_mix_frames": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "ofx_conf_sum": (_i, [_p, _p, _i, _l, _i, _i, _p]),
    "ofx_fb_confidence": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "ofx_warp_and_mask": (_i, [_p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _i, _p]),
    "ofx_conv2d": (_i, [C.POINTER(ConvDesc), _p]),
    "ofx_pack_conv_weight": (_l, [_p, _i, _i, _i, _i, _i, _p]),
    "ofx_split_conv_weight": (_i, [_p, _l, _p]),
    "ofx_split_conv_weight3": (_i, [_p, _l, _p]),
    "ofx_gaussian_blur_u8": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "ofx_resize_bicubic_u8": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ofx_sd_handoff": (_i, [_p] * 9 + [_i] * 5 + [_p]),
    "ofx_detect_edges_scratch_bytes": (_z, [_i, _i, _i]),
    "ofx_detect_edges": (_i, [_p, _p, _p, _z, _i, _i, _i, _i, _p]),
    "ofx_abs_diff_sum_u8": (_i, [_p, _l, _p, _l, _p, _i, _l, _p]),
    "ofx_inorm_stats": (_i, [_p, _i, _p, _p, _p, _i, _l, _i, _f, _p]),
    "ofx_inorm_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p]),
    "ofx_preprocess_u8": (_i, [_p, _p, _l, _i, _p]),
    "ofx_groupnorm_scratch_bytes": (C.c_size_t, [_i, _i]),
    "ofx_groupnorm": (_i, [_p, _p, _p, _p, _p, C.c_size_t, _i, C.c_long, _i, _i, C.c_float, _i, _p]),
    "ofx_softmax_rows": (_i, [_p, C.c_long, C.c_long, _i, C.c_float, _p, C.c_long, C.c_long, _p]),
    "ofx_attention_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "ofx_attention_f32": (_i, [_p, _p, _p, _p, C.c_long, _p, _i, _i, _i, _i, C.c_float, _p, C.c_size_t, _p]),
    "ofx_corr_slice_floats": (_i, [_i, _i]),
    "ofx_corr_volume": (_i, [_p, _p, C.POINTER(_p), _i, _i, _i, _i, _i, _p]),
    "ofx_corr_volume_split": (_i, [_p, _p, C.POINTER(_p), _i, _i, _i, _i, _i, _i, _i, _p]),
    "ofx_corr_lookup": (_i, [C.POINTER(_p), _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ofx_local_corr_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "ofx_local_corr_bwd": (_i, [_p] * 6 + [_i] * 8 + [_p]),
    "ofx_avgpool2_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "ofx_upsample_flow": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "ofx_upsample_flow_warp": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "ofx_raft_create": (_i, [C.POINTER(Tensor), _i, C.POINTER(_p)]),
    "ofx_raft_destroy": (_i, [_p]),
    "ofx_raft_workspace_bytes": (_z, [_p, _i, _i, _i]),
    "ofx_raft_forward": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "ofx_raft_forward_warp": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _f, _p, _p, _z, _p]),
    "ofx_raft_workspace_bytes_pairs": (_z, [_p, _i, _i, _i, _i]),
    "ofx_raft_forward_pairs": (_i, [_p, _p, _i, C.POINTER(_i), C.POINTER(_i), _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "ofx_raft_forward_pairs_warp": (_i, [_p, _p, _i, C.POINTER(_i), C.POINTER(_i), _i, _i, _i, _i, _i, _p, _p, _p, _f, _i, _p, _p, _z, _p]),
    "ofx_raft_buffer": (_i, [_p, C.c_char_p, C.POINTER(_p), C.POINTER(_z)]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libofx.so once and attach the signatures.  Raises OfxLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OfxLibraryError(
            f"{LIB_PATH} not found: the HIP library has not been built. There is no CPU fallback; "
            "run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root.")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise OfxLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise OfxLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def error_string(code: int) -> str:
    return lib().ofx_error_string(int(code)).decode()


def check(code: int, where: str) -> None:
    if code != 0:
        raise OfxError(int(code), where)
