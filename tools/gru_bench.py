#!/usr/bin/env python3
"""Where the GRU convolutions' deficit against the plain 128x128 layers comes from (B = 64 at 512x768: M = 393216 pixels).

The step's z|r convolution (1x5 / 5x1 over hx[:, 0:256], row stride 384, epilogue sigmoid + r*h) runs at ~141 TF, the q
convolution (two input segments: rh with row stride 128, then hx[:, 128:256]; epilogue h = (1-z) h + z tanh(.)) at ~136, the plain
3x3 layers at 142-143.  This times the same launches with one property changed at a time:

    q            as the engine launches it (two segments, GRU_Q epilogue, addend)
    q plain-epi  the same operands, plain store epilogue           -> what the q epilogue costs
    q one-seg    one 256-channel segment (row stride 256), GRU_Q   -> what the second segment costs
    q ld136      rh with row stride 136 instead of 128             -> what the power-of-two row stride costs
    zr           as the engine launches it
    zr plain-epi plain store epilogue
usage: python tools/gru_bench.py [libofx variant .so]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import _lib

EPI_PLAIN, EPI_ZR, EPI_Q = 0, 1, 2
B, h, w = 64, 96, 64
M = B * h * w


def run(libpath):
    lib = C.CDLL(libpath)
    lib.ofx_conv2d.restype = C.c_int
    lib.ofx_conv2d.argtypes = [C.POINTER(_lib.ConvDesc), C.c_void_p]
    dev = "cuda"
    hx = torch.randn((M, 384), device=dev) * 0.5
    rh128 = torch.randn((M, 128), device=dev) * 0.5
    rh136 = torch.randn((M, 136), device=dev) * 0.5
    x256 = torch.randn((M, 256), device=dev) * 0.5
    gadd = torch.randn((M, 768), device=dev) * 0.1
    z = torch.rand((M, 128), device=dev)
    rh_out = torch.empty((M, 128), device=dev)
    out = torch.empty((M, 256), device=dev)
    Kq = 5 * 256
    wq = torch.randn((128, Kq), device=dev) * 0.02
    wzr = torch.randn((256, Kq), device=dev) * 0.02
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def desc(kh, kw, cout, epi, in0, ld0, c0, in1=None, ld1=0, c1=0, wt=None, addend=None):
        d = _lib.ConvDesc()
        d.in0, d.ld0, d.c0 = in0.data_ptr(), ld0, c0
        if in1 is not None:
            d.in1, d.ld1, d.c1 = in1, ld1, c1
        d.w = wt.data_ptr()
        d.B, d.Hin, d.Win, d.Hout, d.Wout, d.Cout = B, h, w, h, w, cout
        d.KH, d.KW, d.stride, d.padH, d.padW = kh, kw, 1, kh // 2, kw // 2
        d.act, d.epi = 0, epi
        if epi == EPI_PLAIN:
            d.out, d.ldo = out.data_ptr(), 256
        else:
            d.aux_z, d.aux_rh, d.aux_h, d.ldh = z.data_ptr(), rh_out.data_ptr(), hx.data_ptr(), 384
        if addend is not None:
            d.addend, d.ldadd = addend.data_ptr(), 768
        return d

    mot = hx.data_ptr() + 128 * 4
    cases = []
    for kh, kw, tag in ((1, 5, "1x5"), (5, 1, "5x1")):
        cases += [
            (f"q {tag}", desc(kh, kw, 128, EPI_Q, rh128, 128, 128, mot, 384, 128, wq, gadd)),
            (f"q {tag} no addend", desc(kh, kw, 128, EPI_Q, rh128, 128, 128, mot, 384, 128, wq)),
            (f"q {tag} plain-epi", desc(kh, kw, 128, EPI_PLAIN, rh128, 128, 128, mot, 384, 128, wq)),
            (f"q {tag} one-seg", desc(kh, kw, 128, EPI_Q, x256, 256, 256, None, 0, 0, wq, gadd)),
            (f"q {tag} one-seg plain", desc(kh, kw, 128, EPI_PLAIN, x256, 256, 256, None, 0, 0, wq)),
            (f"q {tag} ld136", desc(kh, kw, 128, EPI_Q, rh136, 136, 128, mot, 384, 128, wq, gadd)),
            (f"zr {tag}", desc(kh, kw, 256, EPI_ZR, hx, 384, 256, None, 0, 0, wzr, gadd)),
            (f"zr {tag} plain-epi", desc(kh, kw, 256, EPI_PLAIN, hx, 384, 256, None, 0, 0, wzr)),
        ]
    print("==", os.path.basename(libpath))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        for name, d in cases:
            for _ in range(3):
                st = lib.ofx_conv2d(C.byref(d), s)
                assert st == 0, (name, st)
            torch.cuda.synchronize()
            n = 20
            e0.record()
            for _ in range(n):
                lib.ofx_conv2d(C.byref(d), s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = 2.0 * M * d.Cout * Kq
            if rep:
                print(f"  {name:<24}{ms * 1e3:9.1f} us {fl / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    for l in (sys.argv[1:] or [_lib.LIB_PATH]):
        run(l)
