#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernel on the update-block / encoder shapes (B=64 @512x768).
usage: python tools/conv_bench.py [libofx variant .so ...]"""
import ctypes as C, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import _lib
SHAPES = [  # name, B, H, W, Cin, Cout, kh, kw, stride, tile
    ("gru 1x5 256->256", 64, 96, 64, 256, 256, 1, 5, 1, 0),
    ("gru 5x1 256->128", 64, 96, 64, 256, 128, 5, 1, 1, 0),
    ("3x3 256->192", 64, 96, 64, 256, 192, 3, 3, 1, 0),
    ("3x3 128->256", 64, 96, 64, 128, 256, 3, 3, 1, 0),
    ("1x1 324->256", 64, 96, 64, 324, 256, 1, 1, 1, 0),
    ("enc 3x3 64->64 @1/2", 16, 384, 256, 64, 64, 3, 3, 1, 0),
    ("enc 3x3 96->96 @1/4", 16, 192, 128, 96, 96, 3, 3, 1, 0),
]
TILES = [0, 128128, 16128128, 128064, 16128064, 64064, 16064064]
if os.environ.get("CONV_BENCH_N64"):
    SHAPES = [
        ("enc 3x3 64->64 @1/2", 64, 384, 256, 64, 64, 3, 3, 1, 0),
        ("enc 7x7 4->64 s2", 64, 768, 512, 4, 64, 7, 7, 2, 0),
        ("3x3 128->64", 64, 96, 64, 128, 64, 3, 3, 1, 0),
        ("3x3 256->64", 64, 96, 64, 256, 64, 3, 3, 1, 0),
        ("3x3 512->64", 64, 96, 64, 512, 64, 3, 3, 1, 0),
        ("3x3 64->64 @1/8", 64, 96, 64, 64, 64, 3, 3, 1, 0),
        ("1x1 64->64 @1/2", 64, 384, 256, 64, 64, 1, 1, 1, 0),
    ]
    TILES = [0, 16128064, 16256064]
if os.environ.get("CONV_BENCH_B1"):
    SHAPES = [
        ("b1 gru 1x5 256->256", 1, 96, 64, 256, 256, 1, 5, 1, 0),
        ("b1 gru 5x1 256->128", 1, 96, 64, 256, 128, 5, 1, 1, 0),
        ("b1 3x3 256->192", 1, 96, 64, 256, 192, 3, 3, 1, 0),
        ("b1 3x3 128->256", 1, 96, 64, 128, 256, 3, 3, 1, 0),
        ("b1 1x1 324->256", 1, 96, 64, 324, 256, 1, 1, 1, 0),
        ("b1 enc 3x3 64->64 @1/2", 1, 384, 256, 64, 64, 3, 3, 1, 0),
        ("b1 enc 3x3 128->128 @1/8", 1, 96, 64, 128, 128, 3, 3, 1, 0),
    ]


if os.environ.get("CONV_BENCH_96"):      # the encoders' 96-channel stage at the bench batch (quarter resolution)
    SHAPES = [
        ("warm-up 3x3 256->192", 64, 96, 64, 256, 192, 3, 3, 1, 0),
        ("enc 3x3 96->96 @1/4 B64", 64, 192, 128, 96, 96, 3, 3, 1, 0),
    ]
    TILES = [0, 16256096]
if os.environ.get("CONV_BENCH_C1"):      # the motion encoder's 1x1 over the correlation channels, as laid out today and padded to whole chunks
    SHAPES = [
        ("warm-up 3x3 256->192", 64, 96, 64, 256, 192, 3, 3, 1, 0),
        ("1x1 324->256", 64, 96, 64, 324, 256, 1, 1, 1, 0),
        ("1x1 336->256", 64, 96, 64, 336, 256, 1, 1, 1, 0),
        ("1x1 352->256", 64, 96, 64, 352, 256, 1, 1, 1, 0),
        ("1x1 256->576", 64, 96, 64, 256, 576, 1, 1, 1, 0),
    ]
    TILES = [0, 16128064, 32128128, 32128064]
if os.environ.get('CONV_BENCH_TILES'):
    TILES = [int(t) for t in os.environ['CONV_BENCH_TILES'].split(',')]
if os.environ.get("CONV_BENCH_ONLY"):
    SHAPES = [x for x in SHAPES if os.environ["CONV_BENCH_ONLY"] in x[0]]


def run(libpath):
    lib = C.CDLL(libpath)
    lib.ofx_conv2d.restype = C.c_int
    lib.ofx_conv2d.argtypes = [C.POINTER(_lib.ConvDesc), C.c_void_p]
    print("==", os.path.basename(libpath))
    for name, B, H, W, ci, co, kh, kw, st, _ in SHAPES:
      for tile in TILES:
            x = torch.randn((B, H, W, ci), device="cuda")
            if os.environ.get("CONV_BENCH_ZERO"):       # power probe: same instruction stream on all-zero operands
                x.zero_()
            K = kh * kw * ci
            Kp = (K + 31) // 32 * 32
            w = torch.randn((co, Kp), device="cuda") * 0.02
            if os.environ.get("CONV_BENCH_ZERO"):
                w.zero_()
            out = torch.empty((B, H // st, W // st, co), device="cuda")
            d = _lib.ConvDesc()
            d.in0, d.ld0, d.c0 = x.data_ptr(), ci, ci
            d.w = w.data_ptr(); d.out = out.data_ptr(); d.ldo = co
            d.B, d.Hin, d.Win, d.Hout, d.Wout, d.Cout = B, H, W, H // st, W // st, co
            d.KH, d.KW, d.stride, d.padH, d.padW = kh, kw, st, kh // 2, kw // 2
            d.act, d.epi, d.tile = 1, 0, tile
            d.precision = int(os.environ.get('CONV_BENCH_PREC', '0'))
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                assert lib.ofx_conv2d(C.byref(d), s) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = int(os.environ.get('CONV_BENCH_REPS', '10'))
            e0.record()
            for _ in range(n):
                lib.ofx_conv2d(C.byref(d), s)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = 2.0 * B * (H // st) * (W // st) * co * K
            print(f"  {name:<24} tile {tile:>9} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
if __name__ == "__main__":
    libs = sys.argv[1:] or [_lib.LIB_PATH]
    for l in libs:
        run(l)
