#!/usr/bin/env python3
"""Tile sweep of the update-block / encoder convolutions at SMALL batches (B = 1 ... 32 frames at 512x768): which tile each layer wants
when the grid does not fill the chip.  Per (layer, B): the launcher's own choice (tile 0, with split-K scratch like the executor's)
and every forced tile that applies.  Feeds the selection rule in conv.hip (`ofx_conv2d_alpha`, tile selection).
    python tools/small_batch_tune.py [B ...]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_animation_optical_flow_amd import _lib

LAYERS = [  # name, H, W (per image), Cin, Cout, kh, kw, stride
    ("convc1 1x1 324->256", 96, 64, 324, 256, 1, 1, 1),
    ("convc2 3x3 256->192", 96, 64, 256, 192, 3, 3, 1),
    ("conv   3x3 256->126", 96, 64, 256, 126, 3, 3, 1),
    ("gru.zr 1x5 256->256", 96, 64, 256, 256, 1, 5, 1),
    ("gru.q  5x1 256->128", 96, 64, 256, 128, 5, 1, 1),
    ("fh1    3x3 128->256", 96, 64, 128, 256, 3, 3, 1),
    ("convf2 3x3 128->64 ", 96, 64, 128, 64, 3, 3, 1),
    ("enc l1 3x3 64->64 @1/2", 384, 256, 64, 64, 3, 3, 1),
    ("enc l2 3x3 96->96 @1/4", 192, 128, 96, 96, 3, 3, 1),
    ("enc l2.0 3x3s2 64->96", 384, 256, 64, 96, 3, 3, 2),
    ("enc l3 3x3 128->128 @1/8", 96, 64, 128, 128, 3, 3, 1),
    ("enc l3.0 3x3s2 96->128", 192, 128, 96, 128, 3, 3, 2),
    ("enc out 1x1 128->256", 96, 64, 128, 256, 1, 1, 1),
    ("stem 7x7s2 4->64", 768, 512, 4, 64, 7, 7, 2),
]
TILES = [0, 16128128, 16128064, 16128192, 16128096, 32128032, 32064064, 2032064064, 16256064]


def main():
    lib = C.CDLL(os.environ.get("OFX_LIB_PATH", _lib.LIB_PATH))
    lib.ofx_conv2d.restype = C.c_int
    lib.ofx_conv2d.argtypes = [C.POINTER(_lib.ConvDesc), C.c_void_p]
    batches = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
    sk = torch.zeros((65536 + 1024 * 4 * 64 * 64 * 4) // 4, dtype=torch.float32, device="cuda")
    only = os.environ.get("TUNE_ONLY")
    for name, H, W, ci, co, kh, kw, st in LAYERS:
        if only and only not in name:
            continue
        for B in batches:
            x = torch.randn((B, H, W, ci), device="cuda")
            K = kh * kw * ci
            Kp = (K + 31) // 32 * 32
            w = torch.randn((co, Kp), device="cuda") * 0.02
            out = torch.empty((B, H // st, W // st, co), device="cuda")
            fl = 2.0 * B * (H // st) * (W // st) * co * K
            res = []
            for tile in TILES:
                d = _lib.ConvDesc()
                d.in0, d.ld0, d.c0 = x.data_ptr(), ci, ci
                d.w = w.data_ptr(); d.out = out.data_ptr(); d.ldo = co
                d.B, d.Hin, d.Win, d.Hout, d.Wout, d.Cout = B, H, W, H // st, W // st, co
                d.KH, d.KW, d.stride, d.padH, d.padW = kh, kw, st, kh // 2, kw // 2
                d.act, d.epi, d.tile = 1, 0, tile
                if tile == 0:
                    d.splitk_ws, d.splitk_ws_bytes = sk.data_ptr(), sk.numel() * 4
                s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                if lib.ofx_conv2d(C.byref(d), s) != 0:
                    continue
                for _ in range(2):
                    lib.ofx_conv2d(C.byref(d), s)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 20 if B <= 8 else 8
                e0.record()
                for _ in range(n):
                    lib.ofx_conv2d(C.byref(d), s)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                res.append((tile, us))
            best = min(u for _, u in res)
            auto = dict(res).get(0)
            line = "  ".join(f"{t}:{u:7.1f}{'*' if u == best else ' '}" for t, u in res)
            print(f"{name:<26} B={B:<3} auto/best {auto / best:5.2f}  ideal {fl / 157.3e6:7.1f} us | {line}", flush=True)


if __name__ == "__main__":
    main()
