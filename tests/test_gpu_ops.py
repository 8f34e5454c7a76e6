"""-m gpu: operator-level parity of the HIP kernels (through the C ABI) against the CPU oracle.

Tolerances: integer / byte paths are bit-exact; fp32 paths are compared with an absolute tolerance
stated per test (fp32 accumulation order differs between MFMA tiles and the CPU's loops).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mask_oracle as MO
from oracle import raft_oracle as RO
from oracle import warp_oracle as WO


def _diff(a, b):
    """'' when equal, else a short description (keeps pytest output readable)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return f"shape {a.shape} vs {b.shape}"
    bad = a != b
    if not bad.any():
        return ""
    i = tuple(np.argwhere(bad)[0])
    return f"{int(bad.sum())}/{a.size} differ; first at {i}: {a[i]!r} vs {b[i]!r}"


def _ops():
    from sd_animation_optical_flow_amd import ops
    return ops


def nhwc(x):  # NCHW cpu -> NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(x):  # NHWC cuda -> NCHW cpu
    return x.permute(0, 3, 1, 2).contiguous().cpu()


# --------------------------------------------------------------------------------------
# implicit-GEMM convolution
# --------------------------------------------------------------------------------------
CONV_CASES = [
    # (B, H, W, Cin, Cout, kh, kw, stride, act, tile)
    (2, 24, 20, 64, 64, 3, 3, 1, "relu", 0),
    (1, 24, 20, 64, 96, 3, 3, 2, None, 0),
    (2, 17, 13, 128, 126, 3, 3, 1, "relu", 0),       # ragged M and N
    (1, 16, 16, 324, 256, 1, 1, 1, "relu", 0),       # K = 324 (not a multiple of 32)
    (1, 12, 20, 384, 128, 1, 5, 1, "tanh", 0),
    (1, 20, 12, 384, 256, 5, 1, 1, "sigmoid", 0),
    (1, 16, 16, 256, 2, 3, 3, 1, None, 0),           # tiny Cout
    (1, 16, 16, 256, 576, 1, 1, 1, None, 0),
    (3, 40, 24, 64, 64, 1, 1, 2, None, 0),           # 1x1 stride-2 downsample
    (1, 32, 32, 128, 128, 3, 3, 1, None, 128128),
    (1, 32, 32, 128, 128, 3, 3, 1, None, 128064),
    (1, 32, 32, 128, 128, 3, 3, 1, None, 128032),
    (1, 32, 32, 128, 128, 3, 3, 1, None, 64064),
    (1, 32, 32, 128, 128, 3, 3, 1, None, 16064064),          # explicit BK = 16 / 32
    (1, 32, 32, 128, 128, 3, 3, 1, None, 32128128),
    (1, 32, 32, 128, 128, 3, 3, 1, None, 32128064),
    (2, 19, 23, 96, 192, 3, 3, 1, "relu", 16128192),         # 128x192 tile, ragged M (874 rows)
    (2, 21, 19, 64, 96, 3, 3, 1, "relu", 16128096),          # 128x96 tile (B rows staged past the tile), ragged M
    (1, 24, 24, 96, 200, 1, 1, 1, None, 16128096),           # ... three N tiles, ragged N
    (1, 20, 28, 96, 64, 3, 3, 1, None, 2032064064),          # paired K pipelines, odd chunk count (27), ragged M
    (1, 20, 28, 64, 128, 1, 1, 1, "relu", 2032064064),       # ... two chunks: one per pipeline
    (1, 8, 8, 32, 64, 1, 1, 1, None, 2032064064),            # ... one chunk: the second pipeline adds nothing
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(cuda, case):
    ops = _ops()
    B, H, W, ci, co, kh, kw, stride, act, tile = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)))
    x = torch.randn((B, ci, H, W), generator=g)
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    b = torch.randn((co,), generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=(kh // 2, kw // 2))
    ref = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, None: lambda t: t}[act](ref)
    wp = ops.pack_conv_weight(w).cuda()
    out = ops.conv2d_nhwc(nhwc(x), wp, kh, kw, co, stride=stride, shift=b.cuda(), act=act, tile=tile)
    torch.cuda.synchronize()
    err = (nchw(out) - ref).abs().max().item()
    assert err < 2e-5, err      # fp32 accumulate, K <= 3456


@pytest.mark.parametrize("case", [(1, 96, 64, 256, 256, 3, 3, "relu"), (1, 96, 64, 256, 128, 1, 5, "tanh"), (1, 50, 37, 128, 192, 3, 3, None),
                                  (1, 20, 20, 256, 64, 3, 3, "relu")])
def test_conv2d_split_k_small_grids(cuda, case):
    """Grids of a few hundred 64x64 tiles are split along K over workgroups; the last one to arrive adds the parked
    partial tiles and runs the epilogue.  Same result as the unsplit launch up to fp32 summation order, and the
    arrival counters are left at zero so the scratch can be reused launch after launch."""
    ops = _ops()
    B, H, W, ci, co, kh, kw, act = case
    g = torch.Generator().manual_seed(H * W + ci)
    x = torch.randn((B, ci, H, W), generator=g)
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    b = torch.randn((co,), generator=g)
    ref = F.conv2d(x, w, b, padding=(kh // 2, kw // 2))
    ref = {"relu": torch.relu, "tanh": torch.tanh, None: lambda t: t}[act](ref)
    wp = ops.pack_conv_weight(w).cuda()
    ws = torch.zeros((65536 + 512 * 4 * 64 * 64 * 4,), dtype=torch.uint8, device="cuda")
    plain = ops.conv2d_nhwc(nhwc(x), wp, kh, kw, co, shift=b.cuda(), act=act)
    for _ in range(3):                                   # scratch reused without re-zeroing
        out = ops.conv2d_nhwc(nhwc(x), wp, kh, kw, co, shift=b.cuda(), act=act, splitk_ws=ws)
        assert (nchw(out) - ref).abs().max().item() < 2e-5
        assert (out - plain).abs().max().item() < 1e-5
    assert int(ws[:65536].view(torch.int32).abs().max()) == 0
    assert not torch.equal(out, plain) or co == 64        # the split path really ran (different summation order)


@pytest.mark.parametrize("case", [(2, 24, 20, 64, 64, 3, 3, 1), (1, 33, 17, 256, 256, 1, 5, 1), (1, 16, 16, 324, 256, 1, 1, 1),
                                  (1, 40, 24, 96, 192, 3, 3, 2)])
def test_conv2d_bf16x3_mode(cuda, case):
    """Opt-in split-bf16 arithmetic: operands carry ~16 mantissa bits -> relative error ~1e-5 of the output scale."""
    ops = _ops()
    B, H, W, ci, co, kh, kw, stride = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((B, ci, H, W), generator=g)
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    b = torch.randn((co,), generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=(kh // 2, kw // 2)).float()
    wp = ops.pack_conv_weight(w).cuda()
    out32 = ops.conv2d_nhwc(nhwc(x), wp, kh, kw, co, stride=stride, shift=b.cuda())
    out16 = ops.conv2d_nhwc(nhwc(x), wp, kh, kw, co, stride=stride, shift=b.cuda(), precision="bf16x3")
    e32 = (nchw(out32) - ref).abs().max().item()
    e16 = (nchw(out16) - ref).abs().max().item()
    assert e32 < 2e-5 and e16 < 2e-4, (e32, e16)          # outputs are O(1): 16-bit operands -> ~1e-5 .. 1e-4
    assert e16 > 0                                           # and it really is a different arithmetic
    # pre-split weights (what the executor uploads once): same hi / lo values -> bit-identical to the on-the-fly split
    outw = ops.conv2d_nhwc(nhwc(x), ops.split_conv_weight(wp.cpu()).cuda(), kh, kw, co, stride=stride, shift=b.cuda(),
                           precision="bf16x3_w")
    assert torch.equal(outw, out16)


@pytest.mark.parametrize("case", [(2, 24, 20, 64, 64, 3, 3, 1), (1, 33, 17, 256, 256, 1, 5, 1), (1, 16, 16, 324, 256, 1, 1, 1),
                                  (1, 40, 24, 96, 192, 3, 3, 2), (1, 64, 48, 3, 64, 7, 7, 2)])
def test_conv2d_bf16x6_mode_is_fp32_accurate(cuda, case):
    """Opt-in three-piece split: hi + mid + lo is the fp32 operand exactly and only products below 2^-23 are dropped,
    so the error against a float64 convolution must be of the SAME size as the native fp32 path's."""
    ops = _ops()
    B, H, W, ci, co, kh, kw, stride = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn((B, ci, H, W), generator=g)
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    b = torch.randn((co,), generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=(kh // 2, kw // 2)).float()
    wp = ops.pack_conv_weight(w).cuda()
    xin = nhwc(x)
    if ci == 3:
        xin = torch.cat([xin, torch.zeros_like(xin[..., :1])], -1).contiguous()
        wp = ops.pack_conv_weight(torch.cat([w, torch.zeros_like(w[:, :1])], 1)).cuda()
    out32 = ops.conv2d_nhwc(xin, wp, kh, kw, co, stride=stride, shift=b.cuda())
    out6 = ops.conv2d_nhwc(xin, wp, kh, kw, co, stride=stride, shift=b.cuda(), precision="bf16x6")
    e32 = (nchw(out32) - ref).abs().max().item()
    e6 = (nchw(out6) - ref).abs().max().item()
    assert e32 < 2e-5 and e6 < 2e-5 and e6 < 3 * e32 + 1e-6, (e32, e6)
    assert not torch.equal(out6, out32)                      # a different summation, not the fp32 kernel by accident
    # pre-split weights (round 5; what the executor uploads once): the same three pieces -> bit-identical to the on-the-fly split
    out6w = ops.conv2d_nhwc(xin, ops.split_conv_weight3(wp.cpu()).cuda(), kh, kw, co, stride=stride, shift=b.cuda(), precision="bf16x6_w")
    assert torch.equal(out6w, out6)


@pytest.mark.parametrize("case", [  # B, H, W, c0, c1, cout, kh, kw, tile
    (2, 16, 32, 64, 0, 64, 3, 3, 16128064), (1, 24, 48, 256, 0, 192, 3, 3, 16128192), (2, 8, 16, 128, 256, 256, 1, 5, 16128128),
    (1, 40, 16, 128, 128, 128, 5, 1, 16128128), (3, 8, 32, 96, 0, 96, 3, 3, 16128096), (1, 64, 96, 128, 0, 256, 3, 3, 16128128),
    (1, 16, 16, 16, 0, 70, 3, 3, 16128128),
    # the 64x64 small-grid tile: 8x8 patches, 32-channel slabs
    (2, 16, 24, 64, 0, 64, 3, 3, 32064064), (1, 8, 40, 128, 256, 256, 1, 5, 32064064), (1, 24, 8, 128, 128, 128, 5, 1, 32064064),
    (1, 32, 32, 96, 0, 100, 3, 3, 32064064),
    # maps that are not whole 8x16 patches: the last patch of a row / column hangs over
    (2, 13, 22, 64, 0, 64, 3, 3, 16128064), (1, 65, 97, 128, 128, 128, 1, 5, 16128128), (1, 9, 40, 128, 0, 192, 5, 1, 16128192),
    (3, 7, 5, 32, 0, 96, 3, 3, 16128096),
    # the 256x64 tile (tile override only): 16x16 patches, whole and overhanging
    (2, 32, 32, 64, 0, 64, 3, 3, 16256064), (1, 40, 25, 128, 0, 64, 1, 5, 16256064)])
def test_conv2d_halo_patch_kernel(cuda, case):
    """The halo-patch instantiation (stride-1 3x3 / 1x5 / 5x1 on maps made of whole 8x16 patches; forced here through the
    128-row tile override, the executor reaches it by itself at batch size) against F.conv2d: borders, every tile width, one
    and two input segments, addend / residual / scale epilogue, a ragged N; and against the general kernel (OFX_CONV_NO_PATCH
    is read once per process, so the comparison is with the 64x64 BK = 16 tile, which never takes the patch path)."""
    ops = _ops()
    B, H, W, c0, c1, co, kh, kw, tile = case
    g = torch.Generator().manual_seed(sum(case))
    xa = torch.randn((B, c0, H, W), generator=g)
    xb = torch.randn((B, c1, H, W), generator=g) if c1 else None
    ci = c0 + c1
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    sc = torch.rand((co,), generator=g) + 0.5
    sh = torch.randn((co,), generator=g)
    res = torch.randn((B, co, H, W), generator=g)
    xin = xa if xb is None else torch.cat([xa, xb], 1)
    y = F.conv2d(xin.double(), w.double(), None, padding=(kh // 2, kw // 2)) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref = torch.relu(torch.relu(y) + res.double()).float()
    kwargs = dict(shift=sh.cuda(), scale=sc.cuda(), act="relu", x2=None if xb is None else nhwc(xb), res=nhwc(res))
    out = ops.conv2d_nhwc(nhwc(xa), ops.pack_conv_weight(w).cuda(), kh, kw, co, tile=tile, **kwargs)
    assert (nchw(out) - ref).abs().max().item() < 2e-5
    gen = ops.conv2d_nhwc(nhwc(xa), ops.pack_conv_weight(w).cuda(), kh, kw, co, tile=16064064, **kwargs)
    assert (nchw(gen) - ref).abs().max().item() < 2e-5
    assert not torch.equal(out, gen) or ci <= 16            # another summation order: the patch path really ran
    if tile == 32064064:
        # split-K over channel slabs (the executor's small-grid schedule): automatic tile choice + a split-K workspace
        ws = torch.zeros((64 << 20,), dtype=torch.uint8, device="cuda")
        sk = ops.conv2d_nhwc(nhwc(xa), ops.pack_conv_weight(w).cuda(), kh, kw, co, splitk_ws=ws, **kwargs)
        assert (nchw(sk) - ref).abs().max().item() < 2e-5
        sk2 = ops.conv2d_nhwc(nhwc(xa), ops.pack_conv_weight(w).cuda(), kh, kw, co, splitk_ws=ws, **kwargs)
        assert torch.equal(sk, sk2)                          # deterministic reduction order


def test_conv2d_halo_patch_with_fused_instance_norm(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    raw = torch.randn((2, 64, 16, 32), generator=g) * 2 + 0.5
    w = torch.randn((64, 64, 3, 3), generator=g) / 24
    mean, rstd = ops.inorm_stats(nhwc(raw))
    ref = F.conv2d(torch.relu(F.instance_norm(raw)).double(), w.double(), None, padding=1).float()
    out = ops.conv2d_nhwc(nhwc(raw), ops.pack_conv_weight(w).cuda(), 3, 3, 64, nmean=mean, nrstd=rstd, tile=16128064)
    assert (nchw(out) - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 0, 64, 3, 3, 16128064), (1, 24, 48, 128, 128, 256, 1, 5, 16128128), (1, 13, 22, 128, 0, 128, 3, 3, 16128128),
                                  (1, 40, 16, 96, 0, 100, 5, 1, 16128128)])
def test_conv2d_halo_patch_kernel_bf16x3(cuda, case):
    """The split-bf16 arithmetic on the halo patch (conversion once per slab): same error bar as the general bf16x3 kernel, and
    the pre-split weight format is bit-identical to splitting on the fly."""
    ops = _ops()
    B, H, W, c0, c1, co, kh, kw, tile = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    xa = torch.randn((B, c0, H, W), generator=g)
    xb = torch.randn((B, c1, H, W), generator=g) if c1 else None
    ci = c0 + c1
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    b = torch.randn((co,), generator=g)
    xin = xa if xb is None else torch.cat([xa, xb], 1)
    ref = torch.relu(F.conv2d(xin.double(), w.double(), b.double(), padding=(kh // 2, kw // 2))).float()
    wp = ops.pack_conv_weight(w).cuda()
    kwargs = dict(shift=b.cuda(), act="relu", x2=None if xb is None else nhwc(xb))
    out = ops.conv2d_nhwc(nhwc(xa), wp, kh, kw, co, tile=tile, precision="bf16x3", **kwargs)
    e = (nchw(out) - ref).abs().max().item()
    assert 0 < e < 2e-4, e
    outw = ops.conv2d_nhwc(nhwc(xa), ops.split_conv_weight(wp.cpu()).cuda(), kh, kw, co, tile=tile, precision="bf16x3_w", **kwargs)
    assert torch.equal(outw, out)
    gen = ops.conv2d_nhwc(nhwc(xa), wp, kh, kw, co, tile=16064064, precision="bf16x3", **kwargs)      # 64-row tile: general path
    assert (gen - out).abs().max().item() < 2e-4 and not torch.equal(gen, out)


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 0, 64, 3, 3, 16128064), (1, 24, 48, 128, 128, 256, 1, 5, 16128128), (1, 13, 22, 128, 0, 128, 3, 3, 16128128)])
def test_conv2d_halo_patch_kernel_bf16x6(cuda, case):
    """Three-piece split on the halo patch: still at fp32 accuracy (error against a float64 convolution of the size of the
    native kernel's)."""
    ops = _ops()
    B, H, W, c0, c1, co, kh, kw, tile = case
    g = torch.Generator().manual_seed(sum(case) + 6)
    xa = torch.randn((B, c0, H, W), generator=g)
    xb = torch.randn((B, c1, H, W), generator=g) if c1 else None
    ci = c0 + c1
    w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
    b = torch.randn((co,), generator=g)
    xin = xa if xb is None else torch.cat([xa, xb], 1)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=(kh // 2, kw // 2)).float()
    wp = ops.pack_conv_weight(w).cuda()
    kwargs = dict(shift=b.cuda(), x2=None if xb is None else nhwc(xb))
    out32 = ops.conv2d_nhwc(nhwc(xa), wp, kh, kw, co, tile=tile, **kwargs)
    out6 = ops.conv2d_nhwc(nhwc(xa), wp, kh, kw, co, tile=tile, precision="bf16x6", **kwargs)
    e32 = (nchw(out32) - ref).abs().max().item()
    e6 = (nchw(out6) - ref).abs().max().item()
    assert e32 < 2e-5 and e6 < 2e-5 and e6 < 3 * e32 + 1e-6, (e32, e6)
    assert not torch.equal(out6, out32)
    out6w = ops.conv2d_nhwc(nhwc(xa), ops.split_conv_weight3(wp.cpu()).cuda(), kh, kw, co, tile=tile, precision="bf16x6_w", **kwargs)
    assert torch.equal(out6w, out6)                          # pre-split weights on the halo patch: the same pieces, the same bits


def test_conv2d_randomised_sweep_over_schedules(cuda):
    """Seeded sweep: random maps, channel counts (one and two segments), filter shapes, strides and forced tiles, so that the
    general gather, the scalar-coordinate and the halo-patch schedules (whole and overhanging patches, all patch tiles) all get
    shapes nobody hand-picked; every result against a float64 convolution."""
    ops = _ops()
    rng = np.random.default_rng(2024)
    tiles = [0, 16128128, 16128064, 16128192, 16128096, 32064064, 16064064, 16256064, 32128032]
    shapes = [(3, 3), (1, 5), (5, 1), (1, 1), (7, 7)]
    worst = 0.0
    for it in range(48):
        kh, kw = shapes[rng.integers(len(shapes))]
        stride = int(rng.choice([1, 1, 1, 2])) if (kh, kw) in ((3, 3), (7, 7)) else 1
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(5, 45)), int(rng.integers(5, 50))
        c0 = int(rng.choice([4, 16, 32, 48, 64, 96, 128]))
        c1 = int(rng.choice([0, 0, 32, 64])) if c0 % 32 == 0 else 0
        co = int(rng.choice([2, 24, 64, 70, 96, 128, 192, 200]))
        tile = int(tiles[rng.integers(len(tiles))])
        g = torch.Generator().manual_seed(1000 + it)
        xa = torch.randn((B, c0, H, W), generator=g)
        xb = torch.randn((B, c1, H, W), generator=g) if c1 else None
        ci = c0 + c1
        w = torch.randn((co, ci, kh, kw), generator=g) / np.sqrt(ci * kh * kw)
        b = torch.randn((co,), generator=g)
        xin = xa if xb is None else torch.cat([xa, xb], 1)
        ref = torch.relu(F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=(kh // 2, kw // 2))).float()
        out = ops.conv2d_nhwc(nhwc(xa), ops.pack_conv_weight(w).cuda(), kh, kw, co, stride=stride, shift=b.cuda(), act="relu",
                              x2=None if xb is None else nhwc(xb), tile=tile)
        assert tuple(nchw(out).shape) == tuple(ref.shape), (it, kh, kw, stride, B, H, W, c0, c1, co, tile)
        err = (nchw(out) - ref).abs().max().item()
        worst = max(worst, err)
        assert err < 3e-5, (it, err, kh, kw, stride, B, H, W, c0, c1, co, tile)
    assert worst > 0


def test_conv2d_two_segments_residual_and_scale(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    xa = torch.randn((2, 128, 12, 16), generator=g)
    xb = torch.randn((2, 256, 12, 16), generator=g)
    w = torch.randn((128, 384, 3, 3), generator=g) / 60
    sc = torch.rand((128,), generator=g) + 0.5
    sh = torch.randn((128,), generator=g)
    res = torch.randn((2, 128, 12, 16), generator=g)
    y = F.conv2d(torch.cat([xa, xb], 1), w, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = torch.relu(torch.relu(y) + res)
    out = ops.conv2d_nhwc(nhwc(xa), ops.pack_conv_weight(w).cuda(), 3, 3, 128, shift=sh.cuda(), scale=sc.cuda(), act="relu",
                          x2=nhwc(xb), res=nhwc(res))
    assert (nchw(out) - ref).abs().max().item() < 2e-5


def test_conv2d_padded_cin_and_fused_instance_norm(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    # 7x7 stride-2 conv on a 3-channel image padded to 4 channels
    img = torch.randint(0, 256, (2, 40, 48, 3), generator=g, dtype=torch.uint8)
    x0 = ops.preprocess_u8(img.cuda())
    xr = (2 * (img.float() / 255.0) - 1.0)
    assert torch.equal(x0[..., :3].cpu(), xr) and float(x0[..., 3].abs().max()) == 0.0
    xbgr = ops.preprocess_u8(img.cuda(), bgr=True)
    assert torch.equal(xbgr[..., :3].cpu(), xr.flip(-1))
    w = torch.randn((64, 3, 7, 7), generator=g) / 12
    b = torch.randn((64,), generator=g)
    ref = F.conv2d(xr.permute(0, 3, 1, 2), w, b, stride=2, padding=3)
    raw = ops.conv2d_nhwc(x0, ops.pack_conv_weight(w, 4).cuda(), 7, 7, 64, stride=2, shift=b.cuda())
    assert (nchw(raw) - ref).abs().max().item() < 2e-5
    # instance-norm statistics + apply
    mean, rstd = ops.inorm_stats(raw)
    mref = ref.mean((2, 3))
    vref = ref.var((2, 3), unbiased=False)
    assert (mean.cpu() - mref).abs().max().item() < 1e-5
    assert (rstd.cpu() - 1 / torch.sqrt(vref + 1e-5)).abs().max().item() < 1e-4
    y = ops.inorm_apply(raw, mean, rstd, relu=True)
    yref = torch.relu(F.instance_norm(ref))
    assert (nchw(y) - yref).abs().max().item() < 2e-5
    # norm + relu fused into the next conv's operand load == conv of the materialised tensor
    w2 = torch.randn((64, 64, 3, 3), generator=g) / 24
    ref2 = F.conv2d(yref, w2, None, padding=1)
    out2 = ops.conv2d_nhwc(raw, ops.pack_conv_weight(w2).cuda(), 3, 3, 64, nmean=mean, nrstd=rstd)
    assert (nchw(out2) - ref2).abs().max().item() < 5e-5
    # ... in every tile variant that has a fused-norm instantiation (the executor picks them by problem size)
    w3 = torch.randn((96, 64, 3, 3), generator=g) / 24
    ref3 = F.conv2d(yref, w3, None, padding=1)
    for tile in (16128096, 16128128, 16128064, 32128032, 16064064, 2032064064):
        out3 = ops.conv2d_nhwc(raw, ops.pack_conv_weight(w3).cuda(), 3, 3, 96, nmean=mean, nrstd=rstd, tile=tile)
        assert (nchw(out3) - ref3).abs().max().item() < 5e-5, tile
    # residual merge with a normalised shortcut
    r = torch.randn(ref.shape, generator=g)
    rm, rs = ops.inorm_stats(nhwc(r))
    z = ops.inorm_apply(raw, mean, rstd, res=nhwc(r), res_mean=rm, res_rstd=rs)
    zref = torch.relu(F.instance_norm(r) + yref)
    assert (nchw(z) - zref).abs().max().item() < 2e-5


# --------------------------------------------------------------------------------------
# correlation
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 20, 16), (1, 24, 32), (1, 17, 19), (2, 32, 64), (1, 96, 64)])   # the last two: every level in whole blocks -> levels 2, 3 by the register kernel
def test_corr_volume_pyramid_and_lookup(cuda, shape):
    ops = _ops()
    B, h, w = shape
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn((B, 256, h, w), generator=g)
    f2 = torch.randn((B, 256, h, w), generator=g)
    ref_pyr = RO.corr_pyramid(f1, f2)
    pyr = ops.corr_volume(nhwc(f1), nhwc(f2))
    for l in range(4):
        hl, wl = h >> l, w >> l
        assert tuple(pyr[l].shape) == (B * h * w, ops.corr_slice_floats(hl, wl))
        got = ops.corr_unblock(pyr[l], hl, wl).cpu()                     # blocked 4x8 layout -> [M, hl, wl]
        err = (got - ref_pyr[l][:, 0]).abs().max().item()
        # padding elements of the blocked slices are zero (the lookup relies on it)
        assert abs(float(pyr[l].double().abs().sum()) - float(got.double().abs().sum())) <= 1e-6 * max(1.0, float(got.double().abs().sum()))
        assert err < 3e-5, (l, err)
    # lookups: integer, fractional and far out-of-range coordinates
    coords = RO.coords_grid(B, h, w)
    for amp in (0.0, 3.3, 40.0):
        c = coords + (torch.rand((B, 2, h, w), generator=g) - 0.5) * 2 * amp
        ref = RO.corr_lookup(ref_pyr, c)
        out = ops.corr_lookup(pyr, nhwc(c), B, h, w)
        err = (nchw(out) - ref).abs().max().item()
        assert err < 1e-4, (amp, err)


@pytest.mark.parametrize("shape", [(3, 8, 16), (2, 16, 32), (5, 24, 16), (2, 32, 48), (3, 96, 64)])
@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3", "fp32"])
@pytest.mark.parametrize("shared", [False, True])
def test_corr_volume_split_against_float64_and_the_fp32_volume(cuda, shape, precision, shared):
    """`ofx_corr_volume_split` (csrc/corr_split.hip): the CorrBlock pyramid (RAFT/core/corr.py:13-27,52-60) with the volume GEMM on the
    bf16 matrix cores from operands pre-split into bf16 planes -- the A-stationary kernel, LDS-DMA column stream, quad-ordered columns,
    staged whole-line stores.  Every level against a float64 product of the same feature maps: the three-plane form must be as close
    to it as the exact-fp32 GEMM is (it IS an fp32-accurate product), the two-plane form inside 2^-15 of the row scale.  Shapes:
    N = 128 (one row group, two of its four waves past the map), N = 512, a row group that is not whole (N = 384), 1536, and the bench
    geometry; per-pair key frames and one shared key frame (zero batch stride); feature maps with a long-tailed channel scale so the
    low planes carry weight.  Padding-free levels: the blocked slices hold nothing but the values.
    precision 'fp32': the same kernel on v_mfma_f32_32x32x2_f32 -- what the RAFT executor runs by default since round 6 -- must be the
    generic batched GEMM of `ofx_corr_volume` BIT FOR BIT on every level (same k assignment per lane half, same summation order, the
    2^-4 scale folded into an operand, the same (a + b) + (c + d) pooling)."""
    ops = _ops()
    B, h, w = shape
    g = torch.Generator().manual_seed(11)
    chs = torch.exp(torch.randn((1, 256, 1, 1), generator=g))
    f1 = torch.randn((B, 256, h, w), generator=g) * chs
    f2 = torch.randn((1 if shared else B, 256, h, w), generator=g) * chs
    f2e = f2.expand(B, 256, h, w).contiguous()
    ref = RO.corr_pyramid(f1.double(), f2e.double())
    p32 = ops.corr_volume(nhwc(f1), nhwc(f2e))
    pyr = ops.corr_volume_split(nhwc(f1), nhwc(f2), 4, precision)
    scale = ref[0].abs().max().item()
    for l in range(4):
        hl, wl = h >> l, w >> l
        assert tuple(pyr[l].shape) == (B * h * w, ops.corr_slice_floats(hl, wl))
        got = ops.corr_unblock(pyr[l], hl, wl).cpu().double()
        g32 = ops.corr_unblock(p32[l], hl, wl).cpu().double()
        err = (got - ref[l][:, 0]).abs().max().item()
        e32 = (g32 - ref[l][:, 0]).abs().max().item()
        if precision == "fp32":
            assert torch.equal(pyr[l], p32[l]), (l, err, e32)
        elif precision == "bf16x6":
            assert err <= 2.0 * e32 + 1e-7 * scale, (l, err, e32)
        else:
            assert err <= 3.1e-5 * scale, (l, err, scale)
            assert l > 0 or err > e32                                       # two planes: not the fp32 GEMM by accident
    # the lookup reads the split pyramid like any other
    coords = RO.coords_grid(B, h, w) + (torch.rand((B, 2, h, w), generator=g) - 0.5) * 6.6
    out = ops.corr_lookup(pyr, nhwc(coords), B, h, w)
    refl = RO.corr_lookup([r.float() for r in ref], coords)
    assert (nchw(out) - refl).abs().max().item() < (2e-4 if precision == "bf16x6" else 1e-3) * max(1.0, scale / 16)


def test_corr_volume_split_preconditions(cuda):
    """Shapes the split form does not take are refused with OFX_EINVAL (the caller falls back to `ofx_corr_volume`): level 1 must be
    whole blocks (h % 8, w % 16), D = 256."""
    ops = _ops()
    for shape in ((1, 17, 16, 256), (1, 8, 24, 256), (1, 8, 16, 128)):
        f = torch.zeros(shape, device="cuda")
        with pytest.raises(RuntimeError):
            ops.corr_volume_split(f, f, 4, "bf16x6")
    f = torch.zeros((1, 8, 16, 256), device="cuda")
    with pytest.raises(ValueError):
        ops.corr_volume_split(f, f, 4, "fp16")


@pytest.mark.parametrize("shape", [(4, 16, 24), (5, 12, 9), (4, 17, 10), (6, 96, 64), (2, 17, 10)])
@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_upsample_flow_with_the_warp_inside_is_the_two_kernels_bit_for_bit(cuda, shape, sign):
    """`ofx_upsample_flow_warp` samples the shared key frame while the flow of the four fine pixels of a lane is still in
    registers.  It must be the composition it replaces, byte for byte: `ofx_upsample_flow` (raft.py:72-83) then the bilinear
    `ofx_warp_u8` of one shared frame (pdcnet_of.py:34-42 / ofgen_keyframe_inpaint.py:92-98) -- whose own parity against the
    oracles is pinned elsewhere in this file -- flows that leave the frame included."""
    ops = _ops()
    B, h, w = shape
    g = torch.Generator().manual_seed(41)
    coords = RO.coords_grid(B, h, w) + (torch.rand((B, 2, h, w), generator=g) - 0.5) * 9.0
    coords[0, :, 0, 0] += 500.0                                            # far outside: zeros padding
    mask = torch.randn((B, 576, h, w), generator=g) * 2.0
    frame = torch.randint(0, 256, (8 * h, 8 * w, 3), generator=g, dtype=torch.uint8).cuda()
    c, m = nhwc(coords), nhwc(mask)
    flow_ref = ops.upsample_flow(c, m)
    warped_ref = ops.warp(frame, flow_ref, mode="bilinear", sign=sign)
    flow, warped = ops.upsample_flow_warp(c, m, frame, sign=sign)
    assert torch.equal(flow, flow_ref)
    if B >= 4:      # `ofx_warp_u8` hands batches of >= 4 frames to the shared-key-frame kernel whose sampling function the fused kernel calls
        assert torch.equal(warped, warped_ref)
    else:           # smaller batches take the generic bilinear kernel (weights form, contraction off): 1 LSB apart on rounding ties only
        d = (warped.int() - warped_ref.int()).abs()
        assert d.max().item() <= 1 and (d > 0).float().mean().item() < 1e-3
    none, warped2 = ops.upsample_flow_warp(c, m, frame, sign=sign, want_flow=False)
    assert none is None and torch.equal(warped2, warped)
    # against the warp oracle directly (bilinear, zeros outside): u8 rounding of an fp32 blend
    from oracle import warp_oracle as WO2
    ref0 = WO2.warp_frame(frame.cpu().numpy(), flow_ref[0].cpu().numpy(), mode="bilinear", convention="pdcnet" if sign > 0 else "raft")
    d0 = np.abs(warped[0].cpu().numpy().astype(np.int32) - ref0.astype(np.int32))
    assert d0.max() <= 1 and (d0 > 0).mean() < 2e-3
    # and the flow against the oracle (the upsample arithmetic moved into a shared inline function this round)
    ref = RO.upsample_flow(coords - RO.coords_grid(B, h, w), mask)
    assert (nchw(flow) - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


def test_local_corr_backward_matches_oracle(cuda):
    """alt_cuda_corr.backward: gradients w.r.t. both feature maps against the oracle's restatement (itself equal to
    autograd of the pinned forward); C below, equal to and above one lane-quad sweep (64, 256, 320)."""
    from sd_animation_optical_flow_amd import alt_cuda_corr
    g = torch.Generator().manual_seed(16)
    for (B, h, w, h2, w2, C, N, r) in ((2, 9, 11, 5, 6, 64, 2, 3), (1, 6, 5, 6, 5, 256, 1, 4), (1, 4, 6, 3, 4, 320, 2, 1)):
        f1 = torch.randn((B, h, w, C), generator=g)
        f2 = torch.randn((B, h2, w2, C), generator=g)
        base = torch.stack(torch.meshgrid(torch.arange(w).float(), torch.arange(h).float(), indexing="xy"), -1)
        coords = (base[None, None] * (w2 / w) + (torch.rand((B, N, h, w, 2), generator=g) - 0.5) * 7).contiguous()
        gout = torch.randn((B, N, (2 * r + 1) ** 2, h, w), generator=g)
        r1, r2, _ = RO.local_corr_backward(f1, f2, coords, gout, r)
        o1, o2, oc = alt_cuda_corr.backward(f1.cuda(), f2.cuda(), coords.cuda(), gout.cuda(), r)
        s1, s2 = max(1.0, r1.abs().max().item()), max(1.0, r2.abs().max().item())
        assert (o1.cpu() - r1).abs().max().item() < 2e-5 * s1, (C, (o1.cpu() - r1).abs().max().item())
        assert (o2.cpu() - r2).abs().max().item() < 2e-5 * s2, (C, (o2.cpu() - r2).abs().max().item())
        assert tuple(oc.shape) == tuple(coords.shape) and float(oc.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        alt_cuda_corr.backward(f1.cuda(), f2.cuda(), coords.cuda(), gout.cuda()[:, :, :1], r)      # wrong corr_grad shape


def test_local_corr_matches_alt_cuda_corr_semantics(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    B, h, w, C = 2, 12, 10, 64
    f1 = torch.randn((B, h, w, C), generator=g)
    f2 = torch.randn((B, h // 2, w // 2, C), generator=g)
    coords = torch.stack(torch.meshgrid(torch.arange(w).float(), torch.arange(h).float(), indexing="xy"), -1)
    coords = (coords[None, None] / 2 + (torch.rand((B, 1, h, w, 2), generator=g) - 0.5) * 9).contiguous()
    ref = RO.local_corr_level(f1, f2, coords, 4)
    out = ops.local_corr(f1.cuda(), f2.cuda(), coords.cuda(), 4)
    assert tuple(out.shape) == (B, 1, 81, h, w)
    assert (out.cpu() - ref).abs().max().item() < 1e-4
    # property: the volume-free path equals the CorrBlock path
    fa = torch.randn((1, 256, 16, 16), generator=g)
    fb = torch.randn((1, 256, 16, 16), generator=g)
    c = RO.coords_grid(1, 16, 16) + (torch.rand((1, 2, 16, 16), generator=g) - 0.5) * 6
    want = RO.corr_lookup(RO.corr_pyramid(fa, fb), c)[:, :81]
    got = ops.local_corr(nhwc(fa), nhwc(fb), nhwc(c)[:, None].contiguous(), 4)[:, 0] / 16.0
    assert (got.cpu() - want).abs().max().item() < 1e-4
    p = ops.avgpool2_nhwc(nhwc(fb))
    assert (nchw(p) - F.avg_pool2d(fb, 2, 2)).abs().max().item() < 1e-6


def test_local_corr_preconditions_raise(cuda):
    ops = _ops()
    f = torch.zeros((1, 8, 8, 32))
    c = torch.zeros((1, 1, 8, 8, 2))
    with pytest.raises(RuntimeError, match="CUDA"):
        ops.local_corr(f, f.cuda(), c.cuda(), 4)            # correlation.cpp:19
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.local_corr(f.cuda().transpose(1, 2), f.cuda(), c.cuda(), 4)   # correlation.cpp:20


def test_upsample_flow(cuda):
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    B, h, w = 2, 9, 11
    flow = torch.randn((B, 2, h, w), generator=g) * 3
    mask = torch.randn((B, 576, h, w), generator=g) * 2
    ref = RO.upsample_flow(flow, mask)
    coords1 = RO.coords_grid(B, h, w) + flow
    out = ops.upsample_flow(nhwc(coords1), nhwc(mask))
    # coords1 - grid re-derives flow with one extra rounding: tolerance 8 * 2^-18
    assert (nchw(out) - ref).abs().max().item() < 1e-4


# --------------------------------------------------------------------------------------
# warp
# --------------------------------------------------------------------------------------
def _warp_inputs(seed, H=45, W=37, C=3, amp=9.0):
    rng = np.random.default_rng(seed)
    frame = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
    flow = (rng.standard_normal((H, W, 2)) * amp).astype(np.float32)
    flow[0, 0] = (1000.0, -1000.0)          # far outside
    flow[1, 1] = (0.0, 0.0)
    flow[2, 2] = (0.5, 0.5)                 # exact tie for the bilinear uint8 rounding
    flow[3, 3] = (-3.0, -3.0)               # samples the border
    return frame, flow


@pytest.mark.parametrize("sign,conv", [(1.0, "pdcnet"), (-1.0, "raft")])
def test_warp_cv2_cubic_u8_bit_exact(cuda, sign, conv):
    ops = _ops()
    frame, flow = _warp_inputs(11)
    ref = WO.warp_frame(frame, flow, mode="cv2_cubic", convention=conv)
    out = ops.warp(torch.from_numpy(frame).cuda(), torch.from_numpy(flow).cuda(), mode="cv2_cubic", sign=sign)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_warp_cv2_cubic_f32_bit_exact(cuda):
    ops = _ops()
    frame, flow = _warp_inputs(12, C=1)
    f32 = (frame.astype(np.float32) * 0.37 - 20).astype(np.float32)
    ref = WO.warp_frame(f32, flow, mode="cv2_cubic")
    out = ops.warp(torch.from_numpy(f32).cuda(), torch.from_numpy(flow).cuda(), mode="cv2_cubic")
    assert not _diff(out.cpu().numpy(), ref), _diff(out.cpu().numpy(), ref)


@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
def test_warp_float_modes_against_oracle_and_grid_sample(cuda, mode):
    ops = _ops()
    frame, flow = _warp_inputs(13, C=4)
    f32 = frame.astype(np.float32)
    ref = WO.warp_frame(f32, flow, mode=mode)
    out = ops.warp(torch.from_numpy(f32).cuda(), torch.from_numpy(flow).cuda(), mode=mode).cpu().numpy()
    assert np.abs(out - ref).max() < 1e-3                 # values up to 255: ~4e-6 relative
    # independent pin: torch grid_sample with align_corners=True on the same sample positions
    H, W = flow.shape[:2]
    mx, my = WO._maps(flow)
    grid = torch.from_numpy(np.stack([2 * mx / (W - 1) - 1, 2 * my / (H - 1) - 1], -1))[None]
    gs = F.grid_sample(torch.from_numpy(f32).permute(2, 0, 1)[None], grid, mode=mode, padding_mode="zeros", align_corners=True)
    gs = gs[0].permute(1, 2, 0).numpy()
    sane = (np.abs(flow) < 100).all(-1)
    assert np.abs(out - gs)[sane].max() < 5e-3
    # uint8: round-half-even of the float result, at most 1 LSB from the oracle on rounding ties
    ref8 = WO.warp_frame(frame, flow, mode=mode)
    out8 = ops.warp(torch.from_numpy(frame).cuda(), torch.from_numpy(flow).cuda(), mode=mode).cpu().numpy()
    d = np.abs(out8.astype(int) - ref8.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.parametrize("H,W", [(45, 37), (32, 64), (9, 4), (50, 130)])
def test_warp_u8_rgb_fast_path_matches_oracle(cuda, H, W):
    """uint8 x 3 channels takes the 4-pixels-per-thread kernel: borders, ragged widths, far flows."""
    ops = _ops()
    frame, flow = _warp_inputs(16, H=H, W=W, C=3, amp=5.0)
    flow[-1, -1] = (0.25, 0.25)            # bottom-right pixel: taps straddle the end of the frame buffer
    flow[-1, -2] = (0.75, 0.5)
    flow[0, 1] = (-1.5, -0.5)              # top-left border
    for mode in ("bilinear", "cv2_cubic"):
        ref = WO.warp_frame(frame, flow, mode=mode)
        out = ops.warp(torch.from_numpy(frame).cuda(), torch.from_numpy(flow).cuda(), mode=mode).cpu().numpy()
        if mode == "cv2_cubic":
            assert not _diff(out, ref), _diff(out, ref)
        else:
            d = np.abs(out.astype(int) - ref.astype(int))
            assert d.max() <= 1 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())
    # the generic (4-channel) kernel and the fast path agree on the shared channels
    f4 = np.concatenate([frame, frame[:, :, :1]], -1)
    out4 = ops.warp(torch.from_numpy(f4).cuda(), torch.from_numpy(flow).cuda(), mode="bilinear").cpu().numpy()
    out3 = ops.warp(torch.from_numpy(frame).cuda(), torch.from_numpy(flow).cuda(), mode="bilinear").cpu().numpy()
    assert np.array_equal(out4[:, :, :3], out3)


def test_warp_batched_shared_keyframe(cuda):
    ops = _ops()
    frame, flow = _warp_inputs(14)
    flows = np.stack([flow, -flow, flow * 0.25]).astype(np.float32)
    out = ops.warp(torch.from_numpy(frame).cuda(), torch.from_numpy(flows).cuda(), mode="cv2_cubic").cpu().numpy()
    for i in range(3):
        assert np.array_equal(out[i], WO.warp_frame(frame, flows[i], mode="cv2_cubic"))


def test_resize_cubic_and_latent_warp(cuda):
    ops = _ops()
    rng = np.random.default_rng(15)
    lat = rng.standard_normal((1, 4, 12, 8)).astype(np.float32)
    flow = (rng.standard_normal((96, 64, 2)) * 4).astype(np.float32)
    ref = WO.warp_frame_latent(lat, flow, mode="cv2_cubic")
    x = torch.from_numpy(lat).permute(0, 2, 3, 1).contiguous().cuda()
    up = ops.resize_cubic(x, 96, 64)
    assert np.abs(up[0].cpu().numpy() - WO.resize_cubic(np.transpose(lat[0], (1, 2, 0)), 96, 64)).max() < 1e-5
    wp = ops.warp(up, torch.from_numpy(flow)[None].cuda(), mode="cv2_cubic")
    dn = ops.resize_cubic(wp, 12, 8)
    assert np.abs(dn.permute(0, 3, 1, 2).cpu().numpy() - ref).max() < 1e-4


# --------------------------------------------------------------------------------------
# masks (bit-exact)
# --------------------------------------------------------------------------------------
def _conf(seed, H=70, W=53):
    rng = np.random.default_rng(seed)
    c = rng.random((H, W)).astype(np.float32)
    c[::7, ::5] = np.float32(0.95)      # planted exact-threshold values: '<' vs '>' conventions differ here
    c[3::11, 2::9] = np.float32(0.9)
    return c


@pytest.mark.parametrize("thres,ksize", [(0.95, 7), (0.8, 7), (0.9, 15), (0.5, 1), (0.05, 3)])
def test_generate_mask_bit_exact(cuda, thres, ksize):
    ops = _ops()
    conf = _conf(21)
    logc = np.log(conf + 1e-3).astype(np.float32)
    ref_mask, ref_lc = MO.generate_mask(conf, logc, thres, ksize)
    lc = torch.from_numpy(logc.copy())[None].cuda()
    out = ops.generate_mask(torch.from_numpy(conf)[None].cuda(), lc, thres, ksize)
    assert np.array_equal(out[0].cpu().numpy(), ref_mask)
    assert np.array_equal(lc[0].cpu().numpy(), ref_lc)        # in-place reset, like the reference
    # keyframe-path convention: inpaint where NOT(conf > thres)
    ref2 = MO.dilate(np.where(conf > np.float32(thres), 0, 255).astype(np.uint8), MO.ellipse_kernel(ksize))
    out2 = ops.generate_mask(torch.from_numpy(conf)[None].cuda(), None, thres, ksize, cmp_gt=True)
    assert np.array_equal(out2[0].cpu().numpy(), ref2)


def test_generate_mask_edge_shapes(cuda):
    ops = _ops()
    for H, W in ((1, 1), (1, 200), (130, 3), (16, 64), (17, 65)):
        conf = _conf(22, H, W)
        ref, _ = MO.generate_mask(conf, conf.copy(), 0.95, 7)
        out = ops.generate_mask(torch.from_numpy(conf)[None].cuda(), None, 0.95, 7)
        assert np.array_equal(out[0].cpu().numpy(), ref), (H, W)


def test_dilate_expand_merge_mix_travel(cuda):
    ops = _ops()
    rng = np.random.default_rng(23)
    H, W = 66, 50
    m = (rng.random((H, W)) > 0.97).astype(np.uint8) * rng.integers(1, 256, (H, W)).astype(np.uint8)
    for k in (3, 7, 15):
        assert np.array_equal(ops.dilate(torch.from_numpy(m)[None].cuda(), k)[0].cpu().numpy(), MO.dilate(m, MO.ellipse_kernel(k)))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img[10:30, 10:30] = 200             # flat region: no edges
    mask = (rng.random((H, W)) > 0.9).astype(np.uint8) * 255
    ref = MO.expand_mask(mask, img)
    out = ops.expand_mask(torch.from_numpy(mask)[None].cuda(), torch.from_numpy(img)[None].cuda())
    assert np.array_equal(out[0].cpu().numpy(), ref)
    # merge / mix
    a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mk = rng.choice(np.array([0, 127, 128, 254, 255], dtype=np.uint8), (H, W))
    t = lambda x: torch.from_numpy(x)[None].cuda()
    assert np.array_equal(ops.merge_images(t(a), t(b), t(mk))[0].cpu().numpy(), MO.merge_images(a, b, mk))
    for ppw in (1.0, 0.3, 0.5):
        d = _diff(ops.mix_frames(t(a), t(b), t(mk), ppw)[0].cpu().numpy(), MO.mix_propagated_ai_frame(a, b, mk, ppw))
        assert not d, (ppw, d)
    # of_calc distance map
    flow = (rng.standard_normal((H, W, 2)) * 5).astype(np.float32)
    conf = _conf(24, H, W)
    ref_v = MO.travel_distance(flow, conf)
    out_v = ops.travel_distance(t(flow), t(conf))[0].cpu().numpy()
    assert np.array_equal(out_v, ref_v)
    # stateful travel-distance mask (confidence_to_mask)
    travel = (rng.random((H, W)) * 30).astype(np.float32)
    ref_mask, ref_travel = MO.confidence_to_mask(conf, flow, ref_v, travel, 25.0, "cv2_cubic")
    raw, new_travel = ops.travel_mask(t(conf), t(flow), t(ref_v), t(travel), 25.0, "cv2_cubic")
    assert np.array_equal(new_travel[0].cpu().numpy(), ref_travel)
    assert np.array_equal(ops.dilate(raw, 15)[0].cpu().numpy(), ref_mask)
    # keyframe score reduction
    fm = rng.random((5, H, W, 3)).astype(np.float32)
    s = ops.conf_sum(torch.from_numpy(fm).cuda(), 2).cpu().numpy()
    assert np.allclose(s, fm[..., 2].astype(np.float64).sum((1, 2)), rtol=1e-12)


def test_warp_and_mask_full_size_properties(cuda):
    """BASELINE size (512x768): size-independent properties instead of a slow CPU comparison."""
    ops = _ops()
    H, W = 768, 512
    g = torch.Generator(device="cpu").manual_seed(31)
    frame = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).cuda()
    zero = torch.zeros((2, H, W, 2), device="cuda")
    conf = torch.rand((2, H, W), generator=g).cuda()
    for mode in ("bilinear", "bicubic", "cv2_cubic"):
        warped, mask = ops.warp_and_mask(frame, zero, conf, warp_mode=mode, thres=0.95, ksize=7)
        assert torch.equal(warped[0], frame) and torch.equal(warped[1], frame)      # identity flow
    # integer translation == shifted copy with zero fill
    shift = torch.zeros((1, H, W, 2), device="cuda")
    shift[..., 0] = 5.0
    shift[..., 1] = -3.0
    w2 = ops.warp(frame, shift, mode="cv2_cubic")[0]
    assert torch.equal(w2[3:, :W - 5], frame[:H - 3, 5:]) and int(w2[:3].max()) == 0 and int(w2[:, W - 5:].max()) == 0
    # dilation is extensive, idempotent w.r.t. thresholds: mask(thres a) <= mask(thres b) for a <= b
    m1 = ops.generate_mask(conf, None, 0.5, 7)
    m2 = ops.generate_mask(conf, None, 0.6, 7)
    raw = (conf < 0.5).to(torch.uint8) * 255
    assert bool((m1 >= raw).all()) and bool((m2 >= m1).all())
    assert set(torch.unique(m1).tolist()) <= {0, 255}


def test_corr_volume_beyond_2gib_is_split_along_m(cuda):
    """A 1280x1280 frame: the level-0 volume of ONE pair is 25600^2 floats = 2.6 GB, past the 2 GiB reach of
    the GEMM epilogue's 32-bit offsets -> the launcher splits it along M.  Rows either side of the split
    boundary (20736) against an f64 matmul; level 1 against the pooled reference."""
    ops = _ops()
    h = w = 160
    g = torch.Generator().manual_seed(15)
    f1 = torch.randn((1, h, w, 256), generator=g).cuda()
    f2 = torch.randn((1, h, w, 256), generator=g).cuda()
    pyr = ops.corr_volume(f1, f2)
    rows = torch.tensor([0, 1, 12345, 20735, 20736, 20737, 25599], device="cuda")
    ref = (f1.view(-1, 256)[rows].double() @ f2.view(-1, 256).double().T / 16.0).float()
    got = ops.corr_unblock(pyr[0], h, w).view(h * w, h * w)[rows]
    assert (got - ref).abs().max().item() < 3e-5
    ref1 = torch.nn.functional.avg_pool2d(ref.view(len(rows), 1, h, w), 2)[:, 0]
    assert (ops.corr_unblock(pyr[1], h // 2, w // 2)[rows] - ref1).abs().max().item() < 3e-5
    assert torch.isfinite(pyr[3]).all()


def test_top_level_alt_cuda_corr_serves_an_alternate_corr_block_caller(cuda):
    """`import alt_cuda_corr` (the module name RAFT/core/corr.py:5-9 imports) must resolve from the repository root
    and serve a caller shaped like `AlternateCorrBlock.__call__` (corr.py:74-91): channels-last contiguous feature
    maps, coords / 2^i as [B,1,H,W,2], `corr, = alt_cuda_corr.forward(f1, f2, coords, r)`, stack, reshape, / sqrt(dim).
    The result must equal the all-pairs CorrBlock lookup (the equivalence that pins the CUDA original)."""
    import alt_cuda_corr                                   # top-level, exactly the reference's import
    import torch.nn.functional as F
    from oracle import raft_oracle as RO
    g = torch.Generator().manual_seed(5)
    B, D, h, w, r, levels = 2, 64, 24, 32, 4, 4
    fmap1 = torch.randn((B, D, h, w), generator=g)
    fmap2 = torch.randn((B, D, h, w), generator=g)
    coords = RO.coords_grid(B, h, w) + (torch.rand((B, 2, h, w), generator=g) - 0.5) * 9.0

    def caller(fmap1, fmap2, coords):                      # the call pattern of corr.py:63-91, on the device
        pyramid = [fmap2]
        for _ in range(levels):
            pyramid.append(F.avg_pool2d(pyramid[-1], 2, stride=2))
        c = coords.permute(0, 2, 3, 1)
        out = []
        for i in range(levels):
            f1 = fmap1.permute(0, 2, 3, 1).contiguous()
            f2 = pyramid[i].permute(0, 2, 3, 1).contiguous()
            ci = (c / 2 ** i).reshape(B, 1, h, w, 2).contiguous()
            corr, = alt_cuda_corr.forward(f1, f2, ci, r)
            out.append(corr.squeeze(1))
        corr = torch.stack(out, dim=1).reshape(B, -1, h, w)
        return corr / torch.sqrt(torch.tensor(float(D)))

    got = caller(fmap1.cuda(), fmap2.cuda(), coords.cuda()).cpu()
    ref = RO.corr_lookup(RO.corr_pyramid(fmap1, fmap2), coords)
    assert tuple(got.shape) == (B, levels * 81, h, w)
    assert (got - ref).abs().max().item() < 2e-4
    with pytest.raises(RuntimeError):
        alt_cuda_corr.forward(fmap1, fmap2, coords.reshape(B, 1, h, w, 2), r)          # CPU tensors: TORCH_CHECK


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices")
def test_cv2_tables_follow_the_device(cuda):
    """The OpenCV interpolation tables live in device memory: one set per device, picked by the CURRENT device
    (a process that moves its algorithm with `.to(other)` must not hand device 0's table pointers to device 1)."""
    from sd_animation_optical_flow_amd import ops
    g = torch.Generator().manual_seed(3)
    fr = torch.randint(0, 256, (40, 56, 3), dtype=torch.uint8, generator=g)
    fl = (torch.rand((1, 40, 56, 2), generator=g) - 0.5) * 7
    outs = []
    for d in (0, 1, 0):
        with torch.cuda.device(d):
            outs.append(ops.warp(fr.to(f"cuda:{d}"), fl.to(f"cuda:{d}"), mode="cv2_cubic").cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("H,W,B", [(768, 512, 3), (70, 260, 2), (33, 4, 1), (64, 1028, 1), (200, 48, 17), (31, 2052, 2)])
def test_generate_mask_band_kernel_shapes(cuda, H, W, B):
    """The full-row band kernel (W % 4 == 0): widths that are / are not multiples of 16, 32 and 256, bands that end
    mid-way, a batch large enough for the XCD-grouped workgroup order (B >= 16), every structuring element size class,
    both comparison conventions and the in-place log-confidence reset -- all bit-exact against the oracle."""
    ops = _ops()
    rng = np.random.default_rng(H * 7 + W)
    conf = rng.random((B, H, W)).astype(np.float32)
    conf[:, ::5, ::3] = np.float32(0.9)                       # exactly-at-threshold values
    conf[:, 0, :] = 0.0                                       # low first row / last column: border handling
    conf[:, :, -1] = 0.0
    for ksize, thres in ((7, 0.9), (1, 0.9), (3, 0.5), (15, 0.97), (31, 0.995)):
        logc = np.log(conf + 1e-3).astype(np.float32)
        lc = torch.from_numpy(logc.copy()).cuda()
        out = ops.generate_mask(torch.from_numpy(conf).cuda(), lc, thres, ksize)
        out2 = ops.generate_mask(torch.from_numpy(conf).cuda(), None, thres, ksize, cmp_gt=True)
        for b in sorted({0, B - 1}):
            ref_mask, ref_lc = MO.generate_mask(conf[b], logc[b].copy(), thres, ksize)
            assert np.array_equal(out[b].cpu().numpy(), ref_mask), (ksize, b)
            assert np.array_equal(lc[b].cpu().numpy(), ref_lc), (ksize, b)
            ref2 = MO.dilate(np.where(conf[b] > np.float32(thres), 0, 255).astype(np.uint8), MO.ellipse_kernel(ksize))
            assert np.array_equal(out2[b].cpu().numpy(), ref2), (ksize, b)


@pytest.mark.parametrize("H,W,B", [(45, 36, 5), (64, 128, 4), (9, 4, 6), (768, 512, 4)])
def test_warp_bilinear_shared_keyframe_fast_path(cuda, H, W, B):
    """B >= 4 flows against ONE uint8 RGB key frame take the zero-bordered RGBX kernel (warp_fast.hip): borders, far
    and non-finite flows, exact half-pixel ties, both flow conventions -- against the oracle (<= 1 LSB on rounding ties
    only) and against the byte-triplet kernel that serves B < 4."""
    ops = _ops()
    rng = np.random.default_rng(H + W)
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    flows = (rng.standard_normal((B, H, W, 2)) * 4).astype(np.float32)
    flows[0, 0, 1] = (-1.5, -0.5)                     # top-left border
    flows[0, -1, -1] = (0.25, 0.25)                   # bottom-right: taps straddle the frame's last pixel
    flows[1, H // 2, :] = (0.5, 0.5)                  # exact ties
    flows[1, 0, :, 0] = -3.0                          # row 0 pointing left out of the image
    flows[2] = flows[2] * 40                          # far flows: mostly outside
    flows[3, 1, 1] = (1e30, -1e30)
    flows[3, 2, 2] = (np.nan, 0.0)
    fr, fl = torch.from_numpy(frame).cuda(), torch.from_numpy(flows).cuda()
    for sign, conv in ((1.0, "pdcnet"), (-1.0, "raft")):
        out = ops.warp(fr, fl, mode="bilinear", sign=sign).cpu().numpy()
        assert out.shape == (B, H, W, 3)
        for b in range(B if H < 100 else 2):
            ref = WO.warp_frame(frame, np.nan_to_num(flows[b], nan=1e9), mode="bilinear", convention=conv)
            d = np.abs(out[b].astype(int) - ref.astype(int))
            if b == 3:
                d[2, 2] = 0                           # NaN flow: any in-range byte is acceptable, no fault is the point
            assert d.max() <= 1 and (d > 0).mean() < 2e-3, (b, d.max(), (d > 0).mean())
            single = ops.warp(fr, fl[b:b + 1].contiguous(), mode="bilinear", sign=sign)[0].cpu().numpy()
            d2 = np.abs(out[b].astype(int) - single.astype(int))
            if b == 3:
                d2[2, 2] = 0
            assert d2.max() <= 1 and (d2 > 0).mean() < 2e-3


def test_corr_lookup_generic_levels_and_radius(cuda):
    """Anything but the reference's (4 levels, radius 4) takes the one-tap-per-lane kernel on the same blocked pyramid:
    2 and 3 levels, radii 1..3, frame sizes that are not multiples of the 4x8 block, against the oracle."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    for (B, h, w, levels, radius) in ((1, 13, 19, 3, 2), (2, 8, 24, 2, 3), (1, 21, 9, 4, 1)):
        f1 = torch.randn((B, 64, h, w), generator=g)
        f2 = torch.randn((B, 64, h, w), generator=g)
        ref_pyr = RO.corr_pyramid(f1, f2, levels)
        pyr = ops.corr_volume(nhwc(f1), nhwc(f2), levels)
        c = RO.coords_grid(B, h, w) + (torch.rand((B, 2, h, w), generator=g) - 0.5) * 9
        ref = RO.corr_lookup(ref_pyr, c, radius)
        out = ops.corr_lookup(pyr, nhwc(c), B, h, w, radius)
        assert tuple(out.shape) == (B, h, w, levels * (2 * radius + 1) ** 2)
        assert (nchw(out) - ref).abs().max().item() < 1e-4, (B, h, w, levels, radius)
