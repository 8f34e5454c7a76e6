"""CPU: the oracle against the committed golden vectors (tests/golden/*.npz, made by make_golden.py from
the REAL reference RAFT and from torch.grid_sample).  Runs everywhere, no GPU, no /root/reference."""
import hashlib
import os

import numpy as np
import torch

from oracle import mask_oracle as MO
from oracle import raft_oracle as RO
from oracle import warp_oracle as WO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_seeded_state_dict_is_the_pinned_one(raft_sd):
    g = _load("raft_ref_128x160.npz")
    h = hashlib.sha256()
    for k in sorted(raft_sd):
        h.update(k.encode())
        h.update(raft_sd[k].numpy().tobytes())
    assert h.hexdigest() == str(g["state_dict_sha256"])
    assert len(raft_sd) == 179          # every key of the reference's RAFT(args).state_dict()


def test_raft_stages_match_reference_vectors(raft_sd):
    g = _load("raft_ref_128x160.npz")
    img1 = torch.from_numpy(g["image1"]).float()
    img2 = torch.from_numpy(g["image2"]).float()
    tr = {}
    lo, up = RO.raft_forward(raft_sd, img1, img2, 20, trace=tr)
    # half-precision fixtures: tolerance = f16 quantisation of values up to ~20
    for key, name in (("fmap1", "fmap1_f16"), ("fmap2", "fmap2_f16"), ("net0", "net_f16"), ("inp", "inp_f16")):
        ref = torch.from_numpy(g[name].astype(np.float32))
        assert (tr[key] - ref).abs().max().item() < 2e-2, key
    st = g["fmap1_stats"]
    assert abs(float(tr["fmap1"].abs().sum()) / st[1] - 1) < 1e-5 and abs(float(tr["fmap2"].abs().sum()) / st[3] - 1) < 1e-5
    assert (tr["pyramid"][3] - torch.from_numpy(g["pyr3"])).abs().max().item() < 1e-4
    for l in range(4):
        p = tr["pyramid"][l]
        assert abs(float(p.pow(2).sum()) / g["pyr_stats"][l][2] - 1) < 1e-5, l
    # final flow, full resolution, exact fixtures
    assert (lo - torch.from_numpy(g["flow_low"])).abs().max().item() < 1e-3
    epe = (up - torch.from_numpy(g["flow_up"])).pow(2).sum(1).sqrt().mean().item()
    assert epe < 1e-4, epe


def test_lookup_update_upsample_match_reference_vectors(raft_sd):
    g = _load("raft_ref_128x160.npz")
    f1 = torch.from_numpy(g["fmap1_f16"].astype(np.float32))
    img1 = torch.from_numpy(g["image1"]).float()
    img2 = torch.from_numpy(g["image2"]).float()
    tr = {}
    RO.raft_forward(raft_sd, img1, img2, 1, trace=tr)
    pyr = tr["pyramid"]
    h, w = f1.shape[-2:]
    c0 = RO.coords_grid(1, h, w)
    jit = torch.from_numpy(g["lookup_jitter"])
    for name, amp in (("lookup_int", 0.0), ("lookup_frac", 5.0), ("lookup_far", 60.0)):
        out = RO.corr_lookup(pyr, c0 + jit * amp)[:, :, ::3, ::3]
        assert (out - torch.from_numpy(g[name])).abs().max().item() < 2e-4, name
        alt = RO.alternate_corr_lookup(tr["fmap1"], tr["fmap2"], c0 + jit * amp)[:, :, ::3, ::3]
        assert (alt - torch.from_numpy(g[name])).abs().max().item() < 2e-4, name + " (alt)"
    # first update step from zero flow
    corr0 = RO.corr_lookup(pyr, c0)
    n1, m1, d1 = RO.update_block(raft_sd, tr["net0"], tr["inp"], corr0, c0 - c0)
    assert (n1 - torch.from_numpy(g["update_net1_f16"].astype(np.float32))).abs().max().item() < 2e-3
    assert (d1 - torch.from_numpy(g["update_delta1"])).abs().max().item() < 1e-4
    assert abs(float(m1.abs().sum()) / g["update_mask1_stats"][1] - 1) < 1e-5
    assert (RO.upsample_flow(d1, m1)[:, :, ::2, ::2] - torch.from_numpy(g["upsample1"])).abs().max().item() < 1e-4


def test_lookup_channel_order_is_x_major():
    """channel k = l*81 + i*9 + j samples at x+(i-4), y+(j-4)."""
    h = w = 16
    vol = torch.zeros((h * w, 1, h, w))
    p = 5 * w + 6                       # source pixel (y=5, x=6)
    vol[p, 0, 7, 9] = 1.0               # its match at (y=7, x=9): dx=+3, dy=+2
    out = RO.corr_lookup([vol], RO.coords_grid(1, h, w), radius=4)
    k = torch.nonzero(out[0, :, 5, 6]).flatten().tolist()
    assert k == [(3 + 4) * 9 + (2 + 4)]


def test_warp_modes_match_grid_sample_vectors():
    g = _load("warp_grid_sample.npz")
    frame, flow = g["frame"], g["flow"]
    sane = (np.abs(flow) < 100).all(-1)
    for mode in ("bilinear", "bicubic"):
        out = WO.warp_frame(frame.astype(np.float32), flow, mode)
        assert np.abs(out - g[mode])[sane].max() < 5e-3, mode
    # raft convention = pdcnet convention with the flow negated
    assert np.array_equal(WO.warp_frame(frame, flow, "cv2_cubic", convention="raft"), WO.warp_frame(frame, -flow, "cv2_cubic"))


def test_cv2_semantics_frozen():
    g = _load("cv2_semantics.npz")
    frame, flow, conf, img = g["frame"], g["flow"], g["conf"], g["img"]
    assert np.array_equal(WO.warp_frame(frame, flow, "cv2_cubic"), g["warp_cv2_u8"])
    assert np.array_equal(WO.warp_frame(frame, flow, "cv2_cubic", convention="raft"), g["warp_cv2_u8_raft"])
    tabf, tabi = WO.cv2_cubic_tables()
    assert np.array_equal(tabi[[0, 1, 33, 528, 1023]], g["tab_i16_rows"])
    assert (tabi.astype(np.int64).sum(1) == 32768).all()            # weights always sum to 2^15
    assert list(tabi[0]) == [0, 0, 0, 0, 0, 32767, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]   # OpenCV's saturate + residual quirk
    assert np.array_equal(MO.ellipse_kernel(7), g["ellipse7"]) and np.array_equal(MO.ellipse_kernel(15), g["ellipse15"])
    assert list(MO.ellipse_kernel(7).sum(1)) == [1, 5, 7, 7, 7, 5, 1]
    m, _ = MO.generate_mask(conf, conf.copy(), 0.95, 7)
    assert np.array_equal(m, g["mask95"])
    assert np.array_equal(MO.laplacian_edges(img), g["edges"])
    assert np.array_equal(MO.expand_mask(m, img), g["expand"])
    assert np.array_equal(MO.travel_distance(flow, conf), g["travel"])


def test_mask_conventions_at_exact_threshold():
    """generate_mask uses strict '<' (equality keeps the pixel); the key-frame path uses NOT(conf > thres)
    (equality inpaints) -- SURVEY §8 a16 vs a18."""
    conf = np.full((9, 9), 1.0, np.float32)
    conf[4, 4] = np.float32(0.95)
    m, lc = MO.generate_mask(conf, np.full((9, 9), -1.0, np.float32), 0.95, 1)
    assert m.max() == 0 and lc.min() == -1.0
    conf[4, 4] = np.float32(0.9499999)
    m, lc = MO.generate_mask(conf, np.full((9, 9), -1.0, np.float32), 0.95, 7)
    assert int((m == 255).sum()) == 33 and lc[4, 4] == 0.0 and (lc == 0).sum() == 1   # 7x7 ellipse: 1+5+7+7+7+5+1 ones


def test_dilate_matches_brute_force():
    rng = np.random.default_rng(0)
    m = (rng.random((23, 31)) > 0.93).astype(np.uint8) * 255
    k = MO.ellipse_kernel(7)
    ref = np.zeros_like(m)
    for y in range(23):
        for x in range(31):
            best = 0
            for dy in range(-3, 4):
                for dx in range(-3, 4):
                    yy, xx = y + dy, x + dx
                    if k[dy + 3, dx + 3] and 0 <= yy < 23 and 0 <= xx < 31:
                        best = max(best, m[yy, xx])
            ref[y, x] = best
    assert np.array_equal(MO.dilate(m, k), ref)


def test_compose_single_reference_reduces_to_threshold_mask():
    """N = 1 (the live configuration, ofgen_keyframe_inpaint.py:1264): mask2 = 255*NOT(conf>thres) | edges."""
    rng = np.random.default_rng(1)
    H, W = 24, 20
    fm = np.zeros((1, 1, H, W, 3), np.float32)
    fm[0, 0, :, :, 2] = rng.random((H, W))
    ai = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    orig = np.full((H, W, 3), 77, np.uint8)           # flat frame: no Laplacian edges
    ret, mask2, order = MO.compose(fm, [ai], orig, 0.5, warp_mode="bilinear")
    assert order == [0] and np.array_equal(ret, ai)   # zero flow: the warp is the identity
    assert np.array_equal(mask2, np.where(fm[0, 0, :, :, 2] > 0.5, 0, 255).astype(np.uint8))


def test_sd_handoff_primitives_match_pillow_vectors():
    """GaussianBlur / composite / default-resample resize: outputs of the real Pillow (make_golden_handoff.py),
    bit for bit; plus a live comparison when Pillow is importable here."""
    from oracle import handoff_oracle as HO
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sd_handoff_pil.npz"))
    for name in "abcd":
        mask, blur = g[f"{name}_mask"], float(g[f"{name}_blur"])
        H, W = mask.shape
        b = HO.gaussian_blur_u8(mask, blur)
        assert np.array_equal(b, g[f"{name}_pil_blur"]), name
        assert np.array_equal(HO.composite(g[f"{name}_reference_rgb"], g[f"{name}_image_rgb"], b), g[f"{name}_pil_composite"]), name
        assert np.array_equal(HO.resize_bicubic_u8(b, H // 8, W // 8), g[f"{name}_pil_latent"]), name
    # the assembled hand-off: shapes, value sets and the identities the consumer relies on
    r = HO.sd_handoff(g["a_image_rgb"][..., ::-1], g["a_reference_rgb"][..., ::-1], g["a_mask"], 4.0)
    H, W = g["a_mask"].shape
    assert r["image"].shape == (3, H, W) and r["latmask"].shape == (4, H // 8, W // 8)
    assert np.array_equal(r["image"], np.moveaxis(g["a_pil_composite"].astype(np.float32) / 127.5 - 1.0, 2, 0))
    assert set(np.unique(r["latmask"])) <= {0.0, 1.0} and set(np.unique(r["cond_mask"])) <= {0.0, 1.0}
    assert np.array_equal(r["cond_mask"], (g["a_pil_blur"] >= 128).astype(np.float32))      # round(v/255), no ties on uint8
    assert np.array_equal(r["cond_image"], r["image"] * (1 - r["cond_mask"])[None])
    try:
        from PIL import Image, ImageFilter
    except ImportError:
        return
    rng = np.random.default_rng(9)
    m = (rng.random((40, 56)) < 0.4).astype(np.uint8) * 255
    for radius in (1, 3, 4, 6.5):
        assert np.array_equal(HO.gaussian_blur_u8(m, radius), np.array(Image.fromarray(m).filter(ImageFilter.GaussianBlur(radius))))


def test_keyframe_oracle_against_independent_implementations():
    """The key-frame oracle is unpinned (no OpenCV to compare with); its building blocks are at least checked against
    independent implementations: scipy's correlate / grey_dilation and a plain flood fill."""
    from scipy import ndimage
    from oracle import keyframe_oracle as KO
    rng = np.random.default_rng(3)
    lum = ndimage.uniform_filter(rng.integers(0, 256, (37, 45)).astype(np.float32), 5).astype(np.uint8)
    lum[10:25, 8:30] = np.minimum(lum[10:25, 8:30].astype(int) + 90, 255).astype(np.uint8)
    dx, dy = KO._sobel16(lum)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    assert np.array_equal(dx, ndimage.correlate(lum.astype(np.int32), kx, mode="nearest"))
    assert np.array_equal(dy, ndimage.correlate(lum.astype(np.int32), kx.T, mode="nearest"))
    e = (rng.random((30, 41)) < 0.05).astype(np.uint8) * 255
    for k in (1, 3, 5, 7):
        assert np.array_equal(KO.dilate_square(e, k), ndimage.grey_dilation(e, size=(k, k), mode="constant", cval=0))
    low, high = 40, 110
    mp = KO.canny_map(lum, low, high)
    keep = mp == 2                                                  # flood fill from the strong pixels through candidates
    stack = list(zip(*np.nonzero(keep)))
    while stack:
        y, x = stack.pop()
        for yy in range(max(0, y - 1), min(mp.shape[0], y + 2)):
            for xx in range(max(0, x - 1), min(mp.shape[1], x + 2)):
                if mp[yy, xx] == 0 and not keep[yy, xx]:
                    keep[yy, xx] = True
                    stack.append((yy, xx))
    assert np.array_equal(KO.canny(lum, low, high) > 0, keep) and 0 < keep.sum() < keep.size // 2
    assert KO.canny_thresholds(np.array([[10, 20], [30, 41]], np.uint8)) == (16, 33)     # median 25.0 -> int(16.67), int(33.33)
    assert KO.TG22 == 13573 and KO.estimated_kernel_size(512, 768) == 7 and KO.gaps(30.0) == (10, 300)


def test_local_corr_backward_is_the_adjoint_of_the_pinned_forward():
    """alt_cuda_corr.backward restated (cu:122-256) == autograd of the restated forward, whose values are pinned to
    the reference's CorrBlock through the golden file (the CUDA source itself cannot be built here)."""
    g = torch.Generator().manual_seed(12)
    B, H1, W1, H2, W2, C, N, r = 2, 6, 7, 5, 6, 8, 2, 2
    f1 = torch.randn((B, H1, W1, C), generator=g, requires_grad=True)
    f2 = torch.randn((B, H2, W2, C), generator=g, requires_grad=True)
    base = torch.stack(torch.meshgrid(torch.arange(W1).float(), torch.arange(H1).float(), indexing="xy"), -1)
    coords = (base[None, None] * 0.8 + (torch.rand((B, N, H1, W1, 2), generator=g) - 0.5) * 6).contiguous()   # some taps leave fmap2
    out = RO.local_corr_level(f1, f2, coords, r)
    gout = torch.randn(out.shape, generator=g)
    a1, a2 = torch.autograd.grad(out, [f1, f2], grad_outputs=gout)
    g1, g2, gc = RO.local_corr_backward(f1.detach(), f2.detach(), coords, gout, r)
    assert (g1 - a1).abs().max().item() < 1e-5 and (g2 - a2).abs().max().item() < 1e-5
    assert gc.shape == coords.shape and float(gc.abs().max()) == 0.0


def test_raft2_as_written_train_mode_batchnorm_matches_reference_vectors(raft_sd):
    """`RAFT_2` never calls `.eval()` (ofgen_keyframe_inpaint.py:47-60): raft_ref_trainbn_128x160.npz holds what the REAL
    reference module returns when driven exactly like that (train-mode BatchNorm in the context encoder, one image per call).
    The oracle's cnet_norm='batch' mode must reproduce it; the eval-mode oracle must not (the two are pixels apart)."""
    g = _load("raft_ref_trainbn_128x160.npz")
    for tag in ("a", "b"):
        f1, f2, ref = g[f"frame1_{tag}"], g[f"frame2_{tag}"], g[f"flow_{tag}"]
        got = RO.raft2_calc(raft_sd, f1, f2, iters=20, cnet_norm="batch")
        assert got.shape == ref.shape
        assert np.sqrt(((got - ref) ** 2).sum(-1)).mean() < 1e-4
    f1, f2, ref = g["frame1_a"], g["frame2_a"], g["flow_a"]
    ev = RO.raft2_calc(raft_sd, f1, f2, iters=20, cnet_norm="eval")
    assert np.sqrt(((ev - ref) ** 2).sum(-1)).mean() > 1.0
    # a batch in 'batch' mode = as many single-image reference calls (every image normalised by itself)
    a = torch.from_numpy(np.stack([f1[:, :, ::-1], g["frame2_a"][:, :, ::-1]]).copy()).permute(0, 3, 1, 2).float()
    b = torch.from_numpy(np.stack([f2[:, :, ::-1], g["frame1_a"][:, :, ::-1]]).copy()).permute(0, 3, 1, 2).float()
    _, up = RO.raft_forward(raft_sd, a, b, iters=3, cnet_norm="batch")
    _, up0 = RO.raft_forward(raft_sd, a[:1], b[:1], iters=3, cnet_norm="batch")
    assert (up[:1] - up0).abs().max().item() < 1e-4
