"""-m gpu: the reference's Python call surface (pdcnet_of.py / ofgen_*.py names) on top of the HIP path,
checked against the oracle's restatement of the same reference functions."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mask_oracle as MO
from oracle import raft_oracle as RO
from oracle import warp_oracle as WO


def _pair(seed, H=96, W=128):
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 16, W + 16), generator=g), 5, 1, 2)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)
    a = base[0, :, 8:8 + H, 8:8 + W].permute(1, 2, 0).contiguous().numpy()
    b = base[0, :, 10:10 + H, 5:5 + W].permute(1, 2, 0).contiguous().numpy()
    return a, b          # "BGR" uint8 frames


@pytest.fixture(scope="module")
def algo(cuda, raft_sd):
    from sd_animation_optical_flow_amd import pdcnet_of
    return pdcnet_of.create_of_algo(raft_sd)


def test_calc_contract_and_flow_orientation(algo, raft_sd):
    f1, f2 = _pair(1)
    flow, conf, logc = algo.calc(f1, f2)
    H, W = f1.shape[:2]
    assert flow.shape == (H, W, 2) and flow.dtype == np.float32
    assert conf.shape == (H, W) and logc.shape == (H, W) and conf.dtype == np.float32
    assert conf.min() >= 0 and conf.max() <= 1 and logc.max() <= 0
    assert np.allclose(np.exp(logc), conf, atol=1e-6)
    # flow lives on frame2's grid and points into frame1 == RAFT(image1=frame2, image2=frame1) on RGB
    ref = RO.raft2_calc(raft_sd, f2, f1, cnet_norm="eval")   # the pdcnet_of surface runs its network in .eval() (pdcnet_of.py:62)
    assert np.sqrt(((flow - ref) ** 2).sum(-1)).mean() < 1e-3
    # forward-backward confidence (extension) against a numpy restatement built from oracle flows
    bw = RO.raft2_calc(raft_sd, f1, f2, cnet_norm="eval")
    samp = WO.warp_frame(bw, ref, mode="bilinear")
    e = ref + samp
    want = np.exp(-(e ** 2).sum(-1) / (2 * 3.0 ** 2))
    assert np.abs(conf - want).max() < 5e-3


def test_calc_batch_contract(algo):
    f1, f2 = _pair(2)
    src = torch.from_numpy(np.stack([f1[:, :, ::-1], f2[:, :, ::-1]]).copy()).cuda()      # RGB tensors on the device
    tgt = torch.from_numpy(np.stack([f2[:, :, ::-1], f1[:, :, ::-1]]).copy()).cuda()
    flow_est, confidence = algo.calc_batch(src, tgt)
    ret = np.zeros((2, 1, *f1.shape[:2], 3), np.float32)
    for i in range(2):                                   # exactly how PDCNetAux consumes it (:598-599)
        ret[i, 0, :, :, 0:2] = flow_est[i]
        ret[i, 0, :, :, 2] = confidence[i]
    flow, conf, _ = algo.calc(f1, f2)
    assert np.abs(ret[0, 0, :, :, 0:2] - flow).max() < 1e-4 and np.abs(ret[0, 0, :, :, 2] - conf).max() < 1e-4
    assert algo.to(torch.device("cuda:0")) is algo


def test_warp_frame_functions(cuda):
    from sd_animation_optical_flow_amd import ofgen, pdcnet_of
    rng = np.random.default_rng(3)
    H, W = 50, 40
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    flow = (rng.standard_normal((H, W, 2)) * 4).astype(np.float32)
    keep = flow.copy()
    assert np.array_equal(pdcnet_of.warp_frame(frame, flow), WO.warp_frame(frame, flow, "cv2_cubic"))
    assert np.array_equal(ofgen.warp_frame(frame, flow), WO.warp_frame(frame, flow, "cv2_cubic", convention="raft"))
    assert np.array_equal(flow, keep)                                     # inputs are not mutated
    dist = rng.random((H, W)).astype(np.float32) * 9                      # 2-D float map (travel distance)
    assert np.array_equal(pdcnet_of.warp_frame(dist, flow), WO.warp_frame(dist, flow, "cv2_cubic"))
    lat = torch.from_numpy(rng.standard_normal((1, 4, 8, 5)).astype(np.float32))
    big = (rng.standard_normal((64, 40, 2)) * 3).astype(np.float32)
    out = pdcnet_of.warp_frame_latent(lat, big)
    assert out.device.type == "cpu" and tuple(out.shape) == (1, 4, 8, 5)
    assert np.abs(out.numpy() - WO.warp_frame_latent(lat.numpy(), big)).max() < 1e-4
    out2 = ofgen.warp_frame_latent(lat, big)
    assert np.abs(out2.numpy() - WO.warp_frame_latent(lat.numpy(), big, convention="raft")).max() < 1e-4


def test_masks_and_merges_numpy_surface(cuda):
    from sd_animation_optical_flow_amd import ofgen
    rng = np.random.default_rng(4)
    H, W = 60, 44
    conf = rng.random((H, W)).astype(np.float32)
    conf[::6, ::5] = np.float32(0.95)
    logc = np.log(conf + 1e-3).astype(np.float32)
    ref_m, ref_lc = MO.generate_mask(conf, logc.copy(), 0.95, 7)
    lc_in = logc.copy()
    m, lc_out = ofgen.generate_mask(conf, lc_in, thres=0.95)
    assert np.array_equal(m, ref_m) and np.array_equal(lc_out, ref_lc)
    assert lc_out is lc_in and np.array_equal(lc_in, ref_lc)            # the reference mutates its argument (:320)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(ofgen.expand_mask(m, img), MO.expand_mask(m, img))
    a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(ofgen.merge_images(a, b, m), MO.merge_images(a, b, m))
    assert ofgen.mix_propagated_ai_frame(a, b, m, 0.0) is a              # early return (:307-308)
    assert np.array_equal(ofgen.mix_propagated_ai_frame(a, b, m, 0.7), MO.mix_propagated_ai_frame(a, b, m, 0.7))
    flow = (rng.standard_normal((H, W, 2)) * 3).astype(np.float32)
    aux = ofgen.create_mask_aux(H, W, 12.0)
    travel = np.zeros((H, W), np.float32)
    for _ in range(3):                                                   # stateful: three consecutive frames
        dist = MO.travel_distance(flow, conf)
        ref_mask, travel = MO.confidence_to_mask(conf, flow, dist, travel, 12.0, "cv2_cubic")
        got = ofgen.confidence_to_mask(conf, flow, dist, aux)
        assert np.array_equal(got, ref_mask) and np.array_equal(aux.pixel_travel_dist, travel)


def test_of_calc_and_raft2(cuda, raft_sd, algo):
    from sd_animation_optical_flow_amd import ofgen
    f1, f2 = _pair(5, 100, 90)                       # not a multiple of 8: RAFT_2 pads and does not un-pad
    r2 = ofgen.RAFT_2(model=raft_sd)
    flo = r2.calc(f1, f2)
    ref = RO.raft2_calc(raft_sd, f1, f2)
    assert flo.shape == ref.shape == (104, 96, 2)
    assert np.sqrt(((flo - ref) ** 2).sum(-1)).mean() < 1e-3
    g1, g2 = _pair(6)
    flow, conf, v, logc = ofgen.of_calc(g1, g2, algo)
    assert np.array_equal(v, MO.travel_distance(flow, conf))
    # the RAFT-variant of_calc (reference ofgen.py:45-49, live caller :137): an algo whose calc returns a bare flow -> (flow, v),
    # v = sqrt(fx*fx + fy*fy) with no confidence floor
    out = ofgen.of_calc(f1, f2, r2)
    assert isinstance(out, tuple) and len(out) == 2
    flow2, v2 = out
    assert flow2.shape == (104, 96, 2) and v2.shape == (104, 96) and v2.dtype == np.float32
    fx, fy = flow2[:, :, 0], flow2[:, :, 1]
    assert np.array_equal(v2, np.sqrt(fx * fx + fy * fy))
    assert np.array_equal(flow2, flo)


def test_compose_greedy_multi_reference(cuda):
    from sd_animation_optical_flow_amd import ofgen
    rng = np.random.default_rng(7)
    H, W, N = 48, 40, 3
    fm = np.zeros((N, 1, H, W, 3), np.float32)
    fm[..., 0:2] = rng.standard_normal((N, 1, H, W, 2)) * 2
    fm[..., 2] = rng.random((N, 1, H, W))
    fm[1, 0, 10:30, 5:35, 2] = 0.99                                      # reference 1 wins the first round
    ai = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]
    orig = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref_frame, ref_mask, ref_order = MO.compose(fm, ai, orig, 0.6, warp_mode="cv2_cubic")
    fm_in = fm.copy()
    frame, mask2, order = ofgen.compose_warp_and_mask(fm_in, ai, orig, thres=0.6)
    assert order == ref_order and np.array_equal(frame, ref_frame) and np.array_equal(mask2, ref_mask)
    assert set(np.unique(fm_in[..., 2])) <= {0.0, 1.0}                   # flow_mat is updated in place (:995, :1023)


class _Video:
    def __init__(self, frames):
        self.frames = frames
        self.size_hw = frames[0].shape[:2]

    def get_raw_frame(self, i):
        return self.frames[i]


class _Idx:
    def __init__(self, idx):
        self.indices = list(idx)

    def __len__(self):
        return len(self.indices)


def test_pdcnet_aux_pair_cache(algo, tmp_path):
    from sd_animation_optical_flow_amd import ofgen
    a, b = _pair(8, 64, 64)
    c, _ = _pair(9, 64, 64)
    video = _Video([a, b, c])
    aux = ofgen.PDCNetAux(algo, str(tmp_path), batch_size=2)
    mat = aux.calculate_multiple_to_one(video, _Idx([0, 2, 1]), 1)
    assert mat.shape == (3, 1, 64, 64, 3)
    assert float(np.abs(mat[2, 0, :, :, 0:2]).max()) == 0.0 and float(mat[2, 0, :, :, 2].min()) == 1.0    # identity pair
    assert sorted(os.listdir(tmp_path / "pdcnet")) == ["00000-00001.npy", "00002-00001.npy"]
    on_disk = np.load(tmp_path / "pdcnet" / "00000-00001.npy")
    assert on_disk.dtype == np.float32 and on_disk.shape == (64, 64, 3) and np.array_equal(on_disk, mat[0, 0])
    flow, conf, _ = algo.calc(a, b)                                       # (s=0, t=1): source a, target b
    assert np.abs(on_disk[..., 0:2] - flow).max() < 1e-4 and np.abs(on_disk[..., 2] - conf).max() < 1e-4
    aux2 = ofgen.PDCNetAux(algo, str(tmp_path), batch_size=2)             # a new instance finds the cache
    assert aux2.cached_pair == {(0, 1), (2, 1)}
    assert np.array_equal(aux2.calcualte_single(video, 0, 1), on_disk)
    pw = aux2.calculate_pairwise(video, _Idx([0, 1]))
    assert pw.shape == (2, 2, 64, 64, 3) and np.array_equal(pw[0, 1], on_disk)
    scores = aux2.keyframe_scores(pw)
    assert np.allclose(scores, MO.keyframe_scores(pw), rtol=1e-6)
    # background writer: same files, available after flush(); the private copy survives the caller mutating `mat`
    aux3 = ofgen.PDCNetAux(algo, str(tmp_path / "async"), batch_size=2, async_save=True)
    mat3 = aux3.calculate_multiple_to_one(video, _Idx([0, 2, 1]), 1)
    assert np.array_equal(mat3, mat)
    mat3[:] = -1.0
    aux3.flush()
    assert np.array_equal(np.load(tmp_path / "async" / "pdcnet" / "00000-00001.npy"), on_disk)
    assert np.array_equal(aux3.load_cached(2, 1), mat[1, 0])


def test_frame_synthesizer_and_process_clip_device_path(algo, raft_sd):
    """The device-resident tail (flow -> warp -> mask without leaving HBM) and the single-rank clip driver."""
    from sd_animation_optical_flow_amd import clip
    H, W, T = 96, 128, 5
    a, _ = _pair(10, H, W)
    key = torch.from_numpy(a[:, :, ::-1].copy()).cuda()                      # RGB key frame
    frames = torch.stack([torch.roll(key, shifts=(t - 2, 2 - t), dims=(0, 1)) for t in range(T)]).contiguous()
    key_ai = (255 - key).contiguous()
    synth = clip.FrameSynthesizer(algo, warp_mode="cv2_cubic", thres=0.95, ksize=7)
    res = clip.process_clip(frames, key, key_ai, synth, batch_size=2)
    assert res.frame_indices == list(range(T)) and [t.shape[0] for t in res.flow] == [2, 2, 1]
    flow = torch.cat(res.flow)
    warped = torch.cat(res.warped)
    mask = torch.cat(res.mask)
    assert flow.is_cuda and warped.is_cuda and mask.is_cuda                   # nothing left the device
    # same numbers as the host-side reference-style calls
    f0, c0, _ = algo.calc(a, np.ascontiguousarray(frames[0].cpu().numpy()[:, :, ::-1]))      # (source=key, target=frame 0), BGR
    assert np.abs(flow[0].cpu().numpy() - f0).max() < 1e-4
    assert np.array_equal(warped[0].cpu().numpy(), WO.warp_frame(key_ai.cpu().numpy(), flow[0].cpu().numpy(), "cv2_cubic"))
    ref_mask, _ = MO.generate_mask(c0, c0.copy(), 0.95, 7)
    assert (mask[0].cpu().numpy() != ref_mask).mean() < 2e-3                  # confidence agrees to ~1e-6: masks differ only at exact ties


@pytest.mark.parametrize("H,W", [(96, 128), (100, 132)])
def test_frame_synthesizer_bilinear_takes_the_warp_inside_the_upsample(algo, raft_sd, H, W):
    """`FrameSynthesizer(warp_mode='bilinear')` -- what `ClipPipeline` and `bench.py` run -- in its three forms: through the algo
    (flow both ways + forward-backward confidence, `ofx_raft_forward_pairs_warp`), through the engine with a confidence from the
    caller (`ofx_raft_forward_warp`), and with the fusion switched off (upsample, then `ofx_warp_and_mask`): identical flow, warped
    frame and mask, on a frame the network takes as is and on one it pads to a multiple of 8 (where the fused form steps aside)."""
    from sd_animation_optical_flow_amd import clip, ops
    a, _ = _pair(12, H, W)
    key = torch.from_numpy(a[:, :, ::-1].copy()).cuda()
    frames = torch.stack([torch.roll(key, shifts=(t - 1, 1 - 2 * t), dims=(0, 1)) for t in range(3)]).contiguous()
    key_ai = (255 - key).contiguous()
    fused = clip.FrameSynthesizer(algo, warp_mode="bilinear", thres=0.9, ksize=7)
    plain = clip.FrameSynthesizer(algo, warp_mode="bilinear", thres=0.9, ksize=7, fuse_warp=False)
    f1, c1, w1, m1 = fused.synthesize(frames, key, key_ai)
    f2, c2, w2, m2 = plain.synthesize(frames, key, key_ai)
    assert tuple(w1.shape) == (3, H, W, 3) and tuple(m1.shape) == (3, H, W) and tuple(f1.shape) == (3, H, W, 2)
    assert torch.equal(f1, f2) and torch.equal(c1, c2) and torch.equal(w1, w2) and torch.equal(m1, m2)
    assert torch.equal(w1, ops.warp(key_ai, f1.contiguous(), mode="bilinear", sign=1.0))
    # engine + external confidence: one flow per pair, same frame -> key frame flow as the algo's forward half
    conf = torch.rand((3, H, W), generator=torch.Generator().manual_seed(4)).cuda()
    eng = clip.FrameSynthesizer(engine=algo.network, warp_mode="bilinear", thres=0.9, ksize=7)
    f3, w3, m3 = eng(frames, key, key_ai, confidence=conf)
    assert (f3 - f1).abs().max().item() < 1e-3                                   # (pairs call vs shared-key call: other tile schedules)
    assert torch.equal(w3, ops.warp(key_ai, f3.contiguous(), mode="bilinear", sign=1.0))
    assert torch.equal(m3, ops.generate_mask(conf, None, 0.9, 7))
    with pytest.raises(ValueError):
        eng(frames, key, key_ai)                                                 # an engine alone has no confidence to offer


def test_large_frame_1024x1024_single_pair(cuda, raft_sd):
    """BASELINE config #5 frame size (flow part): 128x128 coarse grid, 1.43 GB pyramid, finite and self-consistent."""
    from sd_animation_optical_flow_amd.raft import RaftEngine
    eng = RaftEngine(raft_sd)
    H = W = 1024
    g = torch.Generator().manual_seed(5)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 16, W + 16), generator=g), 7, 1, 3)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)
    a = base[0, :, 8:8 + H, 8:8 + W].permute(1, 2, 0).contiguous().cuda()
    b = base[0, :, 11:11 + H, 6:6 + W].permute(1, 2, 0).contiguous().cuda()
    up, lo = eng.forward(a[None], b[None], iters=4, want_low=True)
    assert tuple(up.shape) == (1, H, W, 2) and torch.isfinite(up).all()
    # the convex upsample of a constant field is the constant field x 8: check on the coarse flow's interior mean
    assert abs(float(up[0, 64:-64, 64:-64].mean()) / max(abs(float(lo[0, 8:-8, 8:-8].mean())), 1e-6) - 8.0) < 1.0


def test_full_hd_frame_single_pair(cuda, raft_sd):
    """1920x1080 (1/8 grid 135x240, odd height): the volume GEMM's M split engages (32400^2 floats = 4.2 GB);
    the level-0 volume must agree with a direct evaluation from the engine's own feature maps."""
    from sd_animation_optical_flow_amd.raft import RaftEngine
    eng = RaftEngine(raft_sd)
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, device="cuda", generator=g)
    up = eng.forward(a[None], b[None], iters=2)
    assert tuple(up.shape) == (1, 1080, 1920, 2) and bool(torch.isfinite(up).all())
    N = 135 * 240
    f1 = eng.buffer("fmap1").view(N, 256)
    f2 = eng.buffer("fmap2").view(N, 256)
    rows = torch.tensor([0, 777, 16383, 16384, N - 1], device="cuda")
    ref = (f1[rows].double() @ f2.double().T / 16.0).float()
    from sd_animation_optical_flow_amd import ops
    slices = eng.buffer("pyr0").view(N, ops.corr_slice_floats(135, 240))[rows]          # 4x8-blocked, 34 x 30 blocks per slice
    got = ops.corr_unblock(slices, 135, 240).reshape(len(rows), N)
    assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_raft2_as_written_against_the_reference_train_mode_vectors(cuda, raft_sd):
    """ofgen.RAFT_2 (default cnet_norm='batch') against what the REAL reference RAFT returned when driven exactly as
    `RAFT_2` drives it -- DataParallel, never `.eval()`, InputPadder, 20 iterations, no un-padding
    (tests/golden/raft_ref_trainbn_128x160.npz, made by make_golden.py from /root/reference).  One hop, no oracle."""
    import os
    from sd_animation_optical_flow_amd import ofgen
    from sd_animation_optical_flow_amd.raft import RaftEngine
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "raft_ref_trainbn_128x160.npz"))
    r2 = ofgen.RAFT_2(model=raft_sd)
    assert r2.model.cnet_norm == "batch"
    for tag in ("a", "b"):
        flo = r2.calc(g[f"frame1_{tag}"], g[f"frame2_{tag}"])
        ref = g[f"flow_{tag}"]
        assert flo.shape == ref.shape and flo.dtype == np.float32          # 132x156 comes back padded to 136x160
        epe = np.sqrt(((flo - ref) ** 2).sum(-1)).mean()
        assert epe < 1e-3 and np.abs(flo - ref).max() < 1e-2, (tag, epe)
    # the context features themselves (half-precision fixture), straight from the engine's state buffer
    hx = r2.model.buffer("hx").cpu().reshape(17 * 20, 384)
    net = hx[:, :128].T.reshape(128, 17, 20)
    inp = hx[:, 256:].T.reshape(128, 17, 20)
    assert (inp - torch.from_numpy(g["inp_f16_b"][0].astype(np.float32))).abs().max().item() < 2e-2
    del net                                                                # h has been through 20 GRU updates by now
    # the opt-in eval mode is the OTHER network on these weights: pixels away from the as-written reference
    ev = ofgen.RAFT_2(model=raft_sd, cnet_norm="eval").calc(g["frame1_a"], g["frame2_a"])
    assert np.sqrt(((ev - g["flow_a"]) ** 2).sum(-1)).mean() > 1.0
    # a batch is as many single-image reference calls: every image normalised by itself, also next to a shared key frame
    eng = RaftEngine(raft_sd, cnet_norm="batch")
    fr = torch.from_numpy(np.stack([g["frame1_a"], g["frame2_a"], g["frame1_a"][::-1].copy()])).cuda()
    key = torch.from_numpy(g["frame2_a"]).cuda()
    up = eng.forward(fr, key, iters=20, bgr=True)
    assert np.sqrt(((up[0].cpu().numpy() - g["flow_a"]) ** 2).sum(-1)).mean() < 1e-3
    for i in (1, 2):
        one = eng.forward(fr[i:i + 1], key[None], iters=20, bgr=True)
        assert (up[i] - one[0]).abs().max().item() < 2e-3
    # key frame as image1 (the PDCNet orientation): ONE context map shared by the batch
    sh = eng.forward(key, fr, iters=20, bgr=True)
    one = eng.forward(key[None], fr[2:3], iters=20, bgr=True)
    assert (sh[2] - one[0]).abs().max().item() < 2e-3
    # indexed pairs (KeyframeConv's sweep) take the same switch
    imgs = torch.cat([fr[:2], key[None]])
    fp = eng.forward_pairs(imgs, [0, 1], [1, 0], iters=20, bgr=True)
    assert np.sqrt(((fp[0].cpu().numpy() - g["flow_a"]) ** 2).sum(-1)).mean() < 1e-3


def test_calc_returns_the_unpadded_size_for_frames_not_divisible_by_8(algo, raft_sd):
    """pdcnet_of.py:72-75 promises [H,W] outputs.  The network runs on the replicate-padded grid (InputPadder 'sintel',
    utils.py:9-16); flow, confidence and log-confidence must be cropped back -- RAFT_2.calc alone keeps the padded size
    (the reference never un-pads there)."""
    f1, f2 = _pair(9, H=100, W=90)
    flow, conf, logc = algo.calc(f1, f2)
    assert flow.shape == (100, 90, 2) and conf.shape == (100, 90) and logc.shape == (100, 90)
    ref = RO.raft2_calc(raft_sd, f2, f1, cnet_norm="eval")   # padded: 104 x 96, offsets (2, 3)
    assert ref.shape[:2] == (104, 96)
    assert np.sqrt(((flow - ref[2:102, 3:93]) ** 2).sum(-1)).mean() < 1e-3
    # warp / mask on the cropped outputs line up with the frame again
    from sd_animation_optical_flow_amd import ofgen, pdcnet_of
    assert pdcnet_of.warp_frame(f1, flow).shape == f1.shape
    m, _ = ofgen.generate_mask(conf, logc.copy(), 0.95)
    assert m.shape == (100, 90)
    dev = torch.from_numpy(np.stack([f1, f2])).cuda()
    fw, cf = algo.calc_pairs(dev, [(0, 1), (1, 0)], bgr=True)
    assert tuple(fw.shape) == (2, 100, 90, 2) and tuple(cf.shape) == (2, 100, 90)
    assert (fw[0].cpu().numpy() - flow).__abs__().max() < 1e-4


def test_calc_pairs_honours_the_per_call_pair_limit(algo, monkeypatch):
    """One executor call addresses its operands with 32-bit offsets (113 pairs at 512x768, 21 at 1080p): calc_pairs must
    slice by RaftEngine.max_pairs, not by a fixed 64.  Mock the limit small and hold the result to the unsliced one."""
    from sd_animation_optical_flow_amd.raft import RaftEngine
    f1, f2 = _pair(10)
    f3, _ = _pair(11)
    dev = torch.from_numpy(np.stack([f1, f2, f3])).cuda()
    pairs = [(s, t) for s in range(3) for t in range(3) if s != t]
    want_f, want_c = algo.calc_pairs(dev, pairs)
    calls = []
    real = RaftEngine.forward_pairs

    def spy(self, images, idx1, idx2, **kw):
        calls.append(len(idx1))
        return real(self, images, idx1, idx2, **kw)
    monkeypatch.setattr(RaftEngine, "max_pairs", staticmethod(lambda H, W: 4))
    monkeypatch.setattr(RaftEngine, "forward_pairs", spy)
    got_f, got_c = algo.calc_pairs(dev, pairs)
    assert calls and max(calls) <= 4 and sum(calls) == 6
    assert (got_f - want_f).abs().max().item() < 1e-4 and (got_c - want_c).abs().max().item() < 1e-4


def test_pair_batches_are_sliced_by_the_memory_that_is_free(algo, monkeypatch):
    """The executor's workspace grows with (H/8 x W/8)^2 per pair; RaftEngine.max_pairs_now bounds a call by the device memory that
    is free (round 6).  Force the budget down to two pairs' worth: calc_pairs and the batched frame path slice accordingly and
    return what the unsliced calls return."""
    from sd_animation_optical_flow_amd import _lib
    from sd_animation_optical_flow_amd.raft import RaftEngine
    f1, f2 = _pair(12)
    f3, _ = _pair(13)
    dev = torch.from_numpy(np.stack([f1, f2, f3])).cuda()
    pairs = [(s, t) for s in range(3) for t in range(3) if s != t]
    want_f, want_c = algo.calc_pairs(dev, pairs)
    want = algo.calc_batch_device(dev[0], dev[1:], bgr=True)
    net = algo.network
    Hp, Wp = (dev.shape[1] + 7) // 8 * 8, (dev.shape[2] + 7) // 8 * 8
    calls = []
    real = RaftEngine.forward_pairs

    def spy(self, images, idx1, idx2, **kw):
        calls.append(len(idx1))
        return real(self, images, idx1, idx2, **kw)
    monkeypatch.setattr(RaftEngine, "forward_pairs", spy)
    try:
        net.ws_budget_bytes = int(_lib.lib().ofx_raft_workspace_bytes_pairs(net._h, 4, 2, Hp, Wp))
        assert net.max_pairs_now(Hp, Wp) == 2 and net.max_pairs_now(Hp, Wp, 1) == 1
        got_f, got_c = algo.calc_pairs(dev, pairs)
        assert max(calls) <= 2 and sum(calls) == 6
        del calls[:]
        got = algo.calc_batch_device(dev[0], dev[1:], bgr=True)    # two frames, both directions: four pairs -> two calls
        assert calls == [2, 2]
    finally:
        net.ws_budget_bytes = None
    assert (got_f - want_f).abs().max().item() < 1e-4 and (got_c - want_c).abs().max().item() < 1e-4
    assert (got[0] - want[0]).abs().max().item() < 1e-4 and (got[1] - want[1]).abs().max().item() < 1e-4


def test_keyframe_conv_device_resident_over_a_workspace(algo, tmp_path):
    """KeyframeConv (ofgen_keyframe_inpaint.py:655-674) over a `workspace.VideoData`: windows from `conv_indices`, scores
    reduced on the device.  Held to the reference's host formulation -- build the [N,N,H,W,3] matrix with
    `calculate_pairwise`, `einops.reduce(..., 's t h w -> s', 'sum')`, np.argmax -- on the same workspace."""
    from sd_animation_optical_flow_amd import ofgen
    from sd_animation_optical_flow_amd.workspace import VideoData, VideoFrameIndices
    H, W = 64, 96
    g = torch.Generator().manual_seed(77)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 40, W + 40), generator=g), 5, 1, 2)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)[0].permute(1, 2, 0).numpy()
    shifts = [(0, 0), (1, 2), (3, 3), (9, 1), (10, 2), (11, 4), (18, 9)]
    frames = [np.ascontiguousarray(base[20 + dy:20 + dy + H, 20 + dx:20 + dx + W]) for dy, dx in shifts]
    video = VideoData(frames, (W, H), str(tmp_path / "ws"))
    aux = ofgen.PDCNetAux(algo, video.workspace_dir, batch_size=4)
    idx = VideoFrameIndices.from_n(video.num_frames)
    kf_dir = str(tmp_path / "ws" / "keyframes")
    got = ofgen.keyframe_conv(aux, kf_dir, video, idx, kernel_size=5, stride=2, dilation=2)
    # nothing but the winners' PNGs is written unless asked: the N*(N-1) pair dumps (4.7 MB each at 512x768) stay off disk
    assert os.listdir(tmp_path / "ws" / "pdcnet") == []
    want = set()
    host = ofgen.PDCNetAux(algo, str(tmp_path / "host"), batch_size=4)
    for window in idx.conv_indices(5, 2, 2):
        mat = host.calculate_pairwise(video, window)
        scores = mat[:, :, :, :, 2].sum(axis=(1, 2, 3), dtype=np.float64)
        dev_scores = aux.keyframe_scores_device(video, window).cpu().numpy()
        assert np.allclose(dev_scores, scores, rtol=1e-5), (window.indices, dev_scores, scores)
        want.add(window.indices[int(np.argmax(scores))])
    assert got.indices == sorted(want)
    assert sorted(os.listdir(kf_dir)) == [f"{i:05d}.png" for i in sorted(want)]
    assert np.array_equal(VideoData(None, (W, H), str(tmp_path / "ws")).get_raw_frame(got.indices[0]), frames[got.indices[0]])
    # a populated result directory short-circuits the computation (:656-660)
    again = ofgen.KeyframeConv(None, kf_dir, None, None)
    assert again.indices == got.indices
    # save_pairs=True leaves a workspace the reference's PDCNetAux can continue from; cached pairs are reused
    aux.keyframe_scores_device(video, VideoFrameIndices([0, 1, 2]), save_pairs=True)
    assert len(os.listdir(tmp_path / "ws" / "pdcnet")) == 6 and (1, 2) in aux.cached_pair
    s1 = aux.keyframe_scores_device(video, VideoFrameIndices([0, 1, 2])).cpu().numpy()
    s0 = host.calculate_pairwise(video, VideoFrameIndices([0, 1, 2]))[..., 2].sum(axis=(1, 2, 3), dtype=np.float64)
    assert np.allclose(s1, s0, rtol=1e-5)


def test_keyframe_conv_ties_go_to_the_earliest_frame(cuda, tmp_path):
    """np.argmax keeps the first maximum (:667): with an of-algo whose confidence is the same for every pair every window
    elects its first member."""
    from sd_animation_optical_flow_amd import ofgen
    from sd_animation_optical_flow_amd.workspace import VideoData, VideoFrameIndices

    class Flat:
        def to(self, device):
            return self

        def calc_pairs(self, frames, pairs, **kw):
            n, h, w, _ = frames.shape
            return (torch.zeros((len(pairs), h, w, 2), device=frames.device), torch.full((len(pairs), h, w), 0.5, device=frames.device))

    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, (16, 24, 3), dtype=np.uint8) for _ in range(9)]
    video = VideoData(frames, (24, 16), str(tmp_path / "ws"))
    aux = ofgen.PDCNetAux(Flat(), video.workspace_dir)
    got = ofgen.keyframe_conv(aux, str(tmp_path / "kf"), video, VideoFrameIndices.from_n(9), kernel_size=4, stride=3, dilation=1)
    assert got.indices == [0, 3, 6]


def test_clip_pipeline_end_to_end_over_a_workspace(algo, tmp_path):
    """The whole non-generative chain on one small clip: workspace PNGs -> key-frame decisions -> per-frame flow /
    confidence against the segment's key frame -> warp + mask -> Pillow-exact SD-inpaint inputs -> first-stage latent;
    every AI frame lands in `ai-frames/` and the per-frame packet agrees with the step-by-step reference-style calls."""
    from sd_animation_optical_flow_amd import handoff, ops, pipeline
    from sd_animation_optical_flow_amd.vae import VaeEncoder, random_vae_state_dict
    from sd_animation_optical_flow_amd.workspace import VideoData
    H, W = 64, 96
    g = torch.Generator().manual_seed(123)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 60, W + 60), generator=g), 5, 1, 2)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)[0].permute(1, 2, 0).numpy()
    scene2 = np.ascontiguousarray(base[::-1, ::-1])                                   # a cut: new key frame
    frames = [np.ascontiguousarray(base[30 + s:30 + s + H, 30 + 2 * s:30 + 2 * s + W]) for s in range(4)]
    frames += [np.ascontiguousarray(scene2[20 + s:20 + s + H, 25:25 + W]) for s in range(3)]
    video = VideoData(frames, (W, H), str(tmp_path / "ws"))
    vae = VaeEncoder(random_vae_state_dict(0))
    pipe = pipeline.ClipPipeline(algo, vae=vae, batch=2, warp_mode="bilinear", thres=0.9, ksize=7)
    flags = [True, False, False, False, True, False, False]
    seen = {}
    for pkt, raw, idx in pipe.packets(video, flags):
        if pkt is not None:
            seen[idx] = pkt
    assert sorted(seen) == [1, 2, 3, 5, 6] and seen[3].key_index == 0 and seen[5].key_index == 4
    p = seen[2]
    assert tuple(p.inpaint["init_latent"].shape) == (4, H // 8, W // 8) and bool(torch.isfinite(p.inpaint["init_latent"]).all())
    # the packet equals the reference-style host calls for that frame: calc(key, frame), warp, mask
    flow, conf, _ = algo.calc(frames[0], frames[2])
    assert np.abs(p.flow.cpu().numpy() - flow).max() < 1e-4 and np.abs(p.confidence.cpu().numpy() - conf).max() < 1e-4
    ref_w = WO.warp_frame(frames[0], p.flow.cpu().numpy(), mode="bilinear")
    d = np.abs(p.warped.cpu().numpy().astype(int) - ref_w.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    ref_m, _ = MO.generate_mask(p.confidence.cpu().numpy(), conf.copy(), 0.9, 7)
    assert np.array_equal(p.mask.cpu().numpy(), ref_m)
    keys = pipe.run(video, flags)
    assert keys == [0, 4] and all(video.generated(i) for i in range(7))
    assert np.array_equal(video.get_ai_frame(4), frames[4])                           # default key render = identity
    out2 = video.get_ai_frame(2)                                                      # default render: raw pixels where masked
    m = p.mask.cpu().numpy() == 255
    assert np.array_equal(out2[m], frames[2][m]) and np.array_equal(out2[~m], p.warped.cpu().numpy()[~m])


def test_clip_pipeline_async_host_side_changes_no_byte(algo, tmp_path):
    """`ClipPipeline.run` with the asynchronous host side (thread-pool PNG decode into pinned staging, H2D on a copy stream one batch
    ahead, D2H + PNG encode off-thread: hostio.py) against the same run with everything inline (`io_threads=0`): every AI frame on disk
    byte for byte the same, every frame written, key frames included."""
    from sd_animation_optical_flow_amd import pipeline
    from sd_animation_optical_flow_amd.workspace import VideoData
    H, W = 64, 96
    g = torch.Generator().manual_seed(321)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 80, W + 80), generator=g), 5, 1, 2)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)[0].permute(1, 2, 0).numpy()
    frames = [np.ascontiguousarray(base[20 + s:20 + s + H, 20 + 2 * s:20 + 2 * s + W]) for s in range(13)]
    flags = [True] + [False] * 6 + [True] + [False] * 5
    outs = []
    for threads in (0, 3):
        video = VideoData(frames, (W, H), str(tmp_path / f"ws{threads}"))
        pipe = pipeline.ClipPipeline(algo, batch=4, warp_mode="bilinear", thres=0.9, ksize=7, io_threads=threads, prefetch=2)
        assert pipe.run(video, flags) == [0, 7]
        assert all(video.generated(i) for i in range(13))
        outs.append([video.get_ai_frame(i) for i in range(13)])
    for i in range(13):
        assert np.array_equal(outs[0][i], outs[1][i]), i
    # round 5: the first and the last batch of the share cut into edge pieces (`edge_batch`: [4, 2] + [4, 1] -> [2, 2, 2] + [4, 1] ...): other
    # executor calls for the same frames.  A pair's flow must not depend on the batch it rides in beyond fp32 summation order, so the
    # rendered bytes agree up to one grey level on a vanishing share of the pixels -- and every frame is still written exactly once
    video = VideoData(frames, (W, H), str(tmp_path / "ws_edge"))
    pipe = pipeline.ClipPipeline(algo, batch=4, warp_mode="bilinear", thres=0.9, ksize=7, io_threads=3, prefetch=2, edge_batch=2)
    assert pipe.run(video, flags) == [0, 7]
    assert all(video.generated(i) for i in range(13))
    for i in range(13):
        d = np.abs(video.get_ai_frame(i).astype(np.int32) - outs[0][i].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (i, int(d.max()), float((d > 0).mean()))


def test_config_c5_1024x1024_flow_warp_mask_against_the_oracle(cuda, raft_sd):
    """BASELINE config #5's frame size, the path's own half of it at full depth: one 1024x1024 pair, 20 iterations, flow
    against the CPU oracle (EPE bar 1e-3 px), then warp + mask of a small batch against their oracles and the hand-off +
    first-stage latent on top (finite, right shapes).  The generative half (UNet / sampler) is out of scope."""
    from sd_animation_optical_flow_amd import handoff, ops
    from sd_animation_optical_flow_amd.raft import RaftEngine
    eng = RaftEngine(raft_sd)
    H = W = 1024
    g = torch.Generator().manual_seed(55)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 16, W + 16), generator=g), 7, 1, 3)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)
    key = base[0, :, 8:8 + H, 8:8 + W].permute(1, 2, 0).contiguous()
    frames = torch.stack([base[0, :, 8 + dy:8 + dy + H, 8 + dx:8 + dx + W].permute(1, 2, 0) for dy, dx in ((3, -2), (-1, 4), (2, 2), (0, -3))]).contiguous()
    flow = eng.forward(frames.cuda(), key.cuda(), iters=20)
    _, up = RO.raft_forward(raft_sd, frames[:1].permute(0, 3, 1, 2).float(), key.permute(2, 0, 1)[None].float(), iters=20)
    epe = (flow[0].cpu() - up[0].permute(1, 2, 0)).pow(2).sum(-1).sqrt().mean().item()
    assert epe < 1e-3, epe
    key_ai = (255 - key).contiguous().cuda()
    conf = torch.rand((4, H, W), generator=g).cuda()
    warped, mask = ops.warp_and_mask(key_ai, flow, conf, warp_mode="bilinear", thres=0.95, ksize=7)
    ref_w = WO.warp_frame(key_ai.cpu().numpy(), flow[0].cpu().numpy(), mode="bilinear")
    d = np.abs(warped[0].cpu().numpy().astype(int) - ref_w.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    c0 = conf[0].cpu().numpy()
    ref_m, _ = MO.generate_mask(c0, c0.copy(), 0.95, 7)
    assert np.array_equal(mask[0].cpu().numpy(), ref_m)
    t = handoff.prepare_inpaint_inputs(warped, frames.cuda(), mask, mask_blur=4)
    assert tuple(t["image"].shape) == (4, 3, H, W) and bool(torch.isfinite(t["image"]).all())
