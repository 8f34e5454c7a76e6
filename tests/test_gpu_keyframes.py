"""-m gpu: key-frame detector kernels against the oracle (bit-exact; the oracle itself is unpinned -- no OpenCV here)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import keyframe_oracle as KO


def _scene(seed, H, W, shift=0):
    """Blocks + gradient + noise: long closed contours (hysteresis across tiles), weak and strong edges."""
    rng = np.random.default_rng(seed)
    img = np.zeros((H, W, 3), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    img += (xx[..., None] * 0.25 + yy[..., None] * 0.1)
    for _ in range(8):
        y0, x0 = rng.integers(0, max(1, H - 10)), rng.integers(0, max(1, W - 10))
        h, w = rng.integers(5, max(6, H // 2)), rng.integers(5, max(6, W // 2))
        img[y0:y0 + h, x0:x0 + w] += rng.integers(-90, 90, 3)
    img = np.roll(img, shift, axis=1)
    img += rng.normal(0, 3.0, img.shape)
    return np.clip(img + 60, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("H,W", [(768, 512), (97, 131), (40, 33), (8, 300)])
def test_detect_edges_matches_oracle(cuda, H, W):
    from sd_animation_optical_flow_amd import keyframes
    frames = np.stack([_scene(s, H, W) for s in (1, 2)])
    frames[1, :, :, :] = frames[1, :, :, :] // 2                      # different median -> different thresholds per image
    out = keyframes.detect_edges(frames).cpu().numpy()
    k = KO.estimated_kernel_size(W, H)
    for b in range(2):
        ref = KO.detect_edges(frames[b], k)
        assert np.array_equal(out[b], ref), (b, int((out[b] != ref).sum()))
    # un-dilated Canny (ksize 1) as well: exercises the map + hysteresis without the dilation hiding differences
    raw = keyframes.detect_edges(frames, ksize=1).cpu().numpy()
    for b in range(2):
        lum = KO.hsv_value(frames[b])
        assert np.array_equal(raw[b], KO.canny(lum, *KO.canny_thresholds(lum))), b


def test_hysteresis_follows_a_long_weak_contour(cuda):
    """A faint spiral that is strong at one end only: the weak part must be kept along its whole length (many
    32x32 tiles, many sweeps)."""
    from sd_animation_optical_flow_amd import keyframes
    H = W = 200
    img = np.full((H, W), 100, np.int32)
    for i in range(4, 96, 8):                                         # nested square rings, one pixel wide, faint
        img[i, i:W - i] += 30; img[H - 1 - i, i:W - i] += 30; img[i:H - i, i] += 30; img[i:H - i, W - 1 - i] += 30
        img[i, i:i + 6] += 120                                        # a short strong segment on each ring
    frame = np.repeat(img.clip(0, 255).astype(np.uint8)[..., None], 3, 2)
    out = keyframes.detect_edges(frame, ksize=1).cpu().numpy()
    lum = KO.hsv_value(frame)
    ref = KO.canny(lum, *KO.canny_thresholds(lum))
    assert np.array_equal(out, ref) and (ref > 0).sum() > 2000


def test_mean_pixel_distance_and_generator_decisions(cuda):
    from sd_animation_optical_flow_amd import keyframes
    H, W = 96, 128
    seq = [_scene(5, H, W, shift=s) for s in (0, 0, 1, 2, 30, 31, 31, 80)]
    e0, e1 = keyframes.detect_edges(seq[0]), keyframes.detect_edges(seq[4])
    assert keyframes.mean_pixel_distance(e0, e1) == KO.mean_pixel_distance(e0.cpu().numpy(), e1.cpu().numpy())
    for th in (8.5, 30.0, 0.0):
        got = [(k, i) for _, k, i in keyframes.frame_generator(seq, fps=30.0, th=th, batch=3)]
        assert [k for k, _ in got] == KO.keyframe_flags(seq, fps=30.0, th=th) and [i for _, i in got] == list(range(len(seq)))
    # keep_every = 3 (the reference's live setting): the gap that relaxes the threshold counts DROPPED frames too
    # (ofgen_keyframe_inpaint.py:346-352), so a small max_gap makes the decisions depend on it
    seq3 = [_scene(5, H, W, shift=s) for s in (0, 0, 0, 1, 1, 1, 3, 3, 3, 30, 30, 30, 31, 32, 33, 80)]
    for th, mg in ((8.5, -1), (30.0, 12), (12.0, 10)):
        got = [(k, i) for _, k, i in keyframes.frame_generator(seq3, fps=30.0, th=th, max_gap=mg, batch=2, keep_every=3)]
        want = KO.keyframe_flags(seq3, fps=30.0, th=th, max_gap=mg, keep_every=3)
        assert [k for k, _ in got] == want and [i for _, i in got] == list(range(len(want))) and len(want) == 6
    assert keyframes.gaps(24.0) == KO.gaps(24.0) == (8, 240) and keyframes.estimated_kernel_size(512, 768) == 7
