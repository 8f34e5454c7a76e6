"""CPU: the reference's workspace formats -- `VideoData` directory layout and PNG round trip,
`VideoFrameIndices` window enumeration (hand-derived from ofgen_keyframe_inpaint.py:483-541)."""
import os

import numpy as np
import pytest

from sd_animation_optical_flow_amd.workspace import VideoData, VideoFrameIndices


def _frames(n, h=24, w=32, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]


def test_video_data_layout_and_round_trip(tmp_path):
    frames = _frames(7)
    ws = str(tmp_path / "ws")
    v = VideoData(frames, (32, 24), ws, keep_every=3)
    assert v.num_frames == 3 and v.size_hw == (24, 32) and v.fps == 10.0
    for d in ("raw-frames", "ai-frames", "pdcnet", "crossattn", "seed"):
        assert os.path.isdir(os.path.join(ws, d))
    assert sorted(os.listdir(os.path.join(ws, "raw-frames"))) == ["00000.png", "00001.png", "00002.png"]
    for k, src in enumerate((0, 3, 6)):
        assert np.array_equal(v.get_raw_frame(k), frames[src])           # lossless, BGR in memory
    # the file on disk is an ordinary RGB PNG (what cv2.imwrite produces from a BGR array)
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(ws, "raw-frames", "00001.png")))
    assert np.array_equal(rgb, frames[3][:, :, ::-1])
    # ai-frames: generated() is file presence
    assert not v.generated(1) and v.get_ai_frame(1) is None
    v.put_ai_frame(1, frames[5])
    assert v.generated(1) and np.array_equal(v.get_ai_frame(1), frames[5])
    with pytest.raises(AssertionError):
        v.get_raw_frame(3)
    # reopening an existing workspace counts the frames instead of re-extracting
    v2 = VideoData(None, (32, 24), ws)
    assert v2.num_frames == 3 and np.array_equal(v2.get_raw_frame(2), frames[6])
    v2.put_kv(2, {"k": [1, 2, 3]})
    assert v2.get_kv(2) == {"k": [1, 2, 3]}
    v2.remove_kv(2)
    assert not os.path.exists(os.path.join(ws, "crossattn", "00002.bin"))
    with pytest.raises(ValueError):
        VideoData(_frames(1, 10, 10), (32, 24), str(tmp_path / "bad"))


def test_max_len_and_existing_files_are_kept(tmp_path):
    ws = str(tmp_path / "ws")
    v = VideoData(_frames(40), (32, 24), ws, keep_every=1, max_len_sec=1, fps=8.0)
    assert v.num_frames == 9                                   # the reference stops after ctr_valid >= fps * seconds
    before = os.path.getmtime(os.path.join(ws, "raw-frames", "00000.png"))
    VideoData(_frames(3, seed=5), (32, 24), ws)                # same workspace again: existing PNGs are not rewritten
    assert os.path.getmtime(os.path.join(ws, "raw-frames", "00000.png")) == before


def test_conv_indices_windows():
    """indices[idx : idx + kernel][0::dilation], idx += stride (:493-497)."""
    idx = VideoFrameIndices.from_n(40)
    got = [w.indices for w in idx.conv_indices(17, 8, 2)]
    assert got == [list(range(0, 17, 2)), list(range(8, 25, 2)), list(range(16, 33, 2)), list(range(24, 40, 2)),
                   [32, 34, 36, 38]]
    assert [w.indices for w in VideoFrameIndices([5, 1, 9, 1]).conv_indices(2, 1, 1)] == [[1, 5], [5, 9], [9]]
    assert [w.indices for w in VideoFrameIndices([]).conv_indices()] == []
    sparse = VideoFrameIndices([3, 10, 11, 40, 41, 42, 90])
    assert [w.indices for w in sparse.conv_indices(4, 3, 1)] == [[3, 10, 11, 40], [40, 41, 42, 90], [90]]


def test_index_set_operations_and_adjacent_frames():
    a = VideoFrameIndices([4, 2, 2, 8])
    assert a.indices == [2, 4, 8] and len(a) == 3
    a.add(6)
    a.add(VideoFrameIndices([1, 8]))
    assert a.indices == [1, 2, 4, 6, 8]
    a.remove(VideoFrameIndices([2, 100]))
    assert a.indices == [1, 4, 6, 8]
    idx = VideoFrameIndices(range(0, 100, 10))
    assert idx.adjacent_frames(34, 3).indices == [20, 30, 40]
    assert idx.adjacent_frames(-5, 2).indices == [0, 10]
    # the reference's range(0, len - n) never offers the last run: the closest run to 95 is [70, 80], not [80, 90]
    assert idx.adjacent_frames(95, 2).indices == [70, 80]
    assert idx.adjacent_frames(50, 20) is idx
    # ties keep the first run
    assert VideoFrameIndices([0, 10, 20, 30]).adjacent_frames(15, 2).indices == [10, 20]
    assert VideoFrameIndices([0, 10, 20, 30]).adjacent_frames(10, 1).indices == [10]


# ------------------------------------------------------------------------------------------------
# hostio: the asynchronous host side of the workspace driver (CPU device: threads, staging ring, error propagation)
# ------------------------------------------------------------------------------------------------
def _small_workspace(tmp_path, n=11, H=12, W=20, seed=7):
    rng = np.random.default_rng(seed)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(n)]
    return VideoData(frames, (W, H), str(tmp_path / "ws")), frames


def test_frame_loader_prefetches_batches_in_any_order(tmp_path):
    import torch
    from sd_animation_optical_flow_amd import hostio
    video, frames = _small_workspace(tmp_path)
    for threads in (0, 3):
        loader = hostio.FrameLoader(video, "cpu", threads=threads, slots=2, batch=4)
        tickets = [loader.request(ids) for ids in ([0, 1, 2, 3], [4], [10, 9, 8], [5, 6, 7, 2])]     # more requests than ring slots
        for t, ids in zip(tickets, ([0, 1, 2, 3], [4], [10, 9, 8], [5, 6, 7, 2])):
            got = loader.fetch(t)
            assert got.dtype == torch.uint8 and tuple(got.shape) == (len(ids), 12, 20, 3)
            assert np.array_equal(got.numpy(), np.stack([frames[i] for i in ids]))
        if threads:
            with pytest.raises(ValueError):
                loader.request([0, 1, 2, 3, 4])                 # over the batch the staging buffers were sized for
            bad = loader.request([3, 99])                       # a frame that does not exist: the failure surfaces at fetch
            with pytest.raises(Exception):
                loader.fetch(bad)
        loader.close()
    # a key-frame request takes a one-frame staging buffer, not a whole batch; the ring is bounded (4 x slots un-fetched requests)
    with hostio.FrameLoader(video, "cpu", threads=2, slots=2, batch=4) as loader:
        t1 = loader.request([5])
        assert t1[2].shape[0] == 1
        assert np.array_equal(loader.fetch(t1).numpy(), frames[5][None])
        held = [loader.request([0, 1]) for _ in range(8)]
        with pytest.raises(RuntimeError):
            loader.request([2, 3])
        for t in held:
            loader.fetch(t)
    assert loader.pool is None                                  # the context manager closed the decode pool


def test_frame_writer_writes_every_frame_and_reports_failures(tmp_path):
    import torch
    from sd_animation_optical_flow_amd import hostio
    video, frames = _small_workspace(tmp_path)
    for threads in (0, 3):
        w = hostio.FrameWriter(video, "cpu", threads=threads, slots=2)             # fewer staging slots than frames: put() must recycle
        for i, f in enumerate(frames):
            w.put(i, torch.from_numpy(255 - f))
        w.close()
        assert all(np.array_equal(video.get_ai_frame(i), 255 - f) for i, f in enumerate(frames))
    w = hostio.FrameWriter(video, "cpu", threads=2, slots=2)
    w.put(0, torch.from_numpy(frames[0]))
    with pytest.raises(ValueError):
        w.put(1, torch.zeros((12, 20), dtype=torch.uint8))      # not a [H,W,3] frame: refused before a staging slot is taken
    # frames of another shape get their own staging buffers (never a broadcast into a recycled one) ...
    for i in range(4):
        w.put(2 + i, torch.full((12, 20, 3), 10 * i, dtype=torch.uint8))
    w.flush()
    w.put(1, torch.full((6, 20, 3), 7, dtype=torch.uint8))
    w.flush()
    assert sorted(w._free) == [(6, 20, 3), (12, 20, 3)]
    # ... a failure on an encoder thread surfaces at flush / close, and its staging slot comes back (no deadlock afterwards)
    real = video.put_ai_frame
    def failing(i, frame):
        if i == 3:
            raise OSError("disk full")
        return real(i, frame)
    video.put_ai_frame = failing
    for i in (3, 4, 5, 6):                                      # more puts than slots after the failure
        w.put(i, torch.from_numpy(frames[i]))
    with pytest.raises(OSError):
        w.close()
    video.put_ai_frame = real
    assert np.array_equal(video.get_ai_frame(6), frames[6])


def test_png_writer_round_trips_through_pillow():
    """`workspace.encode_png_rgb` (Sub filter + one zlib stream; what `put_ai_frame` writes) is a PNG any reader takes: Pillow decodes it
    to the same pixels at every size / compression level, including one-pixel-wide and one-pixel-high images and level 0 (stored)."""
    import io
    from PIL import Image
    from sd_animation_optical_flow_amd.workspace import encode_png_rgb
    rng = np.random.default_rng(3)
    for (h, w) in ((1, 1), (1, 7), (5, 1), (13, 17), (64, 96)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for level in (0, 1, 6):
            data = encode_png_rgb(img, level)
            assert data[:8] == b"\x89PNG\r\n\x1a\n"
            with Image.open(io.BytesIO(data)) as im:
                assert im.mode == "RGB" and im.size == (w, h)
                assert np.array_equal(np.asarray(im), img), (h, w, level)
    flat = np.full((32, 32, 3), 200, dtype=np.uint8)                                  # a flat frame: the Sub filter makes it all zeros
    assert len(encode_png_rgb(flat, 1)) < 200


def test_fast_png_decoder_agrees_with_pillow_and_falls_back(tmp_path):
    """`workspace.decode_png_rgb_fast` takes the PNGs this package writes (8-bit RGB, rows with the None / Sub filter) and returns
    None for everything else, which then goes through Pillow: the decoded pixels are Pillow's in every case -- our own writer at
    levels 0 and 1, Pillow's writer (adaptive filters: Up / Average / Paeth rows), greyscale, RGBA and a truncated file."""
    import io
    from PIL import Image
    from sd_animation_optical_flow_amd import workspace as ws
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    smooth = np.cumsum(rng.integers(0, 3, (40, 64, 3)), axis=1).astype(np.uint8)
    for a in (img, smooth):
        for level in (0, 1):
            data = ws.encode_png_rgb(a, level)
            fast = ws.decode_png_rgb_fast(data)
            assert fast is not None and np.array_equal(fast, a)
            assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), a)
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format="PNG")                 # Pillow picks filters per row
        p = tmp_path / "pil.png"
        p.write_bytes(buf.getvalue())
        assert np.array_equal(ws._read_png_bgr(str(p))[:, :, ::-1], a)          # through the fast path or the fall-back: same pixels
    for mode in ("L", "RGBA"):
        buf = io.BytesIO()
        Image.fromarray(img).convert(mode).save(buf, format="PNG")
        assert ws.decode_png_rgb_fast(buf.getvalue()) is None
    good = ws.encode_png_rgb(img, 1)
    assert ws.decode_png_rgb_fast(good[:len(good) // 2]) is None and ws.decode_png_rgb_fast(b"not a png") is None
