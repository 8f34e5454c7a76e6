"""The reference's on-disk workspace, readable and writable from this package.

`ofgen_keyframe_inpaint.py:372-547` keeps a clip as a directory

    <workspace>/raw-frames/{n:05d}.png     decoded, resized input frames (BGR in memory, ordinary RGB PNG on disk)
    <workspace>/ai-frames/{n:05d}.png      rendered frames; `generated(n)` is "the file exists"
    <workspace>/pdcnet/{s:05d}-{t:05d}.npy flow + confidence of a pair, float32 [H,W,3] (written by ofgen.PDCNetAux)
    <workspace>/crossattn/{n:05d}.bin      pickled key/value history of the SD stage
    <workspace>/seed/

`VideoData` here opens or fills such a directory and serves frames with the reference's method names, so that the
flow / warp / mask path (`ofgen.PDCNetAux`, `ofgen.keyframe_conv`, `clip.FrameSynthesizer`) runs against a workspace
the reference produced, and the reference can continue from one produced here.  Differences, all forced by what is in
the image: OpenCV is absent, so PNGs go through Pillow (same pixels: PNG is lossless, cv2 writes BGR arrays as RGB
files), and video *decoding* (`cv2.VideoCapture`) stays with the caller -- `VideoData` takes an iterable of decoded BGR
frames instead of a path to a movie.  Frames are expected at their final size: the reference's
`cv2.resize(..., INTER_AREA)` at extraction time (:411) is I/O-side preparation, not part of the path.

`VideoFrameIndices` (:483-541) is the sorted index set with the window enumeration `KeyframeConv` walks
(`conv_indices`: `indices[idx : idx + kernel_size][0::dilation]`, idx += stride).
"""
from __future__ import annotations

import glob
import os
import pickle
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

_SUBDIRS = ("raw-frames", "ai-frames", "pdcnet", "crossattn", "seed")


def decode_png_rgb_fast(data: bytes) -> Optional[np.ndarray]:
    """RGB pixels [H,W,3] of an 8-bit truecolour, non-interlaced PNG whose rows all carry filter type 0 (None) or 1 (Sub) -- what
    `encode_png_rgb` writes, i.e. every `ai-frames/` file and the `raw-frames/` of a workspace extracted here: one `zlib.decompress`
    and one running sum per row (uint8 arithmetic wraps modulo 256, as the PNG filter does), both outside the GIL.  3x faster than
    Pillow's decoder (2.7 against 8 ms per 512x768 frame on the GPU boxes' cores).  Anything else -- other colour types, bit depths,
    interlacing, rows with the Up / Average / Paeth filters (what libpng's heuristics may choose) -- returns None: the caller falls
    back to Pillow."""
    import struct
    import zlib
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        return None
    pos, idat, W, H = 8, [], 0, 0
    try:
        while pos + 8 <= len(data):
            n, tag = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            if tag == b"IHDR":
                W, H, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body[:13])
                if (depth, ctype, interlace) != (8, 2, 0):
                    return None
            elif tag == b"IDAT":
                idat.append(body)
            elif tag == b"IEND":
                break
            pos += 12 + n
        if not idat or W <= 0 or H <= 0:
            return None
        raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
        if raw.size != H * (1 + W * 3):
            return None
        raw = raw.reshape(H, 1 + W * 3)
        ft = raw[:, 0]
        if int(ft.max()) > 1:
            return None
        px = raw[:, 1:].reshape(H, W, 3)
        out = np.cumsum(px, axis=1, dtype=np.uint8)                # Sub: byte + the byte one pixel to the left, modulo 256
        if int(ft.min()) == 0:
            out[ft == 0] = px[ft == 0]
        return out
    except (struct.error, zlib.error, ValueError):
        return None


def _read_png_bgr(path: str) -> np.ndarray:
    with open(path, "rb") as fp:
        data = fp.read()
    rgb = decode_png_rgb_fast(data)
    if rgb is None:
        import io
        from PIL import Image
        with Image.open(io.BytesIO(data)) as im:
            rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])                 # what cv2.imread returns


def encode_png_rgb(rgb: np.ndarray, level: int = 1) -> bytes:
    """An 8-bit RGB PNG of `rgb` (uint8 [H,W,3]): every row with the Sub filter (one vectorised numpy subtraction for the whole
    image), one zlib stream.  Same pixels as any PNG writer (lossless); what differs from Pillow's encoder is the time -- it
    tries all five filters on every row, 2.5x the cost of the deflate itself on a 512x768 frame -- and that `zlib.compress`
    releases the GIL, so the writer threads of `hostio.FrameWriter` really run side by side.  level 0 = stored (no deflate).
    The deflate runs with the Z_RLE strategy -- the default of `cv2.imwrite` for PNGs (IMWRITE_PNG_STRATEGY_RLE, compression level 1),
    i.e. of the reference's own writer (ofgen_keyframe_inpaint.py:432): on Sub-filtered rows it is 3x faster than the default strategy
    (10 against 33 ms per 512x768 frame on the build box) and no larger (0.80 against 0.86 MB on the bench clip)."""
    import struct
    import zlib
    H, W, C = rgb.shape
    raw = np.empty((H, 1 + W * C), dtype=np.uint8)
    raw[:, 0] = 1                                               # filter type 1: Sub (byte minus the byte one pixel to the left)
    flat = rgb.reshape(H, W * C)
    raw[:, 1:1 + C] = flat[:, :C]
    np.subtract(flat[:, C:], flat[:, :-C], out=raw[:, 1 + C:])  # uint8 arithmetic wraps modulo 256, as the PNG filter does

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    def deflate(data: bytes) -> bytes:
        if int(level) == 0:
            return zlib.compress(data, 0)
        c = zlib.compressobj(int(level), zlib.DEFLATED, 15, 8, zlib.Z_RLE)
        return c.compress(data) + c.flush()
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", deflate(raw.tobytes())) + chunk(b"IEND", b""))


def _write_png_bgr(path: str, frame_bgr: np.ndarray, level: int = 1) -> None:
    a = np.asarray(frame_bgr)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("frames are uint8 [H,W,3] (BGR)")
    data = encode_png_rgb(np.ascontiguousarray(a[:, :, ::-1]), level)
    with open(path, "wb") as fp:
        fp.write(data)


class VideoData:
    """`VideoData` of ofgen_keyframe_inpaint.py:372-481.

    frames: None (open an existing workspace) or an iterable of decoded BGR uint8 frames [H,W,3] of size `size`
    (w, h); every `keep_every`-th one is kept, like the extraction loop (:399-411), up to `max_len_sec` seconds.
    `num_frames` is the number of PNGs in `raw-frames/` on both paths; the reference's extraction path leaves it at the
    LAST INDEX (`self.num_frames = ctr_valid`, :414, one less than the count its own re-open path reports, :376-381) --
    the count is the consistent reading and the only one under which every extracted frame is reachable."""

    def __init__(self, frames: Optional[Iterable[np.ndarray]], size: Tuple[int, int], workspace_dir: str, keep_every: int = 1,
                 max_len_sec: int = -1, fps: float = 30.0) -> None:
        self.workspace_dir = workspace_dir
        self.size = (int(size[0]), int(size[1]))
        self.fps = float(fps) / max(1, int(keep_every))
        self.kv_hist_map = {}
        for d in _SUBDIRS:
            os.makedirs(os.path.join(workspace_dir, d), exist_ok=True)
        raw = os.path.join(workspace_dir, "raw-frames")
        if frames is None:
            self.num_frames = len(glob.glob(os.path.join(raw, "*.png")))
            return
        target = float("inf") if max_len_sec == -1 else self.fps * max_len_sec
        kept = -1
        for ctr, frame in enumerate(frames):
            if ctr % max(1, int(keep_every)) != 0:
                continue
            kept += 1
            dst = os.path.join(raw, f"{kept:05d}.png")
            if not os.path.exists(dst):
                f = np.asarray(frame)
                if (f.shape[1], f.shape[0]) != self.size:
                    raise ValueError(f"frame {ctr} is {f.shape[1]}x{f.shape[0]}, workspace size is {self.size[0]}x{self.size[1]} "
                                     "(resize before extraction: INTER_AREA in the reference)")
                _write_png_bgr(dst, f)
            if kept >= target:
                break
        self.num_frames = kept + 1

    # ---- frames ------------------------------------------------------------------------------------
    @property
    def size_hw(self) -> Tuple[int, int]:
        return (self.size[1], self.size[0])

    def _path(self, kind: str, n: int, ext: str = "png") -> str:
        return os.path.join(self.workspace_dir, kind, f"{int(n):05d}.{ext}")

    def get_raw_frame(self, n: int) -> np.ndarray:
        assert n < self.num_frames
        return _read_png_bgr(self._path("raw-frames", n))

    def get_ai_frame(self, n: int) -> Optional[np.ndarray]:
        assert n < self.num_frames
        p = self._path("ai-frames", n)
        return _read_png_bgr(p) if os.path.exists(p) else None

    def generated(self, n: int) -> bool:
        return os.path.exists(self._path("ai-frames", n))

    png_level = 1          # zlib level of the PNGs written here (0 = stored: fastest, 1.2 MB per 512x768 frame)

    def put_ai_frame(self, n: int, frame: np.ndarray) -> None:
        assert n < self.num_frames
        _write_png_bgr(self._path("ai-frames", n), frame, self.png_level)

    def raw_frames_device(self, indices: Sequence[int], device="cuda", rgb: bool = True):
        """The frames `indices` as ONE uint8 tensor [n,H,W,3] on the device (RGB by default: what `calc_batch` and
        `calc_pairs` take) -- a single upload for a whole KeyframeConv window."""
        import torch
        arr = np.stack([self.get_raw_frame(i)[:, :, ::-1] if rgb else self.get_raw_frame(i) for i in indices])
        return torch.from_numpy(np.ascontiguousarray(arr)).to(device)

    # ---- key frames (:443-469) -----------------------------------------------------------------------
    def key_frames(self, th: float = 48, min_gap: int = -1, max_gap: int = -1, batch: int = 16, device="cuda") -> Iterator[Tuple[np.ndarray, int]]:
        """Yields (frame, index) for the frames the detector promotes to key frames.  Note that -- unlike
        `frame_generator` -- the reference never advances `gap` here (it stays 0, :455-468), so the threshold is the
        constant `th`; reproduced as is."""
        from . import keyframes
        _, mx = keyframes.gaps(self.fps, min_gap, max_gap)
        key_edges, ksize = None, None
        for i0 in range(0, self.num_frames, batch):
            ids = list(range(i0, min(self.num_frames, i0 + batch)))
            frames = [self.get_raw_frame(i) for i in ids]
            if ksize is None:
                ksize = keyframes.estimated_kernel_size(frames[0].shape[1], frames[0].shape[0])
            edges = keyframes.detect_edges(np.stack(frames), ksize, device=device)
            for j, i in enumerate(ids):
                if key_edges is None:
                    key_edges = edges[j]
                    yield frames[j], i
                    continue
                if th * (mx - 0) / mx < keyframes.mean_pixel_distance(edges[j], key_edges):
                    key_edges = edges[j]
                    yield frames[j], i

    # ---- SD-stage key/value history (:471-481): opaque blobs, kept for workspace compatibility ---------------
    def put_kv(self, frame_idx: int, kv) -> None:
        with open(self._path("crossattn", frame_idx, "bin"), "wb") as fp:
            pickle.dump(kv, fp)

    def get_kv(self, frame_idx: int):
        with open(self._path("crossattn", frame_idx, "bin"), "rb") as fp:
            return pickle.load(fp)

    def remove_kv(self, frame_idx: int) -> None:
        os.remove(self._path("crossattn", frame_idx, "bin"))


class VideoFrameIndices:
    """A sorted set of frame indices into a VideoData (:483-541)."""

    def __init__(self, indices: Iterable[int] = ()) -> None:
        self.indices: List[int] = sorted(set(int(i) for i in indices))

    @staticmethod
    def from_n(n: int) -> "VideoFrameIndices":
        return VideoFrameIndices(range(n))

    def conv_indices(self, kernel_size: int = 17, stride: int = 8, dilation: int = 1) -> Iterator["VideoFrameIndices"]:
        """Sliding windows: `kernel_size` consecutive members starting every `stride` members, thinned to every
        `dilation`-th (the tail windows are shorter; the last may hold a single index)."""
        for start in range(0, len(self.indices), stride):
            yield VideoFrameIndices(self.indices[start:start + kernel_size][::dilation])

    def remove(self, other: "VideoFrameIndices") -> None:
        self.indices = sorted(set(self.indices) - set(other.indices))

    def add(self, other: Union[int, "VideoFrameIndices"]) -> None:
        extra = [other] if isinstance(other, int) else other.indices
        self.indices = sorted(set(self.indices) | set(extra))

    def adjacent_frames(self, idx: int, n: int) -> "VideoFrameIndices":
        """The run of `n` consecutive members closest to `idx` in summed distance; first best run wins, and -- as in the
        reference's `range(0, len - n)` -- the very last run is never a candidate."""
        if len(self) <= n:
            return self
        best, best_dist = None, None
        for i in range(0, len(self) - n):
            run = self.indices[i:i + n]
            dist = int(np.sum(np.abs(np.asarray(run) - idx)))
            if best_dist is None or dist < best_dist:
                best, best_dist = run, dist
        return VideoFrameIndices(best)

    def __len__(self) -> int:
        return len(self.indices)

    def __iter__(self):
        return iter(self.indices)

    def __repr__(self) -> str:
        return f"VideoFrameIndices({self.indices})"
