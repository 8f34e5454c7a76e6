"""Host side of the workspace driver: PNG frames in, PNG frames out, off the critical path.

The reference decodes and encodes on the thread that drives the GPU (`VideoData.get_raw_frame` / `put_ai_frame`,
ofgen_keyframe_inpaint.py:415-432; two `cv2.imread`s per pair at :591-592) and synchronises on every result.  At ~200
frame pairs/s per GPU a rank has ~5 ms of host time per frame in total, and a Pillow decode plus a `compress_level=1` encode of
one 512x768 frame is about that -- with eight ranks sharing one host, input decode / H2D is THE scaling limit SURVEY 8(e)
names.  Two helpers take it off the thread that enqueues kernels:

  `FrameLoader`   a thread pool decodes the PNGs of the NEXT batches straight into pinned staging buffers (Pillow releases the
                  GIL while it inflates), one `non_blocking` H2D copy per batch runs on a copy stream beside the current
                  batch's kernels, and the compute stream waits on the copy's event only.
  `FrameWriter`   a rendered frame is copied D2H into a pinned buffer on a side stream (after the kernels that produced it);
                  a worker waits for that copy's event, encodes the PNG and writes it.  `flush()` joins everything and
                  re-raises the first failure -- the same contract as `ofgen.PDCNetAux`'s `.npy` writer.

Neither touches the device arithmetic: the pipeline's output is byte-identical with them on or off (`ClipPipeline(io_threads=0)`
runs everything inline, which is also what CPU-device test pipelines get for the copies).
"""
from __future__ import annotations

import threading
import time
from collections import deque
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Deque, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def shared_pool(threads: int) -> Optional[ThreadPoolExecutor]:
    """One worker pool for a rank's `FrameLoader` AND `FrameWriter`: 2 x `threads` workers (the two private pools it replaces had
    `threads` each).  The caller shuts it down after both helpers are closed."""
    return ThreadPoolExecutor(2 * int(threads), thread_name_prefix="ofx-io") if int(threads) > 0 else None


class FrameLoader:
    """Prefetching reader of `video.get_raw_frame(i)` batches.

        t = loader.request(ids)        # starts decoding; returns at once
        ...                            # enqueue GPU work of the previous batch here
        frames = loader.fetch(t)       # u8 [len(ids),H,W,3] on `device`, valid on the CURRENT stream

    `threads=0`: no pool, no pinned memory -- `fetch` decodes inline and uploads synchronously (the round-3 behaviour)."""

    def __init__(self, video, device, threads: int = 4, slots: int = 3, batch: int = 64, pool: Optional[ThreadPoolExecutor] = None):
        """`pool`: a worker pool shared with a `FrameWriter` (`shared_pool`): a rank's decoders and encoders are busy at opposite ends
        of its share, so one pool of 2 x threads halves the decode in front of the first kernel and the encode behind the last."""
        self.video, self.device = video, torch.device(device)
        self.threads = max(0, int(threads))
        self.cuda = self.device.type == "cuda"
        H, W = video.size_hw
        self.shape = (H, W, 3)
        self.batch = int(batch)
        self._own_pool = pool is None
        self.pool = (pool if pool is not None else ThreadPoolExecutor(self.threads, thread_name_prefix="ofx-decode")) if self.threads else None
        # two rings of pinned staging buffers: full batches, and single frames (a key-frame request must not pin a whole batch:
        # 75 MB at 64 x 512x768)
        self._free: Dict[int, Deque[torch.Tensor]] = {1: deque(), self.batch: deque()}
        self._slots = max(2, int(slots))
        self._made = {1: 0, self.batch: 0}
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.cuda and self.threads else None
        self._inflight: List[Tuple[torch.Tensor, "torch.cuda.Event"]] = []

    def _staging(self, n: int) -> torch.Tensor:
        """A pinned [cap,H,W,3] buffer (cap = 1 for single frames, else `batch`) from a ring of `slots` per size: a buffer is reused
        only after the H2D copy that read it is done.  A full ring waits for its oldest copy; it grows past `slots` only while every
        buffer sits in a request that was not fetched yet (the caller asked further ahead than it said it would), and never past
        4 x `slots` -- beyond that the request / fetch protocol is broken and pinned memory would grow without bound."""
        cap = 1 if n == 1 and self.batch > 1 else self.batch
        free = self._free[cap]
        still = []
        for buf, ev in self._inflight:                       # reap the copies that have finished
            if ev.query():
                self._free[buf.shape[0]].append(buf)
            else:
                still.append((buf, ev))
        self._inflight = still
        while not free and self._made[cap] >= self._slots:
            k = next((j for j, (b, _) in enumerate(self._inflight) if b.shape[0] == cap), None)
            if k is None:
                if self._made[cap] >= 4 * self._slots:
                    raise RuntimeError(f"FrameLoader: {self._made[cap]} staging buffers of {cap} frame(s) are all held by requests that "
                                       f"were never fetched (slots={self._slots})")
                break
            buf, ev = self._inflight.pop(k)                  # ring full: wait for the oldest copy out of a buffer of this size
            ev.synchronize()
            free.append(buf)
        if free:
            return free.popleft()
        self._made[cap] += 1
        # allocated pinned, not `.pin_memory()`-ed: that is a COPY of the fresh buffer, and a CPU copy of this size runs as an
        # OpenMP region over every core torch sees (192 spinning threads on the 16-core GPU boxes: 8 of a run's 11 CPU-seconds)
        return torch.empty((cap, *self.shape), dtype=torch.uint8, pin_memory=self.cuda)

    def request(self, ids: Sequence[int]):
        ids = [int(i) for i in ids]
        if not self.pool:
            return ("inline", ids, None, None)
        if len(ids) > self.batch:
            raise ValueError(f"a request holds at most {self.batch} frames")
        buf = self._staging(len(ids))
        view = buf.numpy()

        def job(k: int, i: int) -> None:
            np.copyto(view[k], self.video.get_raw_frame(i))
        futs = [self.pool.submit(job, k, i) for k, i in enumerate(ids)]
        return ("async", ids, buf, futs)

    def fetch(self, ticket) -> torch.Tensor:
        kind, ids, buf, futs = ticket
        if kind == "inline":
            return torch.from_numpy(np.stack([self.video.get_raw_frame(i) for i in ids])).to(self.device)
        for f in futs:
            f.result()                                       # decode errors surface here
        n = len(ids)
        if not self.cuda:
            out = buf[:n].clone()
            self._free[buf.shape[0]].append(buf)
            return out
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._copy_stream):
            # allocated ON the copy stream: the caching allocator then orders the block's reuse against this stream's copies, and
            # record_stream below keeps it alive for the consumer's kernels -- the copy never has to wait for compute already enqueued
            out = torch.empty((n, *self.shape), dtype=torch.uint8, device=self.device)
            out.copy_(buf[:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        out.record_stream(cur)
        cur.wait_event(ev)                                   # the compute stream waits for this copy only
        self._inflight.append((buf, ev))
        return out

    def close(self) -> None:
        if self.pool and self._own_pool:
            self.pool.shutdown(wait=True, cancel_futures=True)
        self.pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc) -> None:
        self.close()


class FrameWriter:
    """Asynchronous `video.put_ai_frame(i, frame)` for device-resident frames.  `put` returns as soon as the D2H copy is enqueued
    (it blocks only when all `slots` staging buffers are still waiting for their encoder); `flush()` waits for every file."""

    def __init__(self, video, device, threads: int = 4, slots: int = 32, pool: Optional[ThreadPoolExecutor] = None):
        self.video, self.device = video, torch.device(device)
        self.threads = max(0, int(threads))
        self.cuda = self.device.type == "cuda"
        self._own_pool = pool is None
        self.pool = (pool if pool is not None else ThreadPoolExecutor(self.threads, thread_name_prefix="ofx-encode")) if self.threads else None
        self._stream = torch.cuda.Stream(device=self.device) if self.cuda and self.threads else None
        self._sem = threading.Semaphore(max(2, int(slots)))
        self._free: Dict[Tuple[int, ...], Deque[torch.Tensor]] = {}      # pinned staging buffers, by frame shape
        self._lock = threading.Lock()
        self._futs: List[Future] = []

    def _buffer(self, like: torch.Tensor) -> torch.Tensor:
        shape = tuple(like.shape)
        with self._lock:
            q = self._free.get(shape)
            if q:
                return q.popleft()
        return torch.empty(shape, dtype=torch.uint8, pin_memory=self.cuda)      # (pinned at birth: `.pin_memory()` would copy)

    def _recycle(self, buf: torch.Tensor) -> None:
        with self._lock:
            self._free.setdefault(tuple(buf.shape), deque()).append(buf)

    def put(self, index: int, frame: torch.Tensor) -> None:
        if not self.pool:
            self.video.put_ai_frame(int(index), frame.cpu().numpy())
            return
        if frame.dtype != torch.uint8 or frame.dim() != 3:
            raise ValueError(f"frame {index}: expected a uint8 [H,W,C] tensor, got {frame.dtype} {tuple(frame.shape)}")
        self._sem.acquire()
        buf = ev = None
        try:
            buf = self._buffer(frame)
            if self.cuda and frame.is_cuda:
                cur = torch.cuda.current_stream(self.device)
                self._stream.wait_stream(cur)                # after the kernels that rendered the frame
                with torch.cuda.stream(self._stream):
                    buf.copy_(frame, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                frame.record_stream(self._stream)
            else:
                buf.copy_(frame)

            def job() -> None:
                try:
                    if ev is not None:
                        # hipEventSynchronize spins (also for a "blocking" event on this stack: 2.4 of an encoder pool's 3.9
                        # CPU-seconds per 64 frames were this wait); a sleeping poll gives the core to the decoders
                        while not ev.query():
                            time.sleep(0.0005)
                    self.video.put_ai_frame(int(index), buf.numpy())
                finally:
                    self._recycle(buf)
                    self._sem.release()
            fut = self.pool.submit(job)
        except BaseException:
            # nothing was handed to a worker: give the staging slot (and the buffer, if one was taken) back -- but not while the
            # D2H copy that was already enqueued (pool.submit can fail after it) may still be writing into that buffer
            if buf is not None:
                if ev is not None:
                    ev.synchronize()
                self._recycle(buf)
            self._sem.release()
            raise
        self._futs.append(fut)
        if len(self._futs) > 256:                            # keep the list short; failures still surface (result() re-raises)
            done, self._futs = self._futs[:128], self._futs[128:]
            for f in done:
                f.result()

    def flush(self) -> None:
        futs, self._futs = self._futs, []
        for f in futs:
            f.result()

    def close(self) -> None:
        try:
            self.flush()
        finally:
            if self.pool and self._own_pool:
                self.pool.shutdown(wait=True)
            self.pool = None
