"""CPU: the N > 1 path (frame-parallel sharding + key-frame broadcast) with world_size-2 `gloo`."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sd_animation_optical_flow_amd import clip


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(clip.shard_range(total, r, world))
            assert seen == list(range(total))
            sizes = [len(clip.shard_range(total, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert list(clip.shard_range(512, 3, 8)) == list(range(192, 256))      # BASELINE configs[3]: 64 frames per GPU
    with pytest.raises(ValueError):
        clip.shard_range(4, 2, 2)


def test_single_process_is_a_noop_broadcast():
    k = torch.arange(12, dtype=torch.uint8).reshape(2, 2, 3)
    clip.broadcast_keyframe([k], src=0)
    res = clip.process_clip(torch.zeros((5, 2, 2, 3), dtype=torch.uint8), k, k.clone(),
                            lambda fr, a, b: (fr.float(), fr, fr[..., 0]), batch_size=2)
    assert res.frame_indices == [0, 1, 2, 3, 4] and [t.shape[0] for t in res.flow] == [2, 2, 1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, H, W = 9, 4, 6
        frames = (torch.arange(T * H * W * 3) % 251).to(torch.uint8).reshape(T, H, W, 3)
        # only the source rank holds the rendered key frame; the others hold garbage until the broadcast
        key_raw = torch.full((H, W, 3), 11 if rank == 1 else 0, dtype=torch.uint8)
        key_ai = torch.full((H, W, 3), 200 if rank == 1 else 0, dtype=torch.uint8)
        calls = []

        def step(fr, kr, ka):
            calls.append(fr.shape[0])
            # a stand-in for flow/warp/mask that depends on every input
            flow = fr.float().mean(-1, keepdim=True).repeat(1, 1, 1, 2) + kr.float().mean()
            return flow, (fr // 2 + ka // 2), (fr[..., 0] > 100).to(torch.uint8) * 255

        res = clip.process_clip(frames, key_raw, key_ai, step, batch_size=2, key_src=1)
        assert int(key_raw[0, 0, 0]) == 11 and int(key_ai[0, 0, 0]) == 200           # broadcast reached every rank
        mine = list(clip.shard_range(T, rank, world))
        assert res.frame_indices == mine and sum(calls) == len(mine)
        torch.save({"idx": res.frame_indices, "warped": torch.cat(res.warped), "flow": torch.cat(res.flow)},
                   os.path.join(outdir, f"r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_clip(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    idx = parts[0]["idx"] + parts[1]["idx"]
    assert idx == list(range(9))                                                      # every frame exactly once
    T, H, W = 9, 4, 6
    frames = (torch.arange(T * H * W * 3) % 251).to(torch.uint8).reshape(T, H, W, 3)
    warped = torch.cat([p["warped"] for p in parts])
    assert torch.equal(warped, frames // 2 + 100)                                     # same result as one rank would give
    flow = torch.cat([p["flow"] for p in parts])
    assert torch.allclose(flow[..., 0], frames.float().mean(-1) + 11.0)


# ------------------------------------------------------------------------------------------------
# bench.py's own rank plumbing (self-launch, broadcast, barrier, MAX all-reduce, ONE JSON line)
# ------------------------------------------------------------------------------------------------
def _run_bench(argv, env_extra=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=root)


def test_bench_gpus2_self_launches_two_ranks_gloo():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself; with the hidden CPU stub
    step (gloo) the whole dist path runs here: n_gpus and the collective library's own rank count both say 2."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["ms_per_step"] >= 10.0  # the stub sleeps 10 ms per step


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """No GPU here: `--gpus 2` must exit non-zero with a clear message instead of measuring one (or zero) devices."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], env_extra={"HIP_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus 2" in r.stderr
    # a launcher that started a different number of ranks than --gpus names is refused as well
    r = _run_bench(["--gpus", "4", "--stub-step"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


# ------------------------------------------------------------------------------------------------
# round 3: the rank-aware clip driver over a workspace (pipeline.ClipPipeline + clip.plan_segments)
# ------------------------------------------------------------------------------------------------
def test_plan_segments_balances_content_dependent_segments():
    def flags_of(lengths):                       # lengths = frames following each key frame
        f = []
        for n in lengths:
            f += [True] + [False] * n
        return f
    for lengths, world in (([64] * 8, 8), ([100, 3, 3, 3, 3, 3, 3, 10], 4), ([5], 2), ([0, 0, 7], 2), ([1, 200, 1], 8), ([3, 2], 1)):
        flags = flags_of(lengths)
        plan = clip.plan_segments(flags, world)
        assert [p.key for p in plan] == [i for i, k in enumerate(flags) if k]
        seen = sorted(t for p in plan for per in p.frames for t in per)
        assert seen == [i for i, k in enumerate(flags) if not k]                      # every frame exactly once
        load = [sum(len(p.frames[r]) for p in plan) for r in range(world)]
        assert max(load) <= -(-sum(lengths) // world)                                 # nobody above ceil(total / world)
        for p in plan:
            assert 0 <= p.owner < world
            if any(p.frames):
                assert p.frames[p.owner], "the owner of a key frame works on its segment"
            for per in p.frames:                                                      # pieces are contiguous runs
                assert per == list(range(per[0], per[0] + len(per))) if per else True
    # BASELINE configs[3]: 8 key frames x 64 frames over 8 ranks = one whole segment per rank, no broadcast at all
    plan = clip.plan_segments(flags_of([64] * 8), 8)
    assert sorted(p.owner for p in plan) == list(range(8)) and not any(p.needs_broadcast for p in plan)
    # one long segment has to be cut: it is broadcast, the short ones are not
    plan = clip.plan_segments(flags_of([100, 3, 3]), 2)
    assert plan[0].needs_broadcast and len(plan[0].ranks) == 2
    with pytest.raises(ValueError):
        clip.plan_segments([False, True], 2)


class _StubPipeline:
    """ClipPipeline with the compute step replaced by CPU arithmetic that depends on every input (frame, key frame, rendered
    key frame): the rank logic, the workspace I/O and the broadcast are the real ones."""

    @staticmethod
    def make(device="cpu", flags=None, batch=3, edge_batch=16):
        from sd_animation_optical_flow_amd import pipeline
        rank0_flags = list(_FLAGS if flags is None else flags)

        class P(pipeline.ClipPipeline):
            def process_batch(self, key_raw, key_ai, raws, ids, key_index):
                out = []
                for k, t in enumerate(ids):
                    warped = (raws[k] // 2 + key_ai // 4 + key_raw // 8).to(torch.uint8)
                    mask = ((raws[k][..., 0] > 128).to(torch.uint8) * 255)
                    out.append(pipeline.FramePacket(t, key_index, torch.zeros(1), torch.zeros(1), warped, mask, {}))
                return out
            def key_frame_flags(self, video, th=8.5):
                # what each rank's OWN detector would say: only rank 0's answer is usable -- `shared_flags` must broadcast it
                # (a rank acting on its own list here would plan other segments and the run would hang or write wrong frames)
                from sd_animation_optical_flow_amd import clip
                rank, _ = clip.dist_info()
                return list(rank0_flags) if rank == 0 else [True] + [False] * (video.num_frames - 1)
        render = lambda pkt, raw: torch.where(pkt.mask[..., None] > 0, raw, pkt.warped)
        return P(algo=None, render=render, render_key=lambda raw: 255 - raw, batch=batch, device=torch.device(device),
                 edge_batch=edge_batch, io_threads=2)


def _make_workspace(path, n=23, H=16, W=24):
    import numpy as np
    from sd_animation_optical_flow_amd.workspace import VideoData
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(n)]
    return VideoData(frames, (W, H), path), frames


_FLAGS = [True] + [False] * 11 + [True] + [False] * 2 + [True, True] + [False] * 6      # 23 frames: segments of 11, 2, 0, 6


def _pipeline_worker(rank, world, port, ws):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sd_animation_optical_flow_amd.workspace import VideoData
        video = VideoData(None, (24, 16), ws)
        keys = _StubPipeline.make().run(video)          # flags=None: rank 0's key-frame decisions reach every rank by broadcast
        torch.save(keys, os.path.join(ws, f"keys_r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_pipeline_over_a_workspace(tmp_path):
    """The rank-aware `ClipPipeline.run` over a `VideoData` workspace with world_size 2 (gloo): every frame rendered exactly
    once, by the rank the plan names; the rendered key frame of the segment that had to be cut reaches the other rank through
    the broadcast (its pixels enter every warped frame); the result equals the single-process run byte for byte."""
    import numpy as np
    from sd_animation_optical_flow_amd.workspace import VideoData
    one, frames = _make_workspace(str(tmp_path / "one"))
    keys1 = _StubPipeline.make().run(one, _FLAGS)
    assert keys1 == [0, 12, 15, 16]
    two, _ = _make_workspace(str(tmp_path / "two"))
    world = 2
    mp.spawn(_pipeline_worker, args=(world, _free_port(), str(tmp_path / "two")), nprocs=world, join=True)
    plan = clip.plan_segments(_FLAGS, world)
    assert any(p.needs_broadcast for p in plan)                                       # the 11-frame segment is cut (target 10)
    keys = [torch.load(tmp_path / "two" / f"keys_r{r}.pt") for r in range(world)]
    assert sorted(keys[0] + keys[1]) == keys1 and not set(keys[0]) & set(keys[1])     # each key frame rendered by ONE rank
    for r in range(world):
        assert keys[r] == [p.key for p in plan if p.owner == r]
    a, b = VideoData(None, (24, 16), str(tmp_path / "one")), VideoData(None, (24, 16), str(tmp_path / "two"))
    for i in range(len(_FLAGS)):
        assert b.generated(i), i
        assert np.array_equal(a.get_ai_frame(i), b.get_ai_frame(i)), i


# ------------------------------------------------------------------------------------------------
# round 5: the broadcast ordering of `ClipPipeline.packets` with EIGHT processes (BASELINE configs[3]'s world size), uneven plans,
# several cut segments, ranks that hold no frame of a cut segment (they still join its broadcast), ranks with no frame at all and
# two adjacent key frames.  Every collective carries a timeout: a mis-ordered broadcast fails the test instead of hanging it.
# ------------------------------------------------------------------------------------------------
def _flags_of(lengths):
    f = []
    for n in lengths:
        f += [True] + [False] * n
    return f


_FLAGS8_A = _flags_of([21, 0, 13, 1, 1])      # 41 frames, 36 to warp: target 5 per rank -> the 21- and the 13-frame segments are cut
_FLAGS8_B = _flags_of([3, 0, 2])              # 8 frames, 5 to warp on 8 ranks: both segments cut one frame per rank, three ranks idle


def _pipeline8_worker(rank, world, port, ws_a, ws_b):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        from sd_animation_optical_flow_amd.workspace import VideoData
        va = VideoData(None, (24, 16), ws_a)
        # flags=None: rank 0's detector decisions reach every rank through `shared_flags`' broadcast; edge_batch=1 cuts the first and
        # the last batch of every rank's share
        keys_a = _StubPipeline.make(flags=_FLAGS8_A, batch=2, edge_batch=1).run(va)
        vb = VideoData(None, (24, 16), ws_b)
        keys_b = _StubPipeline.make(batch=2, edge_batch=1).run(vb, list(_FLAGS8_B))     # a second run on the same group: no state left behind
        torch.save((keys_a, keys_b), os.path.join(ws_a, f"keys_r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_pipeline_with_cut_segments_and_idle_ranks(tmp_path):
    import numpy as np
    from sd_animation_optical_flow_amd.workspace import VideoData
    world = 8
    plan_a, plan_b = clip.plan_segments(_FLAGS8_A, world), clip.plan_segments(_FLAGS8_B, world)
    assert sum(p.needs_broadcast for p in plan_a) >= 2 and sum(p.needs_broadcast for p in plan_b) == 2       # >= 2 cut segments each
    assert any(not p.frames[r] for p in plan_a if p.needs_broadcast for r in range(world))                  # bystanders of a broadcast
    assert sum(1 for r in range(world) if not any(p.frames[r] for p in plan_b)) == 3                        # ranks with no frame at all
    assert any(not any(p.frames) for p in plan_a)                                                           # a key frame nobody warps from
    one_a, _ = _make_workspace(str(tmp_path / "one_a"), n=len(_FLAGS8_A))
    one_b, _ = _make_workspace(str(tmp_path / "one_b"), n=len(_FLAGS8_B))
    k1a = _StubPipeline.make(batch=2, edge_batch=1).run(one_a, list(_FLAGS8_A))
    k1b = _StubPipeline.make(batch=2, edge_batch=1).run(one_b, list(_FLAGS8_B))
    _make_workspace(str(tmp_path / "a"), n=len(_FLAGS8_A))
    _make_workspace(str(tmp_path / "b"), n=len(_FLAGS8_B))
    mp.spawn(_pipeline8_worker, args=(world, _free_port(), str(tmp_path / "a"), str(tmp_path / "b")), nprocs=world, join=True)
    keys = [torch.load(tmp_path / "a" / f"keys_r{r}.pt") for r in range(world)]
    for which, plan, k1 in ((0, plan_a, k1a), (1, plan_b, k1b)):
        got = [k[which] for k in keys]
        assert sorted(sum(got, [])) == k1                                             # each key frame rendered by exactly one rank ...
        for r in range(world):
            assert got[r] == [p.key for p in plan if p.owner == r]                    # ... the one the plan names
    for single, multi, n in (("one_a", "a", len(_FLAGS8_A)), ("one_b", "b", len(_FLAGS8_B))):
        x, y = VideoData(None, (24, 16), str(tmp_path / single)), VideoData(None, (24, 16), str(tmp_path / multi))
        for i in range(n):
            assert y.generated(i), (multi, i)
            assert np.array_equal(x.get_ai_frame(i), y.get_ai_frame(i)), (multi, i)   # byte-identical to the single-process run


def test_edge_batches_and_io_thread_default(monkeypatch):
    from sd_animation_optical_flow_amd import pipeline
    sizes = lambda units: [w if not isinstance(w, list) else len(w) for _, w in units]
    one = [("s", "key"), ("s", list(range(64)))]                                     # configs[3]: a rank whose share is ONE 64-frame batch
    assert sizes(pipeline.split_edge_batches(one, 16)) == ["key", 16, 32, 16]
    assert sizes(pipeline.split_edge_batches(one, 0)) == ["key", 64]
    assert sizes(pipeline.split_edge_batches([("s", "key"), ("s", list(range(20)))], 16)) == ["key", 16, 4]
    many = [("s", "key"), ("s", list(range(64))), ("s", list(range(64, 128))), ("t", None), ("t", list(range(200, 240)))]
    cut = pipeline.split_edge_batches(many, 16)
    assert sizes(cut) == ["key", 16, 48, 64, None, 24, 16]
    assert [t for _, w in cut if isinstance(w, list) for t in w] == [t for _, w in many if isinstance(w, list) for t in w]   # order kept
    assert pipeline.split_edge_batches([("s", None)], 16) == [("s", None)]
    # pools sized for the ranks that share the host: min(6, cores // local ranks // 2), at least one
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert pipeline.default_io_threads() == 6
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)), raising=False)
    assert pipeline.default_io_threads() == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert pipeline.default_io_threads() == 6
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    assert pipeline.default_io_threads() == 4


def test_pipeline_run_failures_and_collective_guard(tmp_path, monkeypatch):
    """`ClipPipeline.run`: a failing render hook surfaces as ITSELF (the writer's flush error must not replace it, its threads are
    joined either way); under the `nccl` backend a pipeline built on a CPU device is refused before its first collective."""
    import numpy as np
    from sd_animation_optical_flow_amd import pipeline
    video, _ = _make_workspace(str(tmp_path / "ws"), n=9)
    flags = [True] + [False] * 8
    pipe = _StubPipeline.make(batch=3, edge_batch=1)

    class Boom(RuntimeError):
        pass
    seen = []
    good = pipe.render

    def render(pkt, raw):
        seen.append(pkt.index)
        if pkt.index == 5:
            raise Boom("render failed on frame 5")
        return good(pkt, raw)
    pipe.render = render
    # make the writer fail too: its error must stay in the background of the render failure
    real_put = video.put_ai_frame
    video.put_ai_frame = lambda i, f: (_ for _ in ()).throw(OSError("disk full")) if i == 2 else real_put(i, f)
    with pytest.raises(Boom):
        pipe.run(video, flags)
    assert 5 in seen and 8 not in seen                      # stopped at the failure
    video.put_ai_frame = real_put
    pipe.render = good
    assert pipe.run(video, flags) == [0]                    # the same pipeline object still works afterwards (no leaked slots / threads)
    assert all(video.generated(i) for i in range(9))
    # collective guard
    import torch.distributed as tdist
    monkeypatch.setattr(tdist, "get_backend", lambda group=None: "nccl")
    with pytest.raises(RuntimeError, match="RCCL broadcasts need"):
        pipe._check_collective()


def test_bench_workspace_mode_two_ranks_gloo():
    """`python bench.py --gpus 2 --workspace` (hidden CPU stub step, gloo): the rank-parallel END-TO-END mode -- one shared PNG
    workspace written by rank 0, its path broadcast, every rank running the real `ClipPipeline.run` on its plan share between two
    barriers, per-rank records gathered on rank 0.  One JSON line; `value` is still the step's number; two per-rank records whose
    frames add up to the workspace, every AI frame on disk."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--stub-step", "--workspace", "--ws-segment", "7"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    ws = out["workspace_ranks"]
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert ws["ranks"] == 2 and ws["frames"] == 14 and ws["every_frame_written"] and ws["compute"] == "cpu stub"
    assert [p["rank"] for p in ws["per_rank"]] == [0, 1] and sum(p["frames"] for p in ws["per_rank"]) == 14
    assert all(p["wall_s"] > 0 and p["end_to_end_fps"] > 0 for p in ws["per_rank"]) and ws["max_wall_s"] == max(p["wall_s"] for p in ws["per_rank"])
    assert "no multi-GPU run" in ws["note"]


def _subgroup_worker(rank, world, port, ws):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        from sd_animation_optical_flow_amd import pipeline
        from sd_animation_optical_flow_amd.workspace import VideoData
        grp = dist.new_group(ranks=[1, 2])                # every rank of the default group makes this call
        if rank in (1, 2):
            p = _StubPipeline.make()
            p.group = grp
            keys = p.run(VideoData(None, (24, 16), ws))   # flags=None: the GROUP's rank 0 (global rank 1) detects and broadcasts
            torch.save(keys, os.path.join(ws, f"keys_r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipeline_on_a_subgroup_of_the_ranks(tmp_path):
    """`ClipPipeline(group=...)` (round-5 ADVICE): rank and world come from the GROUP, plan owners are ranks of the group and are mapped
    to global ranks for the `src` of every broadcast.  Three processes, the pipeline runs on the subgroup {1, 2} while rank 0 stays
    out: the cut segment's key frame travels from global rank 1 or 2, the flags come from the group's rank 0 (global rank 1, whose
    stub detector answers like a non-zero rank unless the group mapping is honoured), output byte-identical to one process."""
    import numpy as np
    from sd_animation_optical_flow_amd.workspace import VideoData
    one, _ = _make_workspace(str(tmp_path / "one"))
    # the group's rank 0 is GLOBAL rank 1: its stub detector returns the non-zero-rank answer (one key frame), so that is the plan
    flags = [True] + [False] * 22
    keys1 = _StubPipeline.make().run(one, flags)
    two, _ = _make_workspace(str(tmp_path / "two"))
    mp.spawn(_subgroup_worker, args=(3, _free_port(), str(tmp_path / "two")), nprocs=3, join=True)
    keys = [torch.load(tmp_path / "two" / f"keys_r{r}.pt") for r in (1, 2)]
    assert sorted(keys[0] + keys[1]) == keys1
    a, b = VideoData(None, (24, 16), str(tmp_path / "one")), VideoData(None, (24, 16), str(tmp_path / "two"))
    for i in range(23):
        assert b.generated(i), i
        assert np.array_equal(a.get_ai_frame(i), b.get_ai_frame(i)), i
