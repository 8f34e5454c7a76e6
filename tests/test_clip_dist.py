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
