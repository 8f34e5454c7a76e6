"""-m gpu: the RCCL path on real devices.  One rank: `nccl` init + key-frame broadcast + all-reduce + barrier around the hot
path, the way bench.py and ClipPipeline drive it.  Two ranks: switched on by itself wherever two HIP devices are visible
(the gpurun boxes have one; the driver's 8-GPU node has them)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


_FAST = ["--no-cpu-baseline", "--no-single", "--no-fast", "--no-handoff", "--no-verify"]


def test_rccl_one_rank_bench_path(cuda):
    """bench.py's distributed code path on ONE MI355X: `init_process_group('nccl')`, the per-step key-frame broadcast, the
    barrier around the timed region, the MAX all-reduce of the time and RCCL's own rank count (all-reduce of ones)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "2"] + _FAST,
                       capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=_env(OFX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0",
                                LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["value"] > 0


def test_bench_workspace_mode_on_rccl_one_rank(cuda):
    """`bench.py --workspace` with the collective path forced on (one rank, `nccl` = RCCL): the workspace path travels by
    `broadcast_object_list`, the per-rank records by `all_gather_object`, the real `ClipPipeline.run` (HIP kernels, asynchronous host
    side) writes every AI frame of a 12-frame segment."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "4", "--workspace",
                        "--ws-segment", "12", "--no-volsplit", "--no-pipeline"] + _FAST,
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=_env(OFX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0",
                                LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    ws = json.loads(lines[0])["workspace_ranks"]
    assert ws["ranks"] == 1 and ws["frames"] == 12 and ws["every_frame_written"] and ws["compute"] == "hip"
    assert ws["per_rank"][0]["frames"] == 12 and ws["per_rank"][0]["end_to_end_fps"] > 0


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", device_id=dev)
from sd_animation_optical_flow_amd import clip
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
from sd_animation_optical_flow_amd import ops
eng = RaftEngine(random_state_dict(0), dev)
H, W, T = 128, 160, 6
g = torch.Generator().manual_seed(3)
frames = torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8, generator=g).to(dev)
src = world - 1
key = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, generator=g).to(dev) if rank == src else torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
key_ai = (255 - key) if rank == src else torch.zeros_like(key)
def step(fr, kr, ka):
    flow = eng.forward(fr, kr, iters=4)
    conf = torch.full(flow.shape[:3], 0.5, device=dev)
    w, m = ops.warp_and_mask(ka, flow, conf, warp_mode="bilinear", thres=0.95, ksize=7)
    return flow, w, m
res = clip.process_clip(frames, key, key_ai, step, batch_size=4, key_src=src)
assert res.frame_indices == list(clip.shard_range(T, rank, world))
ks = torch.tensor([float(key.sum()), float(key_ai.sum())], device=dev, dtype=torch.float64)
lo, hi = ks.clone(), ks.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi) and float(lo[0]) > 0, "the broadcast key frame differs between ranks"
n = torch.tensor([float(sum(t.shape[0] for t in res.flow))], device=dev)
dist.all_reduce(n)
assert int(n.item()) == T
assert all(bool(torch.isfinite(f).all()) for f in res.flow)
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank, world)
"""


def _launch(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), ROOT]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env())


def test_rccl_one_rank_clip_shard_and_broadcast(cuda, tmp_path):
    """`clip.process_clip` (shard -> key-frame broadcast -> flow / warp / mask) under a real `nccl` process group of one rank."""
    r = _launch(1, tmp_path)
    assert r.returncode == 0 and "RANK_OK 0 1" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices (auto-enabled where they exist)")
def test_rccl_two_ranks_clip_and_bench(cuda, tmp_path):
    """Two ranks over RCCL: the key frame rendered on rank 1 reaches rank 0, every frame is processed exactly once, and
    `bench.py --gpus 2` self-launches two ranks and reports RCCL's own count."""
    r = _launch(2, tmp_path)
    assert r.returncode == 0 and "RANK_OK 0 2" in r.stdout and "RANK_OK 1 2" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + _FAST,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env())
    assert b.returncode == 0, b.stderr[-3000:]
    out = json.loads([l for l in b.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["scaling"] == "weak" and out["value"] > 0
