"""The executor with the A-stationary fp32 volume kernel (the default) against the executor with the generic batched GEMM
(OFX_VOL_GENERIC=1) on random frame sizes and batches that the new kernel takes: the flows must be IDENTICAL bit for bit (the two
kernels produce the same pyramid bits, everything else is the same code).  Two processes, because the switch is read once:
    OFX_VOL_GENERIC=1 python tools/fuzz_engine_volume.py save /tmp/fev.pt [cases] [seed]
    python tools/fuzz_engine_volume.py compare /tmp/fev.pt [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict

mode, path = sys.argv[1], sys.argv[2]
n_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = random.Random(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
eng = RaftEngine(random_state_dict(0), "cuda")
outs = []
for case in range(n_cases):
    H, W = 64 * rng.randint(1, 4), 128 * rng.randint(1, 3)        # h % 8 == 0, w % 16 == 0 at 1/8 resolution
    N = (H // 8) * (W // 8)
    groups = -(-N // 256)
    B = min(96, max(2, -(-(rng.choice([180, 256, 300, 520])) // groups)))     # enough (pair, row group) tasks to fill the part
    shared = rng.random() < 0.5
    g = torch.Generator().manual_seed(100 + case)
    base = torch.nn.functional.avg_pool2d(torch.rand((1, 3, H + 40, W + 120), generator=g), 5, 1, 2)
    frames = torch.stack([(base[0, :, 10 + (b % 7):10 + (b % 7) + H, 8 + b:8 + b + W] * 255).round().to(torch.uint8).permute(1, 2, 0)
                          for b in range(B)]).contiguous().cuda()
    key = frames[0].contiguous() if shared else frames.flip(0).contiguous()
    flow = eng.forward(frames, key, iters=rng.choice([2, 3, 5]))
    assert torch.isfinite(flow).all()
    outs.append((H, W, B, shared, flow.cpu()))
if mode == "save":
    torch.save(outs, path)
    print(f"saved {n_cases} cases (generic={os.environ.get('OFX_VOL_GENERIC')})")
else:
    ref = torch.load(path)
    bad = 0
    for (H, W, B, sh, a), (_, _, _, _, b) in zip(outs, ref):
        same = torch.equal(a, b)
        bad += 0 if same else 1
        print((H, W, B, sh), "identical" if same else f"DIFFERENT max {float((a - b).abs().max()):.3e}")
    print(f"{n_cases - bad} / {n_cases} identical")
    sys.exit(1 if bad else 0)
