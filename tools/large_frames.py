"""32 frames of 1080x1920 against one key frame through the sliced executor: the 32-bit-offset limit (21 pairs per call, a 130 GB workspace whose
first allocation costs seconds) against forced memory budgets (`RaftEngine.ws_budget_bytes`)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
g = torch.Generator(device="cuda").manual_seed(0)
B, H, W = 32, 1080, 1920
a = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
k = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
ref = None
for budget_gb in (None, 64, 40, 16):
    eng = RaftEngine(random_state_dict(0), "cuda")
    if budget_gb: eng.ws_budget_bytes = budget_gb * 2**30
    fit = eng.pairs_that_fit(min(B, eng.max_pairs(H, W)), H, W)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = eng.forward(a, k, iters=20); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    if ref is None: ref = out
    print(f"budget {budget_gb} GB: {fit} pairs per slice, workspace {eng._ws.numel() / 2**30:.1f} GB, first call {ts[0]:.0f} ms, then {ts[1]:.0f} / {ts[2]:.0f} ms per 32 frames, "
          f"max |diff| to the unbudgeted flows {(out - ref).abs().max().item():.1e} px", flush=True)
    del eng, out
    torch.cuda.empty_cache()
