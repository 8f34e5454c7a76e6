import sys, time, torch
sys.path.insert(0, '/root/repo')
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
eng = RaftEngine(random_state_dict(0), "cuda")
g = torch.Generator(device="cuda").manual_seed(0)
B, H, W = 32, 1080, 1920
a = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
k = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
print("max_pairs", eng.max_pairs(H, W), "fit", eng.pairs_that_fit(min(B, eng.max_pairs(H, W)), H, W), "free GB", torch.cuda.mem_get_info()[0] / 2**30)
t = time.perf_counter(); out = eng.forward(a, k, iters=20); torch.cuda.synchronize()
print(tuple(out.shape), f"{(time.perf_counter() - t) * 1e3:.0f} ms", "ws GB", eng._ws.numel() / 2**30, bool(torch.isfinite(out).all()))
eng.ws_budget_bytes = 40 * 2**30
t = time.perf_counter(); out2 = eng.forward(a, k, iters=20); torch.cuda.synchronize()
print("budget 40 GB: fit", eng.pairs_that_fit(21, H, W), f"{(time.perf_counter() - t) * 1e3:.0f} ms", "max diff", (out - out2).abs().max().item())
