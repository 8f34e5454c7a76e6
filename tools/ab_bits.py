"""Are two builds of the library bit-identical on the small-grid schedules?  Full forwards of 1, 2, 3, 4 and 7 frames against one key
frame at 512x768 and 264x392 (the second: maps that are not whole patches), flows saved by one build and compared by the other:
    python tools/ab_bits.py save /tmp/ab.pt ; OFX_LIB_PATH=tools/variants/libofx_<name>.so python tools/ab_bits.py compare /tmp/ab.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sd_animation_optical_flow_amd.raft import RaftEngine
from sd_animation_optical_flow_amd.weights import random_state_dict
eng = RaftEngine(random_state_dict(0), "cuda")
outs = {}
torch.manual_seed(0)   # (make_clip draws its frame noise from the device generator)
for (H, W) in ((768, 512), (392, 264)):
    frames, key, _, _ = bench.make_clip(7, H, W, torch.device("cuda"))
    for B in (1, 2, 3, 4, 7):
        outs[f"{H}x{W}x{B}"] = eng.forward(frames[:B].contiguous(), key, iters=12).cpu()
if sys.argv[1] == "save":
    torch.save(outs, sys.argv[2])
    print("saved", len(outs), "flows")
else:
    ref = torch.load(sys.argv[2])
    bad = [k for k in outs if not torch.equal(outs[k], ref[k])]
    print(f"{len(outs) - len(bad)} / {len(outs)} identical", bad)
    assert not bad
