"""CPU: the C-ABI library loads and exports every symbol include/ofx.h declares; host-side logic
(weight packing, checkpoint handling, work model of bench.py).  No GPU compute is issued."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ofx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ofx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from sd_animation_optical_flow_amd import _lib
    lib = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libofx.so does not export {name}"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.ofx_version() == 100


def test_error_strings_and_host_side_preconditions():
    from sd_animation_optical_flow_amd import _lib
    lib = _lib.lib()
    assert "aligned" in _lib.error_string(-2) and "workspace" in _lib.error_string(-3)
    assert _lib.error_string(0) == "success"
    # argument validation happens before any HIP call, so it is testable without a device
    assert lib.ofx_conv2d(None, None) == -1
    assert lib.ofx_warp_u8(None, 0, None, None, 1, 8, 8, 3, 0, 1.0, None) == -1
    assert lib.ofx_generate_mask(None, None, None, 1, 8, 8, 0.5, 7, 0, None) == -1
    assert lib.ofx_raft_workspace_bytes(None, 1, 100, 96) == 0          # H not a multiple of 8
    need = lib.ofx_raft_workspace_bytes(None, 64, 768, 512)
    assert 15e9 < need < 40e9                                             # 12.8 GB of pyramids + activations
    d = _lib.ConvDesc()
    buf = (C.c_float * 64)()
    d.in0, d.w, d.c0, d.ld0 = C.addressof(buf), C.addressof(buf), 6, 8    # c0 not a multiple of 4
    assert lib.ofx_conv2d(C.byref(d), None) == -2
    with pytest.raises(_lib.OfxError, match="aligned"):
        _lib.check(-2, "test")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sd_animation_optical_flow_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OfxLibraryError, match="no CPU fallback"):
        _lib.lib()


def test_ops_reject_cpu_and_noncontiguous_tensors():
    from sd_animation_optical_flow_amd import ops
    f = torch.zeros((1, 8, 8, 32))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.local_corr(f, f, torch.zeros((1, 1, 8, 8, 2)), 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.warp(torch.zeros((8, 8, 3), dtype=torch.uint8), torch.zeros((8, 8, 2)))


def test_pack_conv_weight_layout():
    from sd_animation_optical_flow_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn((5, 3, 7, 7), generator=g)
    p = ops.pack_conv_weight(w, 4)
    assert tuple(p.shape) == (5, 224)                      # K = 7*7*4 = 196 -> padded to 224
    ref = torch.zeros((5, 7, 7, 4))
    ref[..., :3] = w.permute(0, 2, 3, 1)                   # k = (ky*KW + kx)*cin_pad + c
    assert torch.equal(p[:, :196], ref.reshape(5, 196)) and float(p[:, 196:].abs().max()) == 0.0
    w2 = torch.randn((2, 324, 1, 1), generator=g)
    p2 = ops.pack_conv_weight(w2)
    assert tuple(p2.shape) == (2, 352) and torch.equal(p2[:, :324], w2.reshape(2, 324))


def test_product_weights_equal_oracle_weights(raft_sd):
    """weights.random_state_dict is the product-side twin of the oracle's seeded init (the product never
    imports oracle/): identical tensors, identical key set."""
    from sd_animation_optical_flow_amd.weights import load_checkpoint, random_state_dict
    sd = random_state_dict(0)
    assert sorted(sd) == sorted(raft_sd)
    assert all(torch.equal(sd[k], raft_sd[k]) for k in sd)
    wrapped = {"module." + k: v for k, v in sd.items()}
    assert sorted(load_checkpoint(wrapped)) == sorted(sd)
    assert sorted(load_checkpoint("random:0")) == sorted(sd)
    with pytest.raises(FileNotFoundError):
        load_checkpoint("/nonexistent/raft-things.pth")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sd_animation_optical_flow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "/root/reference" not in txt, f


def test_bench_work_model_matches_survey():
    import bench
    w = bench.algorithmic_work(768, 512, 1, shared_key=False)
    # volume+pyramid 213.1 MB, 19.33 GFLOP; lookup 17.84 MB/iter; upsample 17.35 MB; warp 5.51; mask 1.97
    assert abs(w["volume_flops"] / 1e9 - 19.33) < 0.05
    total_vol = w["volume_bytes"] + w["pool_bytes"] - w["pool_reread_bytes"]   # the level the pooling kernel reads back is counted once
    assert abs(total_vol / 1e6 - 213.1) < 1.0
    assert abs(w["lookup_bytes"] / 1e6 - 17.84) < 0.1
    assert abs(w["upsample_bytes"] / 1e6 - 17.35) < 0.1
    assert abs((w["warp_bytes"] - 768 * 512 * 3.0 + 768 * 512 * 3.0) / 1e6 - 5.51) < 0.1
    assert abs(w["mask_bytes"] / 1e6 - 1.97) < 0.02
    assert abs(bench.update_flops(768, 512, True) / 1e9 - 38.32) < 0.1
    assert abs(bench.encoder_flops(768, 512) / 1e9 - 53.44) < 0.1


def test_pad_to_8_matches_input_padder():
    from oracle import raft_oracle as RO
    from sd_animation_optical_flow_amd.raft import RaftEngine
    img = torch.randint(0, 256, (1, 37, 50, 3), dtype=torch.uint8)
    got = RaftEngine.pad_to_8(img)
    ref, _ = RO.pad_to_8(img.permute(0, 3, 1, 2).float())
    assert tuple(got.shape) == (1, 40, 56, 3)
    assert torch.equal(got.permute(0, 3, 1, 2).float(), ref)


def test_top_level_alt_cuda_corr_module_resolves():
    """RAFT/core/corr.py:5-9 does `import alt_cuda_corr`: with the repository root on sys.path that import must give
    the HIP-backed forward/backward (no compute here: no GPU)."""
    import importlib
    m = importlib.import_module("alt_cuda_corr")
    from sd_animation_optical_flow_amd import alt_cuda_corr as inner
    assert m.forward is inner.forward and m.backward is inner.backward
    import torch
    with pytest.raises(RuntimeError):
        m.forward(torch.zeros(1, 4, 4, 8), torch.zeros(1, 4, 4, 8), torch.zeros(1, 1, 4, 4, 2), 1)   # CPU tensors
