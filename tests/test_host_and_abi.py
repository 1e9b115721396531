"""Host logic + C-ABI surface (no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

import md2amd
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
syn = md2amd.synthetic


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "md2hip.h")).read()
    declared = sorted(set(re.findall(r"\b(md2_[a-z0-9_A-Z]+)\s*\(", hdr)))
    assert len(declared) >= 40
    L = md2amd._lib.lib()                                   # loads libmd2hip.so and resolves every bound symbol
    for name in declared:
        assert hasattr(L, name), "libmd2hip.so does not export " + name
    assert sorted(md2amd._lib.EXPORTS) == declared          # the ctypes table binds exactly the header


def test_param_layout_agrees_between_python_oracle_and_count():
    for C in (1, 3):
        lay = oracle.OracleNet(C).layout(); pl = syn.param_layout(C)
        assert len(lay) == len(pl)
        off = 0
        for (o, c, _), (_, _, shape) in zip(lay, pl):
            assert o == off and c == int(np.prod(shape)); off += c
        assert off == syn.num_params(C)
    assert syn.num_params(1) == 16225818                    # all-reduce payload


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = md2amd._lib.lib()
    cfg = md2amd._lib.Md2Config(1, 1, 64, 96, 0, 0, 0.1, 100.0, 1e-3, 1e-4)
    h = ctypes.c_void_p()
    rc = L.md2_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc != 0 and b"no HIP device" in L.md2_last_error(None)
    md = md2amd.monodepth
    with pytest.raises((ValueError, RuntimeError)):
        md.SSIM()(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4))        # CPU tensors are rejected, never computed on the host


def test_config_validation_messages():
    L = md2amd._lib.lib()
    h = ctypes.c_void_p()
    bad = md2amd._lib.Md2Config(1, 2, 64, 96, 0, 0, 0.1, 100.0, 1e-3, 1e-4)
    assert L.md2_create(ctypes.byref(bad), ctypes.byref(h)) == 2
    bad = md2amd._lib.Md2Config(1, 1, 60, 96, 0, 0, 0.1, 100.0, 1e-3, 1e-4)
    assert L.md2_create(ctypes.byref(bad), ctypes.byref(h)) == 2


def test_reference_surface_errors():
    md = md2amd.monodepth
    with pytest.raises(ValueError):                          # src/depth_decoder.jl:27-29
        md.DepthDecoder([64, 64, 128, 256, 512], [0, 1])
    enc = md.ResidualNetwork(18, in_channels=1)
    assert enc.stages == (64, 64, 128, 256, 512)
    p = md.Params(target_size=(416, 128), batch_size=4)
    assert p.automasking is True and p.min_depth == 0.1 and p.disparity_smoothness == 1e-3   # src/Monodepth.jl:32-42 defaults


def test_synthetic_inputs():
    x = syn.synthetic_triplets(2, 1, 64, 96, seed=3)
    assert x.shape == (2, 3, 1, 64, 96) and x.dtype == np.float32 and 0 <= x.min() and x.max() <= 1
    assert np.array_equal(x, syn.synthetic_triplets(2, 1, 64, 96, seed=3))
    K, iK = syn.depth10k_intrinsics()
    assert abs(K[0, 0] - 2648.0 / 4.63461538462) < 1e-9 and K[0, 2] == 208 and K[1, 2] == 64   # src/dtk.jl:16-22
    assert np.allclose(K @ iK, np.eye(3))


def test_shard_range_covers_batch():
    par = md2amd.parallel
    for gb in (1, 7, 32, 33):
        for world in (1, 2, 8):
            got = []
            for r in range(world):
                lo, hi = par.shard_range(gb, r, world); got += list(range(lo, hi))
            assert got == list(range(gb))
