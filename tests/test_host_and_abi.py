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


def test_config_struct_layout_matches_the_header():
    """md2_config crosses the boundary by pointer: the ctypes mirror (and through it the generated Julia struct) must have the C layout of include/md2hip.h -- size and every
    field offset, checked against what gcc computes for the header."""
    import subprocess, tempfile
    C = md2amd._lib.Md2Config
    fields = [f[0] for f in C._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "md2hip.h"\nint main(void) { printf("%zu", sizeof(md2_config));' + \
          "".join(' printf(" %%zu", offsetof(md2_config, %s));' % f for f in fields) + ' return 0; }\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert out[0] == ctypes.sizeof(C)
    assert out[1:] == [getattr(C, f).offset for f in fields]
    jl = open(os.path.join(ROOT, "monodepth2.jl_amd", "julia", "Monodepth2HIP.jl")).read()
    struct = jl[jl.index("struct MD2Config"):jl.index("\nend", jl.index("struct MD2Config"))]
    decl = re.findall(r"(\w+)::([\w{}, ]+?)(?:;|\s*#|\n)", struct)
    jl_types = {"Int32": ctypes.c_int32, "Float64": ctypes.c_double, "NTuple{5, Int32}": ctypes.c_int32 * 5, "NTuple{5, Float64}": ctypes.c_double * 5}
    flat = []                                                  # the Julia struct spells source_ids as two Int32 fields: flatten both sides to scalar types
    for _, t in decl:
        ct = jl_types[t.strip()]
        flat += [ct._type_] * ct._length_ if hasattr(ct, "_length_") else [ct]
    cflat = []
    for _, ct in C._fields_:
        cflat += [ct._type_] * ct._length_ if hasattr(ct, "_length_") else [ct]
    assert flat == cflat                                       # same scalar sequence => same C layout (isbits struct, natural alignment)


def test_frame_roles_and_scale_levels_are_validated_before_the_device():
    """TrainCache.target_id / source_ids and DepthDecoder(; scale_levels) (src/Monodepth.jl:44-55, src/depth_decoder.jl:27-29): bad values are refused with status 2 and a message."""
    L = md2amd._lib.lib(); h = ctypes.c_void_p(); M = md2amd._lib.Md2Config
    def create(tid=0, sids=(0, 0), levels=(), scales=()):
        cfg = M(1, 1, 64, 96, 0, 0, 0.1, 100.0, 1e-3, 1e-4, 18, 0, tid, (ctypes.c_int32 * 2)(*sids), (ctypes.c_int32 * 5)(*levels), (ctypes.c_double * 5)(*scales))
        return L.md2_create(ctypes.byref(cfg), ctypes.byref(h)), L.md2_last_error(None)
    for kw in (dict(tid=2, sids=(2, 3)), dict(tid=4, sids=(1, 3)), dict(tid=1, sids=(3, 3)), dict(tid=1, sids=(0, 3))):
        rc, msg = create(**kw); assert rc == 2 and b"source_ids" in msg, kw
    for lv in ((0, 2), (3, 2), (2, 2), (6,), (1, 2, 3, 4, 7)):
        rc, msg = create(levels=lv); assert rc == 2 and b"scale_levels" in msg, lv
    rc, msg = create(levels=(2, 3), scales=(0.5, float("nan"))); assert rc == 2 and b"scales" in msg
    for kw in (dict(), dict(tid=1, sids=(2, 3)), dict(tid=3, sids=(2, 1)), dict(levels=(1, 2, 3, 4, 5)), dict(levels=(5,)), dict(levels=(1, 3, 4), scales=(0.3, 0.7, 1.1)),
               dict(levels=(2, 3), scales=(0.5, -1.0)), dict(levels=(2, 3), scales=(0.0, 0.7))):          # negative = that level's default; an explicit 0 is taken as given
        rc, msg = create(**kw); assert rc in (0, 3), (kw, msg)                 # valid arguments: only the missing device (3) may stop a CPU-only box
        if rc == 0:
            L.md2_destroy(h)
    md = md2amd.monodepth
    with pytest.raises(ValueError):
        md.DepthDecoder([64, 64, 128, 256, 512], [3, 2])                      # ascending, like the reference's constructor walks them
    assert md.DepthDecoder([64, 64, 128, 256, 512], [1, 3, 4]).scale_levels == [1, 3, 4]
    for levels in ((1, 3, 4), (1, 2, 3, 4, 5), (5,)):                          # parameter count / order of a decoder with other levels: Python layout == oracle
        lay = oracle.OracleNet(1, scale_levels=levels).layout(); pl = syn.param_layout(1, 18, levels)
        assert [(c) for (_, c, _) in lay] == [int(np.prod(s_)) for (_, _, s_) in pl] and sum(c for _, c, _ in lay) == syn.num_params(1, 18, levels)


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


def test_conv_product_modes_are_validated_and_round_trip_without_a_device():
    """md2_set_conv_products / md2_get_conv_products (how the fp32 products of the convolutions are formed; the reference computes in Float32, src/Monodepth.jl:72): process-wide host
    state, no device needed.  The load-time default is the per-launch choice between the two FULL-accuracy forms (1, 1); the reduced two-piece form (2) is opt-in only."""
    L = md2amd._lib.lib()
    f = ctypes.c_int32(-7); w = ctypes.c_int32(-7)
    assert L.md2_get_conv_products(ctypes.byref(f), ctypes.byref(w)) == 0
    dflt = (f.value, w.value)
    if not (os.environ.get("MD2_WINO_PIECES") or os.environ.get("MD2_WGRAD_PIECES") or os.environ.get("MD2_WINO_MODE")):
        assert dflt == (1, 1)                                    # full fp32 operand accuracy in every product, nothing reduced by default
    try:
        for bad in ((4, 0), (0, 4), (-2, 0), (0, -3)):
            assert L.md2_set_conv_products(*bad) == 2
            assert L.md2_get_conv_products(ctypes.byref(f), ctypes.byref(w)) == 0 and (f.value, w.value) == dflt      # a refused call changes nothing
        for mode in ((0, 0), (3, 3), (2, 2), (1, 0)):
            assert L.md2_set_conv_products(*mode) == 0
            L.md2_get_conv_products(ctypes.byref(f), ctypes.byref(w)); assert (f.value, w.value) == mode
        assert L.md2_set_conv_products(-1, 3) == 0               # -1 keeps
        L.md2_get_conv_products(ctypes.byref(f), ctypes.byref(w)); assert (f.value, w.value) == (1, 3)
        # the legacy aliases of md2_set_winograd: 4 / 5 = automatic algorithm choice + two / three pieces in the forward / input-gradient kernels
        assert L.md2_set_winograd(5, -1) == 0
        L.md2_get_conv_products(ctypes.byref(f), ctypes.byref(w)); assert (f.value, w.value) == (3, 3)
        assert L.md2_set_winograd(1, -2) == 0                    # -2 restores the workgroup-count threshold; the products stay as set
        L.md2_get_conv_products(ctypes.byref(f), ctypes.byref(w)); assert (f.value, w.value) == (3, 3)
    finally:
        L.md2_set_winograd(1, -2); L.md2_set_conv_products(*dflt)


# Kernels that may keep a private (scratch) segment, with the reason.  Everything else in libmd2hip.so must have `.private_segment_fixed_size: 0` and no spilled
# VGPR: scratch traffic is HBM traffic the roofline does not see (round 4: loss_train_kernel<3,16> moved 4.8x its algorithmic bytes through a 224-byte segment).
SCRATCH_ALLOWED = {
    # opt-in 64-tile Winograd workgroups of the five-row-segment shape (md2_set_winograd NH = 2, not a default route): 5 registers over the 256 budget
    "conv_wino3_kernel<5, 0, 0, 2, 1>", "conv_wino3_kernel<5, 0, 1, 2, 1>",
    # the opt-in REDUCED two-piece weight gradient on multi-row batches (md2_set_conv_products(., 2)): never in `value`
    "conv_wino_wgrad_s_kernel<1, 2, 1>",
}


def test_no_kernel_uses_scratch():
    """Code-object metadata of every gfx950 kernel in the built library (llvm-readelf --notes; tools/kernel_resources.py): no private segment, no spilled vector register,
    outside a short allow-list of opt-in instantiations -- and every loss_train_kernel instantiation is spill-free whatever the list says."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    if not os.path.exists(os.path.join(kernel_resources.LLVM, "llvm-readelf")):
        pytest.skip("no llvm-readelf in this image")
    rows = kernel_resources.kernel_resources(md2amd._lib.LIB_PATH)
    assert len(rows) > 300, "the code objects of libmd2hip.so were not found"
    assert sum("loss_train_kernel" in n for n, _ in rows) >= 2
    bad = []
    for name, r in rows:
        if r["private_segment_fixed_size"] > 0 or r["vgpr_spill_count"] > 0:
            allowed = any(a in name for a in SCRATCH_ALLOWED) and "loss_train_kernel" not in name
            if not allowed:
                bad.append("%s: %d B scratch, %d spilled VGPRs" % (name, r["private_segment_fixed_size"], r["vgpr_spill_count"]))
    assert not bad, "kernels with a private segment: " + "; ".join(bad)


def test_lds_pipeline_order_in_the_isa():
    """The direct-to-LDS operand pipeline of conv_igemm.hip (`buffer_load ... lds`, counted `s_waitcnt vmcnt`, raw `s_barrier`) is correct only if, in the ISA, the operand
    reads of a stage sit between the barrier that publishes it and the barrier that releases it for the next load -- and `llvm.amdgcn.s.barrier` is IntrNoMem, so hipcc is free
    to move plain LDS loads across it.  Round 5's "race of unknown cause" was exactly that: one build had the prologue's stage-0 reads behind the second prologue barrier
    (DESIGN.md section 4).  The source pins the order with data dependences (conv_igemm_body: pin_stage_offsets / lds_wait_reads); this test reads the order back from the
    disassembly of every kernel of the built library that issues LDS loads (tools/lds_pipeline_lint.py: rules R1-R3), so a compiler or source change that reopens the race
    fails the CPU suite instead of one GPU step in a few hundred.  The lint itself is pinned on two hand-made streams: the round-5 failure shape must be flagged."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_pipeline_lint as lint
    dma = "buffer_load_dwordx4 v1, s[8:11], s0 offen lds"; rd = "ds_read_b128 v[2:5], v6"; mf = "v_mfma_f32_32x32x16_bf16 a[0:15], v[2:5], v[6:9], a[0:15]"
    good = [dma] * 6 + ["s_waitcnt vmcnt(2)", "s_barrier"] + [rd] * 4 + ["s_waitcnt lgkmcnt(0)", "s_barrier"] + ([dma] * 2 + [rd] * 4 + [mf] * 6 + ["s_waitcnt vmcnt(2)", "s_waitcnt lgkmcnt(0)", "s_barrier"]) * 3 + ["s_endpgm"]
    sunk = [dma] * 6 + ["s_waitcnt vmcnt(2)", "s_barrier", "s_waitcnt lgkmcnt(0)", "s_barrier"] + [rd] * 4 + ([dma] * 2 + [rd] * 4 + [mf] * 6 + ["s_waitcnt vmcnt(2)", "s_waitcnt lgkmcnt(0)", "s_barrier"]) * 3 + ["s_endpgm"]
    late = [dma] * 6 + ["s_waitcnt vmcnt(2)", "s_barrier"] + [rd] * 4 + ["s_waitcnt lgkmcnt(0)", "s_barrier"] + [dma] * 2 + [rd] * 2 + [mf] * 6 + ["s_waitcnt vmcnt(2)"] + [rd] * 2 + ["s_waitcnt lgkmcnt(0)", "s_barrier", "s_endpgm"]
    assert lint.lint_kernel(good) == [] and lint.lint_kernel(["s_barrier", rd, "s_endpgm"]) == []
    assert any("R2" in v for v in lint.lint_kernel(sunk)) and any("R1" in v for v in lint.lint_kernel(late))
    if not os.path.exists(os.path.join(lint.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    n, bad = lint.lint(md2amd._lib.LIB_PATH)
    assert n >= 30, "the kernels with the direct-to-LDS loader were not found in libmd2hip.so (%d)" % n
    assert not bad, "LDS pipeline out of order in: " + "; ".join("%s: %s" % (k, v) for k, v in bad.items())


def test_bench_convolution_class_accounting(tmp_path):
    """bench.py prices the forward + input-gradient convolution class from the library's per-launch profile records (md2_profile_dump): algorithmic bytes (inputs + weights +
    epilogue operands + outputs, once each) and the matrix-pipe floor of the instructions each launch EXECUTES (fp32 MFMA 157.3 TFLOP/s; exact split: six bf16 MFMAs per product
    group at 2.5 PFLOP/s; Winograd: 16/36 of the direct count).  Host logic on a hand-made dump: no GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("md2_bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    M, co, ci = 1000, 64, 128
    rows = ["cls,ms,gflop,tflops,desc",
            "0,0.1,8.0,80,stem fwd 7x7 s2 (direct)",
            "0,0.1,%g,100,igemm M=%d Cout=%d Cin=%d k=3 s=1 dil=1 pm=0 HxW=10x100 ep=1 alg=winograd_split prod=split3" % (2e-9 * M * co * 9 * ci, M, co, ci),
            "0,0.1,%g,100,igemm M=%d Cout=%d Cin=%d k=1 s=2 dil=1 pm=0 HxW=20x200 ep=0 alg=direct prod=fp32" % (2e-9 * M * co * ci, M, co, ci),
            "1,0.2,5.0,25,wgrad M=1000 Cout=64 Cin=128 k=3 s=1 pm=0 alg=winograd pieces=0"]
    pth = tmp_path / "dump.csv"; pth.write_text("\n".join(rows) + "\n")
    acct = b.conv_class_accounting(str(pth), 4, 1)
    by = 4.0 * ((M * ci + M * co * 2 + 9 * ci * co) + (M * 4 * ci + M * co + ci * co))
    assert abs(acct["algorithmic_bytes_per_step"] - by) < 1e-6 * by
    floor_s = 8.0e9 / 157.3e12 + (2.0 * M * co * 9 * ci) * (16.0 / 36.0) * 6.0 / 2500e12 + (2.0 * M * co * ci) / 157.3e12
    assert abs(acct["floor_ms"] - floor_s * 1e3) < 1e-6 * floor_s * 1e3
    assert acct["launches_by_product_form"] == {"fp32": 2.0, "split3": 1.0}
