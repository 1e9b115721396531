"""The Julia side of the drop-in cannot be executed here (no Julia in the image), so it is checked mechanically: the generated bindings are up to date with
the ctypes table the tests run on, every export is used by the hand-written module, and the parameter table the Julia `flatten_flux_params` walks is the one
the oracle and the library share (/root/reference/src/Monodepth.jl:117-124,191-194, src/utils.jl:1-11)."""
import os
import re
import subprocess
import sys

import numpy as np

import md2amd
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "monodepth2.jl_amd", "julia")
syn = md2amd.synthetic


def test_generated_julia_files_are_up_to_date():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_julia_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_export_is_bound_and_used_by_the_julia_module():
    raw = open(os.path.join(JL, "md2_ccall.jl")).read()
    bound = set(re.findall(r"ccall\(\(:(md2_[A-Za-z0-9_]+), LIB\)", raw))
    assert bound == set(md2amd._lib.EXPORTS)                       # one raw wrapper per export of include/md2hip.h
    mod = open(os.path.join(JL, "Monodepth2HIP.jl")).read()
    used = set(re.findall(r"\b(md2_[A-Za-z0-9_]+)\(", mod))
    missing = sorted(set(md2amd._lib.EXPORTS) - used)
    assert not missing, "exports without a method in Monodepth2HIP.jl: %s" % missing
    assert not (used - bound)
    for name in ("eye_like", "zeros_like", "ones_like"):           # a18: the reference's device seam, src/utils.jl:1-11
        assert re.search(r"^%s\(::ROCArray\{T\}" % name, mod, re.M) and "@non_differentiable %s(::Any...)" % name in mod
    for name in ("flatten_flux_params", "unflatten_flux_params!", "ordered_params", "comm_init!", "broadcast_state!", "testmode!"):
        assert name in mod


def test_julia_param_table_matches_the_shared_layout():
    txt = open(os.path.join(JL, "param_table.jl")).read()
    for depth in (18, 50):
        for C in (1, 3):
            block = txt[txt.index("PARAM_TABLE[(%d, %d)] = [" % (C, depth)):]
            block = block[:block.index("]   #")]
            rows = re.findall(r'\("([^"]+)", :(\w+), \(([^)]*)\), (\d+)\)', block)
            lay = syn.param_layout(C, depth)
            assert len(rows) == len(lay)
            olay = oracle.OracleNet(C, depth=depth).layout()
            off = 0
            for (name, kind, shape, o), (lname, lkind, lshape), (oo, oc, _) in zip(rows, lay, olay):
                jshape = tuple(int(v) for v in shape.replace(" ", "").split(",") if v)
                assert name == lname and kind == lkind and jshape == tuple(reversed(lshape))       # Julia (kW,kH,Cin,Cout) == numpy (Cout,Cin,KH,KW)
                assert int(o) == off == oo and int(np.prod(lshape)) == oc                            # same offsets as the oracle and the library
                off += int(np.prod(lshape))
            assert off == syn.num_params(C, depth)
    # Flux's BatchNorm yields (beta, gamma); the table holds (gamma, beta): the Julia walk must swap them
    mod = open(os.path.join(JL, "Monodepth2HIP.jl")).read()
    assert "β, γ = enc[ei], enc[ei + 1]" in mod and "out[i] = γ; out[i + 1] = β" in mod
