"""Per-kernel register / scratch / LDS figures of the gfx950 code objects inside a built library or object file.

    python tools/kernel_resources.py [path ...] [--scratch] [--grep NAME]

Reads the AMDGPU metadata notes of every device code object bundled into the file (`llvm-objdump --offloading` to unbundle into a
temporary directory, `llvm-readelf --notes` to read them).  `tests/test_host_and_abi.py::test_no_kernel_uses_scratch` runs the same
parser over `libmd2hip.so`.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    if not names or not shutil.which("c++filt"):
        return list(names)
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.strip() for o in out[:len(names)]]


def kernel_resources(path):
    """[(demangled kernel name, {private_segment_fixed_size, vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, group_segment_fixed_size})]"""
    path = os.path.abspath(path)
    tmp = tempfile.mkdtemp(prefix="md2co_")
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)                                    # (llvm-objdump --offloading writes the bundles next to its input)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
        rows = []
        for co in sorted(glob.glob(local + ".*gfx950*")):
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                blk = ".agpr_count:" + blk
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                rec = {}
                for k in ("private_segment_fixed_size", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size"):
                    m = re.search(r"\." + k + r":\s+(\d+)", blk)
                    rec[k] = int(m.group(1)) if m else -1
                rows.append((name, rec))
        names = demangle([r[0] for r in rows])
        return [(n, r[1]) for n, r in zip(names, rows)]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_scratch = "--scratch" in sys.argv
    pat = sys.argv[sys.argv.index("--grep") + 1] if "--grep" in sys.argv else None
    if pat in args:
        args.remove(pat)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in args or [os.path.join(root, "monodepth2.jl_amd", "libmd2hip.so")]:
        rows = kernel_resources(p)
        print("%s: %d kernels" % (p, len(rows)))
        for name, r in rows:
            if only_scratch and r["private_segment_fixed_size"] <= 0 and r["vgpr_spill_count"] <= 0:
                continue
            if pat and pat not in name:
                continue
            print("  scratch %4d B  vgpr %3d  agpr %3d  sgpr %3d  vspill %3d  sspill %3d  lds %6d  %s" % (
                r["private_segment_fixed_size"], r["vgpr_count"], r["agpr_count"], r["sgpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"],
                r["group_segment_fixed_size"], name[:140]))
