"""Order of LDS loads, LDS reads, waits and barriers in the gfx950 ISA of the kernels that stage operands with `buffer_load ... lds` (the direct-to-LDS pipeline of
conv_igemm.hip).

    python tools/lds_pipeline_lint.py [library.so | object.o ...]

Why this exists (round 6): that pipeline synchronises with a raw `s_barrier` and counted `s_waitcnt vmcnt(N)` -- a `__syncthreads()` would drain the
loads in flight -- and `llvm.amdgcn.s.barrier` is IntrNoMem: hipcc may move plain LDS loads across it (and across an `asm volatile("s_waitcnt" ::: "memory")`).  One build
of `conv_igemm_classes_kernel` had the prologue's operand reads of stage 0 BEHIND the barrier from which other waves overwrite stage 0: a write-after-read race on LDS that
no parity test sees (it needs a wave held up for a load's flight time).  The source now pins the order with data dependences; this lint checks the result where it counts,
in the ISA of every build.  `tests/test_host_and_abi.py::test_lds_pipeline_order_in_the_isa` runs it over `libmd2hip.so`.

Rules, per kernel that issues LDS loads, on the linear instruction stream cut into intervals at every `s_barrier`:
  R1  an interval that issues LDS loads waits for them (`s_waitcnt vmcnt`) after the last issue and before its barrier, and no operand read (`ds_read_b128`) or LDS load
      stands between the interval's last such wait and the barrier (a read there was hoisted over the barrier it belongs behind, or has not retired when the barrier releases the stage);
  R2  the first interval of a pipeline (LDS loads, no operand reads: the prologue) is followed by an interval WITH operand reads and without LDS loads (the reads of stage 0
      stay between the two prologue barriers), and those reads are retired (`s_waitcnt lgkmcnt(0)`) before the second barrier;
  R3  in every other interval that issues LDS loads and reads operands (a loop step) the reads are retired before its barrier (a peeled last step has MFMAs and no reads).
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def _code_objects(path, tmp):
    local = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    return sorted(glob.glob(local + ".*gfx950*"))


def disassemble(path, name_filter="conv_igemm"):
    """{mangled kernel name: [instruction text, ...]} for the kernels whose name contains `name_filter`."""
    tmp = tempfile.mkdtemp(prefix="md2lint_")
    out = {}
    try:
        for co in _code_objects(os.path.abspath(path), tmp):
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], check=True, capture_output=True, text=True).stdout
            names = [l.split()[-1] for l in syms.splitlines() if " FUNC " in l and name_filter in l.split()[-1]]
            for i in range(0, len(names), 40):
                txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--disassemble-symbols=" + ",".join(names[i:i + 40]), co],
                                     check=True, capture_output=True, text=True).stdout
                cur = None
                for line in txt.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                    if m:
                        cur = m.group(1) if m.group(1) in names else None          # (branch-target labels inside a kernel look like <name+0x1234> and do not match)
                        if cur is not None: out[cur] = []
                        continue
                    if cur is None or not line.startswith("\t") and not line.startswith(" "): continue
                    ins = line.split("//")[0].strip()
                    if ins: out[cur].append(re.sub(r"\s+", " ", ins))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _is_dma(ins): return (ins.startswith("buffer_load_") and ins.endswith(" lds")) or " lds " in ins and ins.startswith("buffer_load_") or ins.startswith("global_load_lds_")
def _is_read(ins): return ins.startswith("ds_read_b128")
def _is_vm(ins): return ins.startswith("s_waitcnt") and "vmcnt(" in ins
def _is_lgkm0(ins): return ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins


def lint_kernel(ins):
    """List of violations (strings) for one kernel's instruction list; [] when the kernel issues no LDS loads or is in order."""
    if not any(_is_dma(i) for i in ins): return []
    # intervals end at a barrier (the tail after the last barrier is not a pipeline interval)
    ivs, cur = [], []
    for i in ins:
        if i.startswith("s_barrier"): ivs.append(cur); cur = []
        else: cur.append(i)
    bad = []
    def retired(iv):            # the last operand read is followed by an lgkmcnt(0) wait
        last = max(k for k, i in enumerate(iv) if _is_read(i))
        return any(_is_lgkm0(i) for i in iv[last + 1:])
    for n, iv in enumerate(ivs):
        dmas = [k for k, i in enumerate(iv) if _is_dma(i)]
        if not dmas: continue
        reads = [k for k, i in enumerate(iv) if _is_read(i)]
        waits = [k for k, i in enumerate(iv) if _is_vm(i) and k > dmas[-1]]
        if not waits: bad.append("interval %d: LDS loads issued but no s_waitcnt vmcnt before the barrier (R1)" % n); continue
        if any(k > waits[-1] for k in reads): bad.append("interval %d: ds_read_b128 between the last vmcnt wait and the barrier (R1)" % n)
        mfmas = any(i.startswith("v_mfma") for i in iv)
        if not reads and mfmas: continue          # a peeled LAST step: nothing follows it, so its operand reads were dead and are gone
        if not reads:           # prologue
            nxt = ivs[n + 1] if n + 1 < len(ivs) else []
            if not any(_is_read(i) for i in nxt) or any(_is_dma(i) for i in nxt): bad.append("interval %d: prologue not followed by the stage-0 operand reads between its two barriers (R2)" % n)
            elif not retired(nxt): bad.append("interval %d: stage-0 operand reads not retired before the second prologue barrier (R2)" % (n + 1))
        elif not retired(iv): bad.append("interval %d: operand reads of a loop step not retired before its barrier (R3)" % n)
    return bad


def lint(path):
    """(number of kernels with LDS loads, {kernel: [violations]})"""
    ks = disassemble(path)
    n = 0; res = {}
    for name, ins in ks.items():
        if any(_is_dma(i) for i in ins):
            n += 1
            v = lint_kernel(ins)
            if v: res[name] = v
    return n, res


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = 0
    for p in sys.argv[1:] or [os.path.join(root, "monodepth2.jl_amd", "libmd2hip.so")]:
        n, res = lint(p)
        print("%s: %d kernels issue LDS loads, %d out of order" % (p, n, len(res)))
        for k, v in res.items():
            rc = 1
            print("  " + k)
            for line in v: print("     " + line)
    sys.exit(rc)
