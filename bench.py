#!/usr/bin/env python3
"""bench.py — self-supervised training throughput (triplets/s) of the MI355X training step.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run with one
rank per GPU (RCCL) -- or, started plainly with --gpus N > 1 and no WORLD_SIZE in the environment, it starts its own N ranks (child processes, before this
process touches a GPU) and relays rank 0's line.  W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize, MAX over
ranks, rank 0 prints ONE JSON line.  A step = forward + loss + backward + (all-reduce) + ADAM over one batch of synthetic
416x128 triplets already resident in HBM (src/Monodepth.jl:156-176).  Weak scaling: every GPU gets `--batch` triplets
(default 32 = BASELINE.json configs[2]'s per-GPU batch, so that N=1 and N=8 do the same per-GPU work; configs[1]'s
batch-4 rate is measured as well at N=1 and reported under "extra").
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 matrix peak (--precision bf16 prices against this one)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
PROFILE_ROUND = "r06"          # profiles/<round>_*.json: the rocprofv3 PMC summaries of THIS round's kernels (tools/make_profiles.sh + tools/summarize_profiles.sh)
VALU_PEAK_TLANE = 256 * 4 * 32 * 2.4e9 / 1e12   # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-instructions/s (a wave64 VALU instruction takes 2 cycles of its SIMD)
# HBM bytes per launch come from the rocprofv3 PMC passes of THIS command (profiles/<round>_pmc_hbm_traffic.json, written by tools/profile_summary.py from separate
# --pmc FETCH_SIZE / WRITE_SIZE runs: KiB units, FETCH_SIZE x2 because gfx950 tallies 64 B per 128-B request, WRITE_SIZE exact).  PMC counters cannot be read
# from inside bench.py, so the tracked measurement is reported when its workload equals the one being run, null otherwise.
def pmc_traffic(B, C, H, W, depth, precision):
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_pmc_hbm_traffic.json")))
    except Exception:
        return {}
    w = d.get("workload", {})
    if (w.get("batch"), w.get("channels"), w.get("height"), w.get("width"), w.get("depth"), w.get("precision")) != (B, C, H, W, depth, precision):
        return {}
    return d.get("bytes_per_launch", {})


def conv_class_accounting(dump_path, esz, nsteps):
    """Per-step accounting of the forward + input-gradient convolution class from the library's per-launch profile records (md2_profile_dump: one record per launch with
    M output pixels, Cin, Cout, k, stride, the output-sized operands `ep` of the fused epilogue, the algorithm `alg` and the product form `prod`):
      algorithmic_bytes -- each launch reads its input pixels, its weights and the epilogue's output-sized operands once and writes its output once, at `esz` bytes per element;
      floor_ms          -- the time the launches' EXECUTED matrix instructions need at the rate of the pipe they run on: fp32 MFMA 157.3 TFLOP/s; an exact-split launch issues six
                           (two-piece: three) bf16 MFMAs per fp32 product group on the 2.5 PFLOP/s bf16 pipe; Winograd launches execute 16/36 of the direct-convolution count.
    Since round 5 the roofline block's `frac` is floor_ms / ms_per_step (= algorithmic rate / the launch mix's effective peak: the share of the class time its executed matrix
    instructions need at full pipe rate); the rounds-1-to-4 definition, the algorithmic rate against the fp32 MFMA peak alone, is kept as `algorithmic_vs_fp32_mfma_peak`
    (it passed 1.0 once Winograd and the bf16-pipe split multiplied fewer / cheaper products than the direct count)."""
    import re
    by = 0.0; floor_s = 0.0; forms = {}
    for line in open(dump_path).read().splitlines()[1:]:
        f = line.split(",", 4)
        if len(f) != 5 or f[0] != "0":
            continue
        gflop = float(f[2]); desc = f[4]
        m = re.match(r"igemm M=(\d+) Cout=(\d+) Cin=(\d+) k=(\d+) s=(\d+) .* ep=(\d+) alg=(\S+)(?: prod=(\S+))?", desc)
        if m:
            M, co, ci, k, st, ep = (int(v) for v in m.groups()[:6]); alg = m.group(7); prod = m.group(8) or "fp32"
            by += float(esz) * (M * st * st * ci + M * co * (1 + ep) + k * k * ci * co)
            ex = gflop * 1e9 * (16.0 / 36.0 if alg.startswith("winograd") else 1.0)
        else:                                   # the stem: no per-launch shape record, fp32 MFMA (bf16 MFMA in the bf16-tensor mode)
            ex = gflop * 1e9; prod = "bf16" if esz == 2 else "fp32"; alg = "stem"
        rate = {"fp32": MFMA_F32_PEAK_TFLOPS, "split3": MFMA_BF16_PEAK_TFLOPS / 6.0, "split2": MFMA_BF16_PEAK_TFLOPS / 3.0, "bf16": MFMA_BF16_PEAK_TFLOPS}[prod] * 1e12
        floor_s += ex / rate
        forms[prod] = forms.get(prod, 0) + 1
    return {"algorithmic_bytes_per_step": by / nsteps, "floor_ms": floor_s / nsteps * 1e3, "launches_by_product_form": {k: v / nsteps for k, v in forms.items()}}


def bf16_conv_roofline(md, torch, model, step, esz=4, precision="bf16"):
    """The convolutions (forward + input gradient) of a bf16 mode on both rooflines: the class sits far below the bf16 matrix peak and is priced against HBM as well.
    ALGORITHMIC bytes = each launch reads its input pixels, its weights and the output-sized operands of its fused epilogue once and writes its output once at the tensors' storage width `esz` (2: bf16 activation storage,
    precision 'bf16_storage'; 4: the bf16-operand mode on fp32 tensors) -- from the per-launch records of the library profiler (M output pixels, Cin, Cout, k, stride) -- over
    the class time measured with HIP events.  (The stem is in the class time and flops but has no per-launch record: its bytes are not counted.)"""
    import re, tempfile
    md.set_two_streams(model, False); md.profile(model, True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ig = md.profile_get(model, 0); lt = md.profile_get(model, 3)
    tmp = os.path.join(tempfile.gettempdir(), "md2_bf16_layers_%d.csv" % os.getpid())
    md.profile_dump(model, tmp); md.profile(model, False); md.set_two_streams(model, True)
    by = 0.0
    for line in open(tmp).read().splitlines()[1:]:
        f = line.split(",", 4)
        m = re.match(r"igemm M=(\d+) Cout=(\d+) Cin=(\d+) k=(\d+) s=(\d+) .* ep=(\d+)", f[4]) if len(f) == 5 and f[0] == "0" else None
        if m:
            M, co, ci, k, st, ep = (int(v) for v in m.groups())
            by += float(esz) * (M * st * st * ci + M * co * (1 + ep) + k * k * ci * co)          # ep: output-sized tensors the fused epilogue reads (residual, activation source, BatchNorm inputs, accumulated y)
    os.remove(tmp)
    if ig["ms"] <= 0:
        return None
    traffic_step = None; loss_traffic = None; loss_valu = None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_cfg5_" + precision + "_pmc_hbm_traffic.json")))
        if d.get("workload") == {"batch": 8, "channels": 3, "height": 320, "width": 1024, "depth": 18, "precision": precision}:
            traffic_step = d["bytes_per_step"]["conv_igemm"]; loss_traffic = d["bytes_per_launch"].get("loss_train")
        dv = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_cfg5_loss_pmc.json")))
        if dv.get("workload") == {"batch": 8, "channels": 3, "height": 320, "width": 1024}:
            loss_valu = dv.get("loss_train")
    except Exception:
        pass
    tf = ig["flops"] / (ig["ms"] * 1e-3) / 1e12; gbs = by / (ig["ms"] * 1e-3) / 1e9
    loss_block = None
    if lt["ms"] > 0 and lt["launches"] > 0:
        # loss_train_kernel<3, 16> (three colour channels): its own roofline entry since round 5 -- until round 4 this instantiation kept 224 bytes per lane in scratch and moved
        # 4.8 x its algorithmic bytes; `traffic` (PMC) against `algorithmic_bytes_per_launch` = 4 P (2 + 3 C) S B is the check that it no longer does
        ms1 = lt["ms"] / lt["launches"]; a = lt["bytes"] / lt["launches"] / (ms1 * 1e-3) / 1e9
        loss_block = {"kernel": "loss_train_kernel<3, 16> (value + pullback of all four scales from one pass over the warp, C = 3)", "bound": "valu" if loss_valu else "hbm",
                      "ms_per_launch": ms1, "hbm": {"achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": lt["bytes"] / lt["launches"], "traffic": loss_traffic,
                                                    "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/" + PROFILE_ROUND + "_cfg5_" + precision + "_pmc_hbm_traffic.md)"}}
        if loss_valu:
            rate = loss_valu * 64.0 / (ms1 * 1e-3) / 1e12
            loss_block.update({"achieved": rate, "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s", "frac": rate / VALU_PEAK_TLANE, "valu_wave_instructions_per_launch": loss_valu,
                               "counter": "SQ_INSTS_VALU of this kernel at this workload (rocprofv3 --pmc, profiles/" + PROFILE_ROUND + "_cfg5_loss_train_pmc.md)"})
    return {"roofline_loss_train": loss_block,
            "kernel": "forward + input-gradient convolutions (v_mfma_f32_32x32x16_bf16 / 16x16x32, fp32 accumulation) on " + ("bf16 tensors" if esz == 2 else "fp32 tensors"), "bound": "hbm" if gbs / HBM_PEAK_GBS > tf / MFMA_BF16_PEAK_TFLOPS else "mfma",
            "mfma": {"achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS},
            "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_step": by / 3, "traffic": traffic_step,
                    "traffic_unit": "HBM bytes per step of this class (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of tools/prof_config.py 8 18 3 320 1024 " + precision + ", profiles/" + PROFILE_ROUND + "_cfg5_" + precision + "_pmc_hbm_traffic.md)"},
            "ms_per_step": ig["ms"] / 3, "launches_per_step": ig["launches"] / 3,
            "note": "neither roof binds: 64 x 64-pixel-by-channel workgroups with two bf16 MFMAs per wave and 64-byte K-step are issue / latency bound (DESIGN.md section 4, bf16 tensors)"}


def conv_roofline_live(md, torch, model, step, esz=4, peak_tf=MFMA_F32_PEAK_TFLOPS):
    """The forward + input-gradient convolution class of any configuration, measured live: three profiled one-stream steps (HIP events per launch), algorithmic FLOP rate,
    the matrix-pipe floor of the executed instructions (conv_class_accounting) and the class's algorithmic bytes.  No committed counters involved."""
    import tempfile
    md.set_two_streams(model, False); md.profile(model, True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ig = md.profile_get(model, 0); wg = md.profile_get(model, 1)
    tmp = os.path.join(tempfile.gettempdir(), "md2_live_layers_%d.csv" % os.getpid())
    md.profile_dump(model, tmp); acct = conv_class_accounting(tmp, esz, 3); os.remove(tmp)
    md.profile(model, False); md.set_two_streams(model, True)
    if ig["ms"] <= 0:
        return None
    ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12; ms = ig["ms"] / 3
    return {"bound": "mfma", "kernel": "forward + input-gradient convolutions of this configuration (same kernel families as the headline roofline)",
            "achieved": ach, "peak": (ig["flops"] / 3) / (acct["floor_ms"] * 1e-3) / 1e12, "unit": "TFLOP/s", "frac": acct["floor_ms"] / ms, "frac_definition": "floor_ms / ms_per_step",
            "algorithmic_vs_fp32_mfma_peak": ach / peak_tf, "floor_ms": acct["floor_ms"], "ms_per_step": ms, "ms_over_floor": ms / acct["floor_ms"],
            "launches_per_step": ig["launches"] / 3, "launches_by_product_form": acct["launches_by_product_form"],
            "algorithmic_bytes_per_launch": acct["algorithmic_bytes_per_step"] / max(ig["launches"] / 3, 1), "hbm_frac_of_algorithmic_bytes": acct["algorithmic_bytes_per_step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "weight_gradient_ms_per_step": wg["ms"] / 3,
            "timing": "HIP events per launch, 3 one-stream steps inside this run"}


def valu_insts(kernel, B, C, H, W):
    """SQ_INSTS_VALU per launch of a kernel / kernel class from the committed PMC summary (profiles/<round>_loss_pmc.json: {"workload": {...}, "loss_train": n, "conv_wgrad": n, ...});
    None when the workload differs from the one measured."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_loss_pmc.json")))
    except Exception:
        return None
    w = d.get("workload", {})
    if (w.get("batch"), w.get("channels"), w.get("height"), w.get("width")) != (B, C, H, W):
        return None
    return d.get(kernel)


def cpu_baseline(C, H, W):
    """The oracle (CPU restatement of the reference algorithm, op by op and unfused -- not Flux itself) timed on this box's host cores on the reference's own
    training configuration, batch 4 (src/Monodepth.jl:104), 416x128, forward + loss + backward + ADAM: 3 warm-up steps, then 10 timed steps, median
    (~7 s per step on the GPU box's 128 threads, so this leg takes ~1.5 min)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    import md2amd
    syn = md2amd.synthetic
    B = 4
    theta = syn.init_params(C, seed=42)
    x = syn.synthetic_triplets(B, C, H, W, seed=42)
    K, iK = syn.depth10k_intrinsics(W, H)
    cfg = oracle.make_cfg(B, C, H, W, K, iK)
    net = oracle.OracleNet(C); net.set_params(theta)
    threads = oracle.num_threads()
    for _ in range(3):
        net.train_loss(x, cfg, None, True); net.adam_step(1e-4)
    times = []
    for _ in range(10):
        t0 = time.time(); net.train_loss(x, cfg, None, True); net.adam_step(1e-4); times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": B / med, "unit": "triplets/s", "cores": int(threads), "kind": "port",
            "sample": "B=4 triplets (the reference's batch size), %dx%d, C=%d, fwd+loss+bwd+ADAM, 3 warm-up + %d timed steps (median %.2f s/step) on %d OpenMP threads; CPU restatement of the reference algorithm, op by op -- not Flux itself, and not the optimisation target" % (W, H, C, len(times), med, int(threads))}


def spawn_ranks(n, timeout_s=float(os.environ.get("MD2_BENCH_TIMEOUT", "900"))):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (fresh processes; this parent never touches a GPU), relay rank 0's JSON line.
    The children are polled: the first one that exits non-zero (or the overall timeout) ends the others -- a rank that died at init would otherwise leave its siblings
    waiting in a collective, and this parent waiting for them."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=(out0 if r == 0 else subprocess.DEVNULL)))
    t_end = time.time() + timeout_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > t_end:
            rc = abs(bad[0]) if bad else 124
            for p in procs:                          # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill(); p.wait()
            sys.stderr.write("bench.py: %s -- the remaining ranks were ended\n" % ("rank exit codes %s" % codes if bad else "no result after %.0f s" % timeout_s))
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    out0.seek(0); sys.stdout.write(out0.read().decode()); sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="triplets per GPU per step")
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--depth", type=int, default=18, help="encoder: ResidualNetwork(18) or (50)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16", "bf16_storage"], help="f32 = the reference's precision (parity mode); bf16 = bf16 MFMA operands, fp32 everything else; bf16_storage = activations stored in bf16 too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import md2amd
    md = md2amd.monodepth
    par = md2amd.parallel
    syn = md2amd.synthetic

    # one rank per GPU over RCCL.  MD2_BENCH_BACKEND=gloo + MD2_BENCH_DEVICE=0 exist only to rehearse the multi-rank code path on a
    # one-GPU box (two processes sharing the card, all-reduce through the host); the driver's runs never set them.
    rank, world, local = par.init_from_env(os.environ.get("MD2_BENCH_BACKEND", "nccl"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if "MD2_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MD2_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    B, C, H, W = args.batch, args.channels, args.height, args.width

    def setup(batch, depth=args.depth, C=C, H=H, W=W, precision=args.precision, automask=False):
        K, iK = syn.depth10k_intrinsics(W, H)
        params = md.Params(target_size=(W, H), batch_size=batch, automasking=automask)     # train(): automasking=false (src/Monodepth.jl:105)
        cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
        enc = md.ResidualNetwork(depth, in_channels=C)
        model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), lr=1e-4, precision=precision)
        model.load_params(syn.init_params(C, seed=42, depth=depth))                         # same weights on every rank
        x = torch.from_numpy(syn.synthetic_triplets(batch, C, H, W, seed=42 + rank)).cuda()          # `batch` distinct triplets per rank
        return model, params, cache, x

    model, params, cache, x = setup(B)
    # N > 1: the library's own RCCL communicator (md2_comm_*: bucketed all-reduce overlapped with the pullback, ADAM on the mean) -- the path a Julia host uses.
    # It is checked on a known pattern before the timed region; if it cannot be brought up the step falls back to torch.distributed's RCCL all-reduce of the same
    # buffer, and says so in the JSON line.
    comm_note = "single rank"
    backend = os.environ.get("MD2_BENCH_BACKEND", "nccl")
    dp = par.DataParallel(model, comm="torch")
    rccl_ranks = 1
    if world > 1 and backend == "nccl":
        # the library's communicator is VERIFIED before it is timed: every rank steps once on the same batch, first alone (local gradient), then with the communicator
        # attached -- the bucketed, overlapped all-reduce inside md2_backward must return world x the local gradient on every rank (parallel.verify_rccl)
        # Every rank takes the SAME sequence of collectives whatever happens locally: a rank that fails says so in an agreement all-reduce (MIN of a flag over
        # torch.distributed) BEFORE the next library collective would be entered, and then all ranks fall back together.
        def agree(ok):
            flag = torch.tensor([1.0 if ok else 0.0], device="cuda"); torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            return float(flag.item()) > 0.5
        why = ""; dpr = None; g_local = None; xc = None
        try:                                        # (1) local step, no collective
            xc = torch.from_numpy(syn.synthetic_triplets(min(B, 8), C, H, W, seed=7)).cuda().repeat((B + 7) // 8, 1, 1, 1, 1)[:B].contiguous()
            md.train_step(model, xc, None, cache, params, do_update=False)
            g_local = model.grad_buffer().float().clone()
            ok = True
        except Exception as e:                      # noqa: BLE001
            ok = False; why = str(e)[:200]
        if agree(ok):
            try:                                    # (2) communicator: id broadcast, ncclCommInitRank, state broadcast
                dpr = par.DataParallel(model, comm="rccl")
                dpr._setup(x, params, cache)
                ok = model.comm_world()[1] == world
                why = why or ("" if ok else "communicator has %d ranks" % model.comm_world()[1])
            except Exception as e:                  # noqa: BLE001
                ok = False; why = str(e)[:200]
            if agree(ok):
                try:                                # (3) one step through the bucketed, overlapped all-reduce + the MAX all-reduce of verify_rccl
                    ok, err = par.verify_rccl(dpr, xc, cache, params, g_local)
                    why = why or ("" if ok else "reduced gradient differs from world x local gradient (rel %.2e)" % err)
                except Exception as e:              # noqa: BLE001
                    ok = False; why = str(e)[:200]
                ok = agree(ok)
            else:
                ok = False
        else:
            ok = False
        if ok:
            dp = dpr; rccl_ranks = world
            comm_note = "libmd2hip RCCL communicator (md2_comm_init), 4 buckets overlapped with the pullback; verified against the local gradient before timing (rel %.1e)" % err
        else:
            comm_note = "torch.distributed RCCL all-reduce of the library's gradient buffer (library communicator not used: %s)" % (why or "another rank could not verify md2_comm")
            try:
                model.comm_destroy()
            except Exception:                       # noqa: BLE001
                model._comm_spec = None
            dp = par.DataParallel(model, comm="torch")
        if dp.comm != "rccl":
            rccl_ranks = 0
    elif world > 1:
        comm_note = "torch.distributed all-reduce (%s backend)" % backend
        rccl_ranks = 0

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = dp.train_step(x, None, cache, params)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = dp.train_step(x, None, cache, params)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)
    value = world * B * args.steps / dt

    out = {"metric": "self-sup training imgs/sec, 416x128 ResNet-18 (1 img = 1 triplet: 3 frames through the encoder)", "value": value, "unit": "triplets/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
           "dtype_note": ("fp32 storage and fp32 accumulation everywhere; every convolution product carries all 24 mantissa bits of both fp32 operands -- per launch either the fp32 MFMA "
                          "(v_mfma_f32_32x32x2_f32) or its exact decomposition (each operand = hi + mid + lo bf16 pieces, six v_mfma_f32_32x32x16_bf16 with fp32 accumulation), whichever is faster "
                          "at the launch's shape (md2_set_conv_products(1, 1), the library default); no operand is rounded to fewer bits than the reference's Float32 anywhere. "
                          "extra.headline_shape_fp32_mfma_only is the same step with v_mfma_f32_32x32x2_f32 in every convolution") if args.precision == "f32" else "bf16 MFMA operands, fp32 accumulation / storage / master weights",
           "data": "synthetic",
           "config": {"workload": "ResNet-%d depth+pose, %dx%d, C=%d, batch=%d triplets/GPU, %s fwd+loss+bwd+ADAM (BASELINE.json configs[2] per-GPU shape; configs[1] batch=4, configs[3] ResNet-50 640x192 batch=16 and configs[4] 1024x320 bf16 automask under extra)" % (
                          args.depth, W, H, C, B, "fp32" if args.precision == "f32" else "bf16 MFMA operands / fp32 accumulate"),
                      "global_batch": world * B, "parallelism": "dp%d" % world, "gradient_exchange": comm_note, "rccl_ranks": rccl_ranks, "loss": final_loss,
                      "conv_products": "md2_set_conv_products%s: 1 = per launch the faster of fp32 MFMA and the exact three-bf16-piece split (both full fp32 operand accuracy)" % (md.get_conv_products(),)}}

    # ---- roofline of the dominant kernel class, measured live with HIP events on the launch stream (3 profiled steps)
    # The timed region above overlaps the weight-gradient kernels with the rest of the backward pass on a second stream; overlapped
    # launches inflate each other's duration, so the per-kernel roofline is timed with the overlap switched off (same kernels, same
    # data, one at a time).  profiles/ holds the rocprofv3 summaries of both modes.  Every rank runs these steps (the gradient exchange is
    # a collective), rank 0 reports its own kernels.
    md.set_two_streams(model, False)
    md.profile(model, rank == 0)
    for _ in range(3):
        dp.train_step(x, None, cache, params)
    sync()
    if rank == 0:
        import tempfile
        ig, wg, lf, lb = (md.profile_get(model, c) for c in range(4))
        tmp = os.path.join(tempfile.gettempdir(), "md2_bench_layers_%d.csv" % os.getpid())
        md.profile_dump(model, tmp); acct = conv_class_accounting(tmp, 2 if args.precision == "bf16_storage" else 4, 3); os.remove(tmp)
        md.profile(model, False)
    md.set_two_streams(model, True)
    if rank == 0:
        traffic = pmc_traffic(B, C, H, W, args.depth, args.precision)
        ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        peak_tf = MFMA_BF16_PEAK_TFLOPS if args.precision != "f32" else MFMA_F32_PEAK_TFLOPS
        ach_x = ig["flops_executed"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        peak_eff = (ig["flops"] / 3) / (acct["floor_ms"] * 1e-3) / 1e12 if acct["floor_ms"] > 0 else peak_tf
        out["roofline"] = {"bound": "mfma", "kernel": "conv_wino3s_kernel<.,.,3> / conv_wino3_kernel (fused Winograd F(2x2,3x3): the 3x3 stride-1 layers; products as the exact three-bf16-piece split from 96 input channels up, fp32 MFMA below) + conv_igemm_kernel<*> + conv_igemm_classes_kernel<*> + conv3x3_patch_kernel<*> + stem_fwd_kernel (forward + input-gradient convolutions; fp32 v_mfma_f32_32x32x2_f32 / 16x16x4_f32 outside the split launches)",
                           "achieved": ach, "peak": peak_eff, "unit": "TFLOP/s", "frac": ach / peak_eff, "frac_of_executed_floor": ach / peak_eff,
                           "frac_definition": "floor_ms / ms_per_step since round 5 (BENCH_r05 on); rounds 1-4 reported algorithmic_vs_fp32_mfma_peak under this key -- the two are not comparable",
                           "algorithmic": "achieved counts the direct-convolution FLOPs of the layers (SURVEY.md section 8d: 2 M Cout taps Cin), whatever algorithm ran",
                           "peak_note": "peak = the same algorithmic FLOPs / floor_ms: the rate at which the matrix instructions this launch mix EXECUTES could retire (fp32 MFMA launches at 157.3 TFLOP/s; exact-split launches issue six bf16 MFMAs per product group at 2.5 PFLOP/s; Winograd launches execute 16/36 of the direct count) -- since round 5, because the algorithmic rate against the fp32 MFMA peak alone had saturated (round 4: 0.99; it is kept as algorithmic_vs_fp32_mfma_peak)",
                           "algorithmic_vs_fp32_mfma_peak": ach / peak_tf,
                           "executed_achieved": ach_x, "executed_frac": ach_x / peak_tf,
                           "executed": "FLOPs the matrix cores really issued: Winograd launches multiply 16 instead of 36 times per 2x2 output tile -- this is the MFMA pipe utilisation",
                           "floor_ms": acct["floor_ms"], "ms_over_floor": (ig["ms"] / 3) / acct["floor_ms"] if acct["floor_ms"] > 0 else None,
                           "floor": "ms per step the class's EXECUTED matrix instructions need at the rate of the pipe each launch runs on (fp32 MFMA 157.3 TFLOP/s; exact-split launches: six bf16 MFMAs per product group at 2.5 PFLOP/s; Winograd: 16/36 of the direct count) -- ms_per_step / floor_ms is the distance to the matrix-pipe roof; what stands between the two is operand delivery (L2 -> CU bandwidth, DESIGN.md section 4), workgroup prologues / epilogues and launch tails",
                           "launches_by_product_form": acct["launches_by_product_form"],
                           "traffic": traffic.get("conv_igemm"),
                           "algorithmic_bytes_per_launch": acct["algorithmic_bytes_per_step"] / max(ig["launches"] / 3, 1),
                           "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of this command, profiles/" + PROFILE_ROUND + "_pmc_hbm_traffic.md); algorithmic_bytes_per_launch: inputs + weights + epilogue operands + output, each once, fp32",
                           "matrix_pipe": (lambda m: dict(m, counter="SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles) over the class's launches (fp32 and bf16 MFMAs alike; profiles/" + PROFILE_ROUND + "_mfma_pmc.md)") if m else None)(
                               (lambda: (json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_mfma_pmc.json"))).get("conv_igemm") if os.path.exists(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_mfma_pmc.json")) else None))()),
                           "launches_per_step": ig["launches"] / 3, "ms_per_step": ig["ms"] / 3, "avg_launch_ms": ig["ms"] / max(ig["launches"], 1),
                           "timing": "HIP events per launch on the launch stream, 3 steps with the second (weight-gradient) stream disabled; `value` is measured with it enabled"}
        extra = {"frames_per_s": 3.0 * value}         # one triplet = 3 frames through the encoder
        if wg["ms"] > 0:
            a = wg["flops"] / (wg["ms"] * 1e-3) / 1e12
            ax = wg["flops_executed"] / (wg["ms"] * 1e-3) / 1e12
            ent = {"kernel": "conv_wino_wgrad_kernel / conv_wino_wgrad_s_kernel<., 3> (Winograd domain: the 3x3 stride-1 layers) + conv_wgrad_kernel<*, 3> (1x1 / stride-2 layers, exact three-piece split) + conv3x3_wgrad_*_kernel + stem_wgrad_kernel (fp32 MFMA); every product at full fp32 operand accuracy",
                   "fp32_equivalent": {"achieved": a, "peak": peak_tf, "unit": "TFLOP/s", "frac": a / peak_tf, "executed_achieved": ax, "executed_frac": ax / peak_tf,
                                       "note": "algorithmic (direct-convolution) and executed (Winograd 16/36) FLOP rates against the fp32 MFMA peak -- a rate, not a utilisation: the split launches run on the bf16 pipe"},
                   "ms_per_step": wg["ms"] / 3, "launches_per_step": wg["launches"] / 3}
            # The class is issue-bound: most of its time runs the fp32 MFMA kernels, whose matrix instructions and vector instructions (transforms, address arithmetic) share one
            # issue stream (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for v_mfma_f32_32x32x2_f32 on gfx950), so the roof is matrix-pipe busy + vector issue, from the counters of this workload.
            vi = valu_insts("conv_wgrad", B, C, H, W)
            mp = None
            try:
                mp = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_mfma_pmc.json"))).get("conv_wgrad")
            except Exception:
                pass
            ent["bound"] = "issue (matrix pipe + vector instructions, serialised for the fp32 MFMA)"
            if vi:
                rate = vi * 64.0 * (wg["launches"] / 3) / (wg["ms"] / 3 * 1e-3) / 1e12
                ent["valu"] = {"achieved": rate, "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s", "frac": rate / VALU_PEAK_TLANE, "valu_wave_instructions_per_launch": vi,
                               "counter": "SQ_INSTS_VALU, class average per launch at this workload (rocprofv3 --pmc, profiles/" + PROFILE_ROUND + "_loss_train_pmc.md)"}
            if mp:
                ent["matrix_pipe"] = dict(mp, counter="SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles), class total at this workload (profiles/" + PROFILE_ROUND + "_mfma_pmc.md)")
                ent["frac"] = mp["matrix_pipe_busy"] * (1.0 - mp["coexec_share"]) + (ent["valu"]["frac"] if vi else 0.0)
                ent["frac_note"] = "share of SIMD issue time accounted for: matrix-pipe busy (not co-executing) + vector issue; the remainder is waits (LDS / global latency at 2 waves per SIMD, barriers) and launch tails"
            extra["roofline_conv_wgrad"] = ent
        # the training step runs the loss as ONE class: smoothness sums + the fused value-and-pullback kernel ("loss_train"); "loss_fwd" only exists in forward-only calls
        for nm, p in (("roofline_loss_fwd", lf), ("roofline_loss_train", lb)):
            if p["ms"] > 0:
                a = p["bytes"] / (p["ms"] * 1e-3) / 1e9
                ent = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "ms_per_step": p["ms"] / 3,
                       "algorithmic_bytes_per_launch": p["bytes"] / max(p["launches"], 1),
                       "traffic": traffic.get(nm[len("roofline_"):])}
                vi = valu_insts(nm[len("roofline_"):], B, C, H, W)
                if vi:          # the kernel is bound by vector-instruction issue, not by HBM (profiles/r0x_loss_train_pmc.md): report that roofline, keep the HBM fraction beside it
                    rate = vi * 64.0 / (p["ms"] / max(p["launches"], 1) * 1e-3) / 1e12          # T lane-instructions / s
                    ent = {"bound": "valu", "achieved": rate, "peak": VALU_PEAK_TLANE, "unit": "T lane-instr/s", "frac": rate / VALU_PEAK_TLANE,
                           "valu_wave_instructions_per_launch": vi, "counter": "SQ_INSTS_VALU of this kernel at this workload (rocprofv3 --pmc, profiles/)",
                           "ms_per_step": p["ms"] / 3, "hbm": {"achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                                                               "algorithmic_bytes_per_launch": p["bytes"] / max(p["launches"], 1), "traffic": traffic.get(nm[len("roofline_"):])}}
                extra[nm] = ent
        if world == 1 and not args.no_extra and B != 4:
            # BASELINE.json configs[1]: batch = 4 on one MI355X
            del dp, model
            gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()          # (handles freed here, not by a collection inside a timed loop)
            m4, p4, c4, x4 = setup(4)
            for _ in range(5):
                md.train_step(m4, x4, None, c4, p4, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n4 = 20
            for _ in range(n4):
                md.train_step(m4, x4, None, c4, p4, True)
            torch.cuda.synchronize(); d4 = time.perf_counter() - t0
            extra["config1_batch4"] = {"value": 4 * n4 / d4, "unit": "triplets/s", "ms_per_step": d4 / n4 * 1e3}
            # BASELINE.json configs[3]: ResNet-50 encoder, 640x192, batch = 16 per GPU (RGB)
            del m4, x4
            gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()          # (handles freed here, not by a collection inside a timed loop)
            m5, p5, c5, x5 = setup(16, depth=50, C=3, H=192, W=640)
            for _ in range(3):
                md.train_step(m5, x5, None, c5, p5, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n5 = 10
            for _ in range(n5):
                md.train_step(m5, x5, None, c5, p5, True)
            torch.cuda.synchronize(); d5 = time.perf_counter() - t0
            extra["config3_resnet50_640x192_batch16"] = {"value": 16 * n5 / d5, "unit": "triplets/s", "ms_per_step": d5 / n5 * 1e3}
            extra["config3_resnet50_640x192_batch16"]["roofline"] = conv_roofline_live(md, torch, m5, lambda: md.train_step(m5, x5, None, c5, p5, True))
            # ... and the same configuration with bf16 activation storage (md2_config.precision = 2; not a parity mode): the bottleneck trunk is HBM-bound in its 1x1 layers
            del m5, x5
            gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()          # (handles freed here, not by a collection inside a timed loop)
            m5, p5, c5, x5 = setup(16, depth=50, C=3, H=192, W=640, precision="bf16_storage")
            for _ in range(3):
                md.train_step(m5, x5, None, c5, p5, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n5):
                md.train_step(m5, x5, None, c5, p5, True)
            torch.cuda.synchronize(); d5 = time.perf_counter() - t0
            extra["config3_resnet50_640x192_batch16_bf16_storage"] = {"value": 16 * n5 / d5, "unit": "triplets/s", "ms_per_step": d5 / n5 * 1e3, "precision": "bf16_storage",
                                                                      "dtype": "bf16 activation / gradient tensors and weight copies, bf16 MFMA, fp32 accumulate (not a parity mode)"}
            # BASELINE.json configs[4]: 1024x320, bf16 mixed precision + min-reprojection automask, batch = 8 per GPU.  NOT a parity mode (the reference computes in Float32):
            # md2_config.precision = 2 -- activations and their gradients are bf16 tensors in HBM, the convolutions read bf16 weight copies and run bf16 MFMAs with fp32
            # accumulation; master weights, BatchNorm statistics, disparities, loss and ADAM stay fp32; tests bound its loss to 2 % of the fp32 oracle and its gradient's direction.
            # The round-3 definition (precision 1: only the MFMA operands are bf16, every tensor fp32) is timed beside it.
            del m5, x5
            gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()          # (handles freed here, not by a collection inside a timed loop)
            for key, prec, esz, dtype in (("config4_1024x320_bf16_automask_batch8", "bf16_storage", 2, "bf16 activation / gradient tensors and weight copies, bf16 MFMA, fp32 accumulate; fp32 master weights, BatchNorm statistics, loss, ADAM"),
                                          ("config4_1024x320_bf16_operands_fp32_tensors_batch8", "bf16", 4, "bf16 MFMA operands, fp32 tensors and accumulation (md2_config.precision = 1)")):
                m6, p6, c6, x6 = setup(8, depth=18, C=3, H=320, W=1024, precision=prec, automask=True)
                a6 = md.automasking_loss(c6.ssim, x6, x6[:, 1].contiguous(), [1, 3])
                for _ in range(3):
                    md.train_step(m6, x6, a6, c6, p6, True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n6 = 10
                for _ in range(n6):
                    a6 = md.automasking_loss(c6.ssim, x6, x6[:, 1].contiguous(), [1, 3])        # recomputed per batch like src/Monodepth.jl:159-164
                    md.train_step(m6, x6, a6, c6, p6, True)
                torch.cuda.synchronize(); d6 = time.perf_counter() - t0
                extra[key] = {"value": 8 * n6 / d6, "unit": "triplets/s", "ms_per_step": d6 / n6 * 1e3, "dtype": dtype, "precision": prec}
                if esz == 2:
                    rf = bf16_conv_roofline(md, torch, m6, lambda: md.train_step(m6, x6, a6, c6, p6, True), esz=esz, precision=prec)
                    if rf:
                        extra[key]["roofline_loss_train"] = rf.pop("roofline_loss_train", None)
                    extra[key]["roofline"] = rf
                del m6, x6
                gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()          # (handles freed here, not by a collection inside a timed loop)
            # and the headline shape in bf16 activation storage, for reference next to `value` (which is fp32)
            m7, p7, c7, x7 = setup(B, precision="bf16_storage")
            for _ in range(3):
                md.train_step(m7, x7, None, c7, p7, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n7 = 10
            for _ in range(n7):
                md.train_step(m7, x7, None, c7, p7, True)
            torch.cuda.synchronize(); d7 = time.perf_counter() - t0
            extra["headline_shape_bf16_storage"] = {"value": B * n7 / d7, "unit": "triplets/s", "ms_per_step": d7 / n7 * 1e3, "dtype": "bf16 activation / gradient tensors and weight copies, bf16 MFMA, fp32 accumulate (md2_config.precision = 2; not a parity mode)"}
            # the same step under the other md2_set_conv_products modes.  Full accuracy: fp32 MFMA in every convolution (0, 0) and the exact three-piece split wherever it exists
            # (3, 3) -- `value` is their per-launch mix (1, 1).  Reduced precision, opt-in, never part of `value`: two bf16 pieces per operand (~16-bit mantissas) in the weight
            # gradients only (1, 2) -- round 3's default -- and in every Winograd product (2, 2).
            del m7, x7
            gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()          # (handles freed here, not by a collection inside a timed loop)
            dflt = md.get_conv_products()
            m8, p8, c8, x8 = setup(B)
            n8 = 10
            for key, mode, note in (("headline_shape_fp32_mfma_only", (0, 0), "v_mfma_f32_32x32x2_f32 in every convolution (full accuracy)"),
                                    ("headline_shape_exact_split_everywhere", (3, 3), "exact three-bf16-piece products (six bf16 MFMAs, fp32 accumulation) in every Winograd / per-tap weight-gradient launch (full accuracy)"),
                                    ("headline_shape_reduced_two_piece_wgrad", (1, 2), "REDUCED precision (opt-in): two bf16 pieces per operand in the weight gradients only"),
                                    ("headline_shape_reduced_two_piece_all", (2, 2), "REDUCED precision (opt-in): two bf16 pieces per operand in every Winograd and weight-gradient product")):
                md.set_conv_products(*mode)
                gc.collect()
                for _ in range(3):
                    md.train_step(m8, x8, None, c8, p8, True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n8):
                    md.train_step(m8, x8, None, c8, p8, True)
                torch.cuda.synchronize(); d8 = time.perf_counter() - t0
                extra[key] = {"value": B * n8 / d8, "unit": "triplets/s", "ms_per_step": d8 / n8 * 1e3, "conv_products": list(mode), "dtype": note}
            md.set_conv_products(*dflt)
        out["extra"] = extra
        # which fields of this line are NOT measured by this run: hardware counters cannot be read from inside bench.py, so they come from the committed rocprofv3 --pmc
        # summaries of the same command (profiles/<PROFILE_ROUND>_*.json, reported only when their workload equals the one being run)
        out["from_committed_profiles"] = {
            "round": PROFILE_ROUND,
            "fields": ["roofline.traffic", "roofline.matrix_pipe", "extra.roofline_conv_wgrad.valu", "extra.roofline_conv_wgrad.matrix_pipe", "extra.roofline_conv_wgrad.frac",
                       "extra.roofline_loss_train.achieved / .frac / .valu_wave_instructions_per_launch (SQ_INSTS_VALU)", "extra.roofline_loss_train.hbm.traffic", "extra.roofline_loss_fwd.traffic",
                       "extra.config4_1024x320_bf16_automask_batch8.roofline.hbm.traffic", "extra.config4_1024x320_bf16_automask_batch8.roofline_loss_train.hbm.traffic / .valu_wave_instructions_per_launch"],
            "measured_live": "everything else: value, ms_per_step, every class time / launch count / FLOP and byte rate (HIP events), floor_ms, frac, cpu_baseline"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, H, W)
        # the figures a reader needs first, repeated at the END of the line (a log viewer that keeps only the tail of stdout still shows them)
        pick = lambda k: ({"value": extra[k]["value"], "ms_per_step": extra[k]["ms_per_step"]} if k in extra else None)
        out["summary"] = {"value": value, "unit": "triplets/s", "ms_per_step": dt / args.steps * 1e3, "n_gpus": world, "roofline_frac": out["roofline"]["frac"],
                          "conv_ms_per_step": out["roofline"]["ms_per_step"], "conv_floor_ms": out["roofline"]["floor_ms"],
                          "config1_batch4": pick("config1_batch4"), "config3_resnet50_640x192_batch16": pick("config3_resnet50_640x192_batch16"),
                          "config4_1024x320_bf16_automask_batch8": pick("config4_1024x320_bf16_automask_batch8"), "headline_shape_fp32_mfma_only": pick("headline_shape_fp32_mfma_only"),
                          "cpu_baseline": (out.get("cpu_baseline") or {}).get("value")}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
