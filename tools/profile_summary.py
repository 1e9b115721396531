"""Turn rocprofv3 CSV output into the markdown summaries kept under profiles/.

  kernel stats :  profile_summary.py stats  <dir with *_kernel_stats.csv>  <steps in trace>  <title line>  > profiles/rNN_..._kernel_stats.md
  HBM traffic  :  profile_summary.py pmc    <dir of the FETCH_SIZE pass>  <dir of the WRITE_SIZE pass>  <steps in trace>  [out.json B C H W depth precision]  > profiles/rNN_pmc_hbm_traffic.md

FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE tallies a 128-B request of a wide coalesced read as 64 B, so reads are
doubled (MI355X_MICROARCH.md, "HBM" section); the two counters do not fit one pass, hence two runs of the same command.
"""
import csv, glob, os, sys, collections


def klass(name):
    # the classes bench.py's HIP-event profile uses: forward + input-gradient convolutions / weight gradients (incl. the direct stem kernels)
    if "conv_wgrad_kernel" in name or "conv_wgrad_s2_bs_kernel" in name or "conv_wino_wgrad" in name or "conv3x3_wgrad" in name or "stem_wgrad_" in name:
        return "conv_wgrad"
    if "conv_igemm" in name or "conv_wino" in name or "conv3x3_patch_kernel" in name or "stem_fwd_kernel" in name:
        return "conv_igemm"
    if "loss_fwd_kernel" in name:
        return "loss_fwd"
    if "loss_train_kernel" in name or "loss_bwd_kernel" in name:
        return "loss_train"
    return "other"


def stats(d, steps, title):
    f = glob.glob(d + "/*kernel_stats.csv")[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# " + title)
    print()
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in rows:
        if float(r["TotalDurationNs"]) / tot < 0.004:
            continue
        print("| `%s` | %s | %.2f | %.1f | %.2f |" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = klass(r["Name"]); agg[k][0] += int(r["Calls"]); agg[k][1] += float(r["TotalDurationNs"])
    print()
    print("sum of kernel durations: %.2f ms/step over %d steps in the trace" % (tot / 1e6 / steps, steps))
    print()
    print("| class | launches/step | ms/step | avg us |")
    print("|---|---|---|---|")
    for k in ("conv_igemm", "conv_wgrad", "loss_fwd", "loss_train", "other"):
        n, t = agg[k]
        print("| %s | %.1f | %.3f | %.1f |" % (k, n / steps, t / 1e6 / steps, t / 1e3 / max(n, 1)))


def pmc(df, dw, steps):
    def load(d, counter):
        f = glob.glob(d + "/*counter_collection.csv")[0]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = klass(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        return agg
    fe, wr = load(df, "FETCH_SIZE"), load(dw, "WRITE_SIZE")
    print("| kernel class | launches/step | FETCH_SIZE KiB/step (raw) | read MB/step (x2 corrected) | WRITE_SIZE MB/step | read+write MB per launch |")
    print("|---|---|---|---|---|---|")
    for k in ("conv_igemm", "conv_wgrad", "loss_fwd", "loss_train", "other"):
        n = fe[k][0] / steps
        rd = 2 * fe[k][1] * 1024 / 1e6 / steps
        w = wr[k][1] * 1024 / 1e6 / steps
        print("| %s | %.1f | %.0f | %.1f | %.1f | %.2f |" % (k, n, fe[k][1] / steps, rd, w, (rd + w) / max(n, 1e-9)))
    print()
    tab = {k: round((2 * fe[k][1] * 1024 + wr[k][1] * 1024) / max(fe[k][0], 1), -5) for k in ("conv_igemm", "conv_wgrad", "loss_fwd", "loss_train")}
    print("bench.py table (bytes per launch):", tab)
    if len(sys.argv) > 5:                       # machine-readable copy that bench.py reads for `roofline.traffic`
        import json
        w = [int(v) for v in sys.argv[6:11]]
        json.dump({"command": os.environ.get("MD2_PMC_COMMAND", "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra") + " (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
                   "workload": {"batch": w[0], "channels": w[1], "height": w[2], "width": w[3], "depth": w[4], "precision": sys.argv[11]},
                   "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 64 B per 128-B request) + WRITE_SIZE KiB x 1024, per launch", "bytes_per_launch": tab,
                   "bytes_per_step": {k: 2 * fe[k][1] * 1024 / steps + wr[k][1] * 1024 / steps for k in ("conv_igemm", "conv_wgrad", "loss_fwd", "loss_train", "other")}},
                  open(sys.argv[5], "w"), indent=1)


def valu(d, steps, out_json, B, C, H, W):
    """SQ counters of the loss kernels (and the per-class totals) from a --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY pass."""
    import json
    f = glob.glob(d + "/*counter_collection.csv")[0]
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = "loss_train" if "loss_train_kernel" in k else ("loss_smooth_sums" if "loss_smooth_sums" in k else klass(k))
        a = per[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print("# SQ counters per launch (rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra; B=%d C=%d %dx%d)" % (B, C, W, H))
    print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles (MI355X_MICROARCH.md); a wave64 VALU instruction occupies its SIMD-32 for 2 cycles.")
    print()
    print("| kernel / class | launches | SQ_INSTS_VALU per launch | SQ_WAVE_CYCLES | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_ANY |")
    print("|---|---|---|---|---|---|")
    out = {"workload": {"batch": B, "channels": C, "height": H, "width": W}}
    for key in sorted(per):
        c = per[key]
        n = c["SQ_INSTS_VALU"][0]
        row = [c[x][1] / max(c[x][0], 1) for x in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")]
        print("| %s | %d | %.4g | %.4g | %.4g | %.4g |" % (key, n, *row))
        out[key] = row[0]
    px = 4 * B * H * W * (1 + 0.25 + 0.0625 + 0.015625) / 1.328125          # pixel-scales of the four-scale loss (every scale is evaluated at full resolution)
    if "loss_train" in out:
        print()
        print("loss_train_kernel: %.0f lane-instructions per pixel and scale (4 scales x B x H x W = %.3g pixel-scales per launch)" % (out["loss_train"] * 64 / (4 * B * H * W), 4 * B * H * W))
    json.dump(out, open(out_json, "w"), indent=1)


def l2(d, steps):
    """L1 -> L2 read requests and L2 hit rate per kernel (one --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE pass): how many bytes a kernel's compute units
    pull out of the L2 per shader cycle -- the convolution loops of this step are bound by that rate, not by the matrix pipe."""
    f = glob.glob(d + "/*counter_collection.csv")[0]
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        a = per[r["Kernel_Name"].split("(")[0][:90]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print("# L1 -> L2 read requests per kernel, batch 32 (rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra)")
    print("# per launch averages; shader cycles = GRBM_GUI_ACTIVE / 8 XCDs; B / clk / CU = requests x 64 B / (shader cycles x 256 CUs) -- a TCP -> TCC read request moves one 64-byte")
    print("# half line (uncalibrated against a known byte count on gfx950: read the column as a lower bound and compare kernels with each other)")
    print()
    print("| kernel | launches | shader cycles | L1->L2 read requests | B / clk / CU (64 B per request) | L2 hit rate |")
    print("|---|---|---|---|---|---|")
    rows = []
    for k, c in per.items():
        n = c["TCP_TCC_READ_REQ_sum"][0]
        if n == 0:
            continue
        cyc = c["GRBM_GUI_ACTIVE"][1] / max(c["GRBM_GUI_ACTIVE"][0], 1) / 8.0
        req = c["TCP_TCC_READ_REQ_sum"][1] / n
        hit = c["TCC_HIT_sum"][1]; miss = c["TCC_MISS_sum"][1]
        rows.append((req * n, k, n, cyc, req, req * 64.0 / max(cyc * 256.0, 1.0), hit / max(hit + miss, 1.0)))
    for _, k, n, cyc, req, bpc, hr in sorted(rows, reverse=True)[:28]:
        print("| `%s` | %d | %.4g | %.4g | %.1f | %.3f |" % (k, n, cyc, req, bpc, hr))


def mfma(dm, dl, dv):
    """Matrix-pipe / LDS / instruction counters per kernel (three --pmc passes of the same command): how busy the matrix pipe is, whether MFMAs co-execute with vector
    instructions, vector and scalar instructions per MFMA, LDS bank-conflict ratio."""
    def load(d):
        f = glob.glob(d + "/*counter_collection.csv")[0]
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = per[name][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        return per
    pm, pl, pv = load(dm), load(dl), load(dv)
    print("# Matrix-pipe, LDS and instruction counters per kernel, batch 32 (round 4): rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra")
    print("# groups (one pass each): SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS | SQ_INSTS_VALU ...")
    print("# per launch averages; matrix busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); SQ_INSTS_VALU counts the MFMAs too (subtracted in the ratio)")
    print()
    print("| kernel | launches | shader cycles | MFMAs | matrix pipe busy | co-executing with VALU | VALU per MFMA | SALU per MFMA | LDS conflict / active |")
    print("|---|---|---|---|---|---|---|---|---|")
    rows = []
    for name, c in pm.items():
        n = c["SQ_INSTS_MFMA"][0]
        mf = c["SQ_INSTS_MFMA"][1] / max(n, 1)
        if mf <= 0:
            continue
        cyc = c["GRBM_GUI_ACTIVE"][1] / max(c["GRBM_GUI_ACTIVE"][0], 1) / 8.0
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / max(n, 1)
        co = c["SQ_VALU_MFMA_COEXEC_CYCLES"][1] / max(n, 1)
        va = pv.get(name, {}).get("SQ_INSTS_VALU", [0, 0.0]); sa = pl.get(name, {}).get("SQ_INSTS_SALU", [0, 0.0])
        bc = pl.get(name, {}).get("SQ_LDS_BANK_CONFLICT", [0, 0.0]); ia = pl.get(name, {}).get("SQ_LDS_IDX_ACTIVE", [0, 0.0])
        rows.append((busy * n, "| `%s` | %d | %.4g | %.4g | %.3f | %.3f | %.1f | %.1f | %.2f |" % (
            name[:70], n, cyc, mf, busy / (1024.0 * cyc), co / max(busy, 1), (va[1] / max(va[0], 1) - mf) / mf, sa[1] / max(sa[0], 1) / mf, bc[1] / max(ia[1], 1))))
    for _, line in sorted(rows, reverse=True):
        print(line)
    # class totals (bench.py reads them): matrix-pipe busy fraction of the class's shader time, share of it co-executing with vector instructions, vector instructions per MFMA
    import json
    cls = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0.0])          # busy, coexec, cycles, mfmas, valu
    for name, c in pm.items():
        k = klass(name)
        n = max(c["SQ_INSTS_MFMA"][0], 1)
        cls[k][0] += c["SQ_VALU_MFMA_BUSY_CYCLES"][1]; cls[k][1] += c["SQ_VALU_MFMA_COEXEC_CYCLES"][1]; cls[k][2] += c["GRBM_GUI_ACTIVE"][1] / 8.0; cls[k][3] += c["SQ_INSTS_MFMA"][1]
        va = pv.get(name, {}).get("SQ_INSTS_VALU", [0, 0.0]); cls[k][4] += va[1] * (c["SQ_INSTS_MFMA"][0] / max(va[0], 1))
    out = {}
    print()
    print("| class | matrix pipe busy | of it co-executing with VALU | vector instructions per MFMA |")
    print("|---|---|---|---|")
    for k in ("conv_igemm", "conv_wgrad"):
        b_, co, cyc, mf, va = cls[k]
        out[k] = {"matrix_pipe_busy": b_ / (1024.0 * max(cyc, 1)), "coexec_share": co / max(b_, 1), "valu_per_mfma": (va - mf) / max(mf, 1)}
        print("| %s | %.3f | %.3f | %.1f |" % (k, out[k]["matrix_pipe_busy"], out[k]["coexec_share"], out[k]["valu_per_mfma"]))
    if len(sys.argv) > 5:
        json.dump(out, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "l2":
        l2(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "valu":
        valu(sys.argv[2], int(sys.argv[3]), sys.argv[4], *[int(v) for v in sys.argv[5:9]])
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]))
