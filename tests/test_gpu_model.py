"""GPU parity, model level: Model forward, the fused loss (fwd + pullbacks), the whole training step and ADAM through
the C ABI vs the CPU oracle and the committed golden fixtures; plus size-independent properties at the full
BASELINE.json shapes (416x128, B=4 / B=32) where the oracle would take too long."""
import os

import numpy as np
import pytest
import torch

import md2amd
import oracle

pytestmark = pytest.mark.gpu
syn = md2amd.synthetic
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def md():
    return md2amd.monodepth


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def grad_close(a, b, tol=1e-3, frac=0.02, cap=5e-2):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); d = np.abs(a - b) / (np.abs(b).max() + 1e-30)
    return (d > tol).sum() <= max(frac * d.size, 2) and d.max() < cap


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


# With the automask most pixels carry no photometric gradient; the few pixels whose mask / arg-min decision differs between two fp32 summation orders
# then weigh visibly in the small arrays at the END of the chain (the full-resolution head and the last decoder branch).
ALLOW_AUTOMASK = ("dec.head5", "dec.branch5")
REPORT = os.environ.get("MD2_GRAD_REPORT")          # file that collects the per-array tables of every model test (diagnostics)


def per_array(gh, go, C, depth=18, levels=(2, 3, 4, 5)):
    """[(name, cosine, ||gh - go|| / ||go||, ||go||)] for every array of the flat parameter vector (syn.param_layout order)."""
    out = []; off = 0
    gh = np.asarray(gh, np.float64); go = np.asarray(go, np.float64)
    for name, _, shape in syn.param_layout(C, depth, levels):
        n = int(np.prod(shape)); a = gh[off:off + n]; b = go[off:off + n]; off += n
        na, nb = np.linalg.norm(a), np.linalg.norm(b)
        out.append((name, float(a @ b / (na * nb + 1e-300)), float(np.linalg.norm(a - b) / (nb + 1e-300)), float(nb)))
    assert off == gh.size == go.size
    return out


def assert_grads_match(gh, go, C, depth=18, cos_min=0.998, rel_max=5e-2, allow=(), tag="", levels=(2, 3, 4, 5)):
    """Gradient parity as a COMPARISON, array by array: every parameter array (conv weight, bias, BatchNorm gamma / beta -- the 3 136-element stem included)
    must have cosine >= cos_min AND ||g_hip - g_oracle|| / ||g_oracle|| <= rel_max.  Arrays whose oracle gradient is numerically zero (below 1e-7 of the
    largest array norm) are skipped.

    Why 5e-2 and not 1e-4 on arbitrary inputs (profiles/r02_grad_diag.md, tools/grad_diag.py): the gradient of this loss is piecewise smooth -- ReLU, max-pool,
    arg-min over the sources, SSIM / :border clamps, |.|.  Where no decision differs between the two fp32 summation orders the arrays agree to 3e-5 .. 3e-4
    (and test_gradient_parity_tight_on_kink_free_seeds holds them to 4e-3 on such inputs); ONE differing decision moves every array upstream of it by
    ~1e-2, in both implementations alike when measured against an fp64 computation, and the fp32 oracle itself moves by that much under a 1-ulp
    perturbation of its input.  A wrong index / layout / boundary in a kernel shows as O(1) in at least one array and fails this bound.
    `allow`: name prefixes with a documented reason for a looser bound (cos >= 0.99, rel <= 0.15)."""
    tab = per_array(gh, go, C, depth, levels)
    top = max(t[3] for t in tab)
    # an array whose gradient is a near-complete cancellation (e.g. one scalar bias summing signed per-pixel terms to 1e-6 of its siblings' size) has no
    # meaningful RELATIVE error: skip arrays 100x below the median norm of their kind (W / b / gamma / beta)
    kinds = {}
    for (name, _, _, nb), (_, kind, _) in zip(tab, syn.param_layout(C, depth, levels)):
        kinds.setdefault(kind, []).append(nb)
    med = {k: float(np.median(v)) for k, v in kinds.items()}
    kind_of = {name: kind for name, kind, _ in syn.param_layout(C, depth, levels)}
    bad = []
    norm_of = {name: nb for name, _, _, nb in tab}
    size_of = {name: int(np.prod(shape)) for name, _, shape in syn.param_layout(C, depth, levels)}
    for name, cos, rel_, nb in tab:
        if nb < 1e-7 * top or nb < 1e-2 * med[kind_of[name]]:
            continue
        loose = any(name.startswith(a) for a in allow)
        if size_of[name] == 1 and name.endswith(".b"):
            # a ONE-element bias (the disparity heads): its gradient is the plain sum of the per-pixel gradients, a near-cancellation whose own magnitude says nothing about the
            # size of its terms -- the error is measured against the layer's weight gradient instead (same terms, weighted by O(1) activations: 5 % of its norm as the floor)
            rel_ = rel_ * nb / max(nb, 0.05 * norm_of.get(name[:-2] + ".W", 0.0))
        if cos < (0.99 if loose else cos_min) or rel_ > (0.15 if loose else rel_max):
            bad.append((name, cos, rel_, nb))
    if REPORT:                                    # diagnostics: the worst arrays of every call
        with open(REPORT, "a") as f:
            f.write("# %s: %d arrays, worst cos %.6f, worst rel %.3e\n" % (tag, len(tab), min(t[1] for t in tab if t[3] >= 1e-7 * top), max(t[2] for t in tab if t[3] >= 1e-7 * top)))
            for t in sorted(tab, key=lambda t: -t[2])[:12]:
                f.write("  %-28s cos %.6f rel %.3e norm %.3e\n" % t)
    assert not bad, "gradient arrays outside the parity bound: " + ", ".join("%s (cos %.5f, rel %.2e)" % b[:3] for b in bad[:8])


def host(t):
    return t.detach().cpu().numpy()


def make(N, C, H, W, automask=False, pseed=1, lr=1e-4, depth=18, precision="f32"):
    m = md()
    K, iK = syn.depth10k_intrinsics(W, H)
    params = m.Params(target_size=(W, H), batch_size=N, automasking=automask)
    cache = m.TrainCache(m.SSIM(), m.Backproject(W, H), m.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = m.ResidualNetwork(depth, in_channels=C)
    model = m.Model(enc, m.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), m.PoseDecoder(enc.stages[-1]), lr=lr, precision=precision)
    theta = syn.init_params(C, seed=pseed, randomize_affine=True, depth=depth)
    model.load_params(theta)
    return model, params, cache, theta, K, iK


@pytest.mark.parametrize("name", ["loss_c1_64x96", "loss_c3_32x64"])
def test_loss_half_vs_oracle_and_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    N, C, H, W = int(g["N"]), int(g["C"]), int(g["H"]), int(g["W"])
    x = syn.synthetic_triplets(N, C, H, W, seed=int(g["seed"]))
    model, params, cache, _, K, iK = make(N, C, H, W)             # 32x64: a loss-only handle (the network needs >= 64)
    xd = dev(x); model._ensure(xd, params, cache)
    disps = [g["disp%d" % i] for i in range(4)]
    loss, dds, drv, dtv = md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(g["rvec"]), dev(g["tvec"]))
    cfg = oracle.make_cfg(N, C, H, W, K, iK)
    r = oracle.loss_fwd_bwd(x, cfg, disps, g["rvec"], g["tvec"])
    assert abs(float(loss) - r["loss"]) / r["loss"] < 1e-5
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < 1e-4
    for i in range(4):
        assert grad_close(host(dds[i]), r["ddisp"][i], tol=2e-4)
        assert grad_close(host(dds[i]), g["ddisp%d" % i])
    assert rel(host(drv), r["drvec"]) < 2e-3 and rel(host(dtv), r["dtvec"]) < 2e-3
    assert rel(host(drv), g["drvec"]) < 5e-3 and rel(host(dtv), g["dtvec"]) < 5e-3


def test_an_explicit_zero_scale_switches_that_levels_smoothness_off():
    """TrainCache.scales entries are the caller's numbers (src/training.jl:67): a 0 among them is a weight of 0, not "use the default" (round-3 advisor finding); all zeros = no
    smoothness term at all."""
    N, C, H, W = 2, 1, 64, 96
    x = syn.synthetic_triplets(N, C, H, W, seed=31); xd = dev(x)
    rng = np.random.default_rng(32)
    disps = [(0.2 + 0.6 * rng.random((N, 1, H >> (3 - i), W >> (3 - i)))).astype(np.float32) for i in range(4)]
    rv = (0.01 * rng.standard_normal((2, N, 3))).astype(np.float32); tv = (0.05 * rng.standard_normal((2, N, 3))).astype(np.float32)
    K, iK = syn.depth10k_intrinsics(W, H)
    losses = {}
    for scales in ([0.125, 0.25, 0.5, 1.0], [0.125, 0.0, 0.5, 1.0], [0.0, 0.0, 0.0, 0.0]):
        model, params, cache, _, _, _ = make(N, C, H, W)
        params.disparity_smoothness = 0.5          # (large enough that a wrong weight moves the loss far beyond the bound)
        cache.scales = scales
        model._ensure(xd, params, cache)
        loss = float(md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(rv), dev(tv))[0])
        cfg = oracle.make_cfg(N, C, H, W, K, iK, scales=scales, smoothness=0.5)
        lo = oracle.loss_fwd_bwd(x, cfg, disps, rv, tv)["loss"]
        assert abs(loss - lo) / lo < 1e-5, (scales, loss, lo)
        losses[tuple(scales)] = loss
        model.close()
    assert losses[(0.125, 0.25, 0.5, 1.0)] > losses[(0.125, 0.0, 0.5, 1.0)] > losses[(0.0, 0.0, 0.0, 0.0)]


def test_loss_half_c3_64x64_vs_oracle_with_automask_and_clamping():
    N, C, H, W = 2, 3, 64, 64
    rng = np.random.default_rng(5)
    x = syn.synthetic_triplets(N, C, H, W, seed=6)
    model, params, cache, _, K, iK = make(N, C, H, W, automask=True)
    xd = dev(x); model._ensure(xd, params, cache)
    # disparities: a smooth random surface (a 5 x 14 grid of random levels, cubic-resampled to each scale) plus 2 % white noise -- rough enough to send samples
    # across the border and to put real texture under every SSIM window, without turning the warped frames into noise
    from scipy.ndimage import zoom
    def surface(h, w):
        g = rng.random((N, 1, 5, 14))
        return np.clip(zoom(g, (1, 1, h / 5, w / 14), order=3) * 0.9 + 0.05 + 0.02 * (rng.random((N, 1, h, w)) - 0.5), 0.02, 0.98).astype(np.float32)
    disps = [surface(H >> s, W >> s) for s in (3, 2, 1, 0)]
    rvec = ((rng.random((2, N, 3)) - 0.5) * 0.06).astype(np.float32); tvec = ((rng.random((2, N, 3)) - 0.5) * 0.6).astype(np.float32)   # large motion: many :border clamps
    cfg = oracle.make_cfg(N, C, H, W, K, iK, automasking=True)
    al = oracle.automasking_loss(x, cfg)
    ald = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3])
    assert np.abs(host(ald) - al).max() < 2e-5
    loss, dds, drv, dtv = md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(rvec), dev(tvec), auto_loss=dev(al))
    r = oracle.loss_fwd_bwd(x, cfg, disps, rvec, tvec, auto_loss=al)
    assert abs(float(loss) - r["loss"]) / r["loss"] < 1e-5
    for i in range(4):
        assert grad_close(host(dds[i]), r["ddisp"][i], tol=2e-4)
    assert rel(host(drv), r["drvec"]) < 2e-3 and rel(host(dtv), r["dtvec"]) < 2e-3


@pytest.mark.parametrize("case", [(2, 1, 24, 40), (1, 3, 40, 72), (3, 1, 16, 16), (1, 1, 48, 104)])
def test_loss_half_ragged_tiles_vs_oracle(case):
    """Image sizes that are multiples of 8 but not of the loss kernel's 32 x 32 (C = 1) / 32 x 16 (C = 3) tiles, down to the 16 x 16 minimum: partially filled tiles,
    reflect aliases inside a ragged tile, coarse maps of 2 x 2 .. 6 x 13 pixels whose resize adjoint is assembled from several tiles by atomics."""
    N, C, H, W = case
    rng = np.random.default_rng(11 + H)
    x = syn.synthetic_triplets(N, C, H, W, seed=12)
    model, params, cache, _, K, iK = make(N, C, H, W)
    xd = dev(x); model._ensure(xd, params, cache)
    disps = [(rng.random((N, 1, H >> s, W >> s)) * 0.5 + 0.25).astype(np.float32) for s in (3, 2, 1, 0)]
    rvec = ((rng.random((2, N, 3)) - 0.5) * 0.02).astype(np.float32); tvec = ((rng.random((2, N, 3)) - 0.5) * 0.1).astype(np.float32)
    cfg = oracle.make_cfg(N, C, H, W, K, iK)
    loss, dds, drv, dtv = md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(rvec), dev(tvec))
    r = oracle.loss_fwd_bwd(x, cfg, disps, rvec, tvec)
    assert abs(float(loss) - r["loss"]) / r["loss"] < 1e-5
    for i in range(4):
        assert grad_close(host(dds[i]), r["ddisp"][i], tol=5e-4), i
    assert rel(host(drv), r["drvec"]) < 5e-3 and rel(host(dtv), r["dtvec"]) < 5e-3
    # forward-only call (separate kernel) agrees with the fused value
    l2, _, _, _ = md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(rvec), dev(tvec), want_grads=False)
    assert abs(float(l2) - float(loss)) / float(loss) < 1e-6
    # deterministic mode (ordered partial sums, full-resolution maps + the ordered resize adjoint): bit-reproducible, and the same numbers to rounding
    md().set_deterministic(True)
    try:
        la, da, ra, ta = md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(rvec), dev(tvec))
        lb, db, rb, tb = md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(rvec), dev(tvec))
    finally:
        md().set_deterministic(False)
    assert float(la) == float(lb) and all(torch.equal(p, q) for p, q in zip(da, db)) and torch.equal(ra, rb) and torch.equal(ta, tb)
    assert abs(float(la) - float(loss)) / float(loss) < 1e-6 and all(rel(host(p), host(q)) < 1e-5 for p, q in zip(da, dds))
    # the slow_depth objective (one full-resolution disparity, un-normalised smoothness) on the same ragged frame
    d1 = disps[3]
    ls, dd1, dr1, dt1 = md().simple_depth_loss(model, xd, dev(d1), dev(rvec), dev(tvec))
    rs = oracle.loss_fwd_bwd(x, oracle.make_cfg(N, C, H, W, K, iK, simple=True), [d1], rvec, tvec)
    assert abs(float(ls) - rs["loss"]) / rs["loss"] < 1e-5 and grad_close(host(dd1), rs["ddisp"][0], tol=5e-4)
    assert rel(host(dr1), rs["drvec"]) < 5e-3 and rel(host(dt1), rs["dtvec"]) < 5e-3


@pytest.mark.parametrize("name", ["net_c1_64x96", "net_c3_64x96_automask"])
def test_model_and_train_step_vs_oracle_and_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    N, C, H, W = int(g["N"]), int(g["C"]), int(g["H"]), int(g["W"]); automask = bool(int(g["automask"]))
    model, params, cache, theta, K, iK = make(N, C, H, W, automask=automask, pseed=int(g["pseed"]))
    x = syn.synthetic_triplets(N, C, H, W, seed=int(g["xseed"])); xd = dev(x)
    # ---- Model forward (src/model.jl:8-20)
    ds, poses = model(xd, [1, 3], 2)
    cfg = oracle.make_cfg(N, C, H, W, K, iK, automasking=automask)
    net = oracle.OracleNet(C); net.set_params(theta)
    dso, rvo, tvo = net.model_forward(x, cfg)
    for i in range(4):
        assert rel(host(ds[i]), dso[i]) < 1e-4 and rel(host(ds[i]), g["disp%d" % i]) < 1e-4
    for s in range(2):
        assert rel(host(poses[s].rvec), rvo[s]) < 1e-4 and rel(host(poses[s].tvec), tvo[s]) < 1e-4
    # ---- train_loss + gradient (src/training.jl:21-78, src/Monodepth.jl:170-176)
    al = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3]) if automask else None
    loss, _, _, _ = md().train_loss(model, xd, al, cache, params, False)
    lo = net.train_loss(x, cfg, host(al) if automask else None, True)
    assert abs(float(loss) - lo) / lo < 1e-4                     # north-star: forward+loss within 1e-4 relative fp32
    assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < 1e-4
    md().gradient(model)
    gh = model.grads(); go = net.get_grads()
    assert_grads_match(gh, go, C, allow=ALLOW_AUTOMASK if automask else (), tag=name)   # every parameter array against the oracle's
    nh = np.array([t[3] for t in per_array(go, gh, C)])          # (norms of the HIP arrays)
    gn = g["grad_norms"]; bigg = gn > 1e-3 * gn.max()
    assert np.abs(nh[bigg] - gn[bigg]).max() / gn.max() < 2e-2   # and the torch-derived golden norms
    # ---- ADAM (Flux.Optimise.update!)
    md().update(md().ADAM(1e-4), model)
    net.set_grads(gh); net.adam_step(1e-4)                        # same gradients in, so the update itself must agree tightly
    assert rel(model.params() - theta, net.get_params() - theta) < 5e-3   # |delta| = lr = 1e-4; one fp32 ulp of a parameter near 1 is 1.2e-7


def test_gradient_finite_difference_directional():
    """Independent of the oracle: d loss along a random parameter direction vs the analytic gradient."""
    N, C, H, W = 1, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=3)
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=2))
    md().train_step(model, xd, None, cache, params, do_update=False)
    g = model.grads().astype(np.float64)
    rng = np.random.default_rng(0)
    # perturb only the decoder heads / pose tail (smooth in the weights; avoids ReLU/BN kinks dominating the FD error)
    off = 0; d = np.zeros_like(g)
    for name, _, shape in syn.param_layout(C):
        n = int(np.prod(shape))
        if name.startswith("dec.head") or name.startswith("pose.pose3"):
            d[off:off + n] = rng.standard_normal(n)
        off += n
    d /= np.linalg.norm(d); eps = 1e-2
    vals = []
    for sgn in (+1, -1):
        model.load_params((theta + sgn * eps * d).astype(np.float32))
        l, _, _, _ = md().train_loss(model, xd, None, cache, params, False); vals.append(float(l))
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - float(g @ d)) / (abs(fd) + 1e-12) < 5e-2


@pytest.mark.parametrize("B", [4, 32])
def test_full_size_properties(B):
    """BASELINE.json configs #2 (B=4) and #3 (B=32/GPU) at 416x128: the oracle needs ~15 s per sample-step on the host, so
    check properties instead: finite loss/gradients, determinism of the forward, batch-permutation equivariance of the
    loss half, a loss that decreases under ADAM, and agreement of the per-sample loss half with the small-batch result."""
    N, C, H, W = B, 1, 128, 416
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=2, lr=1e-4)
    x = syn.synthetic_triplets(N, C, H, W, seed=10); xd = dev(x)
    l1, _, _, _ = md().train_loss(model, xd, None, cache, params, False)
    l2, _, _, _ = md().train_loss(model, xd, None, cache, params, False)
    assert np.isfinite(float(l1)) and abs(float(l1) - float(l2)) < 1e-6 * abs(float(l1))
    md().gradient(model); g = model.grads()
    assert np.isfinite(g).all() and np.linalg.norm(g) > 0
    # loss half: permuting the batch permutes nothing in the mean
    ds, poses = model(xd, [1, 3], 2)
    rvec = torch.stack([poses[0].rvec, poses[1].rvec]); tvec = torch.stack([poses[0].tvec, poses[1].tvec])
    la, _, _, _ = md().loss_fwd_bwd(model, xd, ds, rvec, tvec, want_grads=False)
    perm = torch.randperm(N, device="cuda")
    lb, _, _, _ = md().loss_fwd_bwd(model, xd[perm].contiguous(), [d[perm].contiguous() for d in ds], rvec[:, perm].contiguous(), tvec[:, perm].contiguous(), want_grads=False)
    assert abs(float(la) - float(lb)) < 2e-6 * abs(float(la))
    # ADAM on a fixed batch: the first step moves EVERY parameter by +-lr (m/sqrt(v) = sign(g)), which can raise the loss once;
    # from there on the loss must come down
    losses = [float(md().train_step(model, xd, None, cache, params, True)) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < max(losses[:3]) and min(losses[6:]) < losses[1]
    if B == 4:
        # sample 0 alone through the oracle's loss half on the GPU's own disparities/poses
        cfg = oracle.make_cfg(1, C, H, W, K, iK)
        ds, poses = model(xd, [1, 3], 2)
        r = oracle.loss_fwd_bwd(x[:1], cfg, [host(d[:1]) for d in ds], np.stack([host(poses[0].rvec[:1]), host(poses[1].rvec[:1])]), np.stack([host(poses[0].tvec[:1]), host(poses[1].tvec[:1])]), do_bwd=False)
        m1, p1, c1, _, _, _ = make(1, C, H, W, pseed=2)
        x1 = dev(x[:1]); m1._ensure(x1, p1, c1)
        lg, _, _, _ = md().loss_fwd_bwd(m1, x1, [d[:1].contiguous() for d in ds], torch.stack([poses[0].rvec[:1], poses[1].rvec[:1]]).contiguous(),
                                        torch.stack([poses[0].tvec[:1], poses[1].tvec[:1]]).contiguous(), want_grads=False)
        assert abs(float(lg) - r["loss"]) / r["loss"] < 1e-5


def test_rccl_allreduce_on_library_gradient_buffer():
    """The DP exchange on the real transport: torch.distributed 'nccl' (= RCCL) all-reduce of the library's flat gradient
    buffer wrapped without a copy.  One GPU per box here, so world_size = 1; ranks > 1 are covered by the gloo test."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    N, C, H, W = 1, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=4)
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=3))
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        dp = md2amd.parallel.DataParallel(model)
        md().train_step(model, xd, None, cache, params, do_update=False)
        g0 = model.grads()
        buf = model.grad_buffer()
        assert buf.is_cuda and buf.numel() >= g0.size
        dist.all_reduce(buf)                                        # sum over 1 rank: unchanged
        torch.cuda.synchronize()
        assert np.array_equal(model.grads(), g0)
        loss = dp.train_step(xd, None, cache, params)               # fwd+bwd -> all-reduce -> ADAM(1/world)
        assert np.isfinite(float(loss)) and not np.array_equal(model.params(), theta)
    finally:
        dist.destroy_process_group()


def test_simple_depth_loss_and_driver():
    """BASELINE config #1 (src/simple_depth.jl): objective + pullbacks vs the oracle, then the ADAM loop lowers the loss."""
    N, C, H, W = 1, 3, 64, 96
    rng = np.random.default_rng(9)
    x = syn.synthetic_triplets(N, C, H, W, seed=14); K, iK = syn.depth10k_intrinsics(W, H)
    model, params, cache, _, _, _ = make(N, C, H, W)
    xd = dev(x); model._ensure(xd, params, cache)
    d = (rng.random((N, 1, H, W)) * 0.8 + 0.1).astype(np.float32)
    rv = ((rng.random((2, N, 3)) - 0.5) * 0.02).astype(np.float32); tv = ((rng.random((2, N, 3)) - 0.5) * 0.2).astype(np.float32)
    loss, dd, dr, dt = md().simple_depth_loss(model, xd, dev(d), dev(rv), dev(tv))
    r = oracle.loss_fwd_bwd(x, oracle.make_cfg(N, C, H, W, K, iK, simple=True), [d], rv, tv)
    assert abs(float(loss) - r["loss"]) / r["loss"] < 1e-5
    assert grad_close(host(dd), r["ddisp"][0], tol=2e-4)
    assert rel(host(dr), r["drvec"]) < 2e-3 and rel(host(dt), r["dtvec"]) < 2e-3
    # the driver: disparity 0.5, rvec (0,0,0.01), ADAM(3e-4), src/simple_depth.jl:8-22
    disp, poses, losses = md().slow_depth(xd, K, iK, iters=60, log_every=20)
    assert disp.shape == (N, 1, H, W) and len(poses) == 2 and np.isfinite(losses).all() and losses[-1] < losses[0]
    # first iteration == oracle objective at the initial point
    d0 = np.full((N, 1, H, W), 0.5, np.float32); r0 = np.zeros((2, N, 3), np.float32); r0[:, :, 2] = 0.01
    l0 = oracle.loss_fwd_bwd(x, oracle.make_cfg(N, C, H, W, K, iK, simple=True), [d0], r0, np.zeros((2, N, 3), np.float32), do_bwd=False)["loss"]
    assert abs(losses[0] - l0) / l0 < 1e-5
    # slow_depth(x, ...; target_id, source_ids) with other frame roles (src/simple_depth.jl:1-6): the poses are optimised directly, so nothing is inverted
    _, _, l31 = md().slow_depth(xd, K, iK, target_id=3, source_ids=(2, 1), iters=1, log_every=1)
    l0 = oracle.loss_fwd_bwd(x, oracle.make_cfg(N, C, H, W, K, iK, simple=True, target_id=3, source_ids=(2, 1)), [d0], r0, np.zeros((2, N, 3), np.float32), do_bwd=False)["loss"]
    assert abs(l31[0] - l0) / l0 < 1e-5 and abs(l31[0] - losses[0]) / l0 > 1e-3


@pytest.mark.parametrize("C,automask", [(1, False), (3, True)])
def test_gradients_do_not_depend_on_previous_step(C, automask):
    """The gradient accumulators are never zeroed (first writer overwrites): a second step on different data must give the same
    gradients as a fresh handle — stale contributions from the previous step would show up here, not on a first step."""
    N, H, W = 2, 64, 96
    xa = dev(syn.synthetic_triplets(N, C, H, W, seed=31)); xb = dev(syn.synthetic_triplets(N, C, H, W, seed=32))
    def grads_after(seq):
        model, params, cache, theta, K, iK = make(N, C, H, W, automask=automask, pseed=8)
        for x in seq:
            al = md().automasking_loss(md().SSIM(), x, x[:, 1].contiguous(), [1, 3]) if automask else None
            md().train_step(model, x, al, cache, params, do_update=False)
        return model.grads()
    g_fresh = grads_after([xb]); g_fresh2 = grads_after([xb]); g_second = grads_after([xa, xb])
    assert np.isfinite(g_second).all()
    # run-to-run noise floor: fp32 atomics (BN statistics, weight gradients) reorder sums, which flips a few ReLU / arg-min decisions
    noise = np.linalg.norm(g_fresh2 - g_fresh) / np.linalg.norm(g_fresh)
    diff = np.linalg.norm(g_second - g_fresh) / np.linalg.norm(g_fresh)
    print("noise floor %.2e, second-step difference %.2e" % (noise, diff))
    # stale accumulators would give O(0.1 .. 1); the atomics noise floor has been observed between 1e-5 and 5e-4
    assert diff < max(5 * noise, 2e-3)


@pytest.mark.parametrize("depth", [18, 50])
def test_eval_disparity_vs_oracle(depth):
    """eval_disparity(m, x), src/model.jl:28, on single frames (encoder batch != 3N, no pose decoder)."""
    N, C, H, W = 3, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=6, depth=depth)
    frames = syn.synthetic_triplets(N, C, H, W, seed=16)[:, 1]
    for n in (3, 2):
        ds = md().eval_disparity(model, dev(frames[:n]))
        net = oracle.OracleNet(C, depth=depth); net.set_params(theta); do = net.eval_disparity(frames[:n])
        for i in range(4):
            assert ds[i].shape == do[i].shape and rel(host(ds[i]), do[i]) < 1e-4
    # the training path still works on the same handle afterwards
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=17))
    loss = md().train_step(model, xd, None, cache, params, do_update=False)
    net = oracle.OracleNet(C, depth=depth); net.set_params(theta)
    lo = net.train_loss(host(xd), oracle.make_cfg(N, C, H, W, K, iK), None, False)
    assert abs(float(loss) - lo) / lo < 1e-4


def test_checkpoint_resume(tmp_path):
    """params + ADAM moments + beta powers round-trip: (2 steps, save, 1 step) == (load, 1 step)."""
    N, C, H, W = 1, 1, 64, 96
    xs = [dev(syn.synthetic_triplets(N, C, H, W, seed=40 + i)) for i in range(3)]
    a, params, cache, theta, K, iK = make(N, C, H, W, pseed=12)
    for x in xs[:2]:
        md().train_step(a, x, None, cache, params, True)
    md().save_checkpoint(a, tmp_path / "ck.npz")
    md().train_step(a, xs[2], None, cache, params, True); pa = a.params()
    b, _, _, _, _, _ = make(N, C, H, W, pseed=99)                      # different weights until the checkpoint is loaded
    b._ensure(xs[2], params, cache)
    md().load_checkpoint(b, str(tmp_path / "ck.npz"))
    md().train_step(b, xs[2], None, cache, params, True); pb = b.params()
    d = np.abs(pa - pb).max()
    assert d < 2.5e-4, d                                                 # lr = 1e-4: at most a sign flip of a near-zero gradient (atomics noise)
    assert np.mean(np.abs(pa - pb) > 1e-6) < 1e-3
    assert np.abs(pa - theta).max() > 1e-4                               # and training really moved the weights


@pytest.mark.parametrize("shape", [(1, 3, 192, 640), (3, 1, 96, 160), (5, 3, 64, 64)])
def test_other_shapes_forward_loss_and_gradient_vs_oracle(shape):
    """Generality of the tiling / halo logic: BASELINE config #4's 640x192 RGB shape, a ragged 160x96 case (tiles overhang in both
    loss-kernel dimensions) and a batch of 5."""
    N, C, H, W = shape
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=21)
    x = syn.synthetic_triplets(N, C, H, W, seed=22); xd = dev(x)
    loss = md().train_step(model, xd, None, cache, params, do_update=False)
    net = oracle.OracleNet(C); net.set_params(theta)
    lo = net.train_loss(x, oracle.make_cfg(N, C, H, W, K, iK), None, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    gh = model.grads(); go = net.get_grads()
    assert np.isfinite(gh).all()
    assert_grads_match(gh, go, C, tag="shape %s" % (shape,))


def test_config5_shape_1024x320_with_automasking_vs_oracle():
    """BASELINE config #5's resolution (1024x320, RGB) with the min-reprojection automask, in fp32 (the parity mode; the bf16 mixed-precision mode is
    bounded separately below): forward + loss within 1e-4 of the oracle, every gradient array compared.  Exercises every kernel's tiling at 2.5x the headline width/height."""
    N, C, H, W = 1, 3, 320, 1024
    model, params, cache, theta, K, iK = make(N, C, H, W, automask=True, pseed=31)
    x = syn.synthetic_triplets(N, C, H, W, seed=32); xd = dev(x)
    al = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3])
    loss = md().train_step(model, xd, al, cache, params, do_update=False)
    cfg = oracle.make_cfg(N, C, H, W, K, iK, automasking=True)
    alo = oracle.automasking_loss(x, cfg)
    assert np.abs(host(al) - alo).max() < 1e-4
    net = oracle.OracleNet(C); net.set_params(theta)
    lo = net.train_loss(x, cfg, alo, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    gh = model.grads(); go = net.get_grads()
    assert np.isfinite(gh).all()
    assert_grads_match(gh, go, C, allow=ALLOW_AUTOMASK, tag="config5 1024x320 automask")


@pytest.mark.parametrize("shape", [(2, 1, 64, 96), (1, 3, 96, 160)])
def test_resnet50_encoder_forward_loss_gradient_and_adam_vs_oracle(shape):
    """BASELINE config "ResNet-50 encoder": Bottleneck trunk (1x1 - 3x3 carrying the stride - 1x1 x4, torchvision convention), stages
    (64,256,512,1024,2048) feeding the decoders.  Forward outputs and loss within 1e-4 of the oracle, gradients in the same direction,
    one ADAM step equal to the oracle's."""
    N, C, H, W = shape
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=41, depth=50)
    assert theta.size == syn.num_params(C, 50)
    x = syn.synthetic_triplets(N, C, H, W, seed=42); xd = dev(x)
    net = oracle.OracleNet(C, depth=50); net.set_params(theta)
    cfg = oracle.make_cfg(N, C, H, W, K, iK)
    ds, poses = model(xd, [1, 3], 2)
    do, rvo, tvo = net.model_forward(x, cfg)
    for i in range(4):
        assert rel(host(ds[i]), do[i]) < 1e-4
    assert rel(host(poses[0].rvec), rvo[0]) < 1e-3 and rel(host(poses[1].tvec), tvo[1]) < 1e-3
    loss = md().train_step(model, xd, None, cache, params, do_update=False)
    lo = net.train_loss(x, cfg, None, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    gh = model.grads(); go = net.get_grads()
    assert np.isfinite(gh).all()
    # (ill-conditioned at fp32 resolution: see test_resnet50_gradients_are_as_close_to_fp64_as_the_fp32_oracle, which is the tight statement for this trunk)
    assert_grads_match(gh, go, C, depth=50, cos_min=0.997, rel_max=6e-2, tag="resnet50 %s" % (shape,))
    md().update(md().ADAM(1e-4), model); net.adam_step(1e-4)
    assert np.abs(model.params() - net.get_params()).max() < 5e-3 * 1e-4 / 1e-4 * 1e-1     # |delta| <= lr per element; sign flips only on ~0 gradients


def test_config4_resnet50_640x192_vs_oracle():
    """BASELINE config "ResNet-50 encoder, 640x192, 4-scale photometric loss" at its full resolution (RGB, one triplet): loss within 1e-4
    of the oracle and the gradient in the same direction."""
    N, C, H, W = 1, 3, 192, 640
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=51, depth=50)
    x = syn.synthetic_triplets(N, C, H, W, seed=52); xd = dev(x)
    loss = md().train_step(model, xd, None, cache, params, do_update=False)
    net = oracle.OracleNet(C, depth=50); net.set_params(theta)
    lo = net.train_loss(x, oracle.make_cfg(N, C, H, W, K, iK), None, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    gh = model.grads(); go = net.get_grads()
    assert np.isfinite(gh).all()
    assert_grads_match(gh, go, C, depth=50, cos_min=0.997, rel_max=6e-2, tag="config4 resnet50 640x192")


_FP32_ORACLE_STEP = {}


def fp32_oracle_step(cfgd, theta, x, K, iK):
    """(loss, gradient vector, disparity maps) of the fp32 oracle's training step on `cfgd` with the parameters / inputs of seeds 61 / 62: the two reduced-precision tests below
    compare against the same oracle step, computed once per configuration (10 s of host time at 1024 x 320)."""
    if cfgd not in _FP32_ORACLE_STEP:
        N, C, H, W, depth, automask = cfgd
        cfg = oracle.make_cfg(N, C, H, W, K, iK, automasking=automask)
        net = oracle.OracleNet(C, depth=depth); net.set_params(theta)
        od, _, _ = net.model_forward(x, cfg)
        lo = net.train_loss(x, cfg, oracle.automasking_loss(x, cfg) if automask else None, True)
        _FP32_ORACLE_STEP[cfgd] = (lo, np.array(net.get_grads(), copy=True), [np.array(d, copy=True) for d in od])
    return _FP32_ORACLE_STEP[cfgd]


@pytest.mark.parametrize("cfgd", [(2, 1, 128, 416, 18, False), (1, 3, 320, 1024, 18, False), (1, 3, 320, 1024, 18, True)])
def test_bf16_mixed_precision_tracks_the_fp32_oracle(cfgd):
    """BASELINE config #5's "bf16 mixed precision": only the MFMA operands are bf16 (8 mantissa bits), accumulation / activations /
    master weights / BatchNorm / loss / ADAM stay fp32.  Not a parity mode: the loss must stay within 2 % of the fp32 oracle and the
    gradient must point the same way (without the automask: cosine > 0.95 at 416x128, measured 0.97; > 0.85 at 1024x320, measured 0.90 .. 0.91); the fp32 mode on the same handle class stays at 1e-4 (the other tests)."""
    N, C, H, W, depth, automask = cfgd
    model, params, cache, theta, K, iK = make(N, C, H, W, automask=automask, pseed=61, depth=depth, precision="bf16")
    x = syn.synthetic_triplets(N, C, H, W, seed=62); xd = dev(x)
    al = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3]) if automask else None
    loss = md().train_step(model, xd, al, cache, params, do_update=False)
    lo, go, _ = fp32_oracle_step(cfgd, theta, x, K, iK)
    assert abs(float(loss) - lo) / lo < 2e-2
    gh = model.grads()
    assert np.isfinite(gh).all()
    # with the automask most pixels carry no gradient and the few mask / arg-min decisions that flip under the bf16 perturbation of the
    # disparities dominate the difference (measured 0.68 .. 0.81 from run to run: the atomics' summation order decides a few of them), without it the
    # directions agree to 0.97 at 416x128 and 0.91 at 1024x320
    cos_all = float(np.dot(gh, go) / (np.linalg.norm(gh) * np.linalg.norm(go)))
    if not automask:
        assert cos_all > (0.95 if W <= 416 else 0.85)
    else:
        # the mask decisions are taken on bf16-perturbed disparities: a handful flip against the fp32 oracle and dominate the small masked gradient.  So the
        # statement is made per array on the part of the network the mask does NOT reach through a few pixels -- the encoder trunk and the pose decoder, whose
        # gradients are sums over every pixel -- and the whole vector only has to keep its direction
        assert cos_all > 0.55
        tab = per_array(gh, go, C, depth)
        top = max(t[3] for t in tab)
        worst = min(t[1] for t in tab if (t[0].startswith("enc.layer") or t[0].startswith("pose.")) and t[0].endswith(".W") and t[3] > 1e-3 * top)
        assert worst > 0.5, "bf16 + automask: an encoder / pose weight gradient lost its direction (worst per-array cosine %.3f)" % worst
    # and it trains: a few ADAM steps reduce the loss
    l0 = float(md().train_step(model, xd, al, cache, params, do_update=True))
    for _ in range(8):
        l1 = float(md().train_step(model, xd, al, cache, params, do_update=True))
    assert np.isfinite(l1) and l1 < l0


@pytest.mark.parametrize("cfgd", [(2, 1, 128, 416, 18, False), (1, 3, 320, 1024, 18, False), (1, 3, 320, 1024, 18, True), (2, 1, 128, 416, 50, False)],
                         ids=["416x128_n2", "1024x320", "1024x320_automask_as_benched", "resnet50_416x128_n2"])
def test_bf16_activation_storage_tracks_the_fp32_oracle(cfgd):
    """md2_config.precision = 2 (BASELINE config #5 "bf16 mixed precision", taken all the way to HBM): every activation / skip tensor and its gradient is a bf16 tensor, the
    convolutions read bf16 copies of the weights (v_mfma_f32_32x32x16_bf16 / 16x16x32, fp32 accumulation); master weights, BatchNorm statistics, disparities, loss and ADAM
    stay fp32.  Not a parity mode (the reference computes in Float32, src/Monodepth.jl:72): against the fp32 oracle the loss stays within 2 %, every disparity map within
    5 % (relative L2; 3.1 % measured on the 1/32-scale map of a single 1024x320 triplet), the gradient keeps its direction (whole vector and every large weight array), and a few ADAM steps reduce the loss."""
    N, C, H, W, depth, automask = cfgd
    model, params, cache, theta, K, iK = make(N, C, H, W, automask=automask, pseed=61, depth=depth, precision="bf16_storage")
    x = syn.synthetic_triplets(N, C, H, W, seed=62); xd = dev(x)
    al = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3]) if automask else None
    disps, _ = model(xd, [1, 3], 2)
    loss = md().train_step(model, xd, al, cache, params, do_update=False)
    lo, go, od = fp32_oracle_step(cfgd, theta, x, K, iK)
    assert abs(float(loss) - lo) / lo < 2e-2
    for a, b in zip(disps, od):
        assert rel(host(a), b) < (5e-2 if depth == 18 else 0.35)
    gh = model.grads()
    assert np.isfinite(gh).all()
    if depth == 50:
        # the randomly initialised bottleneck trunk at batch 2 has no stable gradient direction to compare with the fp32 oracle in ANY reduced precision (|g| ~ 1e2 and the
        # bf16-operand mode of round 3 is as far from the fp32 step as this one: cosines around 0), and its forward pass amplifies a bf16 rounding of the activations through
        # 50 freshly initialised BatchNorm layers (the 1/8-scale disparity map: 8 % .. 23 % from run to run, the finer ones 1 % .. 3 %; the loss 0.8 %).  So the gradient of the
        # bottleneck trunk is checked against the function it is the gradient OF (round-4 advisor finding): see _bf16_storage_gradient_is_the_derivative_of_its_own_loss
        _bf16_storage_gradient_is_the_derivative_of_its_own_loss(model, xd, None, cache, params, theta, C, depth)
    else:
        cos_all = float(np.dot(gh, go) / (np.linalg.norm(gh) * np.linalg.norm(go)))
        tab = per_array(gh, go, C, depth)
        top = max(t[3] for t in tab)
        large = sorted((t[1], t[0]) for t in tab if t[0].endswith(".W") and t[3] > 1e-2 * top)
        worst = large[0]; median = large[len(large) // 2][0]
        print("bf16_storage %s: loss rel %.2e cos_all %.4f large arrays: median %.3f, lowest %s" % (cfgd, abs(float(loss) - lo) / lo, cos_all, median, " ".join("%s %.3f" % (n, c) for c, n in large[:4])))
        if not automask:
            # (the lowest array sits at the noise margin of a single 1024x320 triplet: 0.507 / 0.484 between two builds whose only difference is the summation order of one
            # BatchNorm statistic -- the bulk is what discriminates: a mis-wired kernel leaves its arrays near 0)
            assert cos_all > 0.8 and worst[0] > 0.4 and median > 0.6
        else:
            # ONE triplet under the automask: most pixels carry no gradient, and the mask / arg-min decisions -- taken on bf16-perturbed disparities -- that flip against the
            # fp32 oracle dominate what is left (measured: whole-vector cosine 0.56, single encoder arrays down to 0.2; the bf16-operand mode above sees 0.68 .. 0.81).  Only the
            # direction of the whole vector is asserted here.  (A finite-difference check is no help at this shape: under the automask even the fp32 loss is locally linear only
            # for steps that change it by < 1e-4 relative -- tools/explore_bs.py -- where the bf16 tensors' rounding is the larger effect.)  The gradient of the benched
            # configuration is held against the fp32 step of the same library at its own batch size: test_full_batch_properties_of_configs_4_and_5[...bf16_storage...].
            # (round 6, advisor: a per-array statistic beside the whole-vector one -- the median cosine of the large weight arrays, measured 0.52 with the lowest arrays at
            # 0.20 - 0.31; a mis-wired kernel's arrays sit at 0 +- 0.01 and several of them pull the median down)
            assert cos_all > 0.35 and median > 0.3
    model.load_params(theta)
    l0 = float(md().train_step(model, xd, al, cache, params, do_update=True))
    for _ in range(8):
        l1 = float(md().train_step(model, xd, al, cache, params, do_update=True))
    assert np.isfinite(l1) and l1 < l0


def test_bf16_storage_winograd_opt_in_against_the_default_bf16_step():
    """md2_set_winograd(on + 32, .) (round 6): a precision-2 handle takes conv_wino3b_kernel on its 3x3 / stride-1 layers -- forced here on every eligible layer, 416 x 128,
    batch 2.  Against the default bf16 step of the same library (direct kernels: exact products of the stored bf16 values) on the same inputs: loss within 1 %, every
    disparity map within 4 % (measured 1.4 - 2.9 %), the gradient's direction kept (cosine 0.94); the Winograd launches really ran.  Why it is not the default: V and U are rounded to bf16 in the transform
    domain -- at 1024 x 320, one triplet, the gradient's cosine against the fp32 oracle falls from 0.83 to 0.79 and single decoder arrays lose their direction
    (DESIGN.md section 4, round 6), for 1.7 - 4 % of a step."""
    N, C, H, W = 2, 1, 128, 416
    x = dev(syn.synthetic_triplets(N, C, H, W, seed=62))
    def run(mode):
        md().set_winograd(mode, 1 if mode != 1 else -2)
        try:
            model, params, cache, theta, K, iK = make(N, C, H, W, pseed=61, precision="bf16_storage")
            disps, _ = model(x, [1, 3], 2)
            loss = float(md().train_step(model, x, None, cache, params, do_update=False)); g = model.grads()
            def factory():
                m2, p2, c2, _, _, _ = make(N, C, H, W, pseed=61, precision="bf16_storage")
                return m2, (lambda: md().train_step(m2, x, None, c2, p2, do_update=False))
            routes = _routes_of(factory)
            model.close()
            return loss, [host(d) for d in disps], g, routes
        finally:
            md().set_winograd(True, -2)
    l0, d0, g0, r0 = run(1)
    l1, d1, g1, r1 = run(35)
    assert r0.get("conv winograd_split", 0) == 0 and r1.get("conv winograd_split", 0) >= 20, (r0, r1)
    cos = float(np.dot(g0, g1) / (np.linalg.norm(g0) * np.linalg.norm(g1)))
    print("bf16 tensors, Winograd opt-in vs default: loss %.6f / %.6f, gradient cosine %.4f, maps %s" % (l1, l0, cos, ["%.2e" % rel(a, b) for a, b in zip(d1, d0)]))
    assert np.isfinite(g1).all() and abs(l1 - l0) / l0 < 1e-2 and cos > 0.9
    for a, b in zip(d1, d0):
        assert rel(a, b) < 4e-2


FD_ARRAYS_R18 = ["enc.layer3.0.conv1.W", "enc.layer3.0.down.W", "enc.layer3.1.conv2.W", "enc.layer3.1.bn2.gamma", "enc.layer4.0.conv1.W", "enc.layer4.0.conv2.W", "enc.layer4.0.down.W",
                 "enc.layer4.1.conv1.W", "enc.layer4.1.conv2.W", "enc.layer4.1.bn1.beta", "dec.branch1.c1.W", "dec.branch1.c2.W", "pose.squeezer.W", "pose.pose1.W"]


def _bf16_storage_gradient_is_the_derivative_of_its_own_loss(model, xd, al, cache, params, theta, C, depth, want=None, steps=(3e-3, 1e-3)):
    """A wrongly wired pullback kernel of the bf16-tensor mode (1x1 ONE path, the in-loader BatchNorm transform, the 2048-channel BatchNorm sums ...) changes some parameter
    array's gradient by O(1) while the loss stays right.  For arrays of every kind of bottleneck layer in stages 3 and 4 (1x1 reductions / expansions up to 2048 channels, the
    stride-2 3x3, the strided 1x1 shortcuts, BatchNorm affine parameters) and the decoder / pose layers that read the 2048-channel features the centred finite difference of the handle's OWN loss along that array's gradient direction must reproduce the array's gradient norm:
    (L(theta + e g_a / |g_a|) - L(theta - e g_a / |g_a|)) / 2e  =  |g_a|.  Step sizes: a predicted loss change of 3e-3 and 1e-3 (relative) -- measured with tools/explore_bs.py on
    this configuration: the fp32 handle's ratio is 0.90 .. 1.02 there (1.000 +- 0.006 at 1e-4), the bf16-tensor handle's 0.73 .. 1.00 at one of the two; below 3e-4 the
    rounding of the bf16 tensors, not the slope, decides the difference.  The stem and stages 1 - 2 are left out: perturbing them moves every rounding decision of the 40+ layers
    behind them, and the loss as a function of THOSE weights is rough at every step size (stage 2: ratios 0.25 .. 0.9, stage 1: -35 .. +40): no finite difference exists to
    compare with (their kernels are the same instantiations as the later stages', at other channel counts -- and held to the oracle on bf16 tensors in tests/test_gpu_layers.py).
    Deterministic mode: the rounding of every tensor is then a fixed function of the parameters."""
    want = want or ["enc.layer3.0.down.W", "enc.layer3.1.conv1.W", "enc.layer3.2.conv2.W", "enc.layer3.2.conv3.W", "enc.layer3.5.conv1.W", "enc.layer3.5.bn2.gamma",
            "enc.layer4.0.conv1.W", "enc.layer4.0.conv2.W", "enc.layer4.0.conv3.W", "enc.layer4.0.down.W", "enc.layer4.0.down.bn.beta", "enc.layer4.2.conv1.W", "enc.layer4.2.conv3.W", "enc.layer4.2.bn3.gamma",
            "dec.branch1.c1.W", "dec.branch2.c2.W", "pose.squeezer.W", "pose.pose1.W"]
    layout = syn.param_layout(C, depth); names = [n for n, _, _ in layout]
    assert all(w in names for w in want), [w for w in want if w not in names]
    md().set_deterministic(True)
    try:
        l_ref = float(md().train_step(model, xd, al, cache, params, do_update=False))
        gd = model.grads().astype(np.float64)
        off = 0; rows = []
        for name, _, shape in layout:
            n = int(np.prod(shape))
            if name in want:
                ga = gd[off:off + n]; na = float(np.linalg.norm(ga)); ratios = []
                for step in steps:
                    eps = step * l_ref / na
                    vals = []
                    for sgn in (+1.0, -1.0):
                        th = theta.astype(np.float64).copy(); th[off:off + n] += sgn * eps * ga / na
                        model.load_params(th.astype(np.float32))
                        l, _, _, _ = md().train_loss(model, xd, al, cache, params, False); vals.append(float(l))
                    ratios.append((vals[0] - vals[1]) / (2 * eps) / na)
                rows.append((name, na, ratios[0], ratios[1]))
            off += n
    finally:
        md().set_deterministic(False)
        model.load_params(theta)
    for r in rows:
        print("bf16_storage finite difference: %-24s |g| %.4e  (dL/de) / |g| at predicted relative loss changes %g: %.3f  %g: %.3f" % (r[0], r[1], steps[0], r[2], steps[1], r[3]))
    bad = [r for r in rows if not any(0.6 < v < 1.4 for v in r[2:])]
    assert not bad, "bf16-storage gradient arrays that are not the derivative of the bf16-storage loss: " + ", ".join("%s (ratios %.3f, %.3f)" % (r[0], r[2], r[3]) for r in bad)


def test_bf16_activation_storage_on_the_other_entry_points(tmp_path):
    """The rest of the model surface on a precision = 2 handle: eval_disparity on single frames (encoder batch != 3N), Flux.testmode! (running statistics), the bit-reproducible
    mode (ordered reductions: two fresh handles agree to the last bit, bf16 rounding is deterministic), and a checkpoint round trip.  Bounds as in
    test_bf16_activation_storage_tracks_the_fp32_oracle: 5 % on the disparity maps against the fp32 oracle."""
    N, C, H, W = 3, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=6, precision="bf16_storage")
    frames = syn.synthetic_triplets(N, C, H, W, seed=16)[:, 1]
    net = oracle.OracleNet(C); net.set_params(theta)
    for n in (3, 2):
        ds = md().eval_disparity(model, dev(frames[:n])); do = net.eval_disparity(frames[:n])
        for i in range(4):
            assert ds[i].shape == do[i].shape and rel(host(ds[i]), do[i]) < 5e-2
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=17))
    # bit-reproducible mode
    def run():
        m2, p2, c2, _, _, _ = make(N, C, H, W, pseed=6, precision="bf16_storage")
        l = float(md().train_step(m2, xd, None, c2, p2, do_update=True))
        return l, m2.grads(), m2.params()
    md().set_deterministic(True)
    try:
        a_, b_ = run(), run()
    finally:
        md().set_deterministic(False)
    assert a_[0] == b_[0] and np.array_equal(a_[1], b_[1]) and np.array_equal(a_[2], b_[2])
    c_ = run()          # the default mode takes its BatchNorm statistics from the fp32 accumulators, the ordered mode from the stored (rounded) tensor: same loss to the mode's own bound (0.5 % measured)
    assert abs(c_[0] - a_[0]) < 2e-2 * abs(a_[0])
    # a few steps, then test mode (tracked statistics) and a checkpoint round trip on the same handle
    for _ in range(3):
        md().train_step(model, xd, None, cache, params, do_update=True)
    md().testmode(model, True)
    try:
        d1 = [host(t) for t in md().eval_disparity(model, dev(frames))]
        path = str(tmp_path / "bf16s.npz"); md().save_checkpoint(model, path)
        m3, p3, c3, _, _, _ = make(N, C, H, W, pseed=99, precision="bf16_storage")
        md().train_step(m3, xd, None, c3, p3, do_update=False)          # (creates the handle)
        md().load_checkpoint(m3, path); md().testmode(m3, True)
        d3 = [host(t) for t in md().eval_disparity(m3, dev(frames))]
        for u, v in zip(d1, d3):
            assert np.array_equal(u, v)
    finally:
        md().testmode(model, False)
    assert all(np.isfinite(t).all() for t in d1)


@pytest.mark.parametrize("levels", [(1, 3, 4), (5,), (2, 3)], ids=["l134", "l5", "l23"])
def test_bf16_activation_storage_with_other_scale_levels(levels):
    """DepthDecoder(; scale_levels) subsets on a precision = 2 handle (fewer branches, a 256-channel head at level 1, encoder stages without a decoder gradient): forward and
    loss against the fp32 oracle within the mode's bounds, a finite pullback, and ADAM brings the loss down from its value after the first update."""
    N, C, H, W = 2, 1, 64, 96
    m = md()
    K, iK = syn.depth10k_intrinsics(W, H)
    sc = [1.0 / 2 ** (5 - l) for l in levels]
    params = m.Params(target_size=(W, H), batch_size=N, automasking=False)
    cache = m.TrainCache(m.SSIM(), m.Backproject(W, H), m.Project(W, H), K, iK, 2, [1, 3], sc)
    enc = m.ResidualNetwork(18, in_channels=C)
    model = m.Model(enc, m.DepthDecoder(list(enc.stages), list(levels)), m.PoseDecoder(enc.stages[-1]), lr=1e-4, precision="bf16_storage")
    theta = syn.init_params(C, seed=31, randomize_affine=True, scale_levels=levels); model.load_params(theta)
    x = syn.synthetic_triplets(N, C, H, W, seed=93); xd = dev(x)
    cfg = oracle.make_cfg(N, C, H, W, K, iK, scale_levels=levels, scales=sc)
    net = oracle.OracleNet(C, scale_levels=levels); net.set_params(theta)
    model._ensure(xd, params, cache)
    ds, _ = model(xd, [1, 3], 2)
    dso, _, _ = net.model_forward(x, cfg)
    for i, l in enumerate(levels):
        assert tuple(ds[i].shape) == (N, 1, H >> (5 - l), W >> (5 - l)) and rel(host(ds[i]), dso[i]) < 5e-2
    l0 = float(m.train_step(model, xd, None, cache, params, do_update=True))
    lo = net.train_loss(x, cfg, None, False)
    assert abs(l0 - lo) / lo < 2e-2 and np.isfinite(model.grads()).all()
    ls = [float(m.train_step(model, xd, None, cache, params, do_update=True)) for _ in range(10)]
    # (on these two-triplet configurations the very first ADAM step RAISES the loss in fp32 as well -- 0.05 -> 0.25 for the single-level decoder -- and it falls from there)
    assert np.isfinite(ls).all() and ls[-1] < ls[0]


def test_eval_poses_matches_the_model_call():
    """eval_poses (src/model.jl:22-35): encoder + pose decoder alone give the poses of the full model call."""
    N, C, H, W = 2, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=17)
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=18))
    _, poses = model(xd, [1, 3], 2)
    p2 = md().eval_poses(model, xd)
    for a, b in zip(poses, p2):
        assert rel(host(a.rvec), host(b.rvec)) < 1e-5 and rel(host(a.tvec), host(b.tvec)) < 1e-5
    net = oracle.OracleNet(C); net.set_params(theta)
    _, rvo, tvo = net.model_forward(host(xd), oracle.make_cfg(N, C, H, W, K, iK))
    assert rel(np.stack([host(p.rvec) for p in p2]), rvo) < 1e-4 and rel(np.stack([host(p.tvec) for p in p2]), tvo) < 1e-4


def test_backward_requires_a_training_forward():
    """md2_backward after a plain model forward (which overwrites the activations) must fail loudly, not return stale gradients."""
    N, C, H, W = 1, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W)
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=3))
    loss, vis, _, _ = md().train_loss(model, xd, None, cache, params, True)          # do_visualization path
    assert vis is not None and vis.shape == (N, 1, H, W)
    md().gradient(model)                                                              # fine: the training forward was last
    model(xd, [1, 3], 2)
    with pytest.raises(RuntimeError):
        md().gradient(model)


def test_config2_full_size_parity_vs_oracle():
    """BASELINE.json configs[1] exactly: ResNet-18 depth+pose, 416x128, batch 4, fp32 forward + loss + backward vs the CPU oracle
    (north-star tolerance: forward+loss within 1e-4 relative)."""
    N, C, H, W = 4, 1, 128, 416
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=42)
    x = syn.synthetic_triplets(N, C, H, W, seed=42); xd = dev(x)
    ds, poses = model(xd, [1, 3], 2)
    net = oracle.OracleNet(C); net.set_params(theta)
    cfg = oracle.make_cfg(N, C, H, W, K, iK)
    dso, rvo, tvo = net.model_forward(x, cfg)
    for i in range(4):
        assert rel(host(ds[i]), dso[i]) < 1e-4
    for s in range(2):
        assert rel(host(poses[s].rvec), rvo[s]) < 1e-4 and rel(host(poses[s].tvec), tvo[s]) < 1e-4
    loss = md().train_step(model, xd, None, cache, params, do_update=False)
    lo = net.train_loss(x, cfg, None, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    gh = model.grads(); go = net.get_grads()
    assert_grads_match(gh, go, C, tag="config2 416x128 B=4")


# ---------------------------------------------------------------------------------------------------------------------------------------
# the reference's own real-image fixture: res/image.png, a Depth10k strip of 3 frames of 416x128 RGB (src/dtk.jl:29-46)
# ---------------------------------------------------------------------------------------------------------------------------------------
def _depth10k_triplet(grayscale):
    from PIL import Image
    strip = np.asarray(Image.open(os.path.join(GOLD, "depth10k_strip.png")).convert("RGB")).astype(np.float32) / np.float32(255.0)   # (128, 1248, 3)
    assert strip.shape == (128, 1248, 3)
    if grayscale:                                                  # Gray{Float32}.(frames), src/dtk.jl:33 (BT.601 luma)
        strip = (np.float32(0.299) * strip[..., 0] + np.float32(0.587) * strip[..., 1] + np.float32(0.114) * strip[..., 2])[..., None]
    frames = [strip[:, 416 * j:416 * (j + 1)] for j in range(3)]   # frames[:, (width*j+1):(width*(j+1))], src/dtk.jl:36
    return np.ascontiguousarray(np.stack([f.transpose(2, 0, 1) for f in frames], 0)[None], np.float32)          # (1, 3, C, 128, 416)


@pytest.mark.parametrize("grayscale", [False, True])
def test_real_image_loss_half_vs_oracle(grayscale):
    """Loss half on the reference-held real triplet with rough disparities and poses large enough to clamp against the border: real texture stresses the
    arg-min ties, the SSIM clamps and the :border clamping far harder than the smooth synthetic frames."""
    x = _depth10k_triplet(grayscale); N, _, C, H, W = x.shape
    rng = np.random.default_rng(17)
    model, params, cache, _, K, iK = make(N, C, H, W, automask=True)
    xd = dev(x); model._ensure(xd, params, cache)
    # disparities: a smooth random surface (a 5 x 14 grid of random levels, cubic-resampled to each scale) plus 2 % white noise -- rough enough to send samples
    # across the border and to put real texture under every SSIM window, without turning the warped frames into noise
    from scipy.ndimage import zoom
    def surface(h, w):
        g = rng.random((N, 1, 5, 14))
        return np.clip(zoom(g, (1, 1, h / 5, w / 14), order=3) * 0.9 + 0.05 + 0.02 * (rng.random((N, 1, h, w)) - 0.5), 0.02, 0.98).astype(np.float32)
    disps = [surface(H >> s, W >> s) for s in (3, 2, 1, 0)]
    rvec = ((rng.random((2, N, 3)) - 0.5) * 0.08).astype(np.float32); tvec = ((rng.random((2, N, 3)) - 0.5) * 1.0).astype(np.float32)
    cfg = oracle.make_cfg(N, C, H, W, K, iK, automasking=True)
    al = oracle.automasking_loss(x, cfg)
    ald = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3])
    assert np.abs(host(ald) - al).max() < 1e-4                      # per-pixel photometric values in [0,1]; flat real-image regions put SSIM's denominators at c2 = 9e-4
    for auto in (None, al):
        loss, dds, drv, dtv = md().loss_fwd_bwd(model, xd, [dev(d) for d in disps], dev(rvec), dev(tvec), auto_loss=None if auto is None else dev(auto))
        r = oracle.loss_fwd_bwd(x, cfg, disps, rvec, tvec, auto_loss=auto)
        assert abs(float(loss) - r["loss"]) / r["loss"] < 1e-5
        for i in range(4):
            # real texture + clamping poses: a few pixels per map sit on an arg-min / clamp decision that the two fp32 summation orders resolve differently, and such
            # a pixel's gradient changes completely (a coarse-scale pixel collects ~64 full-resolution ones).  So the statement is: at most 3 % of the pixels off by
            # more than 2e-4 of the map's maximum (measured 0 .. 1 %), the best 95 % of the pixels agree to 2e-4 in norm (measured 4e-6 .. 5e-5), and the flipped ones
            # do not turn the map (cosine > 0.999; measured >= 0.9996)
            a, b = host(dds[i]).astype(np.float64).ravel(), r["ddisp"][i].astype(np.float64).ravel()
            d = np.abs(a - b) / np.abs(b).max()
            keep = d <= np.quantile(d, 0.95)
            stats = ((d > 2e-4).mean(), np.linalg.norm((a - b)[keep]) / np.linalg.norm(b[keep]), float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))))
            print("real image ddisp[%d] auto=%s: frac>2e-4 %.4f  bulk rel %.2e  cos %.6f" % (i, auto is not None, *stats))
            assert stats[0] <= 0.03 and stats[1] < 2e-4 and stats[2] > 0.999, stats
        print("real image drvec rel %.2e dtvec rel %.2e" % (rel(host(drv), r["drvec"]), rel(host(dtv), r["dtvec"])))
        assert rel(host(drv), r["drvec"]) < 3e-2 and rel(host(dtv), r["dtvec"]) < 3e-2          # the same flipped pixels reach the twelve pose numbers (measured 5e-4 .. 1.2e-2)


def test_real_image_full_step_vs_oracle():
    """The whole training step (forward, loss, every gradient array) on the reference's real triplet, RGB, N = 1."""
    x = _depth10k_triplet(False); N, _, C, H, W = x.shape
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=71)
    xd = dev(x)
    ds, poses = model(xd, [1, 3], 2)
    net = oracle.OracleNet(C); net.set_params(theta)
    cfg = oracle.make_cfg(N, C, H, W, K, iK)
    dso, rvo, tvo = net.model_forward(x, cfg)
    for i in range(4):
        assert rel(host(ds[i]), dso[i]) < 1e-4
    loss = md().train_step(model, xd, None, cache, params, do_update=False)
    lo = net.train_loss(x, cfg, None, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    assert_grads_match(model.grads(), net.get_grads(), C, tag="real image 416x128 RGB")


# ---------------------------------------------------------------------------------------------------------------------------------------
# deterministic mode, BatchNorm running statistics / test mode, handle management, the library's own RCCL path
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("depth", [18, 50])
def test_deterministic_mode_is_bit_reproducible(depth):
    """MD2_DETERMINISTIC / md2_set_deterministic: two fresh handles, same inputs -> bit-identical loss, gradients and ADAM update (the default mode differs
    in the last bits from run to run: fp32 atomics in the weight gradients, split-K, 8-slot BatchNorm sums).  Both trunks: the ResNet-50 bottlenecks take the
    folded-BatchNorm kernels with their ordered fallbacks (column passes instead of epilogue sums, no split-K workspace)."""
    N, C, H, W = 2, 3, 64, 96
    x = dev(syn.synthetic_triplets(N, C, H, W, seed=81))
    def run():
        model, params, cache, theta, K, iK = make(N, C, H, W, automask=True, pseed=82, depth=depth)
        al = md().automasking_loss(md().SSIM(), x, x[:, 1].contiguous(), [1, 3])
        l = float(md().train_step(model, x, al, cache, params, do_update=True))
        return l, model.grads(), model.params()
    md().set_deterministic(True)
    try:
        a = run(); b = run()
    finally:
        md().set_deterministic(False)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    c = run()                                                      # and the ordered sums agree with the default mode to rounding
    assert abs(c[0] - a[0]) < (1e-6 if depth == 18 else 2e-5) * abs(a[0])
    assert np.linalg.norm(c[1] - a[1]) / np.linalg.norm(a[1]) < (2e-3 if depth == 18 else 5e-2)      # (ResNet-50: kink noise, see the fp64 criterion test)


@pytest.mark.parametrize("precision", ["f32", "bf16_storage"])
def test_bit_reproducible_mode_repeats_on_one_handle(precision):
    """Six steps of ONE handle in the bit-reproducible mode, three streams at work: every gradient vector equals the first and is finite.  The shape (15 frames of 64 x 64: 2 x 2
    maps at 1/32 scale, row-packed Winograd weight-gradient batches, ragged pixel tiles, padding taps that outnumber the real ones) is the one on which round 5 found the
    row-packed exact-split Winograd weight gradient reading LDS past its allocation: harmless beside most neighbours, non-finite weight gradients in 11 steps of 12 beside a
    kernel that leaves other bit patterns there -- while every single-launch parity test passed.  Both tensor modes are held to the same check."""
    N, C, H, W = 5, 3, 64, 64
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=21, precision=precision)
    x = dev(syn.synthetic_triplets(N, C, H, W, seed=22))
    md().set_deterministic(True)
    try:
        ref = None
        for r in range(6):
            md().train_step(model, x, None, cache, params, do_update=False); g = model.grads()
            assert np.isfinite(g).all(), "step %d: non-finite gradient" % r
            if ref is None: ref = g.copy()
            else: assert np.array_equal(ref, g), "step %d differs from step 0 in %d elements" % (r, int((ref != g).sum()))
    finally:
        md().set_deterministic(False)
    for r in range(4):                                             # default mode (atomics, fused sums, split-K workspace): finite and within rounding of the ordered result
        md().train_step(model, x, None, cache, params, do_update=False); g = model.grads()
        assert np.isfinite(g).all()
        # (bf16 tensors: another summation order rounds activations and gradients to other bf16 neighbours, and at 2 x 2 maps that moves ReLU / max-pool decisions -- measured 0.32)
        assert np.linalg.norm(g - ref) / np.linalg.norm(ref) < (2e-3 if precision == "f32" else 0.6)


STRESS_SHAPES = {"416x128_b4": (4, 1, 128, 416, 18), "resnet50_96x160": (2, 3, 96, 160, 50), "15_frames_64x64": (5, 3, 64, 64, 18)}


@pytest.mark.parametrize("precision", ["f32", "bf16_storage"])
@pytest.mark.parametrize("shape", sorted(STRESS_SHAPES))
def test_bit_reproducible_mode_holds_for_64_consecutive_steps(shape, precision):
    """64 consecutive steps of one handle in the bit-reproducible mode, all three streams at work: every gradient vector is finite and equals the first BIT FOR BIT.
    A parity test sees one step; an intermittent race inside a kernel (round 5: run-to-run differences in 5 steps of 16 with the direct-to-LDS operand loader under its
    counted vmcnt wait; round 6: the cause -- hipcc had sunk the prologue's LDS operand reads below the raw s_barrier behind which other waves overwrite that stage, DESIGN.md
    section 4 -- and the data-dependence pins that close it) shows only as a step that differs from its siblings.  Shapes: BASELINE configs[1] (416 x 128, batch 4: 128 x 64
    tiles of the loader on the encoder's 1 x 1 and stride-2 layers), ResNet-50 at 96 x 160 (bottleneck 1 x 1 layers: one-tap K loops shorter than the three-step pipeline
    on the 64-channel layers) and 15 frames of 64 x 64 (ragged tiles, padding taps that outnumber the real ones, 2 x 2 maps); fp32 tensors (exact-split products, BF = 5) and
    bf16 tensors (BF = 4).  The static counterpart, on every build: tests/test_host_and_abi.py::test_lds_pipeline_order_in_the_isa."""
    N, C, H, W, depth = STRESS_SHAPES[shape]
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=31, precision=precision, depth=depth)
    x = dev(syn.synthetic_triplets(N, C, H, W, seed=32))
    md().set_deterministic(True)
    try:
        ref = None; differing = []
        for r in range(64):
            md().train_step(model, x, None, cache, params, do_update=False); g = model.grads()
            assert np.isfinite(g).all(), "step %d: non-finite gradient" % r
            if ref is None: ref = g.copy()
            elif not np.array_equal(ref, g): differing.append((r, int((ref != g).sum())))
        assert not differing, "%d of 63 steps differ from step 0 (step, elements): %s" % (len(differing), differing[:8])
    finally:
        md().set_deterministic(False)
    # ... and 16 steps in the DEFAULT mode, the mode the bench runs in (atomics in the weight gradients, fused BatchNorm sums, split-K: not bit-reproducible by construction,
    # and other kernel routes than the ordered mode's, so on these kinked random-weight models its result sits 2e-5 .. 3e-2 (fp32) from the ordered one, the same distance
    # at every step).  An intermittent fault there shows as a step that strays from its SIBLINGS: measured spread around the first default-mode step over 64 steps
    # (tools/default_mode_spread.py, profiles/r06_race_probe.txt) <= 1.9e-6 with fp32 tensors, <= 2.1e-2 with bf16 tensors (another summation order rounds activations to
    # other bf16 neighbours); a corrupted operand tile is O(1) in some array.
    first = None; worst = 0.0
    for r in range(16):
        md().train_step(model, x, None, cache, params, do_update=False); g = model.grads()
        assert np.isfinite(g).all(), "default mode, step %d: non-finite gradient" % r
        if first is None: first = g.copy()
        else: worst = max(worst, float(np.linalg.norm(g - first) / np.linalg.norm(first)))
    assert worst < (2e-5 if precision == "f32" else 0.1), worst


@pytest.mark.parametrize("precision", ["f32", "bf16_storage"])
@pytest.mark.parametrize("shape", ["416x128_b4", "resnet50_96x160", "416x128_b32"])
def test_stream_placement_does_not_change_a_bit(shape, precision):
    """The step is enqueued on three HIP streams (the caller's dependency chain; the weight gradients; the pose decoder, the heads of every branch but the last, the loss's
    tail and the 1 x 1 shortcut branches of the down-sampling blocks -- DESIGN.md section 4, "what runs on which stream") or, with md2_set_two_streams(h, 0), on ONE.  In the
    bit-reproducible mode every reduction is ordered, so WHERE a launch is enqueued must not move a single bit: a missing fork / join edge between two streams shows as a
    difference (or as run-to-run noise) between the two programs.  Four training steps WITH the update from one initialisation on each: the loss, every gradient vector, the
    parameters after ADAM and the BatchNorm running statistics are compared bit for bit.  (Both tensor modes; ResNet-18 at BASELINE configs[1], a ResNet-50, and the
    headline batch of 32 -- above the 1.5 M pixels from which the pullback's last weight gradients stay on the caller's stream and the shortcut's pullback leaves the pose stream.)"""
    N, C, H, W, depth = (32, 1, 128, 416, 18) if shape == "416x128_b32" else STRESS_SHAPES[shape]
    xd = dev(np.concatenate([syn.synthetic_triplets(min(N, 4), C, H, W, seed=42 + i) for i in range((N + 3) // 4)])[:N].copy())
    md().set_deterministic(True)
    try:
        runs = []
        for two in (True, False):
            model, params, cache, theta, K, iK = make(N, C, H, W, pseed=41, precision=precision, depth=depth)
            model._ensure(xd, params, cache)
            md().set_two_streams(model, two)
            rec = []
            for r in range(4):
                loss = md().train_step(model, xd, None, cache, params, do_update=True)
                rec.append((np.float32(float(loss)), model.grads().copy()))
            runs.append((rec, model.params().copy(), model.bn_state().copy()))
            model.close()
    finally:
        md().set_deterministic(False)
    (ra, pa, ba), (rb, pb, bb) = runs
    for r, ((la, ga), (lb, gb)) in enumerate(zip(ra, rb)):
        assert np.isfinite(ga).all() and np.isfinite(gb).all(), "step %d: non-finite gradient" % r
        assert la == lb, "step %d: loss %r (three streams) vs %r (one stream)" % (r, la, lb)
        assert np.array_equal(ga, gb), "step %d: %d gradient elements differ between the three-stream and the one-stream program" % (r, int((ga != gb).sum()))
    assert np.array_equal(pa, pb), "%d parameters differ after four updates" % int((pa != pb).sum())
    assert np.array_equal(ba, bb), "%d BatchNorm running statistics differ" % int((ba != bb).sum())


@pytest.mark.parametrize("precision", ["f32", "bf16_storage"])
def test_step_is_immune_to_foreign_lds_contents(precision):
    """LDS is not cleared between workgroups: whatever a neighbour left there is what a kernel reads if it reads LDS it never wrote, or past its allocation -- and 0 x NaN is NaN.
    While a second stream keeps launching a kernel that fills LDS with NaN / Inf bit patterns (tests/native/lds_poison.hip, built by __graft_entry__.build()), steps in the
    bit-reproducible mode must stay finite and equal the quiet step bit for bit.  On the library as it stood before round 5's fix of the row-packed exact-split Winograd weight
    gradient, this test returns non-finite gradients in every step of the first shape (tools/poison_probe.py)."""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "liblds_poison.so")
    assert os.path.exists(so), "tests/native/liblds_poison.so is missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(so)
    lib.lds_poison_launch.restype = ctypes.c_int32
    lib.lds_poison_launch.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p]
    sink = torch.zeros(4, dtype=torch.int32, device="cuda"); side = torch.cuda.Stream()
    for (N, C, H, W) in ((5, 3, 64, 64), (2, 1, 128, 416)):
        model, params, cache, theta, K, iK = make(N, C, H, W, pseed=21, precision=precision)
        x = dev(syn.synthetic_triplets(N, C, H, W, seed=22))
        md().set_deterministic(True)
        try:
            md().train_step(model, x, None, cache, params, do_update=False); ref = model.grads().copy(); torch.cuda.synchronize()
            assert np.isfinite(ref).all()
            for kb, pattern in ((24, 0x7FC00000), (48, 0xFFFFFFFF), (8, 0x7F800000)):
                for r in range(3):
                    assert lib.lds_poison_launch(ctypes.c_void_p(side.cuda_stream), 1500, 512, kb, pattern, ctypes.c_void_p(sink.data_ptr())) == 0
                    md().train_step(model, x, None, cache, params, do_update=False); g = model.grads(); torch.cuda.synchronize()
                    assert np.isfinite(g).all(), "%dx%d, LDS pattern %08x: non-finite gradient" % (W, H, pattern)
                    assert np.array_equal(g, ref), "%dx%d, LDS pattern %08x: the step depends on foreign LDS contents" % (W, H, pattern)
        finally:
            md().set_deterministic(False)
        model.close()


def test_testmode_eval_and_bn_running_statistics_vs_oracle(tmp_path):
    """Flux.testmode! (what eval_image / eval_video run, src/Monodepth.jl:209,237): train-mode passes track the BatchNorm mu / sigma^2, test-mode
    eval_disparity normalises with them (n = 1 frame included), and the statistics travel with the checkpoint."""
    N, C, H, W = 2, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=91)
    net = oracle.OracleNet(C); net.set_params(theta)
    for seed in (92, 93):
        x = syn.synthetic_triplets(N, C, H, W, seed=seed)
        md().train_loss(model, dev(x), None, cache, params, False); net.train_loss(x, oracle.make_cfg(N, C, H, W, K, iK), None, False)
    so, sh = net.bn_state(), model.bn_state()
    assert so.shape == sh.shape and np.abs(so - sh).max() / np.abs(so).max() < 1e-4
    frame = syn.synthetic_triplets(1, C, H, W, seed=94)[:, 1]
    md().testmode(model, True); net.set_testmode(True)
    ds = md().eval_disparity(model, dev(frame)); do = net.eval_disparity(frame)
    for i in range(4):
        assert rel(host(ds[i]), do[i]) < 1e-4
    with pytest.raises(RuntimeError):
        md().train_loss(model, dev(x), None, cache, params, False)       # gradient needs trainmode!
    md().save_checkpoint(model, tmp_path / "ck.npz")
    other, _, _, _, _, _ = make(N, C, H, W, pseed=5)
    other._ensure(dev(x), params, cache); md().load_checkpoint(other, str(tmp_path / "ck.npz")); md().testmode(other, True)
    d2 = md().eval_disparity(other, dev(frame))
    for i in range(4):
        assert np.abs(host(d2[i]) - host(ds[i])).max() < 1e-6        # another handle: same statistics and weights, sums in another order
    md().testmode(model, False)
    dtr = md().eval_disparity(model, dev(frame))                          # batch statistics of ONE frame: a different answer
    assert max(np.abs(host(a) - host(b)).max() for a, b in zip(ds, dtr)) > 1e-3


def test_intrinsics_follow_the_cache_and_state_survives_a_handle_rebuild():
    """TrainCache.K is read on every call (src/training.jl:48-57), and a change of batch size rebuilds the handle WITHOUT restarting ADAM."""
    N, C, H, W = 2, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=95)
    x = syn.synthetic_triplets(N, C, H, W, seed=96); xd = dev(x)
    l1 = float(md().train_loss(model, xd, None, cache, params, False)[0])
    K2 = K.copy(); K2[0, 0] *= 1.3; K2[1, 1] *= 1.3
    cache2 = md().TrainCache(cache.ssim, cache.backprojections, cache.projections, K2, np.linalg.inv(K2), 2, [1, 3], cache.scales)
    l2 = float(md().train_loss(model, xd, None, cache2, params, False)[0])
    net = oracle.OracleNet(C); net.set_params(theta)
    lo2 = net.train_loss(x, oracle.make_cfg(N, C, H, W, K2, np.linalg.inv(K2)), None, False)
    assert abs(l2 - lo2) / lo2 < 1e-4 and abs(l1 - l2) > 1e-6
    # two ADAM steps at batch 2, then a batch-1 call (new handle): moments and beta powers must come along
    for _ in range(2):
        md().train_step(model, xd, None, cache, params, True)
    m_before = model._opt_and_bn_state()
    p1 = md().Params(target_size=(W, H), batch_size=1, automasking=False)
    md().train_loss(model, xd[:1].contiguous(), None, cache, p1, False)
    m_after = model._opt_and_bn_state()
    assert np.array_equal(m_before[0], m_after[0]) and np.array_equal(m_before[1], m_after[1]) and m_before[2:4] == m_after[2:4]
    assert abs(m_after[2] - 0.81) < 1e-12 and np.abs(m_after[0]).max() > 0


@pytest.mark.parametrize("depth", [18, 50])
def test_library_rccl_path_world1(depth):
    """md2_comm_*: the library's own RCCL communicator (what a Julia host uses), here with one rank (one GPU per box): bucketed all-reduce inside md2_backward
    on the communication stream, ADAM on the mean.  With one rank every sum is the identity, so the step must equal the plain step -- which also checks that the
    four buckets tile the gradient buffer of either trunk exactly once (a gap or an overlap would leave / double a slice)."""
    N, C, H, W = 2, 1, 64, 96
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=97))
    a, params, cache, theta, K, iK = make(N, C, H, W, pseed=98, depth=depth)
    b, _, _, _, _, _ = make(N, C, H, W, pseed=98, depth=depth)
    dp = md2amd.parallel.DataParallel(a, comm="rccl")
    md().set_deterministic(True)
    try:
        la = float(dp.train_step(xd, None, cache, params)); lb = float(md().train_step(b, xd, None, cache, params, True))
        la2 = float(dp.train_step(xd, None, cache, params)); lb2 = float(md().train_step(b, xd, None, cache, params, True))
    finally:
        md().set_deterministic(False)
    assert la == lb and la2 == lb2 and np.array_equal(a.grads(), b.grads()) and np.array_equal(a.params(), b.params())
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    assert torch.equal(a.comm_allreduce(t.clone()), t)
    # the unbucketed route (explicit md2_allreduce_grads after a local gradient) gives the same buffer
    md().train_loss(a, xd, None, cache, params, False); md().gradient(a); g1 = a.grads(); a.allreduce_grads(); torch.cuda.synchronize()
    assert np.array_equal(a.grads(), g1)


def _hip_and_oracle_grads(N, C, H, W, sd, depth=18, automask=False):
    model, params, cache, theta, K, iK = make(N, C, H, W, automask=automask, pseed=sd, depth=depth)
    x = syn.synthetic_triplets(N, C, H, W, seed=sd + 1); xd = dev(x)
    al = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3]) if automask else None
    loss = float(md().train_step(model, xd, al, cache, params, do_update=False)); gh = model.grads()
    model.close()
    return loss, gh, theta, x, (host(al) if automask else None), K, iK


# seeds from tools/run_seed_scan.sh (re-scanned in round 5, when the direct 1x1 / stride-2 convolutions got their exact-split form: every one of 20 routes -- algorithm
# automatic / Winograd forced, ordered / default reductions, fp32-MFMA / exact-split / per-launch products in the forward + input-gradient and in the weight-gradient kernels --
# within 2.2e-3 of the oracle in every array; the round-4 seed of the second case, 107, sits next to a decision that the new product form flips: 1.6e-2)
TIGHT_CASES = [(1, 1, 64, 96, 101, False), (2, 1, 64, 96, 104, False), (2, 3, 64, 96, 110, True), (3, 1, 96, 160, 108, False), (5, 3, 64, 64, 115, False),
               (1, 1, 64, 256, 108, False), (2, 3, 64, 256, 106, False),
               # round 6: the HEADLINE resolution (round-5 review: "nothing at 416 x 128 is held tighter than 5e-2").  60 seeds scanned at 416 x 128, one triplet
               # (tools/seed_scan.py 1 1 128 416 18 0 100 60): one is kink-free on all 20 routes -- seed 119, worst array 2.2e-3; the next best sit at 5.1e-3 .. 8e-3
               (1, 1, 128, 416, 119, False)]
# (name, md2_set_winograd arguments, md2_set_conv_products arguments).  "timed": the configuration bench.py's `value` is measured in.  "winograd forced": the shapes a CPU
# oracle can afford sit below the workgroup-count threshold of the automatic choice, so the Winograd forward / input-gradient / weight-gradient kernels -- 7.8 of the
# step's 9.9 ms of convolution time at batch 32 -- are pinned on every layer that qualifies, with the products of the timed configuration and with the other full-accuracy choice.
TIGHT_ROUTES = [("timed", (1, -2), None), ("winograd forced", (3, 1), None), ("winograd forced, fp32 MFMA products", (3, 1), (0, 0)), ("winograd forced, exact-split products", (3, 1), (3, 3)),
                ("winograd forced, exact-split products, 64-channel workgroups", (11, 1), (3, 3))]          # (the opt-in conv_wino3w_kernel of round 5: md2_set_winograd(3 + 8, .))


# The seeds the tight test used until round 5 re-picked them (round-5 advisor finding: re-picking seeds in the change that introduces a new product form leaves the suite
# unable to tell a near-tie decision flip from a regression of the new kernels).  They stay as cases of their own, at the bound their flipped decision needs -- measured in
# round 5 with the exact-split direct convolutions: 1.6e-2 on seed 107 -- instead of the 4e-3 of the kink-free seeds: a kernel regression moves arrays by O(1), a flipped
# ReLU / arg-min decision by ~1e-2 (profiles/r02_grad_diag.md).
RETIRED_TIGHT_SEEDS = [(2, 1, 64, 96, 107, False), (5, 3, 64, 64, 105, False)]


@pytest.mark.parametrize("case", RETIRED_TIGHT_SEEDS)
def test_gradient_parity_on_the_retired_tight_seeds(case):
    N, C, H, W, sd, automask = case
    md().set_deterministic(True)
    try:
        loss, gh, theta, x, al, K, iK = _hip_and_oracle_grads(N, C, H, W, sd, automask=automask)
    finally:
        md().set_deterministic(False)
    net = oracle.OracleNet(C); net.set_params(theta)
    lo = net.train_loss(x, oracle.make_cfg(N, C, H, W, K, iK, automasking=automask), al, True); go = net.get_grads()
    assert abs(loss - lo) / lo < 1e-4
    tab = per_array(gh, go, C)
    top = max(t[3] for t in tab)
    worst = max((t[2], t[0]) for t in tab if t[3] > 1e-3 * top)
    print("retired tight seed %s: worst array %s rel %.3e" % (case, worst[1], worst[0]))
    assert_grads_match(gh, go, C, cos_min=0.9995, rel_max=2.5e-2, tag="retired tight seed %s" % (case,))


_TIGHT_ORACLE = {}


def _routes_of(model_factory):
    """{alg: launches} of one step's convolution launches (md2_profile_dump descriptions)."""
    import csv, tempfile
    model, run = model_factory()
    run(); md().profile(model, True); run(); path = os.path.join(tempfile.mkdtemp(), "p.csv"); md().profile_dump(model, path); md().profile(model, False)
    out = {}
    for row in csv.DictReader(open(path)):
        d = row["desc"]
        if "alg=" in d:
            key = ("wgrad " if d.startswith("wgrad") else "conv ") + d.split("alg=")[1].split()[0]
            out[key] = out.get(key, 0) + 1
    model.close()
    return out


@pytest.mark.parametrize("route", TIGHT_ROUTES, ids=[r[0].replace(" ", "_").replace(",", "") for r in TIGHT_ROUTES])
@pytest.mark.parametrize("case", TIGHT_CASES)
def test_gradient_parity_tight_on_kink_free_seeds(case, route):
    """Inputs on which no ReLU / max-pool / arg-min / clamp decision differs between the HIP and the oracle arithmetic (found with tools/seed_scan.py): there EVERY
    parameter array must agree with the oracle to 4e-3 relative, cosine >= 0.9999 (measured 2e-5 .. 2.5e-3) -- C = 1 and 3, automask on / off, batch 1 .. 5, a
    ragged 160x96 frame, and two 256-wide cases in which all four encoder stages take the BatchNorm-folded convolutions (layer4 is 8 wide there).
    The statement is made on the kernels that are timed: in the default (atomics, split-K) mode AND in deterministic mode -- which since round 4 runs the same
    algorithms with ordered reductions (one workgroup per weight tile) instead of swapping the Winograd / split weight gradients out -- on every route of TIGHT_ROUTES.
    No retry in either mode."""
    N, C, H, W, sd, automask = case
    _, wino, products = route
    dflt = md().get_conv_products()
    md().set_winograd(*wino)
    if products is not None:
        md().set_conv_products(*products)
    try:
        md().set_deterministic(True)
        try:
            loss, gh, theta, x, al, K, iK = _hip_and_oracle_grads(N, C, H, W, sd, automask=automask)
        finally:
            md().set_deterministic(False)
        if case not in _TIGHT_ORACLE:          # (one oracle evaluation per case: the routes share inputs and parameters)
            net = oracle.OracleNet(C); net.set_params(theta)
            _TIGHT_ORACLE[case] = (net.train_loss(x, oracle.make_cfg(N, C, H, W, K, iK, automasking=automask), al, True), net.get_grads())
        lo, go = _TIGHT_ORACLE[case]
        assert abs(loss - lo) / lo < 1e-4
        assert_grads_match(gh, go, C, cos_min=0.9999, rel_max=4e-3, tag="tight deterministic %s %s" % (case, route[0]))
        # default mode (atomics, split-K), same inputs: the SAME bound against the oracle, no retry (round-4 advisor finding: the seeds are re-scanned with a 2x margin instead),
        # and the two modes of the library against each other at the general bound
        loss_d, gd, _, _, _, _, _ = _hip_and_oracle_grads(N, C, H, W, sd, automask=automask)
        assert abs(loss_d - lo) / lo < 1e-4
        assert_grads_match(gd, go, C, cos_min=0.9999, rel_max=4e-3, tag="tight default %s %s" % (case, route[0]))
        assert_grads_match(gd, gh, C, tag="default vs deterministic %s %s" % (case, route[0]))
        if wino[0] in (3, 11) and (H, W) == (96, 160):          # the forced routes really run the Winograd kernels (and the split ones when asked to)
            def factory():
                model, params, cache, theta2, _, _ = make(N, C, H, W, automask=automask, pseed=sd)
                xd = dev(x)
                return model, (lambda: md().train_step(model, xd, None if al is None else dev(al), cache, params, do_update=False))
            r = _routes_of(factory)
            want_pieces = md().get_conv_products()
            assert r.get("wgrad winograd", 0) >= 10 and (r.get("conv winograd_split", 0) if want_pieces[0] else r.get("conv winograd", 0)) >= 20, r
    finally:
        md().set_winograd(True, -2); md().set_conv_products(*dflt)


@pytest.mark.parametrize("products", [(0, 0), (1, 1)], ids=["fp32_mfma", "per_launch_default"])
def test_resnet50_gradients_are_as_close_to_fp64_as_the_fp32_oracle(products):
    """ResNet-50 with randomised affine parameters has no kink-free input (53 BatchNorm + ReLU layers, gradient norms of 1e2: the fp32 oracle moves by 2e-2 under a
    1-ulp input perturbation, tools/grad_diag.py).  So the reference point is an independent fp64 computation (tests/torch_ref.py on the CPU): for every array the HIP
    gradient must be as close to it as the fp32 oracle is (x1.5 + 2e-3) -- with the fp32 MFMA in every convolution and with the per-launch default (since round 5 the
    exact split in every bottleneck 1x1 as well).  Two fp32 summation orders of this network are two samples of a chaotic map, and which of them lands nearer the fp64
    gradient depends on the seed (tools/explore_r50_fp64.py, seeds 100 / 101 / 102: median distance ratio HIP : oracle 0.6-0.9 / 0.4-0.7 / 0.8-1.5, single arrays up to 5x /
    none / 29x over -- in BOTH product forms and with the oracle as the far one just as often): seed 101 is one on which the statement can be made array by array."""
    import torch_ref as tr
    N, C, H, W, sd = 2, 1, 64, 96, 101
    dflt = md().get_conv_products()
    md().set_conv_products(*products)
    try:
        loss, gh, theta, x, _, K, iK = _hip_and_oracle_grads(N, C, H, W, sd, depth=50)
    finally:
        md().set_conv_products(*dflt)
    if sd not in _R50_FP64:
        net = oracle.OracleNet(C, depth=50); net.set_params(theta)
        net.train_loss(x, oracle.make_cfg(N, C, H, W, K, iK), None, True); go = net.get_grads().astype(np.float64)
        th = torch.tensor(theta, dtype=torch.float64, requires_grad=True); x64 = torch.tensor(x, dtype=torch.float64)
        d64, p64 = tr.model_forward(th, x64, C, depth=50)
        l64 = tr.loss_from_outputs(x64, d64, p64, torch.tensor(K), torch.tensor(iK)); l64.backward()
        _R50_FP64[sd] = (go, th.grad.numpy().copy(), float(l64))
    go, g64, l64 = _R50_FP64[sd]
    assert abs(loss - l64) / l64 < 1e-4
    off = 0; bad = []; gh = gh.astype(np.float64)
    top = max(np.linalg.norm(g64[o:o + c]) for o, c in _slices(C, 50))
    for name, _, shape in syn.param_layout(C, 50):
        n = int(np.prod(shape)); s_ = slice(off, off + n); off += n
        nb = np.linalg.norm(g64[s_])
        if nb < 1e-7 * top:
            continue
        eh, eo = np.linalg.norm(gh[s_] - g64[s_]) / nb, np.linalg.norm(go[s_] - g64[s_]) / nb
        if eh > 1.5 * eo + 2e-3:
            bad.append((name, eh, eo))
    assert not bad, "HIP further from the fp64 gradient than the fp32 oracle: " + ", ".join("%s (%.2e vs %.2e)" % b for b in bad[:8])


_R50_FP64 = {}


def _slices(C, depth):
    off = 0
    for _, _, shape in syn.param_layout(C, depth):
        n = int(np.prod(shape)); yield off, n; off += n


def test_bf16_storage_trains_like_fp32(tmp_path):
    """Convergence evidence for md2_config.precision = 2 (round-5 review: "no evidence that it trains to the same place"): ResNet-18, 416 x 128, batch 4, ADAM(1e-4), 16 fixed
    synthetic batches, bit-reproducible mode; the loss is compared per PASS over the 16 batches.
    What the curves look like (tools/convergence_probe.py, 640 steps): 0.213 -> a plateau near 0.09 within two passes, then -- after a run-dependent time -- an escape to a
    second level near 0.057.  The ESCAPE TIME is chaotic: the fp32 run left the plateau at step ~70, the same fp32 run from an initialisation moved by ~8 ulp had not left
    it after 640 steps, the bf16-tensor run left at step ~470 -- and then sat at 0.0570 - 0.0581 against the fp32 run's 0.0557 - 0.0583.  So "within 5 % at steps 50 / 100 /
    150 from one initialisation" is not a property the fp32 step has of ITSELF (fp32 against fp32: up to 60 % while one of them is still on the plateau), and the test asks
    what is one:
      phase 1 -- from scratch, the first four passes (descent onto the plateau): the bf16 curve stays within max(5 %, 2 x the fp32-to-fp32 distance) of the fp32 curve, pass by pass
                 (measured 2.7 / 1.5 / 0.2 / 0.2 % against 0.7 / 5.6 / 3.0 / 0.6 %);
      phase 2 -- "the same place": the fp32 run is continued until it has escaped (pass mean below 0.075, then two more passes), its state (weights, ADAM moments, BatchNorm
                 statistics) is saved, and THREE runs continue from that checkpoint for six passes: fp32, fp32 with the weights moved by ~8 ulp, bf16 tensors.  All three must
                 hold the escaped level, and the bf16 run must stay within 5 % of the fp32 run in every pass (measured: within 1.2 %, the jittered fp32 run within 2.4 %;
                 pass means 0.0565 - 0.0571 / 0.0569 - 0.0586 / 0.0565 - 0.0571).
    If the fp32 run has not escaped after 40 passes (it does at pass 5 in this build) phase 2 is skipped, visibly."""
    B, C, H, W, nb = 4, 1, 128, 416, 16
    K, iK = syn.depth10k_intrinsics(W, H)
    xs = [dev(syn.synthetic_triplets(B, C, H, W, seed=500 + i)) for i in range(nb)]
    theta = syn.init_params(C, seed=42)
    rs = np.random.RandomState(7)
    jitter = lambda th: (th.astype(np.float64) * (1.0 + 1e-6 * rs.standard_normal(th.shape))).astype(np.float32)
    m = md()
    def new_run(prec):
        params = m.Params(target_size=(W, H), batch_size=B, automasking=False)
        cache = m.TrainCache(m.SSIM(), m.Backproject(W, H), m.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
        enc = m.ResidualNetwork(18, in_channels=C)
        model = m.Model(enc, m.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), m.PoseDecoder(enc.stages[-1]), lr=1e-4, precision=prec)
        return model, params, cache
    def passes(run, n, start=0):
        model, params, cache = run
        return np.array([float(m.train_step(model, xs[(start + i) % nb], None, cache, params, True)) for i in range(nb * n)]).reshape(n, nb).mean(1)
    md().set_deterministic(True)
    try:
        # ---- phase 1
        runs = {}
        for name, prec, th in (("f32", "f32", theta), ("f32_jitter", "f32", jitter(theta)), ("bf16", "bf16_storage", theta)):
            runs[name] = new_run(prec); runs[name][0].load_params(th)
        p1 = {k: passes(r, 4) for k, r in runs.items()}
        print("phase 1 pass means: fp32 %s | fp32 jittered %s | bf16 tensors %s" % (np.round(p1["f32"], 5), np.round(p1["f32_jitter"], 5), np.round(p1["bf16"], 5)))
        for k, c in p1.items():
            assert np.isfinite(c).all() and c[0] < 0.13 and (c[1:] < 0.105).all(), (k, c)          # 0.213 at step 1 -> the plateau
        self_dev = np.abs(p1["f32_jitter"] / p1["f32"] - 1.0); bf_dev = np.abs(p1["bf16"] / p1["f32"] - 1.0)
        assert (bf_dev < np.maximum(0.05, 2.0 * self_dev)).all(), (bf_dev, self_dev)
        runs["f32_jitter"][0].close(); runs["bf16"][0].close()
        # ---- phase 2: continue the fp32 run to its escape
        base = runs["f32"]; done = 4; low = 0; hist = []
        while done < 40 and low < 3:
            v = float(passes(base, 1)[0]); hist.append(v); done += 1
            low = low + 1 if v < 0.075 else 0
        print("fp32 run, passes 5..%d: %s" % (done, np.round(hist, 5)))
        if low < 3:
            pytest.skip("the fp32 run has not left the 0.09 plateau after 40 passes in this build: no escaped state to branch from (phase 1 held)")
        ck = str(tmp_path / "escaped.npz")
        md().save_checkpoint(base[0], ck)
        level = float(np.mean(hist[-2:]))
        branches = {"f32": base}
        for name, prec in (("f32_jitter", "f32"), ("bf16", "bf16_storage")):
            r = new_run(prec); r[0]._ensure(xs[0], r[1], r[2]); md().load_checkpoint(r[0], ck)
            if name == "f32_jitter": r[0].load_params(jitter(r[0].params()))
            branches[name] = r
        p2 = {k: passes(r, 6) for k, r in branches.items()}
        print("phase 2 (from the escaped fp32 state, level %.5f) pass means: fp32 %s | fp32 jittered %s | bf16 tensors %s" % (level, np.round(p2["f32"], 5), np.round(p2["f32_jitter"], 5), np.round(p2["bf16"], 5)))
        for k, c in p2.items():
            assert np.isfinite(c).all() and (c < 1.08 * level).all(), (k, c)                         # nobody falls back to the plateau
        assert (np.abs(p2["bf16"] / p2["f32"] - 1.0) < 0.05).all(), p2
        for r in branches.values(): r[0].close()
    finally:
        md().set_deterministic(False)


# ---------------------------------------------------------------------------------------------------------------------------------------
# the configurations bench.py measures, at their FULL batch
# ---------------------------------------------------------------------------------------------------------------------------------------
def test_config3_b32_full_parity_vs_oracle():
    """BASELINE.json configs[2]'s per-GPU shape, the one `bench.py` reports `value` on: ResNet-18, 416x128, C=1, batch 32 (96 frames through the encoder).  Batch 32
    selects other launches than batch 4 (Winograd instead of split-K direct convolutions on most layers, other tile / grid choices, the 8192-block max-pool pullback):
    disparities, poses and loss within 1e-4 of the oracle (the north-star tolerance), every gradient array compared.  (~1.5 min of oracle time on the GPU box's host cores.)"""
    N, C, H, W = 32, 1, 128, 416
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=42)
    x = syn.synthetic_triplets(N, C, H, W, seed=43); xd = dev(x)
    ds, poses = model(xd, [1, 3], 2)
    net = oracle.OracleNet(C); net.set_params(theta)
    cfg = oracle.make_cfg(N, C, H, W, K, iK)
    dso, rvo, tvo = net.model_forward(x, cfg)
    for i in range(4):
        assert rel(host(ds[i]), dso[i]) < 1e-4
    for s_ in range(2):
        assert rel(host(poses[s_].rvec), rvo[s_]) < 1e-4 and rel(host(poses[s_].tvec), tvo[s_]) < 1e-4
    loss = md().train_step(model, xd, None, cache, params, do_update=False)
    lo = net.train_loss(x, cfg, None, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    gh = model.grads(); go = net.get_grads()
    assert np.isfinite(gh).all()
    assert_grads_match(gh, go, C, tag="config3 416x128 B=32")


@pytest.mark.parametrize("cfgd", [(16, 3, 192, 640, 50, "f32", False), (8, 3, 320, 1024, 18, "bf16", True), (8, 3, 320, 1024, 18, "bf16_storage", True)],
                         ids=["config4_resnet50_640x192_b16", "config5_1024x320_bf16_automask_b8", "config5_1024x320_bf16_storage_automask_b8_as_benched"])
def test_full_batch_properties_of_configs_4_and_5(cfgd):
    """BASELINE.json configs[3] / [4] at the per-GPU batch bench.py runs them at (the oracle would need many minutes there; their N = 1 parity tests are above): finite
    loss / gradients, a deterministic forward, batch-permutation invariance of the loss half, sample 0's loss half equal to the one-sample handle's, and a loss that
    comes down under ADAM on a fixed batch."""
    N, C, H, W, depth, precision, automask = cfgd
    model, params, cache, theta, K, iK = make(N, C, H, W, automask=automask, pseed=71, depth=depth, precision=precision)
    x = syn.synthetic_triplets(N, C, H, W, seed=72); xd = dev(x)
    al = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3]) if automask else None
    l1, _, _, _ = md().train_loss(model, xd, al, cache, params, False)
    l2, _, _, _ = md().train_loss(model, xd, al, cache, params, False)
    assert np.isfinite(float(l1)) and abs(float(l1) - float(l2)) < 1e-6 * abs(float(l1))
    md().gradient(model); g = model.grads()
    assert np.isfinite(g).all() and np.linalg.norm(g) > 0
    if precision == "bf16_storage":
        # the benched variant of configuration #5 (md2_config.precision = 2 + automask, batch 8) against the fp32 step of the same library on the same batch and weights (which
        # the N = 1 tests hold to the oracle at this shape): loss within 2 %, the whole gradient and the large weight arrays keep their direction (measured, two seeds:
        # cosine 0.66 / 0.81, median array 0.35 / 0.51, worst 0.14 / 0.34 -- the bf16-operand mode on fp32 tensors: 0.90 / 0.70 / 0.52; at N = 1 the masked gradient is too small for such a statement)
        m32, p32, c32, _, _, _ = make(N, C, H, W, automask=automask, pseed=71, depth=depth, precision="f32")
        l32 = float(md().train_step(m32, xd, al, c32, p32, do_update=False)); g32 = m32.grads(); m32.close()
        assert abs(float(l1) - l32) / l32 < 2e-2
        cos_all = float(np.dot(g, g32) / (np.linalg.norm(g) * np.linalg.norm(g32)))
        tab = per_array(g, g32, C, depth); top = max(t[3] for t in tab)
        big = sorted(t[1] for t in tab if t[0].endswith(".W") and t[3] > 1e-2 * top)
        print("config 5 as benched (bf16 storage + automask, batch %d) vs the fp32 step: loss rel %.2e cos_all %.3f large arrays: worst %.3f median %.3f" % (N, abs(float(l1) - l32) / l32, cos_all, big[0], big[len(big) // 2]))
        assert cos_all > 0.5 and big[len(big) // 2] > 0.25          # (this batch: 0.66 whole vector / 0.35 median array)
        # The per-array statement (round 6; until round 5 this line read `big[0] > 0.0`, i.e. nothing): under the automask single arrays of the bf16-tensor step point up to
        # 80 degrees away from the fp32 step's (mask and arg-min decisions taken on bf16-perturbed disparities), so "same direction as fp32" cannot be asked array by array --
        # but "derivative of its OWN loss" can: with the automask off, at this shape and batch, the centred finite difference of the bf16-tensor handle's loss along each
        # stage-3 / stage-4 / decoder / pose array's gradient must reproduce that array's gradient norm.  Calibration (tools/fd_probe_r18.py, same shape, batch 8): the fp32
        # handle's ratios are 0.76 .. 1.01 at a predicted relative loss change of 1e-3 and 0.95 .. 1.01 at 3e-4; the bf16-tensor handle's 0.72 .. 1.10 and 0.83 .. 1.01.
        # A mis-wired pullback kernel moves its array's ratio by O(1).
        mfd, pfd, cfd, thfd, _, _ = make(N, C, H, W, automask=False, pseed=71, depth=depth, precision="bf16_storage")
        _bf16_storage_gradient_is_the_derivative_of_its_own_loss(mfd, xd, None, cfd, pfd, thfd, C, depth, want=FD_ARRAYS_R18, steps=(1e-3, 3e-4))
        mfd.close()
    ds, poses = model(xd, [1, 3], 2)
    rvec = torch.stack([poses[0].rvec, poses[1].rvec]); tvec = torch.stack([poses[0].tvec, poses[1].tvec])
    la, _, _, _ = md().loss_fwd_bwd(model, xd, ds, rvec, tvec, auto_loss=al, want_grads=False)
    perm = torch.randperm(N, device="cuda")
    lb, _, _, _ = md().loss_fwd_bwd(model, xd[perm].contiguous(), [d[perm].contiguous() for d in ds], rvec[:, perm].contiguous(), tvec[:, perm].contiguous(),
                                    auto_loss=(al[perm].contiguous() if automask else None), want_grads=False)
    assert abs(float(la) - float(lb)) < 2e-6 * abs(float(la))
    # sample 0 alone: the oracle's loss half on the GPU's own disparities / poses (seconds), and the mean of the per-sample values
    cfg1 = oracle.make_cfg(1, C, H, W, K, iK, automasking=automask)
    r = oracle.loss_fwd_bwd(x[:1], cfg1, [host(d[:1]) for d in ds], np.stack([host(poses[0].rvec[:1]), host(poses[1].rvec[:1])]), np.stack([host(poses[0].tvec[:1]), host(poses[1].tvec[:1])]),
                            auto_loss=(host(al[:1]) if automask else None), do_bwd=False)
    m1, p1, c1, _, _, _ = make(1, C, H, W, automask=automask, pseed=71, depth=depth, precision=precision)
    x1 = dev(x[:1]); m1._ensure(x1, p1, c1)
    lg, _, _, _ = md().loss_fwd_bwd(m1, x1, [d[:1].contiguous() for d in ds], torch.stack([poses[0].rvec[:1], poses[1].rvec[:1]]).contiguous(),
                                    torch.stack([poses[0].tvec[:1], poses[1].tvec[:1]]).contiguous(), auto_loss=(al[:1].contiguous() if automask else None), want_grads=False)
    assert abs(float(lg) - r["loss"]) / r["loss"] < 1e-5
    losses = [float(md().train_step(model, xd, al, cache, params, True)) for _ in range(10)]
    # (bf16 tensors: the first update can overshoot into a lucky dip -- 0.1006, 0.0962, 0.0997, ... 0.0998 on this batch -- so the later steps are held against the start)
    assert np.isfinite(losses).all() and losses[-1] < max(losses[:3]) and min(losses[5:]) < (losses[0] if precision == "bf16_storage" else losses[1])


def test_lr_assignment_reaches_an_existing_handle():
    """ADAM's eta lives in the handle: `model.lr = x` on a model whose handle already exists (train(model=resumed, lr=new), src/Monodepth.jl:125) must change the step the
    NEXT update takes.  First ADAM step: |delta| = eta |g| / (|g| + eps), i.e. eta wherever the gradient is not tiny."""
    N, C, H, W = 1, 1, 64, 96
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=5, lr=1e-4)
    xd = dev(syn.synthetic_triplets(N, C, H, W, seed=6))
    md().train_step(model, xd, None, cache, params, do_update=False)          # the handle exists and holds a gradient
    g = model.grads(); p0 = model.params()
    model.lr = 1e-2
    md().update(None, model)
    d = np.abs(model.params() - p0)[np.abs(g) > 1e-6]
    assert d.size > 1000 and abs(float(np.median(d)) - 1e-2) < 2e-4 and float(d.max()) < 1.01e-2


def test_library_rccl_on_every_gpu_of_the_box():
    """N > 1 on the real transport, whenever the box has more than one GPU (the driver's scaling node; skipped on the one-GPU boxes): `bench.py --gpus N` starts its own N
    ranks, verifies the library's bucketed RCCL all-reduce against world x the local gradient before timing, and must report n_gpus = rccl_ranks = N."""
    import json
    import subprocess
    import sys
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU on this box: the world > 1 RCCL path needs the multi-GPU node")
    n = min(n, 4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--batch", "2", "--height", "64", "--width", "96",
                        "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and line["config"]["rccl_ranks"] == n and "verified" in line["config"]["gradient_exchange"]


@pytest.mark.parametrize("frames", [([2, 3], 1), ([2, 1], 3), ([3, 1], 2), ([1, 3], 2)], ids=["target1", "target3_reversed_sources", "sources_swapped", "default"])
@pytest.mark.parametrize("automask", [False, True])
def test_target_and_source_frames_are_the_callers_choice(frames, automask):
    """TrainCache.target_id / source_ids (src/Monodepth.jl:44-55): which frame of the triplet the depth decoder sees, which two are warped onto it, in which order.  A source
    before the target is paired (source, target) and its pose inverted (src/model.jl:30-35, src/training.jl:29); the pose outputs follow the order of source_ids.  Forward,
    loss and every gradient array against the oracle for the non-default roles (and the default as the control)."""
    sids, tid = frames
    N, C, H, W = 2, 1, 64, 96
    m = md()
    K, iK = syn.depth10k_intrinsics(W, H)
    params = m.Params(target_size=(W, H), batch_size=N, automasking=automask)
    cache = m.TrainCache(m.SSIM(), m.Backproject(W, H), m.Project(W, H), K, iK, tid, list(sids), [0.125, 0.25, 0.5, 1.0])
    enc = m.ResidualNetwork(18, in_channels=C)
    model = m.Model(enc, m.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), m.PoseDecoder(enc.stages[-1]))
    theta = syn.init_params(C, seed=91, randomize_affine=True); model.load_params(theta)
    x = syn.synthetic_triplets(N, C, H, W, seed=92); xd = dev(x)
    cfg = oracle.make_cfg(N, C, H, W, K, iK, automasking=automask, target_id=tid, source_ids=sids)
    net = oracle.OracleNet(C); net.set_params(theta)
    ds, poses = model(xd, sids, tid)
    dso, rvo, tvo = net.model_forward(x, cfg)
    for i in range(4):
        assert rel(host(ds[i]), dso[i]) < 1e-4
    for s_ in range(2):
        assert rel(host(poses[s_].rvec), rvo[s_]) < 1e-4 and rel(host(poses[s_].tvec), tvo[s_]) < 1e-4
    al = m.automasking_loss(m.SSIM(), xd, xd[:, tid - 1].contiguous(), sids) if automask else None
    alo = oracle.automasking_loss(x, cfg) if automask else None
    if automask:
        assert np.abs(host(al) - alo).max() < 1e-4
    loss = m.train_step(model, xd, al, cache, params, do_update=False)
    lo = net.train_loss(x, cfg, alo, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    assert_grads_match(model.grads(), net.get_grads(), C, allow=ALLOW_AUTOMASK if automask else (), tag="frames %s" % (frames,))
    with pytest.raises(NotImplementedError):
        model(xd, [1, 2, 3], 2)


@pytest.mark.parametrize("levels,scales,automask", [((1, 3, 4), (0.3, 0.7, 1.1), False), ((1, 2, 3, 4, 5), None, True), ((5,), None, False), ((2, 3), None, False), ((4,), (1.0,), True)],
                         ids=["l134_own_scales", "l12345_automask", "l5", "l23", "l4_automask"])
def test_depth_decoder_scale_levels_vs_oracle(levels, scales, automask):
    """DepthDecoder(; scale_levels) / TrainCache.scales (src/depth_decoder.jl:26-47, src/Monodepth.jl:106-107, src/training.jl:42-67): any ascending subset of 1:5 -- fewer
    branches when the last level is below 5 (encoder stages without a skip consumer get no decoder gradient), a 256-channel head at level 1, a loss without a
    full-resolution level, five loss terms.  Parameter count and order, forward, loss, every gradient array and one ADAM step against the oracle."""
    N, C, H, W = 2, 1, 64, 96
    m = md()
    K, iK = syn.depth10k_intrinsics(W, H)
    sc = list(scales) if scales else [1.0 / 2 ** (5 - l) for l in levels]
    params = m.Params(target_size=(W, H), batch_size=N, automasking=automask)
    cache = m.TrainCache(m.SSIM(), m.Backproject(W, H), m.Project(W, H), K, iK, 2, [1, 3], sc)
    enc = m.ResidualNetwork(18, in_channels=C)
    model = m.Model(enc, m.DepthDecoder(list(enc.stages), list(levels)), m.PoseDecoder(enc.stages[-1]), lr=1e-3)
    theta = syn.init_params(C, seed=31, randomize_affine=True, scale_levels=levels); model.load_params(theta)
    x = syn.synthetic_triplets(N, C, H, W, seed=93); xd = dev(x)
    cfg = oracle.make_cfg(N, C, H, W, K, iK, automasking=automask, scale_levels=levels, scales=sc)
    net = oracle.OracleNet(C, scale_levels=levels); net.set_params(theta)
    model._ensure(xd, params, cache)                             # (TrainCache.scales reach the handle with the cache)
    ds, poses = model(xd, [1, 3], 2)
    assert len(ds) == len(levels) and model.params().size == theta.size == net.nparams
    assert np.array_equal(model.params(), theta)                 # Flux order in, Flux order out, through the internal layout
    dso, rvo, tvo = net.model_forward(x, cfg)
    for i, l in enumerate(levels):
        assert tuple(ds[i].shape) == (N, 1, H >> (5 - l), W >> (5 - l)) and rel(host(ds[i]), dso[i]) < 1e-4
    assert rel(host(poses[0].rvec), rvo[0]) < 1e-4 and rel(host(poses[1].tvec), tvo[1]) < 1e-4
    al = m.automasking_loss(m.SSIM(), xd, xd[:, 1].contiguous(), [1, 3]) if automask else None
    alo = oracle.automasking_loss(x, cfg) if automask else None
    # the loss half on the oracle's disparities and poses, with its pullbacks (a loss without a full-resolution level; five levels)
    # (without the mask here -- it multiplies the pixels whose decision flips between two fp32 summation orders; the training step below runs with it)
    def close(a, b):                              # an arg-min flip of ONE pixel moves a 3x3 window, i.e. up to 4 entries of a coarse level by up to ~0.1 of the largest
        d = np.abs(a.astype(np.float64) - b) / np.abs(b).max(); return (d > 2e-4).sum() <= max(0.03 * d.size, 8) and d.max() < 0.15
    loss, dds, drv, dtv = m.loss_fwd_bwd(model, xd, [dev(d) for d in dso], dev(rvo), dev(tvo))
    cfg0 = oracle.make_cfg(N, C, H, W, K, iK, scale_levels=levels, scales=sc)
    r = oracle.loss_fwd_bwd(x, cfg0, dso, rvo, tvo)
    assert abs(float(loss) - r["loss"]) / r["loss"] < 1e-5
    for i in range(len(levels)):
        assert close(host(dds[i]), r["ddisp"][i]), i
    assert rel(host(drv), r["drvec"]) < 5e-3 and rel(host(dtv), r["dtvec"]) < 5e-3
    # deterministic mode takes the other reduction path of the coarse levels (full-resolution maps + ordered resize adjoint)
    m.set_deterministic(True)
    try:
        _, dd2, _, _ = m.loss_fwd_bwd(model, xd, [dev(d) for d in dso], dev(rvo), dev(tvo))
        for i in range(len(levels)):
            assert close(host(dd2[i]), r["ddisp"][i]), i
    finally:
        m.set_deterministic(False)
    loss = m.train_step(model, xd, al, cache, params, do_update=False)
    lo = net.train_loss(x, cfg, alo, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    last = ("dec.head%d" % levels[-1], "dec.branch%d" % levels[-1])
    assert_grads_match(model.grads(), net.get_grads(), C, allow=last if automask else (), tag="levels %s" % (levels,), levels=levels)
    gh = model.grads(); m.update(m.ADAM(1e-3), model); net.set_grads(gh); net.adam_step(1e-3)          # same gradients in: the update over this parameter layout must agree tightly
    assert rel(model.params() - theta, net.get_params() - theta) < 5e-3


def test_product_modes_against_the_parity_bound():
    """md2_set_conv_products at BASELINE config #2 (416x128, B=4).  The full-accuracy modes -- fp32 MFMA (0) and the exact three-piece split (3), in the forward / input-gradient
    and in the weight-gradient kernels -- must be indistinguishable: disparities within 2e-5 of the oracle, loss 1e-4, every gradient array inside the general bound, and
    within 1e-3 of each other in every weight-gradient array that does not sit behind a kink.  The two-piece mode (2) is reduced precision: it stays inside the 1e-4 forward
    bound but with an order of magnitude less margin, which is why it is opt-in and never part of the measured default."""
    N, C, H, W = 4, 1, 128, 416
    m = md()
    dflt = m.get_conv_products()
    model, params, cache, theta, K, iK = make(N, C, H, W, pseed=7)
    x = syn.synthetic_triplets(N, C, H, W, seed=8); xd = dev(x)
    cfg = oracle.make_cfg(N, C, H, W, K, iK)
    net = oracle.OracleNet(C); net.set_params(theta)
    dso, rvo, tvo = net.model_forward(x, cfg); lo = net.train_loss(x, cfg, None, True); go = net.get_grads()
    worst = {}; grads = {}
    try:
        for mode in ((0, 0), (3, 3), (0, 3), (2, 2)):
            m.set_conv_products(*mode)
            ds, poses = model(xd, [1, 3], 2)
            worst[mode] = max(rel(host(d), o) for d, o in zip(ds, dso))
            assert worst[mode] < 1e-4 and rel(host(poses[0].rvec), rvo[0]) < 1e-4 and rel(host(poses[1].tvec), tvo[1]) < 1e-4
            loss = m.train_step(model, xd, None, cache, params, do_update=False)
            assert abs(float(loss) - lo) / lo < 1e-4
            grads[mode] = model.grads()
            assert_grads_match(grads[mode], go, C, tag="conv products %s" % (mode,))
    finally:
        m.set_conv_products(*dflt)
    assert worst[(0, 0)] < 2e-5 and worst[(3, 3)] < 2e-5 and worst[(0, 0)] < worst[(2, 2)]
    # same forward arithmetic, weight-gradient products fp32 MFMA vs exact split: the gradient arrays agree to accumulation-order noise
    ta = per_array(grads[(0, 3)], grads[(0, 0)], C)
    assert max(t[2] for t in ta if t[3] > 1e-7 * max(u[3] for u in ta)) < 1e-3, sorted(ta, key=lambda t: -t[2])[:4]


def test_scale_levels_and_frame_roles_with_the_resnet50_encoder():
    """The generalised decoder / frame roles on the other encoder (stage channels 64/256/512/1024/2048 feed the branch arithmetic, src/depth_decoder.jl:31-35): levels (1, 3, 4),
    target 3 with sources [2, 1] -- forward, loss and gradient direction against the oracle."""
    N, C, H, W = 2, 1, 96, 160          # (smaller inputs leave the stage-5 BatchNorms with a dozen samples: ill-conditioned, see the ResNet-50 tests above)
    levels, sids, tid = (1, 3, 4), [2, 1], 3
    m = md()
    K, iK = syn.depth10k_intrinsics(W, H)
    sc = [1.0 / 2 ** (5 - l) for l in levels]
    params = m.Params(target_size=(W, H), batch_size=N, automasking=False)
    cache = m.TrainCache(m.SSIM(), m.Backproject(W, H), m.Project(W, H), K, iK, tid, sids, sc)
    enc = m.ResidualNetwork(50, in_channels=C)
    model = m.Model(enc, m.DepthDecoder(list(enc.stages), list(levels)), m.PoseDecoder(enc.stages[-1]))
    theta = syn.init_params(C, seed=61, randomize_affine=True, depth=50, scale_levels=levels); model.load_params(theta)
    x = syn.synthetic_triplets(N, C, H, W, seed=62); xd = dev(x)
    cfg = oracle.make_cfg(N, C, H, W, K, iK, target_id=tid, source_ids=sids, scale_levels=levels, scales=sc)
    net = oracle.OracleNet(C, depth=50, scale_levels=levels); net.set_params(theta)
    assert net.nparams == theta.size == syn.num_params(C, 50, levels)
    model._ensure(xd, params, cache)
    ds, poses = model(xd, sids, tid)
    dso, rvo, tvo = net.model_forward(x, cfg)
    for i in range(len(levels)):
        assert rel(host(ds[i]), dso[i]) < 1e-4
    assert rel(host(poses[0].rvec), rvo[0]) < 1e-4 and rel(host(poses[1].tvec), tvo[1]) < 1e-4
    loss = m.train_step(model, xd, None, cache, params, do_update=False)
    lo = net.train_loss(x, cfg, None, True)
    assert abs(float(loss) - lo) / lo < 1e-4
    gh, go = model.grads(), net.get_grads()
    assert np.isfinite(gh).all() and float(np.dot(gh, go) / (np.linalg.norm(gh) * np.linalg.norm(go))) > 0.999
    assert_grads_match(gh, go, C, depth=50, cos_min=0.997, rel_max=6e-2, tag="resnet50 levels/frames", levels=levels)
