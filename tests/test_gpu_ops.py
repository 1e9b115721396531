"""GPU parity, op level: every L2 op of src/utils.jl / src/training.jl through the C ABI vs the CPU oracle on the
same seeded inputs.  fp32; tolerance 1e-4 relative (BASELINE.json north_star) unless a test states otherwise."""
import numpy as np
import pytest
import torch

import md2amd
import oracle

pytestmark = pytest.mark.gpu
syn = md2amd.synthetic
TOL = 1e-4


def md():
    return md2amd.monodepth


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def colmajor(R):
    return np.ascontiguousarray(np.swapaxes(np.asarray(R, np.float32), -1, -2)).reshape(R.shape[0], 9)


rng = np.random.default_rng(123)


def test_rotations_hat_compose():                              # mirrors test/runtests.jl:14-50 on the device
    r = (rng.random((5, 3)).astype(np.float32) - 0.5) * 2; r[4] = 0      # includes the zero rotation
    R = host(md().so3_exp_map(dev(r))).reshape(5, 3, 3).swapaxes(1, 2)
    assert rel(R, oracle.so3_exp_map(r)) < 1e-6
    from scipy.spatial.transform import Rotation
    assert np.allclose(R[0], Rotation.from_rotvec(r[0].astype(np.float64)).as_matrix(), atol=1e-5)
    S = host(md().hat(dev(r))).reshape(5, 3, 3).swapaxes(1, 2)
    assert np.array_equal(S, oracle.hat(r))
    t = rng.random((5, 3)).astype(np.float32)
    for inv in (False, True):
        Rg, tg = md().composeT(dev(r), dev(t), inv); Ro, to = oracle.composeT(r, t, inv)
        assert rel(host(Rg).reshape(5, 3, 3).swapaxes(1, 2), Ro) < 1e-6 and rel(host(tg), to) < 1e-6
    dR = rng.random((5, 3, 3)).astype(np.float32)
    dr = host(md().so3_exp_map_pullback(dev(r), dev(colmajor(dR))))
    assert np.isfinite(dr).all() and rel(dr, oracle.so3_exp_map_bwd(r, dR)) < 1e-5


@pytest.mark.parametrize("shape", [(1, 1, 2, 2), (2, 1, 2, 2), (2, 3, 17, 23), (1, 1, 128, 416)])
def test_ssim(shape):                                          # includes the reference's 2x2 reflect-pad case (runtests.jl:55)
    x = rng.random(shape).astype(np.float32); y = rng.random(shape).astype(np.float32)
    s = host(md().SSIM()(dev(x), dev(y)))
    assert np.abs(s - oracle.ssim(x, y)).max() < 2e-5
    assert np.allclose(host(md().SSIM()(dev(x), dev(x))), 0.0, atol=1e-5)
    assert np.allclose(s, host(md().SSIM()(dev(y), dev(x))), atol=1e-6)   # symmetry
    if shape == (1, 1, 2, 2):
        o = host(md().SSIM()(dev(np.ones(shape)), dev(np.zeros(shape))))
        assert np.allclose(o, 0.5, atol=1e-1)


def test_photometric_and_prediction_loss():
    N, C, H, W = 2, 3, 20, 30
    t = rng.random((N, C, H, W)).astype(np.float32); p = [np.clip(t + 0.1 * rng.standard_normal(t.shape), 0, 1).astype(np.float32) for _ in range(2)]
    pl = host(md().photometric_loss(md().SSIM(), dev(p[0]), dev(t)))
    assert np.abs(pl - oracle.photometric_loss(p[0], t)).max() < 2e-5
    out = host(md().prediction_loss(md().SSIM(), [dev(p[0]), dev(p[1])], dev(t)))
    assert np.abs(out - oracle.prediction_loss(p, t)[0]).max() < 2e-5
    x = syn.synthetic_triplets(N, C, H, W, seed=4)
    al = host(md().automasking_loss(md().SSIM(), dev(x), dev(x[:, 1]), [1, 3]))
    cfg = oracle.make_cfg(N, C, H, W, *syn.depth10k_intrinsics(W, H))
    assert np.abs(al - oracle.automasking_loss(x, cfg)).max() < 2e-5
    # _apply_mask (src/training.jl:17-19): minimum over cat(mask, warp_loss; dims=3) -- bit-exact (md2_apply_mask)
    wl = host(md().prediction_loss(md().SSIM(), [dev(x[:, 0]), dev(x[:, 2])], dev(x[:, 1])))
    wl2 = (wl + 0.02 * rng.standard_normal(wl.shape)).astype(np.float32)
    assert np.array_equal(host(md()._apply_mask(dev(al), dev(wl2))), np.minimum(al, wl2))


def test_smooth_loss_known_answer_and_random():
    dj = np.array([[0.0, 0.1], [0.2, 0.3]], np.float32); ij = np.array([[0.1, 0.2], [0.3, 0.4]], np.float32)
    sl = float(md().smooth_loss(dev(dj.T[None]), dev(ij.T[None, None])))
    assert abs(sl - 0.2542) < 1e-4                              # test/runtests.jl:80-82
    d = rng.random((3, 33, 47)).astype(np.float32); img = rng.random((3, 3, 33, 47)).astype(np.float32)
    assert abs(float(md().smooth_loss(dev(d), dev(img))) - oracle.smooth_loss(d, img)) / oracle.smooth_loss(d, img) < 1e-5


def test_disparity_to_depth():
    d = rng.random((2, 32, 32)).astype(np.float32)
    z = host(md().disparity_to_depth(dev(d), 0.1, 100.0))
    assert rel(z, oracle.disparity_to_depth(d, 0.1, 100.0)) < 1e-6
    assert z.min() >= 0.1 - 1e-6 and z.max() <= 100.0 + 1e-3      # test/runtests.jl:85-92


def test_backproject_project_identity_warp():                   # test/runtests.jl:94-122
    res, N = 16, 2
    image = rng.random((N, 1, res, res)).astype(np.float32); depth = (rng.random((N, res * res)) * 0.99 + 0.01).astype(np.float32)
    K = np.array([[910.0, 0, res / 2], [0, 910.0, res / 2], [0, 0, 1]]); invK = np.linalg.inv(K)
    R = md().so3_exp_map(dev(np.zeros((N, 3)))); t = dev(np.zeros((N, 3)))
    pts = md().Backproject(res, res)(dev(depth), invK)
    assert rel(host(pts), oracle.backproject(depth, invK.astype(np.float32), res, res)) < 1e-6
    uv = md().Project(res, res)(pts, K, R, t)
    uvo = oracle.project(oracle.backproject(depth, invK.astype(np.float32), res, res), K.astype(np.float32), np.tile(np.eye(3, dtype=np.float32), (N, 1, 1)), np.zeros((N, 3), np.float32), res, res)
    assert np.abs(host(uv) - uvo).max() < 1e-5
    sampled = host(md().grid_sample(dev(image), uv.reshape(N, res, res, 2)))         # default padding_mode = zeros like the reference test
    assert np.allclose(image, sampled, atol=1e-3)


@pytest.mark.parametrize("border", [True, False])
def test_grid_sample_fwd_bwd(border):
    N, C, H, W = 2, 3, 19, 27
    inp = rng.random((N, C, H, W)).astype(np.float32); grid = (rng.random((N, H, W, 2)) * 2.6 - 1.3).astype(np.float32)
    grid[0, 0, 0] = (-1, -1); grid[0, 0, 1] = (1, 1); grid[0, 0, 2] = (1.0, -1.0)                 # exact corners
    mode = "border" if border else "zeros"
    out = host(md().grid_sample(dev(inp), dev(grid), mode))
    assert np.abs(out - oracle.grid_sample(inp, grid, border)).max() < 1e-5
    g = rng.random(out.shape).astype(np.float32)
    dg = host(md().grid_sample_pullback_grid(dev(inp), dev(grid), dev(g), mode))
    assert rel(dg, oracle.grid_sample_bwd(inp, grid, g, border)[0]) < 1e-4


@pytest.mark.parametrize("sizes", [((4, 13), (8, 26)), ((16, 52), (128, 416)), ((64, 208), (128, 416)), ((5, 7), (5, 7))])
def test_upsample_bilinear_fwd_bwd(sizes):
    (h, w), (H, W) = sizes
    x = rng.random((2, 3, h, w)).astype(np.float32)
    y = host(md().upsample_bilinear(dev(x), (W, H)))
    assert np.abs(y - oracle.upsample_bilinear(x, H, W)).max() < 2e-5
    g = rng.random(y.shape).astype(np.float32)
    dx = host(md().upsample_bilinear_pullback(dev(g), (w, h)))
    assert rel(dx, oracle.upsample_bilinear_bwd(g, h, w)) < 1e-4


def test_fused_warp_matches_unfused_sequence():                 # src/training.jl:48-57
    N, C, H, W = 2, 3, 64, 96
    x = syn.synthetic_triplets(N, C, H, W, seed=8); K, iK = syn.depth10k_intrinsics(W, H)
    d = (rng.random((N, 1, H, W)) * 0.8 + 0.1).astype(np.float32)
    r = ((rng.random((N, 3)) - 0.5) * 0.02).astype(np.float32); t = ((rng.random((N, 3)) - 0.5) * 0.2).astype(np.float32)
    R, to = oracle.composeT(r, t, False)
    depth = oracle.disparity_to_depth(d, 0.1, 100.0)
    uv = oracle.project(oracle.backproject(depth.reshape(N, -1), iK.astype(np.float32), H, W), K.astype(np.float32), R, to, H, W)
    ref = oracle.grid_sample(x[:, 0], uv.reshape(N, H, W, 2), True)
    out = host(md().warp(dev(d), dev(x[:, 0]), K, iK, dev(colmajor(R)), dev(to), 0.1, 100.0))
    assert np.abs(out - ref).max() < 1e-4


def test_l2_op_pullbacks_vs_oracle():
    """The op-level pullbacks a Julia `rrule` of SSIM / photometric_loss / prediction_loss / smooth_loss / warp would call (include/md2hip.h `*_bwd`), against the
    oracle's pullbacks of the same functions; shapes with reflect-pad aliases on every border (H, W small) and C = 1 / 3."""
    for (N, C, H, W) in [(2, 3, 20, 30), (1, 1, 5, 7)]:
        x = rng.random((N, C, H, W)).astype(np.float32); y = np.clip(x + 0.1 * rng.standard_normal(x.shape), 0, 1).astype(np.float32)
        g = rng.standard_normal((N, C, H, W)).astype(np.float32)
        dx, dy = md().ssim_pullback(dev(x), dev(y), dev(g)); dxo, dyo = oracle.ssim_bwd(x, y, g)
        assert rel(host(dx), dxo) < 2e-4 and rel(host(dy), dyo) < 2e-4
        g1 = rng.standard_normal((N, 1, H, W)).astype(np.float32)
        dp = md().photometric_loss_pullback(dev(x), dev(y), dev(g1))
        assert rel(host(dp), oracle.photometric_loss_bwd(x, y, g1)) < 2e-4
        # prediction_loss: the arg-min routes the upstream gradient to one prediction per pixel
        p2 = np.clip(y + 0.1 * rng.standard_normal(x.shape), 0, 1).astype(np.float32)
        _, arg = oracle.prediction_loss([x, p2], y)
        dps = md().prediction_loss_pullback([dev(x), dev(p2)], dev(y), dev(g1))
        for j, pj in enumerate([x, p2]):
            ref = oracle.photometric_loss_bwd(pj, y, g1 * (arg == j))
            assert rel(host(dps[j]), ref) < 2e-4
        d = rng.random((N, H, W)).astype(np.float32)
        dd = md().smooth_loss_pullback(dev(d), dev(x), 0.7)
        assert rel(host(dd), oracle.smooth_loss_bwd(d, x, 0.7)) < 1e-4


def test_fused_warp_pullback_vs_unfused_oracle_chain():
    """md2_warp_bwd against the oracle's chain grid_sample_bwd -> project_bwd -> backproject / disparity_to_depth pullbacks (src/training.jl:48-57)."""
    N, C, H, W = 2, 3, 48, 64
    x = syn.synthetic_triplets(N, C, H, W, seed=8); K, iK = syn.depth10k_intrinsics(W, H)
    d = (rng.random((N, 1, H, W)) * 0.8 + 0.1).astype(np.float32)
    r = ((rng.random((N, 3)) - 0.5) * 0.02).astype(np.float32); t = ((rng.random((N, 3)) - 0.5) * 0.2).astype(np.float32)
    R, to = oracle.composeT(r, t, False)
    g = rng.standard_normal((N, C, H, W)).astype(np.float32)
    dd, dR, dt = md().warp_pullback(dev(d), dev(x[:, 0]), K, iK, dev(colmajor(R)), dev(to), 0.1, 100.0, dev(g))
    # reference: the oracle's pullbacks of the unfused chain, in float64 (grid_sample -> Project -> Backproject -> disparity_to_depth)
    d64, R64, t64 = d.astype(np.float64), R.astype(np.float64), to.astype(np.float64)
    depth = oracle.disparity_to_depth(d64, 0.1, 100.0)
    pts = oracle.backproject(depth.reshape(N, -1), iK.astype(np.float64), H, W)
    uv = oracle.project(pts, K.astype(np.float64), R64, t64, H, W)
    dgrid, _ = oracle.grid_sample_bwd(x[:, 0].astype(np.float64), uv.reshape(N, H, W, 2), g.astype(np.float64), True)
    dpts, dRo, dto = oracle.project_bwd(pts, K.astype(np.float64), R64, t64, dgrid.reshape(N, H * W, 2), H, W)
    ys, xs = np.meshgrid(np.arange(1, H + 1, dtype=np.float64), np.arange(1, W + 1, dtype=np.float64), indexing="ij")
    rays = np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)], 0)                     # pixel coordinates are 1-based (src/utils.jl:50-60)
    rays = (iK.astype(np.float64) @ rays).T                                           # (P, 3)
    ddepth = (dpts * rays[None]).sum(-1).reshape(N, 1, H, W)
    mn, mx = 1.0 / 100.0, 1.0 / 0.1
    ddo = -ddepth * (mx - mn) / (d64 * (mx - mn) + mn) ** 2
    assert rel(host(dd), ddo) < 2e-4
    assert rel(host(dt), dto) < 2e-4
    dRh = host(dR).reshape(N, 3, 3).transpose(0, 2, 1)                                # column-major [n][j][i] -> R[i][j]
    assert rel(dRh, dRo) < 2e-4


def test_adam_raw_matches_oracle():
    n = 4096; p = rng.standard_normal(n).astype(np.float32)
    net = oracle.OracleNet(1); theta = np.zeros(net.nparams, np.float32); theta[:n] = p; net.set_params(theta)
    pd, m, v = dev(p), dev(np.zeros(n)), dev(np.zeros(n))
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32); gg = np.zeros(net.nparams, np.float32); gg[:n] = g
        net.set_grads(gg); net.adam_step(1e-3)
        md().adam_raw(pd, dev(g), m, v, step, lr=1e-3)
    assert rel(host(pd), net.get_params()[:n]) < 1e-6
