"""bf16 activation storage (precision 'bf16_storage') against the fp32 step of the same library on the same inputs: loss, disparities, gradient direction (whole vector and the
worst large per-layer array), then step time.  python3 tools/bf16s_check.py [N C H W depth]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
N, C, H, W, depth = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (2, 1, 128, 416, 18)
K, iK = syn.depth10k_intrinsics(W, H)
def build(prec):
    params = md.Params(target_size=(W, H), batch_size=N, automasking=False)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(depth, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec)
    model.load_params(syn.init_params(C, seed=61, depth=depth))
    return model, params, cache
x = torch.from_numpy(syn.synthetic_triplets(N, C, H, W, seed=62)).cuda()
res = {}
for prec in ("f32", "bf16", "bf16_storage"):
    model, params, cache = build(prec)
    disps, poses = model(x, [1, 3], 2)
    loss = float(md.train_step(model, x, None, cache, params, do_update=False))
    g = model.grads().copy()
    res[prec] = (loss, [d.float().cpu().numpy() for d in disps], g)
    print(prec, "loss %.6f  |g| %.4e  finite %s" % (loss, np.linalg.norm(g), np.isfinite(g).all()), flush=True)
    for _ in range(3): md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); print("   %.3f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
    del model
l0, d0, g0 = res["f32"]
for prec in ("bf16", "bf16_storage"):
    l, d, g = res[prec]
    print(prec, "loss rel %.3e" % (abs(l - l0) / l0), "disp rel", ["%.2e" % (np.linalg.norm(a - b) / np.linalg.norm(b)) for a, b in zip(d, d0)],
          "grad cos %.4f" % (np.dot(g, g0) / (np.linalg.norm(g) * np.linalg.norm(g0))), "norm ratio %.3f" % (np.linalg.norm(g) / np.linalg.norm(g0)))
    off = 0; rows = []
    for name, _, shape in syn.param_layout(C, depth, (2, 3, 4, 5)):
        n = int(np.prod(shape)); a = g[off:off + n].astype(np.float64); b = g0[off:off + n].astype(np.float64); off += n
        rows.append((float(np.linalg.norm(a - b)), name, float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)), float(np.linalg.norm(a)), float(np.linalg.norm(b))))
    for r in sorted(rows, reverse=True)[:6]: print("    |diff| %.3e  %-28s cos %.4f  |g| %.3e  |g_f32| %.3e" % r)
