"""Spread of the DEFAULT mode's gradients around the bit-reproducible mode's, step by step (on the GPU box: python3 tools/default_mode_spread.py [steps]).

The default mode (atomics in the weight gradients, fused BatchNorm sums, split-K) is not bit-reproducible by construction, so an intermittent fault there cannot show as
"a step that differs"; it can show as a step whose distance from the ordered result is an outlier among its siblings.  Prints, per shape and tensor mode, the relative
distance |g - g_ordered| / |g_ordered| of every step (minimum, median, maximum) and the distance of every step from the FIRST default-mode step (median, maximum)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
SHAPES = {"416x128_b4": (4, 1, 128, 416, 18), "resnet50_96x160": (2, 3, 96, 160, 50), "15_frames_64x64": (5, 3, 64, 64, 18)}      # tests/test_gpu_model.py STRESS_SHAPES
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for shape in sorted(SHAPES):
    for precision in ("f32", "bf16_storage"):
        N, C, H, W, depth = SHAPES[shape]
        K, iK = syn.depth10k_intrinsics(W, H)
        params = md.Params(target_size=(W, H), batch_size=N, automasking=False)
        cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
        enc = md.ResidualNetwork(depth, in_channels=C)
        model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=precision)
        model.load_params(syn.init_params(C, seed=31, randomize_affine=True, depth=depth))
        x = torch.from_numpy(syn.synthetic_triplets(N, C, H, W, seed=32)).cuda()
        md.set_deterministic(True)
        md.train_step(model, x, None, cache, params, do_update=False); ref = model.grads().copy()
        md.set_deterministic(False)
        d = []; e = []; first = None
        for r in range(steps):
            md.train_step(model, x, None, cache, params, do_update=False); g = model.grads()
            if first is None: first = g.copy()
            ok = np.isfinite(g).all()
            d.append(float(np.linalg.norm(g - ref) / np.linalg.norm(ref)) if ok else float("inf"))
            e.append(float(np.linalg.norm(g - first) / np.linalg.norm(first)) if ok else float("inf"))
        d = np.array(d); e = np.array(e[1:])
        print("%-16s %-12s %d steps: from the ordered result min %.3e median %.3e max %.3e; from the first default-mode step median %.3e max %.3e" %
              (shape, precision, steps, d.min(), np.median(d), d.max(), np.median(e), e.max()), flush=True)
