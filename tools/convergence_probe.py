"""Loss curves of the fp32 step and of the bf16-tensor mode trained from ONE initialisation on the same fixed synthetic batches (bit-reproducible mode).
usage: convergence_probe.py [B depth C H W steps nbatches]    default 4 18 1 128 416 150 16   (the curves behind test_bf16_storage_trains_like_fp32: plateau near 0.09, escape to ~0.057 at a chaotic time -- run with 640 steps to see all three runs)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
B, depth, C, H, W, steps, nb = [int(v) for v in (sys.argv[1:8] if len(sys.argv) > 7 else (4, 18, 1, 128, 416, 150, 16))]
K, iK = syn.depth10k_intrinsics(W, H)
xs = [torch.from_numpy(syn.synthetic_triplets(B, C, H, W, seed=500 + i)).cuda() for i in range(nb)]
theta = syn.init_params(C, seed=42, depth=depth)
curves = {}
md.set_deterministic(True)
rs = np.random.RandomState(7); theta_p = (theta.astype(np.float64) * (1.0 + 1e-6 * rs.standard_normal(theta.shape))).astype(np.float32)          # the same initialisation moved by ~8 ulp: what an fp32 run does to itself
for prec in ("f32", "f32_perturbed_init", "bf16_storage"):
    params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(depth, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), lr=1e-4, precision=prec.split("_perturbed")[0])
    model.load_params(theta_p if "perturbed" in prec else theta)
    curves[prec] = np.array([float(md.train_step(model, xs[i % nb], None, cache, params, True)) for i in range(steps)])
    model.close()
md.set_deterministic(False)
a, b, a2 = curves["f32"], curves["bf16_storage"], curves["f32_perturbed_init"]
for s in range(nb, steps + 1, nb):
    wa, wb, w2 = a[s - nb:s].mean(), b[s - nb:s].mean(), a2[s - nb:s].mean()
    print("steps %3d..%3d (one pass over the %d batches): fp32 %.6f  fp32 from the perturbed init %.6f (ratio %.4f)  bf16 tensors %.6f  ratio %.4f" % (s - nb + 1, s, nb, wa, w2, w2 / wa, wb, wb / wa))
for s in (50, 100, 150):
    if s <= steps: print("window of 16 steps ending at step %d: fp32 %.6f  bf16 tensors %.6f  rel diff %.4f" % (s, a[s - 16:s].mean(), b[s - 16:s].mean(), abs(b[s - 16:s].mean() / a[s - 16:s].mean() - 1)))
print("per-step max rel diff over the run: %.4f; first loss %.6f / %.6f" % (np.abs(b / a - 1).max(), a[0], b[0]))
