"""End-to-end sanity of the training step: ADAM(1e-4) on ONE fixed synthetic batch must drive the loss down steadily (a wrong sign or scale anywhere in the
fused loss pullback, the BatchNorm folding or the split-K paths shows up as divergence or a plateau).  usage: python tools/overfit_check.py [B depth C H W steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
B, depth, C, H, W, steps = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (8, 18, 1, 128, 416, 300))]
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), lr=1e-4)
model.load_params(syn.init_params(C, seed=42, depth=depth))
x = torch.from_numpy(syn.synthetic_triplets(B, C, H, W, seed=7)).cuda()
al = None
losses = []
for i in range(steps):
    losses.append(float(md.train_step(model, x, al, cache, params, True)))
    if i % 50 == 0 or i == steps - 1:
        print("step %4d loss %.6f" % (i, losses[-1]), flush=True)
l = np.array(losses)
assert np.isfinite(l).all()
print("first 10 mean %.6f  last 10 mean %.6f  ratio %.3f  max upward jump %.2e" % (l[:10].mean(), l[-10:].mean(), l[-10:].mean() / l[:10].mean(), np.diff(l).max()))
assert l[-10:].mean() < 0.95 * l[:10].mean()
