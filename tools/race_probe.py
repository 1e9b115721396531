"""Consecutive steps of one handle in the bit-reproducible mode; counts the steps whose gradient vector differs from the first (an intermittent race inside a kernel).
usage: race_probe.py N C H W depth precision [reps]      (tools/run_race_probe.sh runs it with the tree's library and with another build put in its place)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
N, C, H, W, depth = [int(v) for v in sys.argv[1:6]]; precision = sys.argv[6]; reps = int(sys.argv[7]) if len(sys.argv) > 7 else 64
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=N, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=precision)
model.load_params(syn.init_params(C, seed=31, randomize_affine=True, depth=depth))
x = torch.from_numpy(syn.synthetic_triplets(N, C, H, W, seed=32)).cuda()
md.set_deterministic(True)
ref = None; ndiff = 0; nbad = 0; first = []
for r in range(reps):
    md.train_step(model, x, None, cache, params, do_update=False); g = model.grads()
    nbad += not np.isfinite(g).all()
    if ref is None: ref = g.copy()
    elif not np.array_equal(ref, g):
        ndiff += 1
        if len(first) < 4: first.append((r, int((ref != g).sum())))
md.set_deterministic(False)
print("race_probe %dx%dx%dx%d depth %d %s: %d steps, %d differ from step 0, %d non-finite; first (step, elements): %s" % (N, C, H, W, depth, precision, reps, ndiff, nbad, first), flush=True)
