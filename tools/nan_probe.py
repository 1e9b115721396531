"""Repeated training steps of one small configuration, reporting non-finite gradient arrays and run-to-run differences in deterministic mode (diagnostic for loader races).
usage: nan_probe.py N C H W [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
N, C, H, W = [int(v) for v in sys.argv[1:5]]; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=N, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(18, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(512), precision=os.environ.get('NANPROBE_PRECISION', 'f32'))
model.load_params(syn.init_params(C, seed=21, randomize_affine=bool(os.environ.get('NANPROBE_AFFINE'))))
x = torch.from_numpy(syn.synthetic_triplets(N, C, H, W, seed=22)).cuda()
layout = syn.param_layout(C)
if os.environ.get('NANPROBE_WINO'): md.set_winograd(*[int(v) for v in os.environ['NANPROBE_WINO'].split(',')])
def report(g, tag):
    off = 0; bad = []
    for name, _, shape in layout:
        n = int(np.prod(shape)); a = g[off:off + n]; off += n
        if not np.isfinite(a).all():
            bad.append(name)
            if len(shape) == 4 and len(bad) <= 2:          # which (output-channel, input-channel, tap) elements: a workgroup's 64 x 32 block, a row, everything?
                m = ~np.isfinite(a.reshape(shape)); idx = np.argwhere(m)
                print("   ", name, shape, "non-finite: %d of %d; dim ranges" % (m.sum(), m.size), [(int(idx[:, d].min()), int(idx[:, d].max()), len(np.unique(idx[:, d]))) for d in range(4)], flush=True)
    if bad: print(tag, "non-finite arrays:", len(bad), bad[:6], flush=True)
    return len(bad)
nbad = 0
for det in (False, True):
    md.set_deterministic(det); ref = None; ndiff = 0
    for r in range(reps):
        loss = md.train_step(model, x, None, cache, params, do_update=False); g = model.grads()
        nbad += report(g, "det=%d rep %d loss %.6f:" % (det, r, float(loss))) > 0
        if det:
            if ref is None: ref = g.copy()
            elif not np.array_equal(ref, g):
                ndiff += 1; d = np.flatnonzero(ref != g); print("det rep %d: %d elements differ, first at %d, max rel %.3e" % (r, d.size, d[0], np.abs(ref - g).max() / np.abs(ref).max()), flush=True)
    print("deterministic=%s: %d reps, %d with non-finite gradients so far, %d differing from the first" % (det, reps, nbad, ndiff), flush=True)
md.set_deterministic(False)
