"""Diagnostic (not a test): per-array gradient error of the HIP step and of the fp32 oracle against an independent fp64 computation
(tests/torch_ref.py, torch CPU) and the oracle's own sensitivity to a 1-ulp perturbation of the input, on one small case.
usage: python tools/grad_diag.py [N C H W pseed xseed [depth]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import md2amd, oracle, torch_ref as tr
md = md2amd.monodepth; syn = md2amd.synthetic
a = [int(v) for v in sys.argv[1:]] or [1, 1, 64, 96, 1, 5]
N, C, H, W, pseed, xseed = a[:6]; depth = a[6] if len(a) > 6 else 18
K, iK = syn.depth10k_intrinsics(W, H)
theta = syn.init_params(C, seed=pseed, randomize_affine=True, depth=depth)
x = syn.synthetic_triplets(N, C, H, W, seed=xseed)
params = md.Params(target_size=(W, H), batch_size=N, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1])); model.load_params(theta)
xd = torch.from_numpy(x).cuda()
lh = float(md.train_step(model, xd, None, cache, params, do_update=False)); gh = model.grads().astype(np.float64)
cfg = oracle.make_cfg(N, C, H, W, K, iK)
net = oracle.OracleNet(C, depth=depth); net.set_params(theta); lo = net.train_loss(x, cfg, None, True); go = net.get_grads().astype(np.float64)
xp = (x * (1 + 1.2e-7 * np.sign(np.random.default_rng(0).standard_normal(x.shape)))).astype(np.float32)
net.train_loss(xp, cfg, None, True); gp = net.get_grads().astype(np.float64)
th = torch.tensor(theta, dtype=torch.float64, requires_grad=True)
x64 = torch.tensor(x, dtype=torch.float64)
d64, p64 = tr.model_forward(th, x64, C, depth=depth)
l64 = tr.loss_from_outputs(x64, d64, p64, torch.tensor(K), torch.tensor(iK)); l64.backward(); g64 = th.grad.numpy()
print("loss hip %.9f oracle32 %.9f torch64 %.9f" % (lh, lo, float(l64)))
off = 0
print("%-28s %10s %10s %10s %10s %10s" % ("array", "|g64|", "hip-o32", "hip-64", "o32-64", "o32 1ulp"))
for name, _, shape in syn.param_layout(C, depth):
    n = int(np.prod(shape)); s = slice(off, off + n); off += n
    nb = np.linalg.norm(g64[s]) + 1e-300
    print("%-28s %10.3e %10.2e %10.2e %10.2e %10.2e" % (name, nb, np.linalg.norm(gh[s] - go[s]) / nb, np.linalg.norm(gh[s] - g64[s]) / nb, np.linalg.norm(go[s] - g64[s]) / nb, np.linalg.norm(gp[s] - go[s]) / nb))
