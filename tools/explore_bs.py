"""exploration: finite-difference ladders of the loss along gradient directions, fp32 vs bf16-storage handles (deterministic mode)"""
import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, torch, md2amd
from test_gpu_model import make, dev, md, syn, host, per_array
N, C, H, W, depth = 1, 3, 320, 1024, 18
AUTOMASK = True
md().set_deterministic(True)
theta = syn.init_params(C, seed=61, randomize_affine=True, depth=depth).astype(np.float64)
x = syn.synthetic_triplets(N, C, H, W, seed=62); xd = dev(x)
layout = syn.param_layout(C, depth)
al = md().automasking_loss(md().SSIM(), xd, xd[:, 1].contiguous(), [1, 3]) if AUTOMASK else None
for prec in ("f32", "bf16_storage"):
    model, params, cache, _, K, iK = make(N, C, H, W, automask=AUTOMASK, pseed=61, depth=depth, precision=prec)
    def L(th):
        model.load_params(th.astype(np.float32)); l, _, _, _ = md().train_loss(model, xd, al, cache, params, False); return float(l)
    model.load_params(theta.astype(np.float32))
    l0 = float(md().train_step(model, xd, al, cache, params, do_update=False)); g = model.grads().astype(np.float64)
    print(prec, "loss", l0, "|g|", np.linalg.norm(g), "repeat", L(theta), L(theta))
    dirs = [("ALL", g / np.linalg.norm(g), np.linalg.norm(g))]
    off = 0
    for name, _, shape in layout:
        n = int(np.prod(shape))
        if name in ("enc.stem.W", "enc.layer1.0.conv1.W", "enc.layer2.0.conv1.W", "enc.layer3.1.conv2.W", "enc.layer4.0.conv2.W", "enc.layer2.0.down.W", "dec.branch1.c1.W", "dec.branch5.c1.W", "dec.head5.W", "pose.pose2.W", "enc.layer4.1.bn2.gamma"):
            d = np.zeros_like(g); d[off:off + n] = g[off:off + n]; na = np.linalg.norm(d); dirs.append((name, d / na, na))
        off += n
    for name, d, na in dirs:
        out = []
        for rel_change in (1e-2, 3e-3, 1e-3, 3e-4, 1e-4, 3e-5):
            eps = rel_change * l0 / na
            fd = (L(theta + eps * d) - L(theta - eps * d)) / (2 * eps)
            out.append("%.0e:%.3f" % (rel_change, fd / na))
        print("  %-24s |g_a| %.3e  FD/|g_a| by predicted relative loss change: %s" % (name, na, "  ".join(out)), flush=True)
    model.close()
md().set_deterministic(False)
