"""Finite-difference ratios (dL/de along an array's gradient) / |g_a| of the bf16-tensor mode (and of the fp32 step beside it) for ResNet-18 arrays at one configuration, bit-reproducible mode.
usage: fd_probe_r18.py N C H W [automask]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
N, C, H, W = [int(v) for v in sys.argv[1:5]]; automask = len(sys.argv) > 5 and int(sys.argv[5]) != 0
want = ["enc.layer2.1.conv2.W", "enc.layer3.0.conv1.W", "enc.layer3.0.down.W", "enc.layer3.1.conv2.W", "enc.layer3.1.bn2.gamma", "enc.layer4.0.conv1.W", "enc.layer4.0.conv2.W", "enc.layer4.0.down.W", "enc.layer4.1.conv1.W", "enc.layer4.1.conv2.W", "enc.layer4.1.bn1.beta",
        "dec.branch1.c1.W", "dec.branch1.c2.W", "dec.branch2.c1.W", "pose.squeezer.W", "pose.pose1.W"]
K, iK = syn.depth10k_intrinsics(W, H)
x = torch.from_numpy(syn.synthetic_triplets(N, C, H, W, seed=72)).cuda()
theta = syn.init_params(C, seed=71, randomize_affine=True, depth=18)
layout = syn.param_layout(C, 18)
md.set_deterministic(True)
for prec in ("f32", "bf16_storage"):
    params = md.Params(target_size=(W, H), batch_size=N, automasking=automask)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(18, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec)
    model.load_params(theta)
    al = md.automasking_loss(md.SSIM(), x, x[:, 1].contiguous(), [1, 3]) if automask else None
    l_ref = float(md.train_step(model, x, al, cache, params, do_update=False)); gd = model.grads().astype(np.float64)
    off = 0
    for name, _, shape in layout:
        n = int(np.prod(shape))
        if name in want:
            ga = gd[off:off + n]; na = float(np.linalg.norm(ga)); out = []
            for step in (1e-2, 3e-3, 1e-3, 3e-4):
                eps = step * l_ref / na; vals = []
                for sgn in (1.0, -1.0):
                    th = theta.astype(np.float64).copy(); th[off:off + n] += sgn * eps * ga / na
                    model.load_params(th.astype(np.float32)); l, _, _, _ = md.train_loss(model, x, al, cache, params, False); vals.append(float(l))
                out.append((vals[0] - vals[1]) / (2 * eps) / na)
            print("%-12s %-24s |g| %.3e  ratio at 1e-2 %.3f  3e-3 %.3f  1e-3 %.3f  3e-4 %.3f" % (prec, name, na, *out), flush=True)
        off += n
    model.load_params(theta); model.close()
md.set_deterministic(False)
