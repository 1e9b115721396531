"""Step time at the headline shape (and optionally batch 4 / ResNet-50) for every md2_set_conv_products combination: which full-accuracy mode is the default is decided by this
table.  usage: mode_bench.py [all]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic


def run(B, C, H, W, depth, modes, steps=15):
    K, iK = syn.depth10k_intrinsics(W, H)
    params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(depth, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]))
    model.load_params(syn.init_params(C, seed=42, depth=depth))
    base = syn.synthetic_triplets(min(B, 4), C, H, W)
    x = torch.from_numpy(np.concatenate([base] * ((B + 3) // 4))[:B].copy()).cuda()
    md.train_step(model, x, None, cache, params, True)          # (creates the handle: md2_set_two_streams needs one)
    for fp, wp in modes:
        md.set_conv_products(fp, wp)
        res = []
        for two in (True, False):
            md.set_two_streams(model, two)
            for _ in range(4):
                md.train_step(model, x, None, cache, params, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                md.train_step(model, x, None, cache, params, True)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / steps * 1e3)
        md.set_two_streams(model, True)
        print("B=%d %dx%d C=%d R%d products fwd/dgrad=%d wgrad=%d: %.3f ms/step (%.0f triplets/s), one stream %.3f ms" % (B, W, H, C, depth, fp, wp, res[0], B / res[0] * 1e3, res[1]), flush=True)
    model.close()


MODES = [(1, 1), (0, 0), (0, 3), (3, 0), (3, 3), (1, 0), (0, 1), (0, 2), (1, 2), (2, 2)]
run(32, 1, 128, 416, 18, MODES)
if len(sys.argv) > 1:
    run(4, 1, 128, 416, 18, MODES, steps=30)
    run(16, 3, 192, 640, 50, MODES, steps=8)
