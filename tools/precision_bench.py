"""Step time of the two precisions (f32 / bf16 mixed) on a few shapes.  usage: precision_bench.py [all]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic


def run(B, C, H, W, depth, prec, steps=15):
    K, iK = syn.depth10k_intrinsics(W, H)
    params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(depth, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec)
    model.load_params(syn.init_params(C, seed=42, depth=depth))
    base = syn.synthetic_triplets(min(B, 4), C, H, W)
    x = torch.from_numpy(np.concatenate([base] * ((B + 3) // 4))[:B].copy()).cuda()
    for _ in range(4):
        md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        l = md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print("B=%d %dx%d C=%d R%d %s: %.2f ms/step %.0f triplets/s" % (B, W, H, C, depth, prec, dt * 1e3, B / dt), flush=True)
    model.close()


for prec in ("f32", "bf16"):
    run(32, 1, 128, 416, 18, prec)
    if len(sys.argv) > 1:
        run(8, 3, 320, 1024, 18, prec)
        run(16, 3, 192, 640, 50, prec)
