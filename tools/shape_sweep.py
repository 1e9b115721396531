"""Step time of several image sizes in one process: python3 tools/shape_sweep.py B depth C precision HxW [HxW ...]  (per-pixel cost should not depend on the row length)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd, time
md = md2amd.monodepth; syn = md2amd.synthetic
B, depth, C = [int(v) for v in sys.argv[1:4]]; prec = sys.argv[4]
for hw in sys.argv[5:]:
    H, W = [int(v) for v in hw.split("x")]
    K, iK = syn.depth10k_intrinsics(W, H)
    params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(depth, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec, lr=float(os.environ.get("LR", "1e-4")))
    model.load_params(syn.init_params(C, seed=42, depth=depth))
    x = torch.from_numpy(np.concatenate([syn.synthetic_triplets(min(B, 4), C, H, W)] * ((B + 3) // 4))[:B].copy()).cuda()
    for _ in range(3):
        md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); steps = 10
    for _ in range(steps):
        loss = md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    print("B=%d depth=%d C=%d %s %dx%d (HxW): %.3f ms/step, %.3f us per kilo-pixel of the batch, last loss %.5f" % (B, depth, C, prec, H, W, ms, ms * 1e3 / (3 * B * H * W / 1e3), float(loss)), flush=True)
    del model, x
