"""rocprofv3 target: a few training steps of one configuration (used under `rocprofv3 --kernel-trace --stats -- python3 tools/prof_config.py B depth C H W [precision] [steps]`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd, time
md = md2amd.monodepth; syn = md2amd.synthetic
B, depth, C, H, W = [int(v) for v in sys.argv[1:6]]
prec = sys.argv[6] if len(sys.argv) > 6 else "f32"; steps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec)
model.load_params(syn.init_params(C, seed=42, depth=depth))
x = torch.from_numpy(np.concatenate([syn.synthetic_triplets(min(B, 4), C, H, W)] * ((B + 3) // 4))[:B].copy()).cuda()
for _ in range(2):
    md.train_step(model, x, None, cache, params, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    md.train_step(model, x, None, cache, params, True)
torch.cuda.synchronize()
print("B=%d depth=%d %dx%d C=%d %s: %.3f ms/step" % (B, depth, W, H, C, prec, (time.perf_counter() - t0) / steps * 1e3))
