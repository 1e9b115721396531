"""Per-layer timing of the conv kernels inside a real training step (HIP events per launch).
usage: layer_profile.py B out.csv [depth C H W [f32|bf16]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/layers.csv"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 18
C, H, W = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1, 128, 416)
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=(sys.argv[7] if len(sys.argv) > 7 else "f32"))
model.load_params(syn.init_params(C, seed=42, depth=depth))
x = torch.from_numpy(np.concatenate([syn.synthetic_triplets(min(B, 4), C, H, W)] * ((B + 3) // 4))[:B].copy()).cuda()
for _ in range(3):
    md.train_step(model, x, None, cache, params, True)
torch.cuda.synchronize()
md.profile(model, True)
md.train_step(model, x, None, cache, params, True)
torch.cuda.synchronize()
md.profile_dump(model, out)
import time
md.profile(model, False)
t0 = time.perf_counter()
for _ in range(10):
    md.train_step(model, x, None, cache, params, True)
torch.cuda.synchronize()
print("B=%d ms/step %.3f" % (B, (time.perf_counter() - t0) * 100))
t0 = time.perf_counter()
for _ in range(10):
    md.train_loss(model, x, None, cache, params, False)
torch.cuda.synchronize()
print("B=%d forward+loss only ms %.3f" % (B, (time.perf_counter() - t0) * 100))
