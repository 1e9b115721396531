"""Host time to ENQUEUE a training step against the time the GPU takes to run it (is the step launch-bound on the host?).
usage: python3 tools/host_enqueue_time.py B depth C H W [precision] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd, time
md = md2amd.monodepth; syn = md2amd.synthetic
B, depth, C, H, W = [int(v) for v in sys.argv[1:6]]
prec = sys.argv[6] if len(sys.argv) > 6 else "f32"; steps = int(sys.argv[7]) if len(sys.argv) > 7 else 50
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec)
model.load_params(syn.init_params(C, seed=42, depth=depth))
x = torch.from_numpy(np.concatenate([syn.synthetic_triplets(min(B, 4), C, H, W)] * ((B + 3) // 4))[:B].copy()).cuda()
for _ in range(5): md.train_step(model, x, None, cache, params, True)
torch.cuda.synchronize()
# (a) one step at a time, synchronised: host enqueue time of a step with an idle queue, and its GPU latency
enq = []; tot = []
for _ in range(steps):
    t0 = time.perf_counter(); md.train_step(model, x, None, cache, params, True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
# (b) back to back
t0 = time.perf_counter()
for _ in range(steps): md.train_step(model, x, None, cache, params, True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("B=%d depth=%d %dx%d C=%d %s: enqueue (idle queue) median %.3f ms, step latency median %.3f ms; back to back: enqueue %.3f ms/step, total %.3f ms/step" %
      (B, depth, W, H, C, prec, np.median(enq) * 1e3, np.median(tot) * 1e3, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
