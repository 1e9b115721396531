"""Times md2_loss_fwd_bwd (the loss half alone: smoothness sums + fused value-and-pullback kernel + finalize + pose pullback) on random disparities / poses.
usage: python tools/loss_time.py N C [H W]  -- also the target of tools/pmc_kernel.sh for the loss-kernel counters in profiles/r02_loss_train_pmc.md"""
import sys, os, time
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, torch
import md2amd
from test_gpu_model import make, dev, md, syn, host
N, C = int(sys.argv[1]), int(sys.argv[2]); H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (128, 416)
model, params, cache, theta, K, iK = make(N, C, H, W)
x = syn.synthetic_triplets(N, C, H, W, seed=3); xd = dev(x); model._ensure(xd, params, cache)
rng = np.random.default_rng(1)
disps = [dev((rng.random((N, 1, H >> s, W >> s)) * 0.5 + 0.2).astype(np.float32)) for s in (3, 2, 1, 0)]
rvec = dev(((rng.random((2, N, 3)) - 0.5) * 0.02).astype(np.float32)); tvec = dev(((rng.random((2, N, 3)) - 0.5) * 0.2).astype(np.float32))
for it in range(3):
    out = md().loss_fwd_bwd(model, xd, disps, rvec, tvec)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(20):
    out = md().loss_fwd_bwd(model, xd, disps, rvec, tvec)
e1.record(); torch.cuda.synchronize()
print("N", N, "C", C, "loss_fwd_bwd ms", e0.elapsed_time(e1) / 20, "loss", float(out[0]), flush=True)

