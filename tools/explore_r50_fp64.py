"""exploration: per-array distance of the HIP ResNet-50 gradient from an fp64 computation, relative to the fp32 oracle's distance, under the product modes"""
import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, torch, md2amd, oracle
import torch_ref as tr
from test_gpu_model import make, dev, md, syn, host, _hip_and_oracle_grads, _slices
N, C, H, W = 2, 1, 64, 96
for sd in (100, 101, 102):
    theta = None
    res = {}
    for mode in ((0, 0), (1, 1), (3, 3)):
        md().set_conv_products(*mode)
        loss, gh, theta, x, _, K, iK = _hip_and_oracle_grads(N, C, H, W, sd, depth=50)
        res[mode] = gh.astype(np.float64)
    md().set_conv_products(1, 1)
    net = oracle.OracleNet(C, depth=50); net.set_params(theta)
    net.train_loss(x, oracle.make_cfg(N, C, H, W, K, iK), None, True); go = net.get_grads().astype(np.float64)
    th = torch.tensor(theta, dtype=torch.float64, requires_grad=True); x64 = torch.tensor(x, dtype=torch.float64)
    d64, p64 = tr.model_forward(th, x64, C, depth=50)
    l64 = tr.loss_from_outputs(x64, d64, p64, torch.tensor(K), torch.tensor(iK)); l64.backward(); g64 = th.grad.numpy()
    top = max(np.linalg.norm(g64[o:o + c]) for o, c in _slices(C, 50))
    for mode, gh in res.items():
        ratios = []; worst = (0, "")
        off = 0
        for name, _, shape in syn.param_layout(C, 50):
            n = int(np.prod(shape)); s_ = slice(off, off + n); off += n
            nb = np.linalg.norm(g64[s_])
            if nb < 1e-7 * top: continue
            eh, eo = np.linalg.norm(gh[s_] - g64[s_]) / nb, np.linalg.norm(go[s_] - g64[s_]) / nb
            ratios.append(eh / (eo + 1e-12))
            if eh - (1.5 * eo + 2e-3) > worst[0]: worst = (eh - (1.5 * eo + 2e-3), "%s eh=%.2e eo=%.2e" % (name, eh, eo))
        r = np.array(ratios)
        print("seed %d products %s: eh/eo median %.2f  90%% %.2f  max %.2f  arrays over the bound: %d  worst %s" % (sd, mode, np.median(r), np.quantile(r, 0.9), r.max(), int((r > 1.5).sum()), worst[1]), flush=True)
