"""Generates tests/golden/*.npz with the torch-CPU implementation in tests/torch_ref.py (fp64).

Run in the build container:  python tests/golden/make_golden.py
These fixtures are TORCH-DERIVED, NOT FLUX-DERIVED: the reference is Julia with un-vendored, unpinned
dependencies and no Julia toolchain exists here, so they reduce the risk of a wrong
restatement of the documented Flux/NNlib semantics but do not pin parity against the reference itself.
Inputs are regenerated from seeds (monodepth2.jl_amd/synthetic.py); only expected outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import md2amd  # noqa: E402
import torch_ref as tr  # noqa: E402

syn = md2amd.synthetic


def net_case(name, N, C, H, W, pseed, xseed, automask):
    theta = syn.init_params(C, seed=pseed, randomize_affine=True)
    x = syn.synthetic_triplets(N, C, H, W, seed=xseed)
    K, iK = syn.depth10k_intrinsics(W, H)
    flat = torch.tensor(theta, dtype=torch.float64, requires_grad=True); xt = torch.tensor(x, dtype=torch.float64)
    disps, poses = tr.model_forward(flat, xt, C)
    al = tr.automasking_loss(xt) if automask else None
    loss = tr.loss_from_outputs(xt, disps, poses, torch.tensor(K), torch.tensor(iK), auto_loss=al)
    loss.backward()
    g = flat.grad.numpy()
    out = dict(N=N, C=C, H=H, W=W, pseed=pseed, xseed=xseed, automask=int(automask), loss=loss.item(),
               rvec=np.stack([p[0].detach().numpy() for p in poses]), tvec=np.stack([p[1].detach().numpy() for p in poses]))
    for i, d in enumerate(disps):
        out["disp%d" % i] = d.detach().numpy().astype(np.float32)
    # per-array gradient summaries (norm, sum) + full gradients of the small arrays
    off = 0; norms = []; sums = []
    for pname, kind, shape in syn.param_layout(C):
        n = int(np.prod(shape)); a = g[off:off + n]; off += n
        norms.append(np.linalg.norm(a)); sums.append(a.sum())
        if n <= 600:
            out["grad:" + pname] = a.astype(np.float64)
    out["grad_norms"] = np.array(norms); out["grad_sums"] = np.array(sums); out["grad_norm_total"] = np.linalg.norm(g)
    if al is not None:
        out["auto_loss"] = al.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item(), "gnorm", np.linalg.norm(g))


def loss_case(name, N, C, H, W, seed):
    """Loss half alone on random disparities / poses (exercises :border clamping and larger motions)."""
    rng = np.random.default_rng(seed)
    x = syn.synthetic_triplets(N, C, H, W, seed=seed)
    K, iK = syn.depth10k_intrinsics(W, H)
    disps = [rng.random((N, 1, H >> s, W >> s)) * 0.8 + 0.1 for s in (3, 2, 1, 0)]
    rvec = (rng.random((2, N, 3)) - 0.5) * 0.02; tvec = (rng.random((2, N, 3)) - 0.5) * 0.2
    xt = torch.tensor(x, dtype=torch.float64)
    dts = [torch.tensor(d, requires_grad=True) for d in disps]; rt = torch.tensor(rvec, requires_grad=True); tt = torch.tensor(tvec, requires_grad=True)
    loss = tr.loss_from_outputs(xt, dts, [(rt[0], tt[0]), (rt[1], tt[1])], torch.tensor(K), torch.tensor(iK))
    loss.backward()
    out = dict(N=N, C=C, H=H, W=W, seed=seed, loss=loss.item(), rvec=rvec, tvec=tvec, drvec=rt.grad.numpy(), dtvec=tt.grad.numpy())
    for i in range(4):
        out["disp%d" % i] = disps[i].astype(np.float32); out["ddisp%d" % i] = dts[i].grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item())


if __name__ == "__main__":
    net_case("net_c1_64x96", 1, 1, 64, 96, 11, 12, False)
    net_case("net_c3_64x96_automask", 2, 3, 64, 96, 21, 22, True)
    loss_case("loss_c1_64x96", 2, 1, 64, 96, 31)
    loss_case("loss_c3_32x64", 1, 3, 32, 64, 32)
