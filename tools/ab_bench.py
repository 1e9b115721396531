"""A/B step time under md2_set_winograd modes (round 5: 64- vs 32-channel workgroups of the exact-split Winograd kernel), alternating the arms to average out clock drift.
usage: ab_bench.py [config ...]   config = B,depth,C,H,W[,precision]   (default: 32,18,1,128,416  4,18,1,128,416  16,50,3,192,640)
env: AB_ARMS="1:-1,9:-1" arms as winograd_mode:fwd_products (products -1 = keep default)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
ARMS = [tuple(int(v) for v in a.split(":")) for a in os.environ.get("AB_ARMS", "1:-1,9:-1").split(",")]


def run(B, depth, C, H, W, prec="f32", steps=12, rounds=3):
    K, iK = syn.depth10k_intrinsics(W, H)
    params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(depth, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec)
    model.load_params(syn.init_params(C, seed=42, depth=depth))
    base = syn.synthetic_triplets(min(B, 4), C, H, W)
    x = torch.from_numpy(np.concatenate([base] * ((B + 3) // 4))[:B].copy()).cuda()
    md.train_step(model, x, None, cache, params, True)
    dflt = md.get_conv_products()
    res = {a: {True: [], False: []} for a in ARMS}
    for r in range(rounds):
        for a in ARMS:
            md.set_winograd(a[0], -1); md.set_conv_products(a[1] if a[1] >= 0 else dflt[0], -1)
            for two in (True, False):
                md.set_two_streams(model, two)
                for _ in range(3):
                    md.train_step(model, x, None, cache, params, True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    md.train_step(model, x, None, cache, params, True)
                torch.cuda.synchronize(); res[a][two].append((time.perf_counter() - t0) / steps * 1e3)
    md.set_winograd(1, -1); md.set_conv_products(dflt[0], -1); md.set_two_streams(model, True)
    for a in ARMS:
        print("B=%d R%d %dx%d C=%d %s winograd mode %d products %d: two streams %.3f ms (%.0f triplets/s)  one stream %.3f ms   [%s]" % (
            B, depth, W, H, C, prec, a[0], a[1], min(res[a][True]), B / min(res[a][True]) * 1e3, min(res[a][False]), " ".join("%.3f" % v for v in res[a][True])), flush=True)
    model.close()


cfgs = sys.argv[1:] or ["32,18,1,128,416", "4,18,1,128,416", "16,50,3,192,640"]
for c in cfgs:
    p = c.split(",")
    run(int(p[0]), int(p[1]), int(p[2]), int(p[3]), int(p[4]), p[5] if len(p) > 5 else "f32")
