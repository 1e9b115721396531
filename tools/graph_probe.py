"""Probe: what would capturing the whole training step in a HIP graph buy?  The step's launches (three streams, fork / join by events) are captured through torch's graph
capture of the current stream and replayed; ADAM's host-side bias corrections are frozen in the captured arguments, so this is a TIMING probe only.
usage: graph_probe.py B [depth C H W]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
depth, C, H, W = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (18, 1, 128, 416)
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=B, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]))
model.load_params(syn.init_params(C, seed=42, depth=depth))
x = torch.from_numpy(syn.synthetic_triplets(B, C, H, W, seed=42)).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5):
        md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        md.train_step(model, x, None, cache, params, True)
    torch.cuda.synchronize(); print("B=%d eager: %.3f ms/step" % (B, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            md.train_step(model, x, None, cache, params, True)
    except Exception as e:
        print("capture failed:", str(e)[:500]); sys.exit(0)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize(); print("B=%d graph replay: %.3f ms/step" % (B, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
