"""A training step while a second stream keeps filling LDS with NaN bit patterns (tests/native/lds_poison.hip): finite gradients, equal to the quiet step in bit-reproducible mode?
usage: poison_probe.py N C H W [precision] [reps]     env: PP_DEPTH=50, PP_PRODUCTS="f,w" (md2_set_conv_products), PP_WINO="on,min_blocks" (md2_set_winograd)"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, md2amd
md = md2amd.monodepth; syn = md2amd.synthetic
N, C, H, W = [int(v) for v in sys.argv[1:5]]; prec = sys.argv[5] if len(sys.argv) > 5 else "f32"; reps = int(sys.argv[6]) if len(sys.argv) > 6 else 8
lib = ctypes.CDLL(os.path.join(ROOT, "tests", "native", "liblds_poison.so"))
lib.lds_poison_launch.restype = ctypes.c_int32
lib.lds_poison_launch.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p]
K, iK = syn.depth10k_intrinsics(W, H)
params = md.Params(target_size=(W, H), batch_size=N, automasking=False)
cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
depth = int(os.environ.get('PP_DEPTH', '18'))
enc = md.ResidualNetwork(depth, in_channels=C)
model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), precision=prec)
model.load_params(syn.init_params(C, seed=21, randomize_affine=True, depth=depth))
if os.environ.get('PP_PRODUCTS'): md.set_conv_products(*[int(v) for v in os.environ['PP_PRODUCTS'].split(',')])
if os.environ.get('PP_WINO'): md.set_winograd(*[int(v) for v in os.environ['PP_WINO'].split(',')])
x = torch.from_numpy(syn.synthetic_triplets(N, C, H, W, seed=22)).cuda()
sink = torch.zeros(4, dtype=torch.int32, device="cuda"); side = torch.cuda.Stream()
layout = syn.param_layout(C, depth=depth)
md.set_deterministic(True)
try:
    md.train_step(model, x, None, cache, params, do_update=False); ref = model.grads().copy(); torch.cuda.synchronize()
    for kb, pattern in ((24, 0x7FC00000), (48, 0xFFFFFFFF), (8, 0x7F800000)):
        bad = diff = 0
        for r in range(reps):
            lib.lds_poison_launch(ctypes.c_void_p(side.cuda_stream), 1500, 512, kb, pattern, ctypes.c_void_p(sink.data_ptr()))
            md.train_step(model, x, None, cache, params, do_update=False); g = model.grads(); torch.cuda.synchronize()
            if not np.isfinite(g).all():
                bad += 1; off = 0; names = []
                for name, _, shape in layout:
                    n = int(np.prod(shape))
                    if not np.isfinite(g[off:off + n]).all(): names.append(name)
                    off += n
                print("   rep %d non-finite arrays: %s" % (r, names[:8]), flush=True)
            elif not np.array_equal(g, ref): diff += 1
        print("%s depth %d products %s wino %s N=%d %dx%d: poison %d KB pattern %08x: %d reps, %d non-finite, %d differing from the quiet step" % (prec, depth, os.environ.get('PP_PRODUCTS', 'default'), os.environ.get('PP_WINO', 'default'), N, W, H, kb, pattern, reps, bad, diff), flush=True)
finally:
    md.set_deterministic(False)
