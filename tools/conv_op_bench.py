"""One convolution shape through the op-level entry point, repeated (rocprofv3 target for per-kernel counters): python3 tools/conv_op_bench.py N Cin H W Cout k stride pad mode [bf16] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, md2amd
md = md2amd.monodepth
N, Cin, H, W, Cout, k, s, p = [int(v) for v in sys.argv[1:9]]; mode = sys.argv[9]
bs = len(sys.argv) > 10 and sys.argv[10] == "bf16"; reps = int(sys.argv[11]) if len(sys.argv) > 11 else 5
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((N, Cin, H, W)).astype(np.float32)).cuda()
w = torch.from_numpy((rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)).cuda()
b = torch.zeros(Cout, device="cuda")
md.set_op_storage(bs)
for _ in range(reps):
    y = md.conv2d(x, w, b, s, p, mode)
torch.cuda.synchronize()
print("done", tuple(y.shape))
