"""Synthetic inputs, intrinsics and parameter initialisation for the training-step hot path.

Pure numpy; shared by tests, bench.py and __graft_entry__.smoke().  Shapes follow the reference:
a Julia (W,H,C,L,N) batch is the C-contiguous numpy array (N,L,C,H,W).
"""
import numpy as np

DEC_CH = [256, 128, 64, 32, 16]
IN_CH = [512, 256, 128, 64, 32]
SKIP_CH = [256, 128, 64, 64, 0]
BLOCKS = {18: (2, 2, 2, 2), 50: (3, 4, 6, 3)}       # torchvision trunk: BasicBlock x (2,2,2,2) / Bottleneck x (3,4,6,3)


def stages(depth=18):
    """ResidualNetwork(depth).stages (src/depth_decoder.jl:31-35 consumes them)."""
    return (64, 64, 128, 256, 512) if depth == 18 else (64, 256, 512, 1024, 2048)


def decoder_channels(depth=18):
    """(in, dec, skip) channels of the five DepthDecoder branches for an encoder of this depth."""
    st = stages(depth)
    return [st[4]] + DEC_CH[:4], list(DEC_CH), [st[3], st[2], st[1], st[0], 0]


def param_layout(C=1, depth=18, scale_levels=(2, 3, 4, 5)):
    """Flat parameter order shared by the oracle and the HIP library (documented in DESIGN.md).
    Returns a list of (name, kind, shape) with kind in {"W","b","gamma","beta"}; conv weights are in
    Flux layout (kW,kH,Cin,Cout) column-major == numpy (Cout,Cin,KH,KW)."""
    if depth not in BLOCKS:
        raise ValueError("encoder depth must be 18 or 50")
    out = []

    def conv(name, cin, cout, k, bias):
        out.append((name + ".W", "W", (cout, cin, k, k)))
        if bias:
            out.append((name + ".b", "b", (cout,)))

    def bn(name, c):
        out.append((name + ".gamma", "gamma", (c,)))
        out.append((name + ".beta", "beta", (c,)))

    conv("enc.stem", C, 64, 7, False); bn("enc.stem.bn", 64)
    cin = 64
    for li, planes in enumerate((64, 128, 256, 512)):
        for bi in range(BLOCKS[depth][li]):
            s = 2 if (bi == 0 and li > 0) else 1
            p = "enc.layer%d.%d" % (li + 1, bi)
            if depth == 18:
                co = planes
                conv(p + ".conv1", cin, co, 3, False); bn(p + ".bn1", co); conv(p + ".conv2", co, co, 3, False); bn(p + ".bn2", co)
            else:                                  # bottleneck: 1x1 -> 3x3 (carries the stride) -> 1x1 x4
                co = 4 * planes
                conv(p + ".conv1", cin, planes, 1, False); bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3, False); bn(p + ".bn2", planes)
                conv(p + ".conv3", planes, co, 1, False); bn(p + ".bn3", co)
            if s != 1 or cin != co:
                conv(p + ".down", cin, co, 1, False); bn(p + ".down.bn", co)
            cin = co
    inc, dec, skc = decoder_channels(depth)
    for b in range(max(scale_levels)):             # DepthDecoder builds branches 1..max(scale_levels) and one head per listed level (src/depth_decoder.jl:38-46)
        conv("dec.branch%d.c1" % (b + 1), inc[b], dec[b], 3, True)
        conv("dec.branch%d.c2" % (b + 1), dec[b] + skc[b], dec[b], 3, True)
    for l in scale_levels:
        conv("dec.head%d" % l, dec[l - 1], 1, 3, True)
    top = stages(depth)[4]
    conv("pose.squeezer", top, 256, 1, True); conv("pose.pose1", 512, 256, 3, True); conv("pose.pose2", 256, 256, 3, True); conv("pose.pose3", 256, 6, 1, True)
    return out


def num_params(C=1, depth=18, scale_levels=(2, 3, 4, 5)):
    return int(sum(int(np.prod(s)) for _, _, s in param_layout(C, depth, scale_levels)))


def init_params(C=1, seed=42, randomize_affine=False, depth=18, scale_levels=(2, 3, 4, 5)):
    """Flux defaults: conv W Glorot-uniform, bias 0, BN gamma 1 / beta 0.  With
    randomize_affine the biases/gamma/beta are perturbed so tests exercise those gradients."""
    rng = np.random.default_rng(seed)
    parts = []
    for name, kind, shape in param_layout(C, depth, scale_levels):
        if kind == "W":
            cout, cin, kh, kw = shape
            lim = np.sqrt(24.0 / (kh * kw * cin + kh * kw * cout))
            parts.append(((rng.random(shape, dtype=np.float32) - 0.5) * lim).astype(np.float32).reshape(-1))
        elif kind == "gamma":
            g = np.ones(shape, np.float32)
            if randomize_affine:
                g = g + 0.2 * (rng.random(shape, dtype=np.float32) - 0.5)
            parts.append(g.astype(np.float32))
        else:
            parts.append((0.1 * (rng.random(shape, dtype=np.float32) - 0.5)).astype(np.float32) if randomize_affine else np.zeros(shape, np.float32))
    return np.concatenate(parts).astype(np.float32)


def depth10k_intrinsics(W=416, H=128):
    """Depth10k intrinsics, src/dtk.jl:16-22 (focal 2648/4.63461538462, principal point at the centre),
    rescaled with the image size when not 416x128."""
    f = 2648.0 / 4.63461538462 * (W / 416.0)
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float64)
    return K, np.linalg.inv(K)


def synthetic_triplets(N, C=1, H=128, W=416, seed=42, shift=2):
    """Smooth base image (4 low-frequency sinusoids + 5 % noise) in [0,1]; frames 1 and 3 are the base
    shifted by -/+ `shift` px horizontally so the warps are near-identity like real video.
    Returns float32 (N,3,C,H,W)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W + 2 * shift, dtype=np.float64), indexing="ij")
    x = np.zeros((N, 3, C, H, W), np.float32)
    for n in range(N):
        for c in range(C):
            base = np.zeros_like(xx)
            for _ in range(4):
                fx, fy = rng.uniform(0.5, 4.0, 2); ph = rng.uniform(0, 2 * np.pi); a = rng.uniform(0.5, 1.0)
                base += a * np.sin(2 * np.pi * (fx * xx / W + fy * yy / H) + ph)
            base = (base - base.min()) / (base.max() - base.min() + 1e-9) * 0.9
            for l, off in enumerate((0, shift, 2 * shift)):
                x[n, l, c] = (base[:, off:off + W] + 0.05 * rng.random((H, W))).astype(np.float32)
    return np.clip(x, 0, 1)
