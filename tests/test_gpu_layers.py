"""GPU parity, layer level: Flux Conv / BatchNorm / MaxPool semantics through the MFMA implicit-GEMM and layer kernels
vs the CPU oracle.  Shapes cover every tile configuration the network uses (Cout 16..512, stride 1/2, reflect pad,
ragged pixel counts, channel padding 6 -> 16)."""
import numpy as np
import pytest
import torch

import md2amd
import oracle

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(7)


def md():
    return md2amd.monodepth


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, pad_mode
    (2, 64, 16, 26, 64, 3, 1, 1, "zeros"),        # layer1
    (3, 64, 16, 26, 128, 3, 2, 1, "zeros"),       # layer2 downsampling conv
    (3, 64, 16, 26, 128, 1, 2, 0, "zeros"),       # 1x1 s2 shortcut
    (2, 128, 8, 13, 256, 3, 1, 1, "zeros"),
    (1, 512, 4, 13, 256, 1, 1, 0, "zeros"),       # pose squeezer
    (2, 256, 4, 13, 6, 1, 1, 0, "zeros"),         # pose head: Cout 6 -> padded 16
    (2, 32, 9, 11, 16, 3, 1, 0, "reflect"),       # decoder, Cout 16
    (1, 96, 12, 20, 32, 3, 1, 0, "reflect"),      # decoder c2 with skip concat (Cin 96)
    (1, 16, 33, 35, 16, 3, 1, 0, "reflect"),      # ragged pixel count
    (7, 128, 16, 13, 64, 3, 1, 0, "reflect"),
    (40, 64, 32, 52, 64, 3, 1, 1, "zeros"),       # enough tiles to take the 128-wide tile path
    (8, 128, 64, 128, 128, 3, 2, 1, "zeros"),     # stride 2 with enough 4 x 16 output tiles per weight block for the transposed-read weight gradient of the bf16 tensors (round 5)
    (8, 128, 64, 128, 128, 1, 2, 0, "zeros"),     # ... and its 1x1 shortcut
    # large enough (Ho >= 64, Wo >= 128) for the LDS-patch 3x3 kernel (forward, and the input gradient with the channel roles swapped)
    (1, 16, 66, 135, 16, 3, 1, 0, "reflect"),     # decoder branch 5: ragged 8x32 tiles, one channel chunk (weights loaded once)
    (1, 96, 64, 128, 32, 3, 1, 0, "reflect"),     # decoder c2 with skip: 6 channel chunks
    (2, 32, 69, 131, 16, 3, 1, 0, "reflect"),     # several images per persistent block
    (1, 32, 64, 136, 32, 3, 1, 1, "zeros"),       # zero padding: masked halo loads
    # all-taps weight gradient (every 3x3 / stride-1 layer with Cin, Cout % 32 == 0): odd sizes -> partial tile rows, a masked last column
    (3, 32, 5, 9, 64, 3, 1, 1, "zeros"),          # Ho = 5: second tile row has one valid row; Wo = 9 -> TW = 10
    (2, 64, 7, 33, 32, 3, 1, 0, "reflect"),       # Wo = 33 -> two tiles of 18 columns, the second ragged; reflected halo
    # one-tap launches (LDS-transposed epilogue) and split-K through the workspace (few tiles, long K loop)
    (3, 256, 12, 20, 64, 1, 1, 0, "zeros"),       # ResNet-50 bottleneck conv1: 1x1, Cout 64
    (48, 64, 24, 40, 256, 1, 1, 0, "zeros"),      # ... conv3: enough pixels for the 128-wide 1x1 tile; its input gradient contracts 256 channels into 64
    (2, 1024, 3, 5, 512, 1, 1, 0, "zeros"),       # 1x1 with 64 K-steps on 30 pixels: split over the channel chunks
    (1, 512, 4, 13, 512, 3, 1, 1, "zeros"),       # the 1/32-scale 3x3 at batch 1: 8 tiles, 288 K-steps -> split 8 ways
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    N, Cin, H, W, Cout, k, s, p, mode = case
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    xin = oracle.pad_reflect(x) if mode == "reflect" else x
    yo = oracle.conv2d_fwd(xin, w, b, s, 0 if mode == "reflect" else p)
    y = host(md().conv2d(dev(x), dev(w), dev(b), s, p, mode))
    assert y.shape == yo.shape and rel(y, yo) < 1e-4
    dy = rng.standard_normal(yo.shape).astype(np.float32)
    dxo, dwo, dbo = oracle.conv2d_bwd(xin, w, dy, s, 0 if mode == "reflect" else p)
    if mode == "reflect":                                       # pull back through pad_reflect with torch's reflect-pad adjoint
        t = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.pad(t, (1, 1, 1, 1), mode="reflect").backward(torch.tensor(dxo, dtype=torch.float64))
        dxo = t.grad.numpy()
    dx, dw, db = md().conv2d_pullback(dev(x), dev(w), dev(dy), s, p, mode)
    assert rel(host(dx), dxo) < 1e-4 and rel(host(dw), dwo) < 1e-4 and rel(host(db), dbo) < 1e-4


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[4] % 64 == 0 or c[1] % 64 == 0])
@pytest.mark.parametrize("products", [0, 3], ids=["fp32_mfma", "exact_split"])
def test_direct_conv_product_forms_vs_oracle(case, products):
    """The direct implicit GEMM (Winograd off) with both full-accuracy product forms of md2_set_conv_products: the fp32 MFMA and -- round 5, conv_igemm_kernel<..., BF = 3> --
    the exact three-bf16-piece split of both operand fragments on the bf16 matrix pipe (every launch with 64-channel-wide tiles: 1x1, stride 2 and its residue-class input
    gradients, split-K, the BatchNorm-sum epilogues).  Same 1e-4 bound against the oracle; the two forms agree with each other to fp32 rounding (2e-6 of the largest output)."""
    N, Cin, H, W, Cout, k, s, p, mode = case
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    xin = oracle.pad_reflect(x) if mode == "reflect" else x
    yo = oracle.conv2d_fwd(xin, w, b, s, 0 if mode == "reflect" else p)
    dy = rng.standard_normal(yo.shape).astype(np.float32)
    dxo, dwo, dbo = oracle.conv2d_bwd(xin, w, dy, s, 0 if mode == "reflect" else p)
    if mode == "reflect":
        dxo = _reflect_adjoint(dxo, N, Cin, H, W)
    dflt = md().get_conv_products()
    md().set_winograd(False)
    try:
        md().set_conv_products(products, products)
        y = host(md().conv2d(dev(x), dev(w), dev(b), s, p, mode))
        dx, dw, db = md().conv2d_pullback(dev(x), dev(w), dev(dy), s, p, mode)
        assert y.shape == yo.shape and rel(y, yo) < 1e-4
        assert rel(host(dx), dxo) < 1e-4 and rel(host(dw), dwo) < 1e-4 and rel(host(db), dbo) < 1e-4
        if products == 3:
            md().set_conv_products(0, 0)
            y0 = host(md().conv2d(dev(x), dev(w), dev(b), s, p, mode))
            dx0, _, _ = md().conv2d_pullback(dev(x), dev(w), dev(dy), s, p, mode)
            assert rel(y, y0) < 2e-6 and rel(host(dx), host(dx0)) < 2e-6
            if Cout % 64 == 0:
                assert not np.array_equal(y, y0), "the exact-split form did not run"
    finally:
        md().set_winograd(True, -2); md().set_conv_products(*dflt)


def bf16_round(a):
    """fp32 -> nearest bf16 (ties to even), returned as fp32: what a bf16 tensor holds."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad_on_bf16_tensors(case):
    """The convolution kernels of md2_config.precision = 2 (bf16 activation tensors, bf16 weight copies, fp32 accumulation: 64-byte K-steps of 32 channels, paired taps for
    16 channels, channel-major bf16 tiles in the all-taps weight gradient) against the oracle on the bf16-rounded operands.  Products of bf16 values are exact in fp32, so what
    remains is the summation order (weight / bias gradients: 1e-4) and the final rounding of a bf16 output tensor (2^-9 per element; the reflect-pad pullback rounds twice)."""
    N, Cin, H, W, Cout, k, s, p, mode = case
    x = bf16_round(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = bf16_round((rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    xin = oracle.pad_reflect(x) if mode == "reflect" else x
    yo = oracle.conv2d_fwd(xin, w, b, s, 0 if mode == "reflect" else p)
    dy = bf16_round(rng.standard_normal(yo.shape).astype(np.float32))
    dxo, dwo, dbo = oracle.conv2d_bwd(xin, w, dy, s, 0 if mode == "reflect" else p)
    if mode == "reflect":
        t = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.pad(t, (1, 1, 1, 1), mode="reflect").backward(torch.tensor(dxo, dtype=torch.float64))
        dxo = t.grad.numpy()
    md().set_op_storage(True); md().set_winograd(False)          # the DIRECT kernels (exact products of the stored bf16 values); the Winograd form of these tensors has a test of its own
    try:
        y = host(md().conv2d(dev(x), dev(w), dev(b), s, p, mode))
        dx, dw, db = md().conv2d_pullback(dev(x), dev(w), dev(dy), s, p, mode)
        dx, dw, db = host(dx), host(dw), host(db)
    finally:
        md().set_op_storage(False); md().set_winograd(True, -2)
    assert y.shape == yo.shape and rel(y, yo) < 4e-3
    assert rel(dx, dxo) < 8e-3 and rel(dw, dwo) < 1e-4 and rel(db, dbo) < 1e-4          # (5.6e-3 measured on the 12 x 20 reflect-padded map: border pixels sum two or four rounded values)


@pytest.mark.parametrize("hw", [(10, 14), (65, 131)])           # the second shape takes the LDS-patch kernel
@pytest.mark.parametrize("act", ["relu", "elu", "sigmoid"])
def test_conv_epilogue_activations(act, hw):
    x = rng.standard_normal((2, 32) + hw).astype(np.float32); w = (rng.standard_normal((32, 32, 3, 3)) * 0.1).astype(np.float32); b = rng.standard_normal(32).astype(np.float32)
    yo = oracle.conv2d_fwd(x, w, b, 1, 1)
    ref = {"relu": np.maximum(yo, 0), "elu": np.where(yo > 0, yo, np.expm1(yo)), "sigmoid": 1 / (1 + np.exp(-yo))}[act]
    assert rel(host(md().conv2d(dev(x), dev(w), dev(b), 1, 1, "zeros", act)), ref) < 1e-4


@pytest.mark.parametrize("shape", [(6, 64, 16, 26), (3, 512, 2, 3), (12, 128, 8, 13)])
def test_batchnorm_train(shape):
    x = (rng.standard_normal(shape) * 2 + 0.7).astype(np.float32); C = shape[1]
    gam = (rng.random(C) + 0.5).astype(np.float32); bet = rng.standard_normal(C).astype(np.float32)
    yo = oracle.bn_train_fwd(x, gam, bet)[0]
    assert rel(host(md().batchnorm_train(dev(x), dev(gam), dev(bet))), yo) < 1e-4
    assert rel(host(md().batchnorm_train(dev(x), dev(gam), dev(bet), relu=True)), np.maximum(yo, 0)) < 1e-4
    dy = rng.standard_normal(shape).astype(np.float32)
    dxo, dgo, dbo = oracle.bn_train_bwd(x, gam, dy)
    dx, dg, db = md().batchnorm_train_pullback(dev(x), dev(gam), dev(dy))
    assert rel(host(dx), dxo) < 2e-4 and rel(host(dg), dgo) < 1e-4 and rel(host(db), dbo) < 1e-4


def test_maxpool():
    x = np.maximum(rng.standard_normal((3, 64, 18, 26)), 0).astype(np.float32)     # ReLU output: exact-zero ties exercise the first-max rule
    assert np.array_equal(host(md().maxpool3s2(dev(x))), oracle.maxpool3s2_fwd(x))
    dy = rng.standard_normal((3, 64, 9, 13)).astype(np.float32)
    assert rel(host(md().maxpool3s2_pullback(dev(x), dev(dy))), oracle.maxpool3s2_bwd(x, dy)) < 1e-6


# ---- the direct kernels the network uses besides the implicit GEMM (never reached through md2_conv2d_*) --------------------------------
def _reflect_adjoint(dxpad, N, C, H, W):
    t = torch.zeros(N, C, H, W, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.pad(t, (1, 1, 1, 1), mode="reflect").backward(torch.tensor(dxpad, dtype=torch.float64))
    return t.grad.numpy()


@pytest.mark.parametrize("case", [(3, 1, 64, 96), (6, 3, 64, 96), (3, 1, 34, 70), (3, 3, 18, 130), (12, 1, 128, 416)])
def test_stem_direct_kernels_vs_oracle(case):
    """stem_fwd_kernel / stem_wgrad_kernel (7x7, stride 2, pad 3 on the boundary tensor) incl. the BatchNorm statistics of the epilogue:
    C in {1,3}, B*3 images, ragged tiles (output 17x35, 9x65), and the headline frame size."""
    N, C, H, W = case
    x = rng.random((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((64, C, 7, 7)) / np.sqrt(49 * C)).astype(np.float32)
    zo = oracle.conv2d_fwd(x, w, None, 2, 3)
    z, mean, invstd = md().stem_conv(dev(x), dev(w))
    assert z.shape == zo.shape and rel(host(z), zo) < 1e-4
    m64 = zo.astype(np.float64).mean((0, 2, 3)); v64 = zo.astype(np.float64).var((0, 2, 3))
    assert np.abs(host(mean) - m64).max() < 1e-4 * max(1.0, np.abs(m64).max()) and rel(host(invstd), 1.0 / np.sqrt(v64 + 1e-5)) < 1e-4
    dz = rng.standard_normal(zo.shape).astype(np.float32)
    _, dwo, _ = oracle.conv2d_bwd(x, w, dz, 2, 3)
    dw = host(md().stem_conv_wgrad(dev(x), dev(dz)))
    assert dw.shape == dwo.shape and rel(dw, dwo) < 1e-4          # max over ALL 64*C*49 weights: a wrong MFMA-layout index would move whole rows


@pytest.mark.parametrize("case", [(3, 1, 64, 96), (6, 3, 64, 96), (3, 1, 34, 70), (3, 3, 18, 130), (12, 1, 128, 416)])
def test_stem_kernels_with_bf16_products_vs_oracle(case):
    """The stem of the bf16 modes (md2_config.precision >= 1) on v_mfma_f32_32x32x16_bf16, the contraction re-ordered into (channel, filter row) blocks of 8.  Forward: bf16
    weights, the image as hi + lo bf16 planes (16 mantissa bits) -- against the oracle on (fp32 image, bf16-rounded weights) the fp32 accumulators' statistics agree to 1e-4
    and the stored bf16 z to its final rounding (2^-9 per element).  Weight gradient: image and dz as single bf16 values, channel-major bf16 dz tile -- products of bf16 values
    are exact in fp32, so on bf16-rounded inputs it agrees with the oracle to summation order (1e-4)."""
    N, C, H, W = case
    x = rng.random((N, C, H, W)).astype(np.float32)
    w = bf16_round((rng.standard_normal((64, C, 7, 7)) / np.sqrt(49 * C)).astype(np.float32))
    zo = oracle.conv2d_fwd(x, w, None, 2, 3)
    dz = bf16_round(rng.standard_normal(zo.shape).astype(np.float32))
    xr = bf16_round(x)
    _, dwo, _ = oracle.conv2d_bwd(xr, w, dz, 2, 3)
    md().set_op_storage(True)
    try:
        z, mean, invstd = md().stem_conv(dev(x), dev(w))
        z, mean, invstd = host(z), host(mean), host(invstd)
        dw = host(md().stem_conv_wgrad(dev(xr), dev(dz)))
    finally:
        md().set_op_storage(False)
    assert z.shape == zo.shape and rel(z, zo) < 4e-3
    m64 = zo.astype(np.float64).mean((0, 2, 3)); v64 = zo.astype(np.float64).var((0, 2, 3))
    assert np.abs(mean - m64).max() < 1e-4 * max(1.0, np.abs(m64).max()) and rel(invstd, 1.0 / np.sqrt(v64 + 1e-5)) < 1e-4          # (the statistics come from the fp32 accumulators)
    assert dw.shape == dwo.shape and rel(dw, dwo) < 1e-4


@pytest.mark.parametrize("case", [(2, 16, 20, 37, "elu"), (1, 32, 9, 21, "none"), (3, 64, 6, 11, "none"), (2, 128, 5, 7, "none"), (1, 16, 64, 208, "elu")])
def test_disparity_head_direct_kernels_vs_oracle(case):
    """head_fwd_kernel / head_bwd_kernel (reflect-pad 3x3 conv to one channel + sigmoid, src/depth_decoder.jl:43): all four channel counts,
    ragged tiles, the ELU-derivative variant of the last branch."""
    N, Cin, H, W, xact = case
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    if xact == "elu":
        x = np.where(x > 0, x, np.expm1(x)).astype(np.float32)          # an ELU output
    w = (rng.standard_normal((1, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32); b = rng.standard_normal(1).astype(np.float32)
    xp = oracle.pad_reflect(x)
    yo = 1.0 / (1.0 + np.exp(-oracle.conv2d_fwd(xp, w, b, 1, 0).astype(np.float64)))
    y = md().head_conv(dev(x), dev(w), dev(b))
    assert rel(host(y), yo) < 1e-5
    dy = rng.standard_normal(yo.shape).astype(np.float32)
    dzo = (dy * yo * (1 - yo)).astype(np.float32)
    dxp, dwo, dbo = oracle.conv2d_bwd(xp, w, dzo, 1, 0)
    dxo = _reflect_adjoint(dxp, N, Cin, H, W)
    if xact == "elu":
        dxo = dxo * np.where(x > 0, 1.0, x + 1.0)
    dx, dw, db = md().head_conv_pullback(dev(x), dev(w), y, dev(dy), xact)
    assert rel(host(dx), dxo) < 1e-4 and rel(host(dw), dwo) < 1e-4 and rel(host(db), dbo) < 1e-4


@pytest.mark.parametrize("case", [(40, 64, 32, 52, 64, 3, 1, 1), (6, 64, 16, 26, 128, 3, 2, 1), (6, 64, 16, 26, 128, 1, 2, 0), (3, 128, 8, 13, 256, 3, 1, 1), (2, 256, 4, 13, 512, 3, 1, 1),
                                  (2, 32, 9, 11, 16, 3, 1, 1)])
def test_conv_bn_statistics_epilogue_vs_oracle(case):
    """The BatchNorm batch statistics accumulated in the conv epilogue (128x64 and 64x64 tiles: butterfly + 8-slot double atomics; the
    last case has 16 output channels and takes the separate statistics pass)."""
    N, Cin, H, W, Cout, k, s, p = case
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    yo = oracle.conv2d_fwd(x, w, None, s, p)
    y, mean, invstd = md().conv2d_bnstats(dev(x), dev(w), s, p)
    assert rel(host(y), yo) < 1e-4
    m64 = yo.astype(np.float64).mean((0, 2, 3)); v64 = yo.astype(np.float64).var((0, 2, 3))
    assert np.abs(host(mean) - m64).max() < 1e-5 * max(1.0, np.abs(yo).max()) and rel(host(invstd), 1.0 / np.sqrt(v64 + 1e-5)) < 1e-4


@pytest.mark.parametrize("case", [(40, 64, 32, 52, 64, 3, 1, 1), (48, 64, 20, 37, 128, 3, 1, 1), (6, 64, 16, 26, 128, 3, 2, 1), (3, 128, 8, 13, 256, 3, 1, 1)])
def test_conv_bn_statistics_epilogue_on_bf16_tensors(case):
    """The same statistics out of the bf16-tensor implicit GEMM (32-channel K-steps; ragged pixel tiles, a stride-2 layer, a split-K candidate).  They come from the fp32
    accumulators, so against the oracle on the bf16-rounded operands they agree to summation order; the stored bf16 output carries its final rounding."""
    N, Cin, H, W, Cout, k, s, p = case
    x = bf16_round(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = bf16_round((rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
    yo = oracle.conv2d_fwd(x, w, None, s, p)
    md().set_op_storage(True); md().set_winograd(False)
    try:
        y, mean, invstd = md().conv2d_bnstats(dev(x), dev(w), s, p)
        y, mean, invstd = host(y), host(mean), host(invstd)
    finally:
        md().set_op_storage(False); md().set_winograd(True, -2)
    assert rel(y, yo) < 4e-3
    m64 = yo.astype(np.float64).mean((0, 2, 3)); v64 = yo.astype(np.float64).var((0, 2, 3))
    assert np.abs(mean - m64).max() < 1e-5 * max(1.0, np.abs(yo).max()) and rel(invstd, 1.0 / np.sqrt(v64 + 1e-5)) < 1e-4


# ---- fused Winograd F(2x2,3x3) kernel (conv_winograd.hip): every 3x3 / stride-1 layer with Cout % 32 == 0, forward and input gradient ---------------------
WINO_CASES = [
    # N, Cin, H, W, Cout, pad, pad_mode, act
    (2, 64, 16, 26, 64, 1, "zeros", "none"),        # encoder layer 1 shape family
    (3, 128, 8, 13, 128, 1, "zeros", "none"),       # odd width: half-empty last tile column
    (2, 256, 4, 13, 512, 1, "zeros", "none"),       # 1/32-scale map: 7x2 tiles per image, blocks span images
    (5, 512, 4, 13, 256, 0, "reflect", "elu"),      # decoder branch 1 conv 1 (reflect pad, bias + ELU)
    (2, 96, 12, 20, 32, 0, "reflect", "elu"),       # Cout 32: one 32-channel sub-tile per block; Cin 96 (6 chunks)
    (1, 32, 7, 9, 96, 1, "zeros", "relu"),          # Cout 96 = 3 sub-tile blocks; odd height AND width, a single partly filled block
    (3, 16, 5, 33, 64, 1, "zeros", "none"),         # one channel chunk (no steady-state loop iteration after the prologue)
    (1, 64, 2, 2, 64, 0, "reflect", "none"),        # the smallest reflect-padded map: every patch row / column is a mirror
    (7, 128, 16, 13, 64, 0, "reflect", "sigmoid"),
    (2, 8, 6, 140, 64, 1, "zeros", "none"),         # tile rows wider than a block (70 tiles): blocks start mid-row; one 8-channel chunk
    (1, 24, 40, 6, 128, 1, "zeros", "relu"),        # 3 tiles per row: a block spans 22 row segments
]


# (algorithm pin, pieces per operand of the forward / input-gradient products, pieces of the weight-gradient products): md2_set_winograd / md2_set_conv_products
WINO_ROUTES = {"tile32x64": (2, 0, 0), "tile64x32": (3, 0, 3), "split_bf16": (3, 2, 2), "split3_bf16": (3, 3, 3), "split3_bf16_persistent": (19, 3, 3), "split3_bf16_64_channel_workgroups": (11, 3, 3), "split_bf16_64_channel_workgroups": (11, 2, 2)}


def _set_route(route):
    md().set_winograd(route[0], 1); md().set_conv_products(route[1], route[2])


@pytest.fixture(params=list(WINO_ROUTES), ids=list(WINO_ROUTES))
def forced_winograd(request):
    dflt = md().get_conv_products()
    _set_route(WINO_ROUTES[request.param])       # every eligible launch takes the Winograd kernel, however small its grid: the 256-thread / 32-tile one, or the
    yield WINO_ROUTES[request.param]             # 64-tile x 32-channel one with row-segment patches (the one the training step uses), wherever it applies; products by
    md().set_winograd(True, -2); md().set_conv_products(*dflt)          # fp32 MFMA, the exact three-piece split, or the reduced two-piece one


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv_fwd_and_input_gradient_vs_oracle(case, forced_winograd):
    """conv_wino_kernel against the oracle's direct convolution (1e-4 like every other layer kernel), and bit-for-bit agreement of the layer's OTHER outputs with the
    direct implicit GEMM is NOT expected (different summation order) -- so the direct path is checked against the same oracle numbers in the same test."""
    N, Cin, H, W, Cout, p, mode, act = case
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    xin = oracle.pad_reflect(x) if mode == "reflect" else x
    zo = oracle.conv2d_fwd(xin, w, b, 1, 0 if mode == "reflect" else p)
    ref = {"none": zo, "relu": np.maximum(zo, 0), "elu": np.where(zo > 0, zo, np.expm1(zo)), "sigmoid": 1 / (1 + np.exp(-zo))}[act]
    y = host(md().conv2d(dev(x), dev(w), dev(b), 1, p, mode, act))
    assert y.shape == ref.shape and rel(y, ref) < 1e-4
    md().set_winograd(False)
    y_direct = host(md().conv2d(dev(x), dev(w), dev(b), 1, p, mode, act))
    _set_route(forced_winograd)
    # the two algorithms really are different code paths (the split-bf16 kernel declines maps with one tile per row and channel counts that are not multiples of 16)
    split_declines = forced_winograd[1] != 0 and (Cin % 16 != 0 or (W + 2 * (p if mode != "reflect" else 1) - 2 + 1) // 2 < 2)
    assert rel(y_direct, ref) < 1e-4 and (split_declines or not np.array_equal(y, y_direct))
    dy = rng.standard_normal(zo.shape).astype(np.float32)
    dxo, dwo, dbo = oracle.conv2d_bwd(xin, w, dy, 1, 0 if mode == "reflect" else p)
    if mode == "reflect":
        dxo = _reflect_adjoint(dxo, N, Cin, H, W)
    dx, dw, db = md().conv2d_pullback(dev(x), dev(w), dev(dy), 1, p, mode)          # the input gradient is the same kernel with the channel roles swapped (Cin % 32 == 0 cases)
    assert rel(host(dx), dxo) < 1e-4 and rel(host(dw), dwo) < 1e-4 and rel(host(db), dbo) < 1e-4


@pytest.mark.parametrize("case", WINO_CASES + [(40, 64, 32, 52, 64, 1, "zeros", "none"), (6, 128, 32, 104, 128, 1, "zeros", "none")])
def test_winograd_on_bf16_tensors_vs_oracle(case):
    """conv_wino3b_kernel (round 6): Winograd F(2x2,3x3) on bf16 tensors -- bf16 patches widened exactly, fp32 input transform, V rounded to bf16 ONCE, U = G g G^T as one
    bf16 piece from the fp32 weights, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 output transform, one rounding on the store -- forward and input gradient against
    the oracle on the bf16-rounded operands, beside the direct kernel on the same tensors.  The direct kernel multiplies the stored bf16 values exactly (4e-3: the output's own
    rounding, measured 2.0e-3 .. 2.8e-3); the Winograd form adds the roundings of V and U (2^-9 relative each, in the transform domain): measured 3.1e-3 .. 5.2e-3 forward,
    2.2e-3 .. 5.4e-3 input gradient; bounds 7e-3 / 8e-3 of the largest output.
    The BatchNorm statistics of its epilogue come from the fp32 accumulators and are checked on the two large cases."""
    N, Cin, H, W, Cout, p, mode, act = case
    x = bf16_round(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = bf16_round((rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    xin = oracle.pad_reflect(x) if mode == "reflect" else x
    zo = oracle.conv2d_fwd(xin, w, b, 1, 0 if mode == "reflect" else p)
    ref = {"none": zo, "relu": np.maximum(zo, 0), "elu": np.where(zo > 0, zo, np.expm1(zo)), "sigmoid": 1 / (1 + np.exp(-zo))}[act]
    dy = bf16_round(rng.standard_normal(zo.shape).astype(np.float32))
    dxo = oracle.conv2d_bwd(xin, w, dy, 1, 0 if mode == "reflect" else p)[0]
    if mode == "reflect":
        dxo = _reflect_adjoint(dxo, N, Cin, H, W)
    md().set_op_storage(True)
    try:
        md().set_winograd(35, 1)          # 3 + 32: the Winograd kernel pinned, bf16 tensors included (opt-in: md2_set_winograd(on + 32, .))
        y = host(md().conv2d(dev(x), dev(w), dev(b), 1, p, mode, act)); dx = host(md().conv2d_pullback(dev(x), dev(w), dev(dy), 1, p, mode)[0])
        st = md().conv2d_bnstats(dev(x), dev(w), 1, p) if (mode == "zeros" and N >= 6) else None
        md().set_winograd(False)
        y_direct = host(md().conv2d(dev(x), dev(w), dev(b), 1, p, mode, act)); dx_direct = host(md().conv2d_pullback(dev(x), dev(w), dev(dy), 1, p, mode)[0])
    finally:
        md().set_op_storage(False); md().set_winograd(True, -2)
    declines = (W + 2 * (p if mode != "reflect" else 1) - 2 + 1) // 2 < 2          # one tile per row: the 32-tile kernel does not stage such maps
    print("bf16 winograd %s: y rel %.2e (direct %.2e) dx rel %.2e (direct %.2e)%s" % (case, rel(y, ref), rel(y_direct, ref), rel(dx, dxo), rel(dx_direct, dxo), "  [declined]" if declines else ""))
    assert rel(y_direct, ref) < 4e-3 and rel(dx_direct, dxo) < 8e-3
    assert rel(y, ref) < 7e-3 and rel(dx, dxo) < 8e-3
    assert declines or not np.array_equal(y, y_direct)                                # the two algorithms really are different code paths
    if st is not None:
        yo = oracle.conv2d_fwd(x, w, None, 1, p)
        m64 = yo.astype(np.float64).mean((0, 2, 3)); v64 = yo.astype(np.float64).var((0, 2, 3))
        assert rel(host(st[0]), yo) < 7e-3
        assert np.abs(host(st[1]) - m64).max() < 2e-4 * max(1.0, np.abs(yo).max()) and rel(host(st[2]), 1.0 / np.sqrt(v64 + 1e-5)) < 2e-3


PERSISTENT_CASES = [
    # N, Cin, H, W, Cout, pad, pad_mode, act       -- every one has more (32-tile, 32-channel) items than the 512 resident workgroups, so some workgroups walk two items
    (2, 64, 64, 128, 160, 1, "zeros", "elu"),       # 640 items, four chunks: the loop is nothing but the two tail chunks that carry the next item's head
    (4, 128, 64, 104, 128, 0, "reflect", "none"),   # 832 items forward AND input gradient, eight chunks, reflect padding
    (11, 96, 30, 70, 96, 1, "zeros", "relu"),       # 543 items: 31 workgroups take a second item; ragged last tile group (5775 tiles), six chunks, tile rows wider than a group
]


@pytest.mark.parametrize("pieces", [3, 2])
@pytest.mark.parametrize("case", PERSISTENT_CASES)
def test_persistent_winograd_kernel_equals_the_one_item_kernel_bit_for_bit(case, pieces):
    """conv_wino3p_kernel (round 6, md2_set_winograd(on + 16, .): the resident set of workgroups walks the launch's items, the head of the next item rides in the last two
    chunks of the current one) against the oracle's direct convolution, and BIT FOR BIT against conv_wino3s_kernel: same loader, same arithmetic order, another schedule.
    (Measured equal to the one-item workgroups launch by launch, so it is not the default route; this test keeps it correct.)"""
    N, Cin, H, W, Cout, p, mode, act = case
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy_shape = (N, Cout, H, W)
    dy = rng.standard_normal(dy_shape).astype(np.float32)
    dflt = md().get_conv_products()
    try:
        md().set_winograd(19, 1); md().set_conv_products(pieces, pieces)
        y = host(md().conv2d(dev(x), dev(w), dev(b), 1, p, mode, act)); dx = host(md().conv2d_pullback(dev(x), dev(w), dev(dy), 1, p, mode)[0])
        md().set_winograd(3, 1)
        y1 = host(md().conv2d(dev(x), dev(w), dev(b), 1, p, mode, act)); dx1 = host(md().conv2d_pullback(dev(x), dev(w), dev(dy), 1, p, mode)[0])
    finally:
        md().set_winograd(True, -2); md().set_conv_products(*dflt)
    assert np.array_equal(y, y1) and np.array_equal(dx, dx1)
    xin = oracle.pad_reflect(x) if mode == "reflect" else x
    zo = oracle.conv2d_fwd(xin, w, b, 1, 0 if mode == "reflect" else p)
    ref = {"none": zo, "relu": np.maximum(zo, 0), "elu": np.where(zo > 0, zo, np.expm1(zo))}[act]
    dxo = oracle.conv2d_bwd(xin, w, dy, 1, 0 if mode == "reflect" else p)[0]
    if mode == "reflect":
        dxo = _reflect_adjoint(dxo, N, Cin, H, W)
    tol = 1e-4 if pieces == 3 else 2e-3                             # (two pieces: the reduced-precision opt-in form)
    assert rel(y, ref) < tol and rel(dx, dxo) < tol


@pytest.mark.parametrize("case", [(40, 64, 32, 52, 64), (3, 128, 8, 13, 256), (2, 256, 4, 13, 512), (4, 64, 7, 9, 32), (6, 128, 32, 104, 128)])
def test_winograd_bn_statistics_epilogue_vs_oracle(case, forced_winograd):
    """BatchNorm batch statistics out of the Winograd epilogue (column sums through LDS + 8-slot double atomics), incl. partly filled blocks and odd maps."""
    N, Cin, H, W, Cout = case
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    yo = oracle.conv2d_fwd(x, w, None, 1, 1)
    y, mean, invstd = md().conv2d_bnstats(dev(x), dev(w), 1, 1)
    assert rel(host(y), yo) < 1e-4
    m64 = yo.astype(np.float64).mean((0, 2, 3)); v64 = yo.astype(np.float64).var((0, 2, 3))
    assert np.abs(host(mean) - m64).max() < 1e-5 * max(1.0, np.abs(yo).max()) and rel(host(invstd), 1.0 / np.sqrt(v64 + 1e-5)) < 1e-4
