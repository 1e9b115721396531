// Standalone timing harness for the implicit-GEMM conv kernel (ablations + forced tiles). Built with -DMD2_ABLATE.
// usage: conv_bench N H W Cin Cout k [tile] [ablate]
#include "../monodepth2.jl_amd/csrc/md2_internal.h"
#include <vector>
#include <cstdlib>
#include <algorithm>
using namespace md2;
int main(int argc, char** argv) {
    int N = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]), k = atoi(argv[6]);
    int tile = argc > 7 ? atoi(argv[7]) : 0, abl = argc > 8 ? atoi(argv[8]) : 0;
    size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout, nw = (size_t)Cout * k * k * Cin;
    float *x, *y, *w; hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, nw * 4);
    std::vector<float> hx(nx), hw(nw);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    ConvArgs a{};
    a.x = x; a.x_bs = (long)H * W * Cin; a.x_ps = Cin; a.H = H; a.W = W; a.Cin = Cin; a.w = w;
    const int st = getenv("STRIDE") ? atoi(getenv("STRIDE")) : 1, Ho = (H + 2 * (k / 2) - k) / st + 1, Wo = (W + 2 * (k / 2) - k) / st + 1;          // STRIDE=2: the encoder's down-sampling convolutions
    a.y = y; a.y_bs = (long)Ho * Wo * Cout; a.y_ps = Cout; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.N = N; a.KH = k; a.KW = k; a.stride = st; a.pad = k / 2; a.dil = 1; a.padmode = 0;
    a.M = N * Ho * Wo; a.oy0 = 0; a.ox0 = 0; a.ostep = 1; a.Hl = Ho; a.Wl = Wo; a.kh0 = 0; a.khs = 1; a.nkh = k; a.kw0 = 0; a.kws = 1; a.nkw = k;
    if (getenv("SPLIT3")) a.split3 = 1;       // exact three-piece products in the direct implicit GEMM (conv_igemm_kernel<..., BF = 3, ...>)
#ifdef MD2_ABLATE
    a.ablate = abl; a.force_tile = tile;
#endif
    a.accumulate_c00 = -1;
    if (getenv("BF16S")) a.bf16 = 2;          // bf16 tensors (the kernels of md2_config.precision = 2): the buffers are read as bf16, their values do not matter for timing
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (getenv("WGRAD")) {
        float* dw; hipMalloc(&dw, nw * 4); hipMemset(dw, 0, nw * 4);
        hipMemcpy(y, hx.data(), std::min(nx, ny) * 4, hipMemcpyHostToDevice);
        WgradArgs g{};
        g.x = x; g.x_bs = a.x_bs; g.x_ps = Cin; g.H = H; g.W = W; g.Cin = Cin; g.dy = y; g.dy_bs = a.y_bs; g.dy_ps = Cout; g.Ho = H; g.Wo = W; g.Cout = Cout;
        g.dw = dw; g.N = N; g.KH = k; g.KW = k; g.stride = 1; g.pad = k / 2; g.padmode = 0; g.M = a.M;
#ifdef MD2_ABLATE
        g.ablate = abl;
#endif

        for (int i = 0; i < 3; ++i) launch_conv_wgrad(g, 0);
        hipEventRecord(e0); const int reps = 20; for (int i = 0; i < reps; ++i) launch_conv_wgrad(g, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("WGRAD N=%d HxW=%dx%d Cin=%d Cout=%d k=%d abl=%d : %.3f ms  %.1f TFLOP/s\n", N, H, W, Cin, Cout, k, abl, ms, 2.0 * a.M * Cout * k * k * Cin / (ms * 1e-3) / 1e12);
        return 0;
    }
    if (getenv("WINO")) {            // fused Winograd F(2x2,3x3): U = G g G^T once, then the same launch entry (WINO=2: also time the weight transform)
        float* U; hipMalloc(&U, conv_wino_u_floats(Cout, Cin) * 4);
        launch_wino_weights(w, U, Cout, Cin, g_wino_pieces, 0); a.wino_u = U; a.wino_pieces = wino_resolve_pieces(g_wino_pieces, Cin); g_wino_min_blocks = 1; if (getenv("WMODE")) g_wino_on = atoi(getenv("WMODE")); if (getenv("WIDE")) g_wino_wide = 1; if (getenv("CB1")) g_wino_cb2 = 0;          // WMODE=2 / 3: pin the 32-tile / 64-tile kernel
        if (getenv("XF")) { float* sc; hipMalloc(&sc, Cin * 8); std::vector<float> h2(2 * Cin, 0.5f); hipMemcpy(sc, h2.data(), Cin * 8, hipMemcpyHostToDevice); a.in_scale = sc; a.in_shift = sc + Cin; }
        if (atoi(getenv("WINO")) == 2) {
            hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch_wino_weights(w, U, Cout, Cin, g_wino_pieces, 0); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); printf("wino weight transform %dx%d: %.1f us\n", Cout, Cin, ms / 20 * 1e3);
        }
    }
    for (int i = 0; i < 3; ++i) launch_conv_igemm(a, 0);
    hipEventRecord(e0); const int reps = 20; for (int i = 0; i < reps; ++i) launch_conv_igemm(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("N=%d HxW=%dx%d Cin=%d Cout=%d k=%d tile=%d abl=%d : %.3f ms  %.1f TFLOP/s\n", N, H, W, Cin, Cout, k, tile, abl, ms, 2.0 * a.M * Cout * k * k * Cin / (ms * 1e-3) / 1e12);
    return 0;
}
