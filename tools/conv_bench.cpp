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
        g.dw = dw; g.N = N; g.KH = k; g.KW = k; g.stride = 1; g.pad = k / 2; g.padmode = 0; g.M = a.M; if (getenv("BF16S")) g.bf16 = 2;
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
        launch_wino_weights(w, U, Cout, Cin, g_wino_pieces, 0); a.wino_u = U; a.wino_pieces = wino_resolve_pieces(g_wino_pieces, Cin); g_wino_min_blocks = 1; if (getenv("WMODE")) g_wino_on = atoi(getenv("WMODE")); if (getenv("WIDE")) g_wino_wide = 1; if (getenv("CB1")) g_wino_cb2 = 0; if (getenv("PERSIST")) g_wino_persist = atoi(getenv("PERSIST"));          // WMODE=2 / 3: pin the 32-tile / 64-tile kernel
        if (getenv("XF")) { float* sc; hipMalloc(&sc, Cin * 8); std::vector<float> h2(2 * Cin, 0.5f); hipMemcpy(sc, h2.data(), Cin * 8, hipMemcpyHostToDevice); a.in_scale = sc; a.in_shift = sc + Cin; }
        if (atoi(getenv("WINO")) == 2) {
            hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch_wino_weights(w, U, Cout, Cin, g_wino_pieces, 0); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); printf("wino weight transform %dx%d: %.1f us\n", Cout, Cin, ms / 20 * 1e3);
        }
    }
#ifdef MD2_ABLATE
    if (getenv("STAMPS") && a.wino_pieces) {          // per-wave s_memtime stamps of one launch of conv_wino3s_kernel: where a workgroup's life goes, and what runs beside it on its CU
        for (int i = 0; i < 3; ++i) launch_conv_igemm(a, 0);
        const int tiles = N * ((H + 1) / 2) * ((W + 1) / 2), nwg = ((tiles + 31) / 32) * (Cout / 32);
        unsigned long long* dbg; hipMalloc(&dbg, (size_t)nwg * 4 * 8 * 8); hipMemset(dbg, 0, (size_t)nwg * 4 * 8 * 8);
        a.dbg = dbg; launch_conv_igemm(a, 0); hipDeviceSynchronize(); a.dbg = nullptr;
        std::vector<unsigned long long> h((size_t)nwg * 32); hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0; for (int g = 0; g < nwg; ++g) { t0 = std::min(t0, h[(size_t)g * 32]); t1 = std::max(t1, h[(size_t)g * 32 + 5]); }
        double ph[5] = {0, 0, 0, 0, 0}, life = 0;          // wave 0 of every workgroup: entry -> table barrier -> loop entry -> loop exit -> transform barrier -> end
        for (int g = 0; g < nwg; ++g) { const unsigned long long* w = &h[(size_t)g * 32]; for (int k = 0; k < 5; ++k) ph[k] += (double)(w[k + 1] - w[k]); life += (double)(w[5] - w[0]); }
        printf("stamps: %d workgroups, launch span %llu ticks; per workgroup (wave 0): geometry %.0f, first loads + commit %.0f, chunk loop %.0f (%d chunks: %.0f each), output transform %.0f, epilogue %.0f, life %.0f ticks\n",
               nwg, t1 - t0, ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, Cin / 16, ph[2] / nwg / (Cin / 16), ph[3] / nwg, ph[4] / nwg, life / nwg);
        // per CU (xcc, se, sh, cu from the hardware id): workgroups in order of entry, the gap between one's end and the next's entry on the same SIMD-0 slot
        struct Ev { unsigned long long a, b; int slot; };
        std::vector<std::vector<Ev>> cu(16 * 8 * 2 * 16);
        for (int g = 0; g < nwg; ++g) {
            const unsigned long long id = h[(size_t)g * 32 + 7]; const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 15;
            const int cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, wave = hw & 15;
            cu[((xcc * 8 + se) * 2 + sh) * 16 + cuid].push_back({h[(size_t)g * 32], h[(size_t)g * 32 + 5], wave});
        }
        double busy2 = 0, busy1 = 0, busy0 = 0; int ncu = 0;
        for (auto& v : cu) if (!v.empty()) {
            ++ncu; std::vector<std::pair<unsigned long long, int>> ev; for (auto& e : v) { ev.push_back({e.a, +1}); ev.push_back({e.b, -1}); }
            std::sort(ev.begin(), ev.end()); int lvl = 0; unsigned long long last = t0;
            for (auto& e : ev) { const double d = (double)(e.first - last); if (lvl >= 2) busy2 += d; else if (lvl == 1) busy1 += d; else busy0 += d; lvl += e.second; last = e.first; }
            busy0 += (double)(t1 - last);
        }
        printf("stamps: %d CUs seen; share of the launch span with >= 2 / 1 / 0 workgroups resident per group: %.3f / %.3f / %.3f\n", ncu, busy2 / ((double)(t1 - t0) * ncu), busy1 / ((double)(t1 - t0) * ncu), busy0 / ((double)(t1 - t0) * ncu));
        return 0;
    }
#endif
    for (int i = 0; i < 3; ++i) launch_conv_igemm(a, 0);
    hipEventRecord(e0); const int reps = 20; for (int i = 0; i < reps; ++i) launch_conv_igemm(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("N=%d HxW=%dx%d Cin=%d Cout=%d k=%d tile=%d abl=%d : %.3f ms  %.1f TFLOP/s\n", N, H, W, Cin, Cout, k, tile, abl, ms, 2.0 * a.M * Cout * k * k * Cin / (ms * 1e-3) / 1e12);
    return 0;
}
