// md2_oracle.cpp — C entry points of the CPU oracle (ctypes-loadable).  TEST INFRASTRUCTURE ONLY.
// See md2_oracle.h for the parity-pinning statement and md2_oracle_ops.hpp / md2_oracle_net.hpp for
// the per-function reference citations.
#include "md2_oracle.h"
#include "md2_oracle_net.hpp"
#include <omp.h>

using namespace md2o;

#define BOTH(name, ...)                                                                  \
    O_API void name##_f32(__VA_ARGS__);                                                  \
    O_API void name##_f64(__VA_ARGS__);

// ---- L2 ops, float and double (the reference's unit tests run in Float64) ----------------------
#define DEF_OPS(T, SFX)                                                                                                                  \
    O_API void o_hat_##SFX(const T* r, T* S, int N) { hat<T>(r, S, N); }                                                                 \
    O_API void o_hat_pullback_##SFX(const T* d, T* dv, int N) { hat_pullback<T>(d, dv, N); }                                             \
    O_API void o_so3_exp_map_##SFX(const T* r, T* R, int N) { so3_exp_map<T>(r, R, N); }                                                 \
    O_API void o_so3_exp_map_bwd_##SFX(const T* r, const T* dR, T* dr, int N) { so3_exp_map_bwd<T>(r, dR, dr, N); }                      \
    O_API void o_composeT_##SFX(const T* r, const T* t, int inv, T* R, T* to, int N) { composeT<T>(r, t, inv, R, to, N); }               \
    O_API void o_composeT_bwd_##SFX(const T* r, const T* t, int inv, const T* dR, const T* dto, T* dr, T* dt, int N) {                   \
        composeT_bwd<T>(r, t, inv, dR, dto, dr, dt, N); }                                                                                \
    O_API void o_ssim_##SFX(const T* x, const T* y, T* out, int planes, int H, int W) {                                                  \
        ssim_fwd_bwd<T>(x, y, out, nullptr, nullptr, nullptr, planes, H, W); }                                                           \
    O_API void o_ssim_bwd_##SFX(const T* x, const T* y, const T* dout, T* dx, T* dy, int planes, int H, int W) {                         \
        ssim_fwd_bwd<T>(x, y, nullptr, dout, dx, dy, planes, H, W); }                                                                    \
    O_API void o_photometric_##SFX(const T* pred, const T* tgt, T* out, int N, int C, int H, int W) {                                    \
        photometric_fwd_bwd<T>(pred, tgt, out, nullptr, nullptr, N, C, H, W); }                                                          \
    O_API void o_photometric_bwd_##SFX(const T* pred, const T* tgt, const T* dout, T* dpred, int N, int C, int H, int W) {               \
        photometric_fwd_bwd<T>(pred, tgt, nullptr, dout, dpred, N, C, H, W); }                                                           \
    O_API void o_min_stack_##SFX(const T* stack, int k, long n, T* out, int32_t* arg) { min_stack<T>(stack, k, (size_t)n, out, arg); }   \
    O_API double o_smooth_loss_##SFX(const T* d, const T* img, int N, int C, int H, int W) {                                             \
        return (double)smooth_loss_fwd_bwd<T>(d, img, T(1), nullptr, N, C, H, W); }                                                      \
    O_API double o_smooth_loss_bwd_##SFX(const T* d, const T* img, double dl, T* dd, int N, int C, int H, int W) {                       \
        return (double)smooth_loss_fwd_bwd<T>(d, img, T(dl), dd, N, C, H, W); }                                                          \
    O_API void o_disparity_to_depth_##SFX(const T* d, T* out, long n, double mn, double mx) { disparity_to_depth<T>(d, out, n, mn, mx); }\
    O_API void o_disparity_to_depth_bwd_##SFX(const T* d, const T* dz, T* dd, long n, double mn, double mx) {                            \
        disparity_to_depth_bwd<T>(d, dz, dd, n, mn, mx); }                                                                               \
    O_API void o_backproject_##SFX(const T* depth, const T* invK, T* out, int N, int H, int W) { backproject<T>(depth, invK, out, N, H, W); } \
    O_API void o_backproject_bwd_##SFX(const T* dout, const T* invK, T* dz, int N, int H, int W) { backproject_bwd<T>(dout, invK, dz, N, H, W); } \
    O_API void o_project_##SFX(const T* pts, const T* K, const T* R, const T* t, T* uv, int N, int H, int W) {                           \
        project<T>(pts, K, R, t, uv, N, H, W); }                                                                                         \
    O_API void o_project_bwd_##SFX(const T* pts, const T* K, const T* R, const T* t, const T* duv, T* dpts, T* dR, T* dt, int N, int H, int W) { \
        project_bwd<T>(pts, K, R, t, duv, dpts, dR, dt, N, H, W); }                                                                      \
    O_API void o_grid_sample_##SFX(const T* in, const T* grid, T* out, int N, int C, int H, int W, int Ho, int Wo, int border) {         \
        grid_sample_fwd_bwd<T>(in, grid, out, nullptr, nullptr, nullptr, N, C, H, W, Ho, Wo, border); }                                  \
    O_API void o_grid_sample_bwd_##SFX(const T* in, const T* grid, const T* dout, T* dgrid, T* din, int N, int C, int H, int W, int Ho,  \
                                       int Wo, int border) {                                                                             \
        grid_sample_fwd_bwd<T>(in, grid, nullptr, dout, dgrid, din, N, C, H, W, Ho, Wo, border); }                                       \
    O_API void o_upsample_bilinear_##SFX(const T* x, T* out, int planes, int H, int W, int Ho, int Wo) {                                 \
        upsample_bilinear<T>(x, out, planes, H, W, Ho, Wo); }                                                                            \
    O_API void o_upsample_bilinear_bwd_##SFX(const T* dout, T* dx, int planes, int H, int W, int Ho, int Wo) {                           \
        upsample_bilinear_bwd<T>(dout, dx, planes, H, W, Ho, Wo); }                                                                      \
    O_API void o_pad_reflect_##SFX(const T* x, T* out, int planes, int H, int W) { pad_reflect1<T>(x, out, planes, H, W); }

DEF_OPS(float, f32)
DEF_OPS(double, f64)

// ---- layer ops (fp32) --------------------------------------------------------------------------
static Tensor wrap(const float* p, int n, int c, int h, int w) { Tensor t(n, c, h, w); std::memcpy(t.p(), p, sizeof(float) * t.size()); return t; }

O_API void o_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W, int Cout, int k, int s, int p) {
    Tensor X = wrap(x, N, Cin, H, W), Y; conv2d_fwd(X, w, b, Y, Cout, k, s, p); std::memcpy(y, Y.p(), sizeof(float) * Y.size());
}
// dx/dw/db may be null; dw/db are OVERWRITTEN (zeroed first)
O_API void o_conv2d_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int N, int Cin, int H, int W, int Cout, int k, int s, int p) {
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    Tensor X = wrap(x, N, Cin, H, W), DY = wrap(dy, N, Cout, Ho, Wo), DX;
    if (dw) std::memset(dw, 0, sizeof(float) * (size_t)Cout * Cin * k * k);
    if (db) std::memset(db, 0, sizeof(float) * Cout);
    conv2d_bwd(X, w, DY, dx ? &DX : nullptr, dw, db, k, s, p);
    if (dx) std::memcpy(dx, DX.p(), sizeof(float) * DX.size());
}
O_API void o_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var, int N, int C, int H, int W) {
    Tensor X = wrap(x, N, C, H, W), Y; BNSave sv; bn_train_fwd(X, gamma, beta, Y, sv); std::memcpy(y, Y.p(), sizeof(float) * Y.size());
    if (mean) std::memcpy(mean, sv.mean.data(), sizeof(float) * C);
    if (var) std::memcpy(var, sv.var.data(), sizeof(float) * C);
}
O_API void o_bn_train_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int N, int C, int H, int W) {
    Tensor X = wrap(x, N, C, H, W), Y, DY = wrap(dy, N, C, H, W), DX; BNSave sv; Vec<float> beta(C, 0.f);
    bn_train_fwd(X, gamma, beta.data(), Y, sv);
    std::memset(dgamma, 0, sizeof(float) * C); std::memset(dbeta, 0, sizeof(float) * C);
    bn_train_bwd(DY, gamma, sv, DX, dgamma, dbeta); std::memcpy(dx, DX.p(), sizeof(float) * DX.size());
}
O_API void o_maxpool3s2_fwd(const float* x, float* y, int N, int C, int H, int W) {
    Tensor X = wrap(x, N, C, H, W), Y; Vec<int32_t> arg; maxpool3s2_fwd(X, Y, arg); std::memcpy(y, Y.p(), sizeof(float) * Y.size());
}
O_API void o_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int N, int C, int H, int W) {
    Tensor X = wrap(x, N, C, H, W), Y, DX; Vec<int32_t> arg; maxpool3s2_fwd(X, Y, arg);
    Tensor DY = wrap(dy, N, C, Y.h, Y.w); maxpool3s2_bwd(DY, arg, DX, H, W); std::memcpy(dx, DX.p(), sizeof(float) * DX.size());
}
O_API void o_act(float* y, long n, int act) { Tensor t(1, 1, 1, (int)n); std::memcpy(t.p(), y, sizeof(float) * n); act_fwd(t, act); std::memcpy(y, t.p(), sizeof(float) * n); }

// ---- network handle ---------------------------------------------------------------------------
struct o_cfg {
    int N, C, H, W, automasking, simple;
    double min_depth, max_depth, smoothness;
    float K[9], invK[9];
    int target_id, source_ids[2];          // TrainCache.target_id / source_ids (1-based, src/Monodepth.jl:44-55); 0 = the defaults 2 / {1, 3}
    int scale_levels[5]; double scales[5]; // DepthDecoder scale_levels (0-terminated; all 0 = 2:5) and TrainCache.scales (all 0 = 1 / 2^(5 - level))
};
static LossCfg to_cfg(const o_cfg* c) {
    LossCfg l; l.N = c->N; l.C = c->C; l.H = c->H; l.W = c->W; l.automasking = c->automasking; l.simple = c->simple; if (c->simple) l.nscales = 1;
    if (c->target_id) { l.target_id = c->target_id; l.source_ids[0] = c->source_ids[0]; l.source_ids[1] = c->source_ids[1]; }
    if (c->scale_levels[0] && !c->simple) { l.nscales = 0; for (int i = 0; i < 5 && c->scale_levels[i]; ++i) { l.levels[i] = c->scale_levels[i]; l.nscales = i + 1; } }
    bool scales_given = false; for (int i = 0; i < 5; ++i) scales_given |= c->scales[i] != 0.0;          // all zero = not given; otherwise as given (an explicit 0 is a weight of 0, negative = default)
    if (!c->simple) for (int i = 0; i < l.nscales; ++i) l.scales[i] = (scales_given && c->scales[i] >= 0.0) ? c->scales[i] : 1.0 / (double)(1 << (5 - l.levels[i]));
    l.min_depth = c->min_depth; l.max_depth = c->max_depth; l.smoothness = c->smoothness;
    std::memcpy(l.K, c->K, sizeof(l.K)); std::memcpy(l.invK, c->invK, sizeof(l.invK)); return l;
}
struct o_net { Net net; StepOut out; explicit o_net(int C, int depth = 18, const int* levels = nullptr) : net(C, depth, levels) {} };

O_API o_net* o_net_create(int C) { return new o_net(C); }
O_API o_net* o_net_create_depth(int C, int depth) { return (depth == 18 || depth == 50) ? new o_net(C, depth) : nullptr; }
// DepthDecoder(; scale_levels): ascending levels in 1..5, 0-terminated
O_API o_net* o_net_create_levels(int C, int depth, const int* levels) { return (depth == 18 || depth == 50) ? new o_net(C, depth, levels) : nullptr; }
O_API void o_net_destroy(o_net* h) { delete h; }
O_API long o_net_nparams(o_net* h) { return (long)h->net.np; }
O_API void o_net_params_set(o_net* h, const float* p) { std::memcpy(h->net.th.data(), p, sizeof(float) * h->net.np); }
O_API void o_net_params_get(o_net* h, float* p) { std::memcpy(p, h->net.th.data(), sizeof(float) * h->net.np); }
O_API void o_net_grads_get(o_net* h, float* p) { std::memcpy(p, h->net.g.data(), sizeof(float) * h->net.np); }
O_API void o_net_grads_set(o_net* h, const float* p) { std::memcpy(h->net.g.data(), p, sizeof(float) * h->net.np); }
// layout query: fills (offset, count) of every parameter array in flat order; returns the number of arrays.
O_API int o_net_layout(o_net* h, long* offs, long* counts, int* kinds, int cap) {
    int n = 0; Net& N = h->net;
    auto conv = [&](const ConvP& c) { if (n < cap) { offs[n] = (long)c.w; counts[n] = (long)c.cin * c.cout * c.k * c.k; kinds[n] = 0; } ++n; if (c.has_b) { if (n < cap) { offs[n] = (long)c.b; counts[n] = c.cout; kinds[n] = 1; } ++n; } };
    auto bn = [&](const BNP& b) { if (n < cap) { offs[n] = (long)b.g; counts[n] = b.c; kinds[n] = 2; } ++n; if (n < cap) { offs[n] = (long)b.b; counts[n] = b.c; kinds[n] = 3; } ++n; };
    conv(N.stem); bn(N.stem_bn);
    for (int i = 0; i < N.nblk; ++i) { conv(N.blk[i].c1); bn(N.blk[i].b1); conv(N.blk[i].c2); bn(N.blk[i].b2); if (N.blk[i].bottleneck) { conv(N.blk[i].c3); bn(N.blk[i].b3); } if (N.blk[i].down) { conv(N.blk[i].cd); bn(N.blk[i].bd); } }
    for (int b = 0; b < N.nbr; ++b) { conv(N.dc1[b]); conv(N.dc2[b]); }
    for (int l = 0; l < N.ns; ++l) conv(N.head[l]);
    conv(N.squeeze); for (int i = 0; i < 3; ++i) conv(N.pose[i]);
    return n;
}
// conv descriptor list in flat order: (cin, cout, k) per conv; returns count
O_API int o_net_convs(o_net* h, int* desc, int cap) {
    int n = 0; Net& N = h->net;
    auto conv = [&](const ConvP& c) { if (n < cap) { desc[n * 6 + 0] = c.cin; desc[n * 6 + 1] = c.cout; desc[n * 6 + 2] = c.k; desc[n * 6 + 3] = c.s; desc[n * 6 + 4] = c.p; desc[n * 6 + 5] = c.has_b; } ++n; };
    conv(N.stem);
    for (int i = 0; i < N.nblk; ++i) { conv(N.blk[i].c1); conv(N.blk[i].c2); if (N.blk[i].bottleneck) conv(N.blk[i].c3); if (N.blk[i].down) conv(N.blk[i].cd); }
    for (int b = 0; b < N.nbr; ++b) { conv(N.dc1[b]); conv(N.dc2[b]); }
    for (int l = 0; l < N.ns; ++l) conv(N.head[l]);
    conv(N.squeeze); for (int i = 0; i < 3; ++i) conv(N.pose[i]);
    return n;
}
// Model forward: disparities (4 buffers [N][1][h][w] coarse->fine), rvec/tvec [2][N][3]
O_API void o_net_model_forward(o_net* h, const float* x, const o_cfg* cfg, float* d0, float* d1, float* d2, float* d3, float* d4, float* rvec, float* tvec) {
    LossCfg c = to_cfg(cfg); model_loss(h->net, x, c, nullptr, false, h->out, true);
    float* ds[5] = {d0, d1, d2, d3, d4};
    for (int i = 0; i < h->net.ns; ++i) if (ds[i]) std::memcpy(ds[i], h->out.disp[i].p(), sizeof(float) * h->out.disp[i].size());
    for (int s = 0; s < 2; ++s) { if (rvec) std::memcpy(rvec + (size_t)s * c.N * 3, h->out.rvec[s].data(), sizeof(float) * c.N * 3); if (tvec) std::memcpy(tvec + (size_t)s * c.N * 3, h->out.tvec[s].data(), sizeof(float) * c.N * 3); }
}
// train_loss forward (+ pullback into the handle's gradient buffer when do_bwd)
O_API float o_net_train_loss(o_net* h, const float* x, const o_cfg* cfg, const float* auto_loss, int do_bwd) {
    LossCfg c = to_cfg(cfg); return model_loss(h->net, x, c, auto_loss, do_bwd != 0, h->out);
}
O_API void o_net_adam_step(o_net* h, double eta, double grad_scale) { adam_step(h->net, eta, grad_scale); }
// Flux.testmode! / the BatchNorm running statistics (mean[c], var[c] per BatchNorm in parameter order)
O_API void o_net_set_testmode(o_net* h, int on) { h->net.testmode = on != 0; }
O_API long o_net_bn_state_count(o_net* h) { return (long)h->net.nrs; }
O_API void o_net_bn_state_get(o_net* h, float* p) { std::memcpy(p, h->net.rs.data(), sizeof(float) * h->net.nrs); }
O_API void o_net_bn_state_set(o_net* h, const float* p) { std::memcpy(h->net.rs.data(), p, sizeof(float) * h->net.nrs); }
O_API void o_net_eval_disparity(o_net* h, const float* x, int N, int C, int H, int W, float* d0, float* d1, float* d2, float* d3, float* d4) {
    Tensor disp[5]; eval_disparity(h->net, x, N, C, H, W, disp);
    float* ds[5] = {d0, d1, d2, d3, d4};
    for (int i = 0; i < h->net.ns; ++i) if (ds[i]) std::memcpy(ds[i], disp[i].p(), sizeof(float) * disp[i].size());
}
O_API void o_net_opt_state_get(o_net* h, float* m, float* v, double* b1p, double* b2p) {
    std::memcpy(m, h->net.m.data(), sizeof(float) * h->net.np); std::memcpy(v, h->net.v.data(), sizeof(float) * h->net.np); *b1p = h->net.beta1p; *b2p = h->net.beta2p;
}
O_API void o_automasking_loss(const float* x, const o_cfg* cfg, float* out) { LossCfg c = to_cfg(cfg); Tensor t = automasking_loss(x, c); std::memcpy(out, t.p(), sizeof(float) * t.size()); }

// The loss half alone: disparities + poses in, loss (+ pullbacks) out.  Any output pointer may be null.
O_API float o_loss_fwd_bwd(const float* x, const o_cfg* cfg, const float* d0, const float* d1, const float* d2, const float* d3, const float* d4,
                           const float* rvec, const float* tvec, const float* auto_loss, int do_bwd,
                           float* dd0, float* dd1, float* dd2, float* dd3, float* dd4, float* drvec, float* dtvec, float* warped_out, float* warp_loss_out) {
    LossCfg c = to_cfg(cfg);
    const float* ds[5] = {d0, d1, d2, d3, d4}; Tensor disp[5], ddisp[5];
    for (int i = 0; i < c.nscales; ++i) { const int sh = c.simple ? 0 : 5 - c.levels[i]; disp[i] = wrap(ds[i], c.N, 1, c.H >> sh, c.W >> sh); }
    const float* rv[2] = {rvec, rvec + (size_t)c.N * 3}; const float* tv[2] = {tvec, tvec + (size_t)c.N * 3};
    Vec<float> drv((size_t)2 * c.N * 3, 0.f), dtv((size_t)2 * c.N * 3, 0.f);
    float* drp[2] = {drv.data(), drv.data() + (size_t)c.N * 3}; float* dtp[2] = {dtv.data(), dtv.data() + (size_t)c.N * 3};
    const float loss = loss_fwd_bwd(x, c, disp, rv, tv, auto_loss, do_bwd != 0, ddisp, drp, dtp, warped_out, warp_loss_out);
    if (do_bwd) {
        float* dds[5] = {dd0, dd1, dd2, dd3, dd4};
        for (int i = 0; i < c.nscales; ++i) if (dds[i]) std::memcpy(dds[i], ddisp[i].p(), sizeof(float) * ddisp[i].size());
        if (drvec) std::memcpy(drvec, drv.data(), sizeof(float) * drv.size());
        if (dtvec) std::memcpy(dtvec, dtv.data(), sizeof(float) * dtv.size());
    }
    return loss;
}
O_API int o_num_threads() { return omp_get_max_threads(); }
O_API void o_set_num_threads(int n) { omp_set_num_threads(n); }
