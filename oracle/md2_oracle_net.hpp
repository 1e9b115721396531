// md2_oracle_net.hpp — CPU restatement of the model + training step:
//   src/model.jl, src/depth_decoder.jl, src/pose_decoder.jl, src/training.jl:21-78,
//   src/Monodepth.jl:117-126,170 (ResNet-18 encoder, ADAM), with the Flux/NNlib layer semantics of
//   SURVEY.md §8c.  TEST INFRASTRUCTURE ONLY (see md2_oracle.h).  fp32 storage like the reference
//   (`precision = f32`, src/Monodepth.jl:72); long reductions accumulate in double.
//
// Encoder: ResNet.jl is not vendored in the reference tree; the torchvision trunks are assumed: ResNet-18
// (7x7 s2 p3 conv no-bias + BN + ReLU | 3x3 s2 p1 maxpool | 4 layers x 2 BasicBlocks, 1x1-s2 conv+BN
// shortcut), stages = (64,64,128,256,512) as required by src/depth_decoder.jl:31-35, and ResNet-50
// (Bottleneck x (3,4,6,3): 1x1 -> 3x3 carrying the stride -> 1x1 x4; stages (64,256,512,1024,2048);
// BASELINE config "ResNet-50 encoder, 640x192").  PARITY UNPINNED.
#pragma once
#include "md2_oracle_ops.hpp"
#include <memory>
#include <cstdio>

namespace md2o {

struct Tensor {
    Vec<float> v; int n = 0, c = 0, h = 0, w = 0;
    Tensor() {}
    Tensor(int n_, int c_, int h_, int w_) { reset(n_, c_, h_, w_); }
    void reset(int n_, int c_, int h_, int w_) { n = n_; c = c_; h = h_; w = w_; v.assign((size_t)n * c * h * w, 0.f); }
    size_t size() const { return v.size(); }
    float* p() { return v.data(); }
    const float* p() const { return v.data(); }
    size_t plane() const { return (size_t)h * w; }
};

// ---- Flux Conv: TRUE convolution (kernel flipped), weight (kW,kH,Cin,Cout) col-major == [Cout][Cin][KH][KW].
//      y[n,co,oh,ow] = b[co] + sum w[co,ci,kh,kw] * x[n,ci, oh*s - p + (KH-1-kh), ow*s - p + (KW-1-kw)]
inline void conv2d_fwd(const Tensor& x, const float* w, const float* b, Tensor& y, int cout, int k, int s, int p) {
    const int H = x.h, W = x.w, Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1, cin = x.c;
    y.reset(x.n, cout, Ho, Wo);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < x.n; ++n)
        for (int co = 0; co < cout; ++co) {
            float* yp = y.p() + ((size_t)n * cout + co) * Ho * Wo;
            const float bias = b ? b[co] : 0.f;
            for (int i = 0; i < Ho * Wo; ++i) yp[i] = bias;
            for (int ci = 0; ci < cin; ++ci) {
                const float* xp = x.p() + ((size_t)n * cin + ci) * H * W;
                for (int kh = 0; kh < k; ++kh)
                    for (int kw = 0; kw < k; ++kw) {
                        const float wv = w[(((size_t)co * cin + ci) * k + kh) * k + kw];
                        const int dh = (k - 1 - kh) - p, dw = (k - 1 - kw) - p;
                        int olo = 0; while (olo < Wo && olo * s + dw < 0) ++olo;
                        int ohi = Wo; while (ohi > olo && (ohi - 1) * s + dw >= W) --ohi;
                        for (int oh = 0; oh < Ho; ++oh) {
                            const int ih = oh * s + dh; if (ih < 0 || ih >= H) continue;
                            const float* xr = xp + (size_t)ih * W + dw; float* yr = yp + (size_t)oh * Wo;
                            for (int ow = olo; ow < ohi; ++ow) yr[ow] += wv * xr[ow * s];
                        }
                    }
            }
        }
}
// pullbacks: dx (may be null), dw += , db +=
inline void conv2d_bwd(const Tensor& x, const float* w, const Tensor& dy, Tensor* dx, float* dw, float* db, int k, int s, int p) {
    const int H = x.h, W = x.w, Ho = dy.h, Wo = dy.w, cin = x.c, cout = dy.c, N = x.n;
    if (dx) {
        dx->reset(N, cin, H, W);
#pragma omp parallel for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int ci = 0; ci < cin; ++ci) {
                float* gx = dx->p() + ((size_t)n * cin + ci) * H * W;
                for (int co = 0; co < cout; ++co) {
                    const float* gy = dy.p() + ((size_t)n * cout + co) * Ho * Wo;
                    for (int kh = 0; kh < k; ++kh)
                        for (int kw = 0; kw < k; ++kw) {
                            const float wv = w[(((size_t)co * cin + ci) * k + kh) * k + kw];
                            const int dh = (k - 1 - kh) - p, dw_ = (k - 1 - kw) - p;
                            int olo = 0; while (olo < Wo && olo * s + dw_ < 0) ++olo;
                            int ohi = Wo; while (ohi > olo && (ohi - 1) * s + dw_ >= W) --ohi;
                            for (int oh = 0; oh < Ho; ++oh) {
                                const int ih = oh * s + dh; if (ih < 0 || ih >= H) continue;
                                float* xr = gx + (size_t)ih * W + dw_; const float* yr = gy + (size_t)oh * Wo;
                                for (int ow = olo; ow < ohi; ++ow) xr[ow * s] += wv * yr[ow];
                            }
                        }
                }
            }
    }
    if (dw) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int kh = 0; kh < k; ++kh)
                    for (int kw = 0; kw < k; ++kw) {
                        const int dh = (k - 1 - kh) - p, dw_ = (k - 1 - kw) - p;
                        int olo = 0; while (olo < Wo && olo * s + dw_ < 0) ++olo;
                        int ohi = Wo; while (ohi > olo && (ohi - 1) * s + dw_ >= W) --ohi;
                        double acc = 0;
                        for (int n = 0; n < N; ++n) {
                            const float* xp = x.p() + ((size_t)n * cin + ci) * H * W;
                            const float* gy = dy.p() + ((size_t)n * cout + co) * Ho * Wo;
                            for (int oh = 0; oh < Ho; ++oh) {
                                const int ih = oh * s + dh; if (ih < 0 || ih >= H) continue;
                                const float* xr = xp + (size_t)ih * W + dw_; const float* yr = gy + (size_t)oh * Wo;
                                float a = 0; for (int ow = olo; ow < ohi; ++ow) a += xr[ow * s] * yr[ow];
                                acc += a;
                            }
                        }
                        dw[(((size_t)co * cin + ci) * k + kh) * k + kw] += (float)acc;
                    }
    }
    if (db)
        for (int co = 0; co < cout; ++co) {
            double acc = 0;
            for (int n = 0; n < N; ++n) { const float* gy = dy.p() + ((size_t)n * cout + co) * Ho * Wo; for (int i = 0; i < Ho * Wo; ++i) acc += gy[i]; }
            db[co] += (float)acc;
        }
}

// ---- Flux BatchNorm, train mode: batch mean, BIASED variance over (W,H,N), eps = 1e-5.
struct BNSave { Tensor xhat; Vec<float> invstd, mean, var; };
inline void bn_train_fwd(const Tensor& x, const float* gamma, const float* beta, Tensor& y, BNSave& sv) {
    const int C = x.c; const size_t P = x.plane(); const double M = (double)x.n * P;
    y.reset(x.n, C, x.h, x.w); sv.xhat.reset(x.n, C, x.h, x.w); sv.invstd.assign(C, 0); sv.mean.assign(C, 0); sv.var.assign(C, 0);
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        double s = 0, ss = 0;
        for (int n = 0; n < x.n; ++n) { const float* xp = x.p() + ((size_t)n * C + c) * P; for (size_t i = 0; i < P; ++i) s += xp[i]; }
        const double mean = s / M;
        for (int n = 0; n < x.n; ++n) { const float* xp = x.p() + ((size_t)n * C + c) * P; for (size_t i = 0; i < P; ++i) { const double d = xp[i] - mean; ss += d * d; } }
        const double var = ss / M; const float is = (float)(1.0 / std::sqrt(var + 1e-5));
        sv.invstd[c] = is; sv.mean[c] = (float)mean; sv.var[c] = (float)var;
        for (int n = 0; n < x.n; ++n) {
            const size_t o = ((size_t)n * C + c) * P;
            for (size_t i = 0; i < P; ++i) { const float xh = (x.v[o + i] - (float)mean) * is; sv.xhat.v[o + i] = xh; y.v[o + i] = gamma[c] * xh + beta[c]; }
        }
    }
}
inline void bn_train_bwd(const Tensor& dy, const float* gamma, const BNSave& sv, Tensor& dx, float* dgamma, float* dbeta) {
    const int C = dy.c; const size_t P = dy.plane(); const double M = (double)dy.n * P;
    dx.reset(dy.n, C, dy.h, dy.w);
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        double s1 = 0, s2 = 0;
        for (int n = 0; n < dy.n; ++n) { const size_t o = ((size_t)n * C + c) * P; for (size_t i = 0; i < P; ++i) { s1 += dy.v[o + i]; s2 += (double)dy.v[o + i] * sv.xhat.v[o + i]; } }
        dgamma[c] += (float)s2; dbeta[c] += (float)s1;
        const float m1 = (float)(s1 / M), m2 = (float)(s2 / M), k = gamma[c] * sv.invstd[c];
        for (int n = 0; n < dy.n; ++n) { const size_t o = ((size_t)n * C + c) * P; for (size_t i = 0; i < P; ++i) dx.v[o + i] = k * (dy.v[o + i] - m1 - sv.xhat.v[o + i] * m2); }
    }
}

// ---- MaxPool((3,3); pad=1, stride=2): padding never wins.
inline void maxpool3s2_fwd(const Tensor& x, Tensor& y, Vec<int32_t>& arg) {
    const int Ho = (x.h + 2 - 3) / 2 + 1, Wo = (x.w + 2 - 3) / 2 + 1;
    y.reset(x.n, x.c, Ho, Wo); arg.assign(y.size(), 0);
#pragma omp parallel for
    for (int pl = 0; pl < x.n * x.c; ++pl)
        for (int oh = 0; oh < Ho; ++oh)
            for (int ow = 0; ow < Wo; ++ow) {
                float best = -INFINITY; int bi = -1;
                for (int dh = 0; dh < 3; ++dh)
                    for (int dw = 0; dw < 3; ++dw) {
                        const int ih = oh * 2 - 1 + dh, iw = ow * 2 - 1 + dw;
                        if (ih < 0 || ih >= x.h || iw < 0 || iw >= x.w) continue;
                        const float v = x.v[(size_t)pl * x.h * x.w + (size_t)ih * x.w + iw];
                        if (v > best) { best = v; bi = ih * x.w + iw; }
                    }
                y.v[((size_t)pl * Ho + oh) * Wo + ow] = best; arg[((size_t)pl * Ho + oh) * Wo + ow] = bi;
            }
}
inline void maxpool3s2_bwd(const Tensor& dy, const Vec<int32_t>& arg, Tensor& dx, int H, int W) {
    dx.reset(dy.n, dy.c, H, W);
    const size_t Po = dy.plane();
    for (int pl = 0; pl < dy.n * dy.c; ++pl)
        for (size_t i = 0; i < Po; ++i) dx.v[(size_t)pl * H * W + arg[(size_t)pl * Po + i]] += dy.v[(size_t)pl * Po + i];
}

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SIGMOID = 3 };
inline void act_fwd(Tensor& y, int act) {
    for (auto& v : y.v) {
        if (act == ACT_RELU) v = v > 0 ? v : 0.f;
        else if (act == ACT_ELU) v = v > 0 ? v : std::expm1(v);     // NNlib.elu, alpha = 1
        else if (act == ACT_SIGMOID) v = 1.f / (1.f + std::exp(-v));
    }
}
// derivative from the OUTPUT value y
inline void act_bwd(const Tensor& y, Tensor& dy, int act) {
    for (size_t i = 0; i < y.size(); ++i) {
        const float o = y.v[i];
        if (act == ACT_RELU) dy.v[i] = o > 0 ? dy.v[i] : 0.f;
        else if (act == ACT_ELU) dy.v[i] *= o > 0 ? 1.f : (o + 1.f);
        else if (act == ACT_SIGMOID) dy.v[i] *= o * (1.f - o);
    }
}

// reflect-pad / bilinear x2 / cat / slice on Tensors
inline Tensor t_pad_reflect(const Tensor& x) { Tensor y(x.n, x.c, x.h + 2, x.w + 2); pad_reflect1(x.p(), y.p(), x.n * x.c, x.h, x.w); return y; }
inline Tensor t_pad_reflect_bwd(const Tensor& dy, int H, int W) { Tensor dx(dy.n, dy.c, H, W); pad_reflect1_bwd(dy.p(), dx.p(), dy.n * dy.c, H, W); return dx; }
inline Tensor t_up2(const Tensor& x) { Tensor y(x.n, x.c, x.h * 2, x.w * 2); upsample_bilinear(x.p(), y.p(), x.n * x.c, x.h, x.w, y.h, y.w); return y; }
inline Tensor t_up2_bwd(const Tensor& dy) { Tensor dx(dy.n, dy.c, dy.h / 2, dy.w / 2); upsample_bilinear_bwd(dy.p(), dx.p(), dy.n * dy.c, dx.h, dx.w, dy.h, dy.w); return dx; }
inline Tensor t_cat(const Tensor& a, const Tensor& b) {
    Tensor y(a.n, a.c + b.c, a.h, a.w); const size_t P = a.plane();
    for (int n = 0; n < a.n; ++n) {
        std::memcpy(y.p() + (size_t)n * y.c * P, a.p() + (size_t)n * a.c * P, sizeof(float) * a.c * P);
        std::memcpy(y.p() + ((size_t)n * y.c + a.c) * P, b.p() + (size_t)n * b.c * P, sizeof(float) * b.c * P);
    }
    return y;
}
inline void t_uncat(const Tensor& y, Tensor& a, Tensor& b, int ca) {
    const size_t P = y.plane(); a.reset(y.n, ca, y.h, y.w); b.reset(y.n, y.c - ca, y.h, y.w);
    for (int n = 0; n < y.n; ++n) {
        std::memcpy(a.p() + (size_t)n * ca * P, y.p() + (size_t)n * y.c * P, sizeof(float) * ca * P);
        std::memcpy(b.p() + (size_t)n * b.c * P, y.p() + ((size_t)n * y.c + ca) * P, sizeof(float) * b.c * P);
    }
}
// f[:, :, :, l, :] of a (w,h,c,L,N) array stored as [N*L][c][h][w] with batch index n*L + l
inline Tensor t_frame(const Tensor& f, int L, int l) {
    Tensor y(f.n / L, f.c, f.h, f.w); const size_t S = (size_t)f.c * f.plane();
    for (int n = 0; n < y.n; ++n) std::memcpy(y.p() + n * S, f.p() + ((size_t)n * L + l) * S, sizeof(float) * S);
    return y;
}
inline void t_frame_add(Tensor& df, const Tensor& dy, int L, int l) {
    const size_t S = (size_t)df.c * df.plane();
    for (int n = 0; n < dy.n; ++n) for (size_t i = 0; i < S; ++i) df.v[((size_t)n * L + l) * S + i] += dy.v[n * S + i];
}

// ------------------------------------------------------------------------------------------------
// Parameter table.  Flat order (documented in DESIGN.md, mirrored by the HIP library):
//   encoder: stem.W, stem.bn(g,b); layer{1..4}.block{0,1}: conv1.W, bn1(g,b), conv2.W, bn2(g,b)[, down.W, down.bn(g,b)]
//   depth decoder (src/depth_decoder.jl:26-47): branches 1..max(scale_levels): c1.W, c1.b, c2.W, c2.b ; one head per scale level (2..5 by default): W, b
//   pose decoder (src/pose_decoder.jl:13-21): squeezer W,b ; pose[1] W,b ; pose[2] W,b ; pose[3] W,b
// Conv weights in Flux layout (kW,kH,Cin,Cout) col-major.
// ------------------------------------------------------------------------------------------------
struct ConvP { int cin, cout, k, s, p; size_t w, b; bool has_b; };
struct BNP { int c; size_t g, b; size_t rs; };            // rs: offset of (running mean[c], running var[c]) in Net::rs

struct Net {
    int C = 1;                         // image channels
    Vec<float> th, g, m, v;            // params, grads, Adam moments
    double beta1p = 1.0, beta2p = 1.0; // running beta powers (Flux ADAM state)
    size_t np = 0, nrs = 0;
    // Flux BatchNorm state: tracked mu / sigma^2 (momentum 0.1, `_track_stats!`), used instead of the batch statistics under Flux.testmode!
    // (the reference's eval_image / eval_video, src/Monodepth.jl:209,237).  For every BatchNorm in parameter order: mean[c], var[c].
    mutable Vec<float> rs; bool testmode = false;
    ConvP stem; BNP stem_bn;
    struct Block { ConvP c1, c2, c3, cd; BNP b1, b2, b3, bd; bool down; bool bottleneck; bool stage_end; };
    int depth = 18, nblk = 8;
    Block blk[16];
    ConvP dc1[5], dc2[5], head[5], squeeze, pose[3];
    // DepthDecoder(; scale_levels): ns outputs at ascending levels lv[i] in 1..5; branches 1..nbr = lv[ns-1]; out_of[b] = output index of the head after branch b, or -1
    int ns = 4, lv[5] = {2, 3, 4, 5, 0}, nbr = 5, out_of[5] = {-1, 0, 1, 2, 3};
    int dec_ch[5] = {256, 128, 64, 32, 16}, in_ch[5] = {512, 256, 128, 64, 32}, skip_ch[5] = {256, 128, 64, 64, 0};

    ConvP mk(int cin, int cout, int k, int s, int p, bool bias) { ConvP c{cin, cout, k, s, p, np, 0, bias}; np += (size_t)cin * cout * k * k; if (bias) { c.b = np; np += cout; } return c; }
    BNP mkbn(int c) { BNP b{c, np, np + (size_t)c, nrs}; np += 2 * (size_t)c; nrs += 2 * (size_t)c; return b; }
    explicit Net(int C_, int depth_ = 18, const int* levels = nullptr) : C(C_), depth(depth_) {
        if (levels && levels[0]) { ns = 0; for (int b = 0; b < 5; ++b) out_of[b] = -1; for (int i = 0; i < 5 && levels[i]; ++i) { lv[i] = levels[i]; out_of[levels[i] - 1] = i; ns = i + 1; } nbr = lv[ns - 1]; }
        stem = mk(C, 64, 7, 2, 3, false); stem_bn = mkbn(64);
        const int planes[4] = {64, 128, 256, 512}; int cin = 64;
        const int nb18[4] = {2, 2, 2, 2}, nb50[4] = {3, 4, 6, 3};
        const int* nb = depth == 50 ? nb50 : nb18;
        nblk = 0;
        for (int l = 0; l < 4; ++l)
            for (int b = 0; b < nb[l]; ++b) {
                Block& B = blk[nblk++]; const int s = (b == 0 && l > 0) ? 2 : 1;
                B.bottleneck = depth == 50; B.stage_end = (b == nb[l] - 1);
                int co;
                if (!B.bottleneck) { co = planes[l]; B.c1 = mk(cin, co, 3, s, 1, false); B.b1 = mkbn(co); B.c2 = mk(co, co, 3, 1, 1, false); B.b2 = mkbn(co); }
                else { co = 4 * planes[l]; B.c1 = mk(cin, planes[l], 1, 1, 0, false); B.b1 = mkbn(planes[l]); B.c2 = mk(planes[l], planes[l], 3, s, 1, false); B.b2 = mkbn(planes[l]);
                       B.c3 = mk(planes[l], co, 1, 1, 0, false); B.b3 = mkbn(co); }
                B.down = (s != 1 || cin != co);
                if (B.down) { B.cd = mk(cin, co, 1, s, 0, false); B.bd = mkbn(co); }
                cin = co;
            }
        if (depth == 50) { const int st[5] = {64, 256, 512, 1024, 2048}; in_ch[0] = st[4]; skip_ch[0] = st[3]; skip_ch[1] = st[2]; skip_ch[2] = st[1]; skip_ch[3] = st[0]; }
        for (int b = 0; b < nbr; ++b) { dc1[b] = mk(in_ch[b], dec_ch[b], 3, 1, 0, true); dc2[b] = mk(dec_ch[b] + skip_ch[b], dec_ch[b], 3, 1, 0, true); }
        for (int i = 0; i < ns; ++i) head[i] = mk(dec_ch[lv[i] - 1], 1, 3, 1, 0, true);
        squeeze = mk(cin, 256, 1, 1, 0, true); pose[0] = mk(512, 256, 3, 1, 1, true); pose[1] = mk(256, 256, 3, 1, 1, true); pose[2] = mk(256, 6, 1, 1, 0, true);
        th.assign(np, 0); g.assign(np, 0); m.assign(np, 0); v.assign(np, 0);
        rs.assign(nrs, 0);
        auto init_rs = [&](const BNP& b) { for (int i = 0; i < b.c; ++i) rs[b.rs + b.c + i] = 1.f; };       // Flux: mu = 0, sigma^2 = 1
        init_rs(stem_bn);
        for (int i = 0; i < nblk; ++i) { init_rs(blk[i].b1); init_rs(blk[i].b2); if (blk[i].bottleneck) init_rs(blk[i].b3); if (blk[i].down) init_rs(blk[i].bd); }
    }
    const float* W(const ConvP& c) const { return th.data() + c.w; }
    const float* B(const ConvP& c) const { return c.has_b ? th.data() + c.b : nullptr; }
    float* gW(const ConvP& c) { return g.data() + c.w; }
    float* gB(const ConvP& c) { return c.has_b ? g.data() + c.b : nullptr; }
};

// ---- saved state of one forward pass -----------------------------------------------------------
struct ConvBN { Tensor x, z, a; BNSave sv; };                     // a = act(bn(conv(x)))
struct BlockSave { ConvBN c1, c2, c3, cd; Tensor out; };
struct EncSave { ConvBN stem; Tensor pooled; Vec<int32_t> arg; BlockSave blk[16]; Tensor feat[5]; };
struct DecConv { Tensor xin, xpad, y; };                          // y = act(conv(pad_reflect(xin)))
struct DepthSave { DecConv c1[5], c2[5], head[5]; Tensor up[5], cat[5]; };
struct PoseSave { Tensor f[2], sq[2], cat, y1, y2, y3; int order[2]; };   // f: the (first,second) feature pair

// Flux BatchNorm under testmode!: (x - mu) / sqrt(sigma^2 + eps) with the tracked statistics
inline void bn_test_fwd(const Tensor& x, const float* gamma, const float* beta, const float* rmean, const float* rvar, Tensor& y) {
    const int C = x.c; const size_t P = x.plane();
    y.reset(x.n, C, x.h, x.w);
    for (int c = 0; c < C; ++c) {
        const float is = 1.f / std::sqrt(rvar[c] + 1e-5f);
        for (int n = 0; n < x.n; ++n) { const size_t o = ((size_t)n * C + c) * P; for (size_t i = 0; i < P; ++i) y.v[o + i] = gamma[c] * ((x.v[o + i] - rmean[c]) * is) + beta[c]; }
    }
}
inline void convbn_fwd(const Net& N, const ConvP& c, const BNP& b, const Tensor& x, ConvBN& s, int act) {
    s.x = x; conv2d_fwd(x, N.W(c), nullptr, s.z, c.cout, c.k, c.s, c.p);
    if (N.testmode) { bn_test_fwd(s.z, N.th.data() + b.g, N.th.data() + b.b, N.rs.data() + b.rs, N.rs.data() + b.rs + b.c, s.a); act_fwd(s.a, act); return; }
    bn_train_fwd(s.z, N.th.data() + b.g, N.th.data() + b.b, s.a, s.sv); act_fwd(s.a, act);
    // Flux `_track_stats!`: mu <- 0.9 mu + 0.1 mean ; sigma^2 <- 0.9 sigma^2 + 0.1 * m/(m-1) * var (m = elements per channel)
    const double m = (double)s.z.n * s.z.plane(); const float corr = m > 1 ? (float)(m / (m - 1)) : 1.f;
    for (int i = 0; i < b.c; ++i) {
        N.rs[b.rs + i] = 0.9f * N.rs[b.rs + i] + 0.1f * s.sv.mean[i];
        N.rs[b.rs + b.c + i] = 0.9f * N.rs[b.rs + b.c + i] + 0.1f * corr * s.sv.var[i];
    }
}
// d(out of bn, before act) -> dx ; accumulates dW, dgamma, dbeta
inline void convbn_bwd(Net& N, const ConvP& c, const BNP& b, const ConvBN& s, const Tensor& da_preact, Tensor* dx) {
    Tensor dz; bn_train_bwd(da_preact, N.th.data() + b.g, s.sv, dz, N.g.data() + b.g, N.g.data() + b.b);
    conv2d_bwd(s.x, N.W(c), dz, dx, N.gW(c), nullptr, c.k, c.s, c.p);
}

inline void encoder_fwd(const Net& N, const Tensor& x, EncSave& S) {
    convbn_fwd(N, N.stem, N.stem_bn, x, S.stem, ACT_RELU);
    S.feat[0] = S.stem.a;
    maxpool3s2_fwd(S.stem.a, S.pooled, S.arg);
    const Tensor* cur = &S.pooled;
    int stage = 1;
    for (int i = 0; i < N.nblk; ++i) {
        const Net::Block& B = N.blk[i]; BlockSave& bs = S.blk[i];
        convbn_fwd(N, B.c1, B.b1, *cur, bs.c1, ACT_RELU);
        if (!B.bottleneck) { convbn_fwd(N, B.c2, B.b2, bs.c1.a, bs.c2, ACT_NONE); bs.out = bs.c2.a; }
        else { convbn_fwd(N, B.c2, B.b2, bs.c1.a, bs.c2, ACT_RELU); convbn_fwd(N, B.c3, B.b3, bs.c2.a, bs.c3, ACT_NONE); bs.out = bs.c3.a; }
        if (B.down) { convbn_fwd(N, B.cd, B.bd, *cur, bs.cd, ACT_NONE); for (size_t j = 0; j < bs.out.size(); ++j) bs.out.v[j] += bs.cd.a.v[j]; }
        else for (size_t j = 0; j < bs.out.size(); ++j) bs.out.v[j] += cur->v[j];
        act_fwd(bs.out, ACT_RELU);
        cur = &bs.out;
        if (B.stage_end) S.feat[stage++] = bs.out;
    }
}
// dfeat[5]: pullbacks of the five stage outputs (consumed; modified in place)
inline void encoder_bwd(Net& N, EncSave& S, Tensor dfeat[5]) {
    Tensor dcur;   // gradient wrt the running activation (output of block i)
    int stage = 4;
    for (int i = N.nblk - 1; i >= 0; --i) {
        const Net::Block& B = N.blk[i]; BlockSave& bs = S.blk[i];
        if (B.stage_end) { if (dcur.size() == 0) dcur = dfeat[stage]; else for (size_t j = 0; j < dcur.size(); ++j) dcur.v[j] += dfeat[stage].v[j]; --stage; }
        act_bwd(bs.out, dcur, ACT_RELU);                       // through the final ReLU
        Tensor din;                                             // gradient wrt the block input
        if (B.down) convbn_bwd(N, B.cd, B.bd, bs.cd, dcur, &din); else din = dcur;
        Tensor da1;
        if (!B.bottleneck) convbn_bwd(N, B.c2, B.b2, bs.c2, dcur, &da1);
        else { Tensor da2; convbn_bwd(N, B.c3, B.b3, bs.c3, dcur, &da2); act_bwd(bs.c2.a, da2, ACT_RELU); convbn_bwd(N, B.c2, B.b2, bs.c2, da2, &da1); }
        act_bwd(bs.c1.a, da1, ACT_RELU);
        Tensor dmain; convbn_bwd(N, B.c1, B.b1, bs.c1, da1, &dmain);
        for (size_t j = 0; j < din.size(); ++j) din.v[j] += dmain.v[j];
        dcur = din;
    }
    Tensor dstem; maxpool3s2_bwd(dcur, S.arg, dstem, S.stem.a.h, S.stem.a.w);
    for (size_t j = 0; j < dstem.size(); ++j) dstem.v[j] += dfeat[0].v[j];
    act_bwd(S.stem.a, dstem, ACT_RELU);
    convbn_bwd(N, N.stem, N.stem_bn, S.stem, dstem, nullptr);   // the image needs no gradient
}

inline void decconv_fwd(const Net& N, const ConvP& c, const Tensor& x, DecConv& s, int act) {   // DecoderBlock, src/depth_decoder.jl:1-5
    s.xin = x; s.xpad = t_pad_reflect(x); conv2d_fwd(s.xpad, N.W(c), N.B(c), s.y, c.cout, c.k, 1, 0); act_fwd(s.y, act);
}
inline Tensor decconv_bwd(Net& N, const ConvP& c, const DecConv& s, Tensor dy, int act) {
    act_bwd(s.y, dy, act); Tensor dxp; conv2d_bwd(s.xpad, N.W(c), dy, &dxp, N.gW(c), N.gB(c), c.k, 1, 0);
    return t_pad_reflect_bwd(dxp, s.xin.h, s.xin.w);
}
// DepthDecoder forward, src/depth_decoder.jl:49-65.  feat: the 5 TARGET-frame features; disp[i] = output of scale level lv[i].
inline void depth_fwd(const Net& N, const Tensor feat[5], DepthSave& S, Tensor disp[5]) {
    Tensor x = feat[4];
    for (int b = 0; b < N.nbr; ++b) {
        decconv_fwd(N, N.dc1[b], x, S.c1[b], ACT_ELU);
        S.up[b] = t_up2(S.c1[b].y);
        S.cat[b] = (b < 4) ? t_cat(S.up[b], feat[3 - b]) : S.up[b];
        decconv_fwd(N, N.dc2[b], S.cat[b], S.c2[b], ACT_ELU);
        x = S.c2[b].y;
        if (const int o = N.out_of[b]; o >= 0) { decconv_fwd(N, N.head[o], x, S.head[o], ACT_SIGMOID); disp[o] = S.head[o].y; }
    }
}
inline void depth_bwd(Net& N, DepthSave& S, const Tensor ddisp[5], Tensor dfeat[5]) {   // dfeat: target-frame feature grads (output; empty where the decoder does not read the stage)
    Tensor dx;   // gradient wrt the running x (output of branch b)
    for (int b = N.nbr - 1; b >= 0; --b) {
        Tensor dout;
        if (const int o = N.out_of[b]; o >= 0) dout = decconv_bwd(N, N.head[o], S.head[o], ddisp[o], ACT_SIGMOID);
        if (dx.size()) { if (dout.size()) for (size_t j = 0; j < dout.size(); ++j) dout.v[j] += dx.v[j]; else dout = dx; }
        Tensor dcat = decconv_bwd(N, N.dc2[b], S.c2[b], dout, ACT_ELU);
        Tensor dup;
        if (b < 4) { Tensor dskip; t_uncat(dcat, dup, dskip, N.dec_ch[b]); dfeat[3 - b] = dskip; } else dup = dcat;
        Tensor dc1 = t_up2_bwd(dup);
        dx = decconv_bwd(N, N.dc1[b], S.c1[b], dc1, ACT_ELU);
    }
    dfeat[4] = dx;
}
// PoseDecoder forward, src/pose_decoder.jl:23-32.  f0,f1: (13,4,512,N) pair in temporal order -> rvec [N][3], tvec [N][3]
inline void pose_fwd(const Net& N, const Tensor& f0, const Tensor& f1, PoseSave& S, float* rvec, float* tvec) {
    S.f[0] = f0; S.f[1] = f1;
    for (int i = 0; i < 2; ++i) { conv2d_fwd(S.f[i], N.W(N.squeeze), N.B(N.squeeze), S.sq[i], 256, 1, 1, 0); act_fwd(S.sq[i], ACT_RELU); }
    S.cat = t_cat(S.sq[0], S.sq[1]);
    conv2d_fwd(S.cat, N.W(N.pose[0]), N.B(N.pose[0]), S.y1, 256, 3, 1, 1); act_fwd(S.y1, ACT_RELU);
    conv2d_fwd(S.y1, N.W(N.pose[1]), N.B(N.pose[1]), S.y2, 256, 3, 1, 1); act_fwd(S.y2, ACT_RELU);
    conv2d_fwd(S.y2, N.W(N.pose[2]), N.B(N.pose[2]), S.y3, 6, 1, 1, 0);
    const size_t P = S.y3.plane();
    for (int n = 0; n < S.y3.n; ++n)
        for (int c = 0; c < 6; ++c) {
            double s = 0; for (size_t i = 0; i < P; ++i) s += S.y3.v[((size_t)n * 6 + c) * P + i];
            const float val = 0.01f * (float)(s / (double)P);
            if (c < 3) rvec[n * 3 + c] = val; else tvec[n * 3 + c - 3] = val;
        }
}
inline void pose_bwd(Net& N, PoseSave& S, const float* drvec, const float* dtvec, Tensor& df0, Tensor& df1) {
    Tensor dy3(S.y3.n, 6, S.y3.h, S.y3.w); const size_t P = dy3.plane();
    for (int n = 0; n < dy3.n; ++n)
        for (int c = 0; c < 6; ++c) { const float g = 0.01f * (c < 3 ? drvec[n * 3 + c] : dtvec[n * 3 + c - 3]) / (float)P; for (size_t i = 0; i < P; ++i) dy3.v[((size_t)n * 6 + c) * P + i] = g; }
    Tensor dy2; conv2d_bwd(S.y2, N.W(N.pose[2]), dy3, &dy2, N.gW(N.pose[2]), N.gB(N.pose[2]), 1, 1, 0);
    act_bwd(S.y2, dy2, ACT_RELU);
    Tensor dy1; conv2d_bwd(S.y1, N.W(N.pose[1]), dy2, &dy1, N.gW(N.pose[1]), N.gB(N.pose[1]), 3, 1, 1);
    act_bwd(S.y1, dy1, ACT_RELU);
    Tensor dcat; conv2d_bwd(S.cat, N.W(N.pose[0]), dy1, &dcat, N.gW(N.pose[0]), N.gB(N.pose[0]), 3, 1, 1);
    Tensor ds[2]; t_uncat(dcat, ds[0], ds[1], 256);
    Tensor* dfs[2] = {&df0, &df1};
    for (int i = 0; i < 2; ++i) { act_bwd(S.sq[i], ds[i], ACT_RELU); conv2d_bwd(S.f[i], N.W(N.squeeze), ds[i], dfs[i], N.gW(N.squeeze), N.gB(N.squeeze), 1, 1, 0); }
}

// ------------------------------------------------------------------------------------------------
// The loss half of train_loss (src/training.jl:28-77) given disparities + poses, with full pullbacks.
// x: [N][L=3][C][H][W];  disp[i]: [N][1][h_i][w_i];  rvec/tvec[s]: [N][3] for the two sources.
// Returns loss; if do_bwd, fills ddisp[i], drvec[s], dtvec[s].
// ------------------------------------------------------------------------------------------------
struct LossCfg {
    int N, C, H, W, L = 3, target_id = 2, nsrc = 2; int source_ids[2] = {1, 3};
    double min_depth = 0.1, max_depth = 100.0, smoothness = 1e-3; int automasking = 0;
    double scales[5] = {0.125, 0.25, 0.5, 1.0, 0.0}; int nscales = 4; int levels[5] = {2, 3, 4, 5, 0};      // TrainCache.scales; the DepthDecoder's scale levels (output i is H >> (5 - level))
    // mode 1 = slow_depth (src/simple_depth.jl:25-42): ONE full-resolution scale, smooth_loss on the raw disparity (no mean
    // normalisation) with weight 1
    int simple = 0;
    float K[9], invK[9];
};
inline Tensor x_frame(const float* x, const LossCfg& c, int id1) {   // x[:, :, :, id, :], 1-based id
    Tensor y(c.N, c.C, c.H, c.W); const size_t S = (size_t)c.C * c.H * c.W;
    for (int n = 0; n < c.N; ++n) std::memcpy(y.p() + n * S, x + ((size_t)n * c.L + (id1 - 1)) * S, sizeof(float) * S);
    return y;
}
// automasking_loss, src/training.jl:9-11
inline Tensor automasking_loss(const float* x, const LossCfg& c) {
    Tensor tgt = x_frame(x, c, c.target_id); const size_t NP = (size_t)c.N * c.H * c.W;
    Vec<float> stack(NP * c.nsrc);
    for (int s = 0; s < c.nsrc; ++s) { Tensor src = x_frame(x, c, c.source_ids[s]); photometric_fwd_bwd<float>(src.p(), tgt.p(), stack.data() + s * NP, nullptr, nullptr, c.N, c.C, c.H, c.W); }
    Tensor out(c.N, 1, c.H, c.W); min_stack<float>(stack.data(), c.nsrc, NP, out.p(), nullptr); return out;
}
inline float loss_fwd_bwd(const float* x, const LossCfg& c, const Tensor disp[5], const float* const rvec[2], const float* const tvec[2],
                          const float* auto_loss, bool do_bwd, Tensor ddisp[5], float* drvec[2], float* dtvec[2],
                          float* warped_out /* optional [nscales][nsrc][N][C][H][W] */, float* warp_loss_out /* optional [nscales][N][H][W] */) {
    const int N = c.N, C = c.C, H = c.H, W = c.W; const size_t P = (size_t)H * W, NP = N * P;
    Tensor tgt = x_frame(x, c, c.target_id);
    Tensor src[2]; for (int s = 0; s < c.nsrc; ++s) src[s] = x_frame(x, c, c.source_ids[s]);
    // Ps = composeT(rvec, tvec, source_id < target_id), :29-32
    Vec<float> R[2], t[2], dRacc[2], dtacc[2];
    for (int s = 0; s < c.nsrc; ++s) {
        R[s].assign((size_t)N * 9, 0); t[s].assign((size_t)N * 3, 0); dRacc[s].assign((size_t)N * 9, 0); dtacc[s].assign((size_t)N * 3, 0);
        composeT<float>(rvec[s], tvec[s], c.source_ids[s] < c.target_id, R[s].data(), t[s].data(), N);
    }
    float loss = 0.f;
    const float inv_scales = 1.f / (float)c.nscales;   // d(loss/S)/d(loss)
    for (int i = 0; i < c.nscales; ++i) {
        const Tensor& d0 = disp[i];
        Tensor d(N, 1, H, W);
        const bool up = d0.w != W || d0.h != H;
        if (up) upsample_bilinear(d0.p(), d.p(), N, d0.h, d0.w, H, W); else d = d0;
        Tensor depth(N, 1, H, W); disparity_to_depth<float>(d.p(), depth.p(), NP, c.min_depth, c.max_depth);
        Vec<float> coords(NP * 3); backproject<float>(depth.p(), c.invK, coords.data(), N, H, W);
        Vec<float> uv[2]; Tensor warped[2]; Vec<float> stack(NP * c.nsrc);
        for (int s = 0; s < c.nsrc; ++s) {
            uv[s].assign(NP * 2, 0); project<float>(coords.data(), c.K, R[s].data(), t[s].data(), uv[s].data(), N, H, W);
            warped[s].reset(N, C, H, W);
            grid_sample_fwd_bwd<float>(src[s].p(), uv[s].data(), warped[s].p(), nullptr, nullptr, nullptr, N, C, H, W, H, W, 1);
            photometric_fwd_bwd<float>(warped[s].p(), tgt.p(), stack.data() + s * NP, nullptr, nullptr, N, C, H, W);
            if (warped_out) std::memcpy(warped_out + ((size_t)i * c.nsrc + s) * NP * C, warped[s].p(), sizeof(float) * NP * C);
        }
        Vec<float> wl(NP); Vec<int32_t> arg(NP);
        min_stack<float>(stack.data(), c.nsrc, NP, wl.data(), arg.data());
        Vec<uint8_t> masked(NP, 0);
        if (c.automasking && auto_loss)
            for (size_t j = 0; j < NP; ++j) if (!(wl[j] < auto_loss[j])) { wl[j] = auto_loss[j]; masked[j] = 1; }   // min(cat(mask, warp_loss)): ties -> mask (first)
        if (warp_loss_out) std::memcpy(warp_loss_out + (size_t)i * NP, wl.data(), sizeof(float) * NP);
        double wsum = 0; for (size_t j = 0; j < NP; ++j) wsum += wl[j];
        // normalized_disparity = d ./ (mean(d; dims=(1,2)) .+ 1e-7), :64-65
        Vec<float> mean_d(N); Tensor nd(N, 1, H, W);
        for (int n = 0; n < N; ++n) { double s = 0; for (size_t j = 0; j < P; ++j) s += d.v[n * P + j]; mean_d[n] = (float)(s / (double)P); for (size_t j = 0; j < P; ++j) nd.v[n * P + j] = c.simple ? d.v[n * P + j] : d.v[n * P + j] / (mean_d[n] + 1e-7f); }
        const float sm_w = c.simple ? 1.f : (float)c.smoothness * (float)c.scales[i];
        Tensor dnd; if (do_bwd) dnd.reset(N, 1, H, W);
        const float sl = smooth_loss_fwd_bwd<float>(nd.p(), tgt.p(), inv_scales * sm_w, do_bwd ? dnd.p() : nullptr, N, C, H, W);
        loss += (float)(wsum / (double)NP) + sl * sm_w;
        if (!do_bwd) continue;
        // ---------------- pullback of scale i ----------------
        Tensor dd(N, 1, H, W);                                   // wrt the (upsampled) disparity
        for (int n = 0; n < N; ++n) {                            // through d / (mean + eps)
            if (c.simple) { for (size_t j = 0; j < P; ++j) dd.v[n * P + j] = dnd.v[n * P + j]; continue; }
            const float den = mean_d[n] + 1e-7f; double dot = 0;
            for (size_t j = 0; j < P; ++j) dot += (double)dnd.v[n * P + j] * d.v[n * P + j];
            const float corr = (float)(dot / ((double)den * den) / (double)P);
            for (size_t j = 0; j < P; ++j) dd.v[n * P + j] = dnd.v[n * P + j] / den - corr;
        }
        Vec<float> dcoords(NP * 3, 0.f);
        for (int s = 0; s < c.nsrc; ++s) {
            Vec<float> dpl(NP);                                   // d mean(warp_loss) / d photometric_s
            for (size_t j = 0; j < NP; ++j) dpl[j] = (!masked[j] && arg[j] == s) ? inv_scales / (float)NP : 0.f;
            Tensor dwarp(N, C, H, W);
            photometric_fwd_bwd<float>(warped[s].p(), tgt.p(), nullptr, dpl.data(), dwarp.p(), N, C, H, W);
            Vec<float> duv(NP * 2);
            grid_sample_fwd_bwd<float>(src[s].p(), uv[s].data(), nullptr, dwarp.p(), duv.data(), nullptr, N, C, H, W, H, W, 1);
            Vec<float> dc(NP * 3), dR((size_t)N * 9), dt((size_t)N * 3);
            project_bwd<float>(coords.data(), c.K, R[s].data(), t[s].data(), duv.data(), dc.data(), dR.data(), dt.data(), N, H, W);
            for (size_t j = 0; j < NP * 3; ++j) dcoords[j] += dc[j];
            for (size_t j = 0; j < dR.size(); ++j) dRacc[s][j] += dR[j];
            for (size_t j = 0; j < dt.size(); ++j) dtacc[s][j] += dt[j];
        }
        Vec<float> ddepth(NP), dd2(NP);
        backproject_bwd<float>(dcoords.data(), c.invK, ddepth.data(), N, H, W);
        disparity_to_depth_bwd<float>(d.p(), ddepth.data(), dd2.data(), NP, c.min_depth, c.max_depth);
        for (size_t j = 0; j < NP; ++j) dd.v[j] += dd2[j];
        ddisp[i].reset(N, 1, d0.h, d0.w);
        if (up) upsample_bilinear_bwd(dd.p(), ddisp[i].p(), N, d0.h, d0.w, H, W); else ddisp[i] = dd;
    }
    if (do_bwd)
        for (int s = 0; s < c.nsrc; ++s)
            composeT_bwd<float>(rvec[s], tvec[s], c.source_ids[s] < c.target_id, dRacc[s].data(), dtacc[s].data(), drvec[s], dtvec[s], N);
    return loss / (float)c.nscales;
}

// ------------------------------------------------------------------------------------------------
// Model forward (src/model.jl:8-20) + train_loss (src/training.jl:21-78) + pullback into N.g.
// ------------------------------------------------------------------------------------------------
struct StepOut { Tensor disp[5]; Vec<float> rvec[2], tvec[2]; float loss = 0; };
inline float model_loss(Net& net, const float* x, const LossCfg& c, const float* auto_loss, bool do_bwd, StepOut& out, bool model_only = false) {
    const int N = c.N, L = c.L;
    Tensor xf(N * L, c.C, c.H, c.W); std::memcpy(xf.p(), x, sizeof(float) * xf.size());   // reshape (w,h,c,l*n): frame fastest
    auto enc = std::make_unique<EncSave>(); encoder_fwd(net, xf, *enc);
    Tensor tf[5]; for (int i = 0; i < 5; ++i) tf[i] = t_frame(enc->feat[i], L, c.target_id - 1);
    auto dep = std::make_unique<DepthSave>(); depth_fwd(net, tf, *dep, out.disp);
    PoseSave ps[2];
    for (int s = 0; s < c.nsrc; ++s) {                       // _get_pose_features, src/model.jl:30-35
        const int sid = c.source_ids[s];
        ps[s].order[0] = sid < c.target_id ? sid : c.target_id; ps[s].order[1] = sid < c.target_id ? c.target_id : sid;
        out.rvec[s].assign((size_t)N * 3, 0); out.tvec[s].assign((size_t)N * 3, 0);
        pose_fwd(net, t_frame(enc->feat[4], L, ps[s].order[0] - 1), t_frame(enc->feat[4], L, ps[s].order[1] - 1), ps[s], out.rvec[s].data(), out.tvec[s].data());
    }
    if (model_only) return 0.f;
    const float* rv[2] = {out.rvec[0].data(), out.rvec[1].data()}; const float* tv[2] = {out.tvec[0].data(), out.tvec[1].data()};
    Tensor ddisp[5]; Vec<float> drv[2], dtv[2]; float* drp[2]; float* dtp[2];
    for (int s = 0; s < 2; ++s) { drv[s].assign((size_t)N * 3, 0); dtv[s].assign((size_t)N * 3, 0); drp[s] = drv[s].data(); dtp[s] = dtv[s].data(); }
    out.loss = loss_fwd_bwd(x, c, out.disp, rv, tv, auto_loss, do_bwd, ddisp, drp, dtp, nullptr, nullptr);
    if (!do_bwd) return out.loss;
    std::fill(net.g.begin(), net.g.end(), 0.f);
    Tensor dfeat[5]; for (int i = 0; i < 5; ++i) dfeat[i].reset(enc->feat[i].n, enc->feat[i].c, enc->feat[i].h, enc->feat[i].w);
    Tensor dtf[5]; depth_bwd(net, *dep, ddisp, dtf);
    for (int i = 0; i < 5; ++i) if (dtf[i].size()) t_frame_add(dfeat[i], dtf[i], L, c.target_id - 1);
    for (int s = 0; s < c.nsrc; ++s) {
        Tensor df0, df1; pose_bwd(net, ps[s], drp[s], dtp[s], df0, df1);
        t_frame_add(dfeat[4], df0, L, ps[s].order[0] - 1); t_frame_add(dfeat[4], df1, L, ps[s].order[1] - 1);
    }
    encoder_bwd(net, *enc, dfeat);
    return out.loss;
}

// eval_disparity(m, x) = m.depth_decoder(m.encoder(x, Val(:stages))), src/model.jl:28.  x: [N][C][H][W] single frames.
// BatchNorm follows the model's mode like Flux's: batch statistics under trainmode! (train(), src/Monodepth.jl:126), the tracked running statistics
// under testmode! -- which eval_image / eval_video call explicitly (src/Monodepth.jl:209,237) -- see Net::testmode.
inline void eval_disparity(Net& net, const float* x, int N, int C, int H, int W, Tensor disp[5]) {
    Tensor xf(N, C, H, W); std::memcpy(xf.p(), x, sizeof(float) * xf.size());
    auto enc = std::make_unique<EncSave>(); encoder_fwd(net, xf, *enc);
    auto dep = std::make_unique<DepthSave>(); depth_fwd(net, enc->feat, *dep, disp);
}

// Flux.Optimise.ADAM(eta, (0.9, 0.999)), eps = 1e-8 (src/Monodepth.jl:125,170)
inline void adam_step(Net& net, double eta, double grad_scale = 1.0) {
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    net.beta1p *= b1; net.beta2p *= b2;
    const float c1 = (float)(1.0 - net.beta1p), c2 = (float)(1.0 - net.beta2p);
#pragma omp parallel for
    for (size_t i = 0; i < net.np; ++i) {
        const float g = net.g[i] * (float)grad_scale;
        net.m[i] = (float)b1 * net.m[i] + (1.f - (float)b1) * g;
        net.v[i] = (float)b2 * net.v[i] + (1.f - (float)b2) * g * g;
        net.th[i] -= net.m[i] / c1 / (std::sqrt(net.v[i] / c2) + (float)eps) * (float)eta;
    }
}

}  // namespace md2o
