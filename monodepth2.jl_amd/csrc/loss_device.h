// loss_device.h — device helpers shared by loss.hip and loss_train.hip (the fused photometric-loss kernels, gfx950): projection chain, bilinear sampling, SSIM arithmetic,
// row reductions.  Split out in round 6 so that loss_train.hip can be compiled without SLP vectorisation (Makefile) while every other loss kernel keeps the code it had.
#pragma once
#include "md2_internal.h"
#include <algorithm>

namespace md2 {

constexpr int TW = 32, TH = 16;           // tile
constexpr int FW = TW + 2, FH = TH + 2;   // fwd region (halo 1)
constexpr int MAXC = 3;

struct Pose12 { float R[9]; float t[3]; };   // column-major R

// v_rcp_f32 / v_exp_f32 (1 ulp / ~2 ulp): an IEEE fp32 division is ~10 VALU instructions and the fused loss kernels were
// instruction-bound (~20 divisions per pixel and scale); the 1e-7 relative differences are far inside the 1e-4 parity budget.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fexp(float x) { return __expf(x); }

__device__ __forceinline__ void up_src_f(int o, int in, int out, int& i0, int& i1, float& w1) {
    const float ratio = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float s = ratio * (float)o;
    i0 = min((int)floorf(s), in - 1);
    i1 = min(i0 + 1, in - 1);
    w1 = s - (float)i0;
}
// same with the ratio precomputed on the host (identical fp32 value: (in-1)/(out-1) rounded once)
__device__ __forceinline__ void up_src_r(int o, int in, float ratio, int& i0, int& i1, float& w1) {
    const float s = ratio * (float)o;
    i0 = min((int)floorf(s), in - 1);
    i1 = min(i0 + 1, in - 1);
    w1 = s - (float)i0;
}
// disparity of scale `sc` at full-resolution pixel (py,px): identity at full res, else NNlib.upsample_bilinear (src/training.jl:44-46)
__device__ __forceinline__ float disp_at(const float* __restrict__ d, int dh, int dw, int H, int W, int py, int px) {
    if (dh == H && dw == W) return d[(long)py * W + px];
    int h0, h1, w0, w1; float fh, fw;
    const float rh = H > 1 ? (float)(dh - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(dw - 1) / (float)(W - 1) : 0.f;   // block-uniform -> scalar unit
    up_src_r(py, dh, rh, h0, h1, fh); up_src_r(px, dw, rw, w0, w1, fw);
    return (1.f - fh) * ((1.f - fw) * d[h0 * dw + w0] + fw * d[h0 * dw + w1]) + fh * ((1.f - fw) * d[h1 * dw + w0] + fw * d[h1 * dw + w1]);
}
struct Proj { float x, y, mx, my; float P[3]; float c[3]; float denom; };
// disparity_to_depth + Backproject + Project + grid-sample coordinate (src/utils.jl:175-179,63-65,95-99; NNlib unnormalize + :border clamp)
__device__ __forceinline__ void project_px(float d, int py, int px, const float* __restrict__ K, const float* __restrict__ invK, const Pose12& T,
                                           float min_disp, float max_disp, int H, int W, Proj& o) {
    const float depth = frcp(d * (max_disp - min_disp) + min_disp);
    const float cx = (float)(px + 1), cy = (float)(py + 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) o.P[i] = depth * (invK[i] * cx + invK[3 + i] * cy + invK[6 + i] * 1.f);
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = (T.R[i] * o.P[0] + T.R[3 + i] * o.P[1] + T.R[6 + i] * o.P[2]) + T.t[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.c[i] = K[i] * r[0] + K[3 + i] * r[1] + K[6 + i] * r[2];
    o.denom = frcp(o.c[2] + 1e-7f);
    const float rW = 1.f / (float)(W - 1), rH = 1.f / (float)(H - 1);        // uniform: evaluated once on the scalar/constant path
    const float gx = (((o.c[0] * o.denom - 1.f) * rW) - 0.5f) * 2.f;
    const float gy = (((o.c[1] * o.denom - 1.f) * rH) - 0.5f) * 2.f;
    float x = ((gx + 1.f) * 0.5f) * (float)(W - 1), y = ((gy + 1.f) * 0.5f) * (float)(H - 1);
    o.mx = 0.5f * (float)(W - 1); o.my = 0.5f * (float)(H - 1);
    if (!(x > 0.f)) { x = 0.f; o.mx = 0.f; } else if (x >= (float)(W - 1)) { x = (float)(W - 1); o.mx = 0.f; }
    if (!(y > 0.f)) { y = 0.f; o.my = 0.f; } else if (y >= (float)(H - 1)) { y = (float)(H - 1); o.my = 0.f; }
    o.x = x; o.y = y;
}
struct Samp { int i00, i01, i10, i11; float wx0, wx1, wy0, wy1; bool v01, v10, v11; };
__device__ __forceinline__ void sample_setup(float x, float y, int H, int W, Samp& s) {
    const float fx = floorf(x), fy = floorf(y);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    s.wx1 = x - fx; s.wx0 = 1.f - s.wx1; s.wy1 = y - fy; s.wy0 = 1.f - s.wy1;
    const bool vx1 = x1 < W, vy1 = y1 < H;          // x0,y0 always valid after the border clamp
    s.v01 = vx1; s.v10 = vy1; s.v11 = vx1 && vy1;
    s.i00 = y0 * W + x0; s.i01 = y0 * W + (vx1 ? x1 : x0); s.i10 = (vy1 ? y1 : y0) * W + x0; s.i11 = (vy1 ? y1 : y0) * W + (vx1 ? x1 : x0);
}
// Branch-free gather: a corner that falls outside (only possible for x == W-1 / y == H-1 after the border clamp) is clamped
// onto a valid texel by sample_setup and its bilinear weight is exactly 0, so it contributes 0 like NNlib's bounds check.
__device__ __forceinline__ float sample_val(const float* __restrict__ im, const Samp& s) {
    const float v00 = im[s.i00], v01 = im[s.i01], v10 = im[s.i10], v11 = im[s.i11];
    return v00 * s.wy0 * s.wx0 + v01 * s.wy0 * s.wx1 + v10 * s.wy1 * s.wx0 + v11 * s.wy1 * s.wx1;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct SsimOut { float val; float g_mx, g_pxx, g_pxy; };   // SSIM value and d val / d(pool x, pool x^2, pool xy) (0 outside the clamp)
// SSIM functor arithmetic (src/utils.jl:28-38) from the nine-sample window sums
__device__ __forceinline__ void ssim_from_sums(float sx, float sy, float sxx, float syy, float sxy, bool want_grad, SsimOut& o) {
    const float r9 = 1.f / 9.f;
    const float mx = sx * r9, my = sy * r9, pxx = sxx * r9, pyy = syy * r9, pxy = sxy * r9;
    const float c1 = 0.0001f, c2 = 0.0009f;
    const float vx = pxx - mx * mx, vy = pyy - my * my, vxy = pxy - mx * my;
    const float n1 = 2.f * mx * my + c1, n2 = 2.f * vxy + c2, d1 = mx * mx + my * my + c1, d2 = vx + vy + c2;
    const float sn = n1 * n2, sd = d1 * d2, rsd = frcp(sd);
    const float v = (1.f - sn * rsd) * 0.5f;
    o.val = fminf(fmaxf(v, 0.f), 1.f);
    if (want_grad) {
        const float g = (v >= 0.f && v <= 1.f) ? 1.f : 0.f;
        const float gsn = -0.5f * g * rsd, gsd = 0.5f * g * sn * rsd * rsd;
        const float gn1 = gsn * n2, gn2 = gsn * n1, gd1 = gsd * d2, gd2 = gsd * d1;
        o.g_pxx = gd2; o.g_pxy = 2.f * gn2;
        o.g_mx = gn1 * 2.f * my + gd1 * 2.f * mx - gd2 * 2.f * mx - 2.f * gn2 * my;
    }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
// forward-only projection: c = K (R P + t) = (K R) P + K t, P = depth * ray(px,py): 12 scalars per source instead of K, R, t
struct ProjF { float M[9]; float v[3]; };       // M = K*R (column-major), v = K*t
__device__ __forceinline__ void project_fwd(float depth, const float ray[3], const ProjF& T, int H, int W, float& x, float& y) {
    const float P0 = depth * ray[0], P1 = depth * ray[1], P2 = depth * ray[2];
    const float c0 = T.M[0] * P0 + T.M[3] * P1 + T.M[6] * P2 + T.v[0];
    const float c1 = T.M[1] * P0 + T.M[4] * P1 + T.M[7] * P2 + T.v[1];
    const float c2 = T.M[2] * P0 + T.M[5] * P1 + T.M[8] * P2 + T.v[2];
    const float denom = frcp(c2 + 1e-7f);
    const float rW = 1.f / (float)(W - 1), rH = 1.f / (float)(H - 1);
    const float gx = (((c0 * denom - 1.f) * rW) - 0.5f) * 2.f, gy = (((c1 * denom - 1.f) * rH) - 0.5f) * 2.f;
    x = ((gx + 1.f) * 0.5f) * (float)(W - 1); y = ((gy + 1.f) * 0.5f) * (float)(H - 1);
    x = !(x > 0.f) ? 0.f : fminf(x, (float)(W - 1));       // :border clamp (NaN -> 0)
    y = !(y > 0.f) ? 0.f : fminf(y, (float)(H - 1));
}


template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
// sum over each row of 16 lanes, result in every lane of the row: xor-1 / xor-2 quad permutes, then mirror within 8 and within 16 (4 v_add_f32_dpp)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return v;
}

}  // namespace md2
