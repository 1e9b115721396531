// loss.hip — fused photometric loss of the self-supervised step (gfx950), forward and backward.
//
// Replaces the ~400 small kernels of src/training.jl:42-75 (SURVEY.md §2a k5,k10-k18): per scale
//   upsample_bilinear(disp) -> disparity_to_depth -> Backproject -> Project (x2 sources) -> grid_sample(:border)
//   -> SSIM (pad_reflect + 5 MeanPools) + L1 -> minimum over sources [-> automask] -> mean ; + smooth_loss
// with ONE launch for all 4 scales (grid.z = scale*N + n).  Nothing is materialised in HBM: a 32x16 pixel tile
// (+1 px halo, reflect-mapped like pad_reflect) of both warped sources and the target is built in LDS, the 3x3
// window statistics are read from LDS, and only per-(scale,sample) partial sums leave the block.
// Algorithmic HBM bytes per sample per scale (fwd): 4P(1+3C).
// Training uses loss_train_kernel instead: the same tile (+2 px halo) built once, loss value AND pullback out of it -- d loss / d disparity
// (4P per scale, the coarse scales already resized to their own resolution) plus 24 pose-gradient partials per block (DPP + LDS reduction, no per-pixel atomics).
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

template <int C>
__global__ __launch_bounds__(256) void loss_fwd_kernel(LossArgs a, double* __restrict__ part, double* __restrict__ blk) {
    __shared__ float wv[2][C][FH * FW];      // warped sources (padded positions)
    __shared__ float tg[C][FH * FW];         // target
    __shared__ float dl[FH * FW];            // disparity
    __shared__ float red[4][4];
    const int zi = blockIdx.z, sc = zi / a.N, n = zi % a.N;
    const int H = a.H, W = a.W;
    const long P = (long)H * W;
    const int ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const float* xn = a.x + (long)n * 3 * C * P;
    const float* tgt = xn + (long)a.tgt * C * P;
    const float* srcs[2] = {xn + (long)a.src0 * C * P, xn + (long)a.src1 * C * P};
    const int dh = a.dh[sc], dw = a.dw[sc];
    const float* dsp = a.disp[sc] + (long)n * dh * dw;
    ProjF T[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* rt = a.Rt + ((long)s * a.N + n) * 12;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) T[s].M[j * 3 + i] = a.K[i] * rt[j * 3] + a.K[3 + i] * rt[j * 3 + 1] + a.K[6 + i] * rt[j * 3 + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) T[s].v[i] = a.K[i] * rt[9] + a.K[3 + i] * rt[10] + a.K[6 + i] * rt[11];
    }
    // ---- phase 1: warped / target / disparity at the padded positions of tile + 1
    for (int i = threadIdx.x; i < FH * FW; i += 256) {
        const int ry = i / FW, rx = i - ry * FW;
        const int py = refl_idx(min(ty0 - 1 + ry, H), H), px = refl_idx(min(tx0 - 1 + rx, W), W);   // positions past the padded image are never used
        const float d = disp_at(dsp, dh, dw, H, W, py, px);
        dl[i] = d;
        const float depth = frcp(d * (a.max_disp - a.min_disp) + a.min_disp);
        const float cx = (float)(px + 1), cy = (float)(py + 1);
        const float ray[3] = {a.invK[0] * cx + a.invK[3] * cy + a.invK[6], a.invK[1] * cx + a.invK[4] * cy + a.invK[7], a.invK[2] * cx + a.invK[5] * cy + a.invK[8]};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float x, y; project_fwd(depth, ray, T[s], H, W, x, y);
            Samp sm; sample_setup(x, y, H, W, sm);
#pragma unroll
            for (int c = 0; c < C; ++c) wv[s][c][i] = sample_val(srcs[s] + (long)c * P, sm);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) tg[c][i] = tgt[(long)c * P + (long)py * W + px];
    }
    __syncthreads();
    // ---- phase 2: per pixel SSIM + L1 -> photometric -> min over sources [-> automask]; smoothness terms
    float s_photo = 0.f, s_d = 0.f, s_x = 0.f, s_y = 0.f;
    const float alpha = 0.85f, beta = 1.f - alpha, invC = 1.f / (float)C;
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW, py = ty0 + ly, px = tx0 + lx;
        if (py >= H || px >= W) continue;
        const int ctr = (ly + 1) * FW + lx + 1;
        float pl[2] = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float yw[9], sy = 0.f, syy = 0.f;                  // target window: shared by both sources
#pragma unroll
            for (int k = 0; k < 9; ++k) { yw[k] = tg[c][ctr + (k / 3 - 1) * FW + (k % 3 - 1)]; sy += yw[k]; syy += yw[k] * yw[k]; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float sx = 0.f, sxx = 0.f, sxy = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) { const float xv = wv[s][c][ctr + (k / 3 - 1) * FW + (k % 3 - 1)]; sx += xv; sxx += xv * xv; sxy += xv * yw[k]; }
                SsimOut so; ssim_from_sums(sx, sy, sxx, syy, sxy, false, so);
                pl[s] += alpha * invC * so.val + beta * invC * fabsf(yw[4] - wv[s][c][ctr]);
            }
        }
        float wl = pl[1] < pl[0] ? pl[1] : pl[0];
        if (a.auto_loss) { const float al = a.auto_loss[(long)n * P + (long)py * W + px]; if (!(wl < al)) wl = al; }
        s_photo += wl;
        const float d = dl[ctr];
        s_d += d;
        if (px + 1 < W) { float gi = 0.f; for (int c = 0; c < C; ++c) gi += fabsf(tg[c][ctr] - tg[c][ctr + 1]); if (C > 1) gi *= invC; s_x += fabsf(d - dl[ctr + 1]) * fexp(-gi); }
        if (py + 1 < H) { float gi = 0.f; for (int c = 0; c < C; ++c) gi += fabsf(tg[c][ctr] - tg[c][ctr + FW]); if (C > 1) gi *= invC; s_y += fabsf(d - dl[ctr + FW]) * fexp(-gi); }
    }
    s_photo = wave_sum(s_photo); s_d = wave_sum(s_d); s_x = wave_sum(s_x); s_y = wave_sum(s_y);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = s_photo; red[wave][1] = s_d; red[wave][2] = s_x; red[wave][3] = s_y; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double v = (double)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (blk) blk[((long)zi * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = v;     // deterministic mode: summed in order afterwards
        else atomicAdd(part + (long)zi * 4 + threadIdx.x, v);
    }
}
// deterministic mode: out[g*nv + k] = sum over the nb blocks of group g, in block order (one thread per value)
__global__ void loss_part_reduce_kernel(const double* __restrict__ blk, double* __restrict__ out, int groups, int nb, int nv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= groups * nv) return;
    const int g = i / nv, k = i - g * nv;
    double s = 0;
    for (int b = 0; b < nb; ++b) s += blk[((long)g * nb + b) * nv + k];
    out[i] = s;
}
int loss_blocks_per_image(const LossArgs& a) { return ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH); }
void launch_loss_fwd(const LossArgs& a, double* part, double* blk, hipStream_t s) {
    if ((a.C != 1 && a.C != 3) || a.H < 2 || a.W < 2) { fprintf(stderr, "md2: loss_fwd: unsupported C=%d H=%d W=%d\n", a.C, a.H, a.W); abort(); }
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, a.N * a.nscales);
    if (a.C == 1) hipLaunchKernelGGL(loss_fwd_kernel<1>, grid, dim3(256), 0, s, a, part, blk);
    else hipLaunchKernelGGL(loss_fwd_kernel<3>, grid, dim3(256), 0, s, a, part, blk);
    if (blk) { const int groups = a.N * a.nscales; hipLaunchKernelGGL(loss_part_reduce_kernel, dim3((groups * 4 + 63) / 64), dim3(64), 0, s, blk, part, groups, (int)(grid.x * grid.y), 4); }
}

// totals[(sc*N+n)*4 + {0: photo sum, 1: mean d, 2: Sx, 3: Sy}] ; loss = mean over scales of [mean(warp_loss) + smooth]
__global__ void loss_finalize_kernel(LossArgs a, const double* __restrict__ part, float* __restrict__ totals, float* __restrict__ loss) {
    // one lane per (scale, sample), then a wave sum in double (<= 64 entries per pass; the order of the sum is fixed)
    const double P = (double)a.H * a.W, cntx = (double)(a.W - 1) * a.H * a.N, cnty = (double)a.W * (a.H - 1) * a.N;
    double total = 0;
    for (int i = threadIdx.x; i < a.nscales * a.N; i += 64) {
        const int sc = i / a.N;
        const double* p = part + (long)i * 4;
        const float mean_d = (float)(p[1] / P);
        float* t = totals + (long)i * 4;
        t[0] = (float)p[0]; t[1] = mean_d; t[2] = (float)p[2]; t[3] = (float)p[3];
        total += p[0] / (P * a.N) + (p[2] / cntx + p[3] / cnty) / (a.normalize ? (double)(mean_d + 1e-7f) : 1.0) * (double)a.smooth_w[sc];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if (threadIdx.x == 0) *loss = (float)(total / a.nscales);
}
void launch_loss_finalize(const LossArgs& a, const double* part, float* totals, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, a, part, totals, loss);
}

// =====================================================================================================================
// training step: loss value AND pullback from one pass over the warp (md2_train_loss / md2_loss_fwd_bwd with gradients)
// =====================================================================================================================
// The smoothness pullback is normalised by per-(scale, sample) sums of the whole image (mean disparity, Sx, Sy), so those come first from a
// cheap pass that touches only the disparities and the target frame (no warp); the main kernel then builds the warped tile ONCE and emits the
// photometric sum (for the loss value), d loss / d disparity and the pose-gradient partials.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
// sum over each row of 16 lanes, result in every lane of the row: xor-1 / xor-2 quad permutes, then mirror within 8 and within 16 (4 v_add_f32_dpp)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return v;
}

// part[(sc*N+n)*4 + {1: sum d, 2: Sx, 3: Sy}]: 32x32 pixel tiles, 4 pixels per thread, the scales in a loop (the edge weights exp(-|dI|) are computed once;
// the upsampled disparities of tile + 1 go through LDS so that every position is interpolated once)
template <int C>
__global__ __launch_bounds__(256) void loss_smooth_sums_kernel(LossArgs a, double* __restrict__ part, double* __restrict__ blk) {
    __shared__ float dt[33 * 33];
    __shared__ float red[16][3 * MD2_MAX_SCALES];
    const int n = blockIdx.z, H = a.H, W = a.W;
    const long P = (long)H * W;
    const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5, tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32, px = tx0 + lx;
    const float* tgt = a.x + ((long)n * 3 + a.tgt) * C * P;
    const float invC = 1.f / (float)C;
    float v[3 * MD2_MAX_SCALES], ex[4], ey[4];
#pragma unroll
    for (int k = 0; k < 3 * MD2_MAX_SCALES; ++k) v[k] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int py = ty0 + ly0 + r * 8;
        ex[r] = 0.f; ey[r] = 0.f;                    // 0 where the pixel or its neighbour does not exist
        if (px >= W || py >= H) continue;
        const long p = (long)py * W + px;
        if (px + 1 < W) { float gi = 0.f; for (int c = 0; c < C; ++c) gi += fabsf(tgt[c * P + p] - tgt[c * P + p + 1]); if (C > 1) gi *= invC; ex[r] = fexp(-gi); }
        if (py + 1 < H) { float gi = 0.f; for (int c = 0; c < C; ++c) gi += fabsf(tgt[c * P + p] - tgt[c * P + p + W]); if (C > 1) gi *= invC; ey[r] = fexp(-gi); }
    }
#pragma unroll
    for (int sc = 0; sc < MD2_MAX_SCALES; ++sc) {
        if (sc >= a.nscales) break;
        const int dh = a.dh[sc], dw = a.dw[sc];
        const float* dsp = a.disp[sc] + (long)n * dh * dw;
        if (sc) __syncthreads();
        for (int i = threadIdx.x; i < 33 * 33; i += 256) { const int ry = i / 33, rx = i - ry * 33; dt[i] = disp_at(dsp, dh, dw, H, W, min(ty0 + ry, H - 1), min(tx0 + rx, W - 1)); }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ly = ly0 + r * 8;
            if (px >= W || ty0 + ly >= H) continue;
            const float d = dt[ly * 33 + lx];
            v[sc * 3] += d;
            v[sc * 3 + 1] += fabsf(d - dt[ly * 33 + lx + 1]) * ex[r];       // ex / ey are 0 on the last column / row
            v[sc * 3 + 2] += fabsf(d - dt[(ly + 1) * 33 + lx]) * ey[r];
        }
    }
    const int row = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < 3 * MD2_MAX_SCALES; ++k) { const float r = row16_sum(v[k]); if ((threadIdx.x & 15) == 0) red[row][k] = r; }
    __syncthreads();
    if (threadIdx.x < 3 * a.nscales) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
        const int sc = threadIdx.x / 3, k = threadIdx.x - sc * 3;
        if (blk) blk[((long)n * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * (3 * MD2_MAX_SCALES) + threadIdx.x] = (double)t;
        else atomicAdd(part + ((long)sc * a.N + n) * 4 + 1 + k, (double)t);
    }
}
// deterministic mode: the sums above in block order
__global__ void loss_smooth_reduce_kernel(const double* __restrict__ blk, double* __restrict__ part, int N, int nscales, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= N * nscales * 3) return;
    const int k = i % 3, sc = (i / 3) % nscales, n = i / (3 * nscales);
    double s = 0;
    for (int b = 0; b < nb; ++b) s += blk[((long)n * nb + b) * (3 * MD2_MAX_SCALES) + sc * 3 + k];
    part[((long)sc * N + n) * 4 + 1 + k] = s;
}

struct LossTrainOut { double* part; float* dd[MD2_MAX_SCALES]; float* dlo[MD2_MAX_SCALES]; double* dRt_acc; double* blk; };
// One block = `tpb` consecutive TW x TH tiles of one (scale, sample).  Per tile:
//  1. warped sources / target / disparity at the padded positions of tile + 2 -> LDS                           (the only global gathers of the forward)
//  2. window owners q in tile + 1, three per thread in a column (row sums of the five window moments shared between them):
//     photometric of both sources, arg-min [, automask], SSIM pullback coefficients -> LDS; owners inside the tile add to the loss sum
//  3. per tile pixel r, TH/8 per thread in a column: d loss / d warped_s[r] = box sum of the winning owners' coefficients (row sums shared
//     down the column; pad_reflect aliases are weights 2 on the affected row / column), then the pullback through grid_sample, Project,
//     Backproject and disparity_to_depth, plus the smoothness pullback.
// The 24 pose-gradient partials and the loss sum stay in registers across the block's tiles and are reduced once.
#ifndef MD2_LOSS_HOISTED_PROJECTION
#define MD2_LOSS_HOISTED_PROJECTION 0      // 1: sampling coordinates from the hoisted (K R) P + K t (12 scalars per source, ~4 % faster) -- moves them by ~1e-5 px against
#endif                                     // the reference's K (R P + t), which on real texture flips a few more arg-min decisions than the oracle comparison likes
#ifndef MD2_LOSS_WAVES
#define MD2_LOSS_WAVES 2
#endif
template <int C, int TH>
__global__ __launch_bounds__(256, MD2_LOSS_WAVES) void loss_train_kernel(LossArgs a, LossTrainOut o, int tpb, int ntx) {
    constexpr int OSH = C == 1 ? 3 : 2;          // window owners per thread in phase 2 (three colour channels keep 2 x 3 x OSH coefficient triples alive: two owners fit the registers)
    constexpr int VW = TW + 4, VH = TH + 4, OW = TW + 2, OH = TH + 2, SH = TH / 8, NSR = (OH + OSH - 1) / OSH;
    __shared__ float wv[2][C][VH * VW];          // warped sources at the padded positions of tile + 2
    __shared__ float tg[C][VH * VW];
    __shared__ float dl[VH * VW];
    __shared__ float coef[C][3][OH * OW];        // d photometric_{s*} / d (pool x, pool x^2, pool xy) at q in tile + 1, pre-scaled
    __shared__ signed char amin[OH * OW];        // winning source at q, -1: none (outside / masked)
    __shared__ float red[16][25];
    constexpr int CW = TW / 2 + 2, CH = TH / 2 + 2;
    __shared__ float crow[CH * TW];              // coarse scales, resize adjoint of the tile's d loss / d disparity: after the row direction [coarse row][tile column] ...
    __shared__ float cacc[CH * CW];              // ... and after the column direction [coarse row][coarse column]
    const int zi = blockIdx.z, sc = zi / a.N, n = zi % a.N;
    const int H = a.H, W = a.W;
    const long P = (long)H * W;
    const int ty0 = blockIdx.y * TH;
    const float* xn = a.x + (long)n * 3 * C * P;
    const float* tgt = xn + (long)a.tgt * C * P;
    const float* srcs[2] = {xn + (long)a.src0 * C * P, xn + (long)a.src1 * C * P};
    const int dh = a.dh[sc], dw = a.dw[sc];
    const float* dsp = a.disp[sc] + (long)n * dh * dw;
    float* dd = o.dd[sc] + (long)n * P;
    float* dlo = o.dlo[sc] ? o.dlo[sc] + (long)n * dh * dw : nullptr;
    const float rh = H > 1 ? (float)(dh - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(dw - 1) / (float)(W - 1) : 0.f;
    Pose12 T[2]; ProjF TF[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* rt = a.Rt + ((long)s * a.N + n) * 12;
#pragma unroll
        for (int i = 0; i < 9; ++i) T[s].R[i] = rt[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) T[s].t[i] = rt[9 + i];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) TF[s].M[j * 3 + i] = a.K[i] * rt[j * 3] + a.K[3 + i] * rt[j * 3 + 1] + a.K[6 + i] * rt[j * 3 + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) TF[s].v[i] = a.K[i] * rt[9] + a.K[3 + i] * rt[10] + a.K[6 + i] * rt[11];
    }
    const float alpha = 0.85f, beta = 1.f - alpha, invC = 1.f / (float)C;
    const float wpix = 1.f / ((float)a.nscales * (float)a.N * (float)P);      // d loss / d warp_loss[q]
    // smoothness normalisation from the sums of the first pass (same arithmetic as loss_finalize_kernel)
    const double* pt = o.part + (long)zi * 4;
    const double cntx = (double)(W - 1) * H * a.N, cnty = (double)W * (H - 1) * a.N;
    const float mean_d = (float)(pt[1] / (double)P);
    const float den = a.normalize ? mean_d + 1e-7f : 1.f;
    const float sw = a.smooth_w[sc] / (float)a.nscales;
    const float ax = sw / den / (float)cntx, ay = sw / den / (float)cnty;
    const float corr = a.normalize ? sw * (float)((double)(float)pt[2] / cntx + (double)(float)pt[3] / cnty) / (den * den) / (float)P : 0.f;
    float acc[24], s_photo = 0.f;
#pragma unroll
    for (int k = 0; k < 24; ++k) acc[k] = 0.f;
    if constexpr (C > 1) {                               // three channels: the 24 pose sums live only inside phase 3 -- per tile they are row-reduced into `red` (held across phases 1-2 they went to scratch)
        for (int i = threadIdx.x; i < 16 * 25; i += 256) (&red[0][0])[i] = 0.f;
    }

    for (int t = 0; t < tpb; ++t) {
        int tid = threadIdx.x;
        if (C > 1) asm volatile("" : "+v"(tid));      // three channels: keeps hipcc from hoisting the per-thread tile indices of every phase out of the tile loop (they lived in scratch: 56 spilled registers)
        const int tix = blockIdx.x * tpb + t;
        if (tix >= ntx) break;
        const int tx0 = tix * TW;
        if (t) __syncthreads();                  // the previous tile's phases 3 / 4 still read the LDS tiles
        int ci_lo = 0, cj_lo = 0;
        if (dlo) {
            int i1; float f;
            up_src_r(ty0, dh, rh, ci_lo, i1, f); up_src_r(tx0, dw, rw, cj_lo, i1, f);                 // first coarse row / column this tile touches
            for (int i = tid; i < CH * TW; i += 256) crow[i] = 0.f;
            for (int i = tid; i < CH * CW; i += 256) cacc[i] = 0.f;
        }
        // ---- phase 1: values at padded positions u in tile + 2 (u clamped into the padded domain [-1, size])
        for (int i = tid; i < VH * VW; i += 256) {
            const int ry = i / VW, rx = i - ry * VW;
            const int uy = max(-1, min(ty0 - 2 + ry, H)), ux = max(-1, min(tx0 - 2 + rx, W));
            const int py = refl_idx(uy, H), px = refl_idx(ux, W);
            const float d = disp_at(dsp, dh, dw, H, W, py, px);
            dl[i] = d;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#if MD2_LOSS_HOISTED_PROJECTION
                const float depth = frcp(d * (a.max_disp - a.min_disp) + a.min_disp);
                const float cx = (float)(px + 1), cy = (float)(py + 1);
                const float ray[3] = {a.invK[0] * cx + a.invK[3] * cy + a.invK[6], a.invK[1] * cx + a.invK[4] * cy + a.invK[7], a.invK[2] * cx + a.invK[5] * cy + a.invK[8]};
                float x, y; project_fwd(depth, ray, TF[s], H, W, x, y);
                Samp sm; sample_setup(x, y, H, W, sm);
#else
                Proj pr; project_px(d, py, px, a.K, a.invK, T[s], a.min_disp, a.max_disp, H, W, pr);      // the reference's order of operations: K (R P + t)
                Samp sm; sample_setup(pr.x, pr.y, H, W, sm);
#endif
#pragma unroll
                for (int c = 0; c < C; ++c) wv[s][c][i] = sample_val(srcs[s] + (long)c * P, sm);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) tg[c][i] = tgt[(long)c * P + (long)py * W + px];
        }
        __syncthreads();
        // ---- phase 2: window owners q in tile + 1, OSH vertically adjacent ones per thread
#pragma unroll 1
        for (int st = tid; st < OW * NSR; st += 256) {
            const int k3 = st / OW, rx = st - k3 * OW, ry0 = k3 * OSH;
            float pl[OSH][2], gmx[OSH][2][C], gpxx[OSH][2][C], gpxy[OSH][2][C];
#pragma unroll
            for (int oo = 0; oo < OSH; ++oo) { pl[oo][0] = 0.f; pl[oo][1] = 0.f; }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (C > 1 && c) __builtin_amdgcn_sched_barrier(0);      // one channel's window rows at a time: hoisting all three channels' LDS reads pushed the pose sums into scratch
                float Ry[OSH + 2], Ryy[OSH + 2], yc[OSH + 2], Rx[2][OSH + 2], Rxx[2][OSH + 2], Rxy[2][OSH + 2], xc[2][OSH + 2];
#pragma unroll
                for (int j = 0; j < OSH + 2; ++j) {
                    const int b = min(ry0 + j, VH - 1) * VW + rx;       // rows past the value region belong to owners that are never stored
                    const float y0 = tg[c][b], y1 = tg[c][b + 1], y2 = tg[c][b + 2];
                    Ry[j] = y0 + y1 + y2; Ryy[j] = y0 * y0 + y1 * y1 + y2 * y2; yc[j] = y1;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float x0 = wv[s][c][b], x1 = wv[s][c][b + 1], x2 = wv[s][c][b + 2];
                        Rx[s][j] = x0 + x1 + x2; Rxx[s][j] = x0 * x0 + x1 * x1 + x2 * x2; Rxy[s][j] = x0 * y0 + x1 * y1 + x2 * y2; xc[s][j] = x1;
                    }
                }
#pragma unroll
                for (int oo = 0; oo < OSH; ++oo) {
                    const float sy = Ry[oo] + Ry[oo + 1] + Ry[oo + 2], syy = Ryy[oo] + Ryy[oo + 1] + Ryy[oo + 2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        SsimOut so;
                        ssim_from_sums(Rx[s][oo] + Rx[s][oo + 1] + Rx[s][oo + 2], sy, Rxx[s][oo] + Rxx[s][oo + 1] + Rxx[s][oo + 2], syy, Rxy[s][oo] + Rxy[s][oo + 1] + Rxy[s][oo + 2], true, so);
                        pl[oo][s] += alpha * invC * so.val + beta * invC * fabsf(yc[oo + 1] - xc[s][oo + 1]);
                        gmx[oo][s][c] = so.g_mx; gpxx[oo][s][c] = so.g_pxx; gpxy[oo][s][c] = so.g_pxy;
                    }
                }
            }
#pragma unroll
            for (int oo = 0; oo < OSH; ++oo) {
                const int ry = ry0 + oo;
                if (ry >= OH) break;
                const int qi = ry * OW + rx, qy = ty0 - 1 + ry, qx = tx0 - 1 + rx;
                signed char win = -1;
                float m = 0.f;
                if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
                    win = pl[oo][1] < pl[oo][0] ? 1 : 0;                                       // first arg-min (Zygote `minimum`)
                    float wl = win ? pl[oo][1] : pl[oo][0];
                    if (a.auto_loss) { const float al = a.auto_loss[(long)n * P + (long)qy * W + qx]; if (!(wl < al)) { wl = al; win = -1; } }   // mask wins ties -> no gradient
                    if (ry >= 1 && ry <= TH && rx >= 1 && rx <= TW) s_photo += wl;          // owners of this tile (the ring belongs to the neighbours)
                    m = win < 0 ? 0.f : wpix * alpha * invC * (1.f / 9.f);                     // through mean_c, alpha, and the /9 of MeanPool
                }
                const bool w1 = win == 1;
#pragma unroll
                for (int c = 0; c < C; ++c) {                                                  // select by value: a runtime index would put the arrays in scratch
                    coef[c][0][qi] = m * (w1 ? gmx[oo][1][c] : gmx[oo][0][c]);
                    coef[c][1][qi] = m * (w1 ? gpxx[oo][1][c] : gpxx[oo][0][c]);
                    coef[c][2][qi] = m * (w1 ? gpxy[oo][1][c] : gpxy[oo][0][c]);
                }
                amin[qi] = win;
            }
        }
        __syncthreads();
        // ---- phase 3: tile pixels, SH per thread down a column
        {
            const int lx = tid & 31, ly0 = (tid >> 5) * SH, px = tx0 + lx;
            const float wxm = px == 1 ? 2.f : 1.f, wxp = px == W - 2 ? 2.f : 1.f;          // pad_reflect: column px-1 / px+1 is seen a second time through the padded column
            float rs[3][2][C][3];                                                          // ring of owner-row sums: [row][source][channel][coefficient]
            auto rowsum = [&](int orow, float (&r)[2][C][3]) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int c = 0; c < C; ++c) { r[s][c][0] = 0.f; r[s][c][1] = 0.f; r[s][c][2] = 0.f; }
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int qi = orow * OW + lx + 1 + dx;
                    const int am = amin[qi];
                    const float w = dx < 0 ? wxm : (dx > 0 ? wxp : 1.f);
                    const float m0 = am == 0 ? w : 0.f, m1 = am == 1 ? w : 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int k = 0; k < 3; ++k) { const float cv = coef[c][k][qi]; r[0][c][k] += m0 * cv; r[1][c][k] += m1 * cv; }
                }
            };
            float gwp[C > 1 ? SH : 1][2][C];                                              // three channels: d loss / d warped_s[r][c] of the thread's SH pixels, one channel at a time
            if constexpr (C > 1) {                                                        // (the ring of all channels' row sums is 54 registers; per channel it is (SH + 2) x 6)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (c) __builtin_amdgcn_sched_barrier(0);
                    float r1[SH + 2][2][3];
#pragma unroll
                    for (int r = 0; r < SH + 2; ++r) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) { r1[r][0][k] = 0.f; r1[r][1][k] = 0.f; }
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int qi = (ly0 + r) * OW + lx + 1 + dx;
                            const int am = amin[qi];
                            const float w = dx < 0 ? wxm : (dx > 0 ? wxp : 1.f);
                            const float m0 = am == 0 ? w : 0.f, m1 = am == 1 ? w : 0.f;
#pragma unroll
                            for (int k = 0; k < 3; ++k) { const float cv = coef[c][k][qi]; r1[r][0][k] += m0 * cv; r1[r][1][k] += m1 * cv; }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < SH; ++i) {
                        const int py = ty0 + ly0 + i, cb = (ly0 + i + 2) * VW + lx + 2;
                        const float wym = py == 1 ? 2.f : 1.f, wyp = py == H - 2 ? 2.f : 1.f;
                        const int rwin = amin[(ly0 + i + 1) * OW + lx + 1];
                        const float xv[2] = {wv[0][c][cb], wv[1][c][cb]}, tv = tg[c][cb];
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const float g0 = wym * r1[i][s][0] + r1[i + 1][s][0] + wyp * r1[i + 2][s][0];
                            const float g1 = wym * r1[i][s][1] + r1[i + 1][s][1] + wyp * r1[i + 2][s][1];
                            const float g2 = wym * r1[i][s][2] + r1[i + 1][s][2] + wyp * r1[i + 2][s][2];
                            float g = g0 + 2.f * xv[s] * g1 + tv * g2;
                            if (rwin == s) { const float df = tv - xv[s]; g += -(df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * beta * invC * wpix; }
                            gwp[i][s][c] = g;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            } else { rowsum(ly0, rs[0]); rowsum(ly0 + 1, rs[1]); }                          // owner rows of pixel rows ly0-1 and ly0
#pragma unroll
            for (int i = 0; i < SH; ++i) {
                if constexpr (C == 1) rowsum(ly0 + i + 2, rs[(i + 2) % 3]);
                const int ly = ly0 + i, py = ty0 + ly;
                if (py >= H || px >= W) continue;
                const float wym = py == 1 ? 2.f : 1.f, wyp = py == H - 2 ? 2.f : 1.f;
                const int cb = (ly + 2) * VW + lx + 2;          // r in the value region
                const float d = dl[cb];
                float gd = 0.f;                                   // d loss / d disparity[r]
                const int rwin = amin[(ly + 1) * OW + lx + 1];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float gw[C];
                    bool any = false;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float g;
                        if constexpr (C > 1) g = gwp[i][s][c];
                        else {
                            const float g0 = wym * rs[i % 3][s][c][0] + rs[(i + 1) % 3][s][c][0] + wyp * rs[(i + 2) % 3][s][c][0];
                            const float g1 = wym * rs[i % 3][s][c][1] + rs[(i + 1) % 3][s][c][1] + wyp * rs[(i + 2) % 3][s][c][1];
                            const float g2 = wym * rs[i % 3][s][c][2] + rs[(i + 1) % 3][s][c][2] + wyp * rs[(i + 2) % 3][s][c][2];
                            g = g0 + 2.f * wv[s][c][cb] * g1 + tg[c][cb] * g2;
                            if (rwin == s) { const float df = tg[c][cb] - wv[s][c][cb]; g += -(df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * beta * invC * wpix; }
                        }
                        gw[c] = g; any = any || g != 0.f;
                    }
                    if (!any) continue;
                    // grid_sample pullback wrt the sampling coordinate
                    Proj pr; project_px(d, py, px, a.K, a.invK, T[s], a.min_disp, a.max_disp, H, W, pr);
                    Samp sm; sample_setup(pr.x, pr.y, H, W, sm);
                    float ggx = 0.f, ggy = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float* im = srcs[s] + (long)c * P;
                        const float v00 = im[sm.i00], v01 = sm.v01 ? im[sm.i01] : 0.f, v10 = sm.v10 ? im[sm.i10] : 0.f, v11 = sm.v11 ? im[sm.i11] : 0.f;   // NNlib: outside corners read as 0 in the coordinate pullback
                        ggx += gw[c] * ((v01 - v00) * sm.wy0 + (v11 - v10) * sm.wy1);
                        ggy += gw[c] * ((v10 - v00) * sm.wx0 + (v11 - v01) * sm.wx1);
                    }
                    ggx *= pr.mx; ggy *= pr.my;                  // unnormalise: (size-1)/2, 0 where clamped
                    // Project pullback (normalise: *2/(size-1))
                    const float gu = ggx * (2.f / (float)(W - 1)), gv = ggy * (2.f / (float)(H - 1));
                    const float gc[3] = {gu * pr.denom, gv * pr.denom, -(gu * pr.c[0] + gv * pr.c[1]) * pr.denom * pr.denom};
                    float gr[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) gr[j] = a.K[j * 3 + 0] * gc[0] + a.K[j * 3 + 1] * gc[1] + a.K[j * 3 + 2] * gc[2];
                    float gP[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        gP[j] = T[s].R[j * 3 + 0] * gr[0] + T[s].R[j * 3 + 1] * gr[1] + T[s].R[j * 3 + 2] * gr[2];
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc[s * 12 + j * 3 + k] += gr[k] * pr.P[j];
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[s * 12 + 9 + k] += gr[k];
                    // Backproject + disparity_to_depth pullback
                    const float cx = (float)(px + 1), cy = (float)(py + 1);
                    float gdepth = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) gdepth += gP[k] * (a.invK[k] * cx + a.invK[3 + k] * cy + a.invK[6 + k]);
                    const float ddn = d * (a.max_disp - a.min_disp) + a.min_disp;
                    const float rdd = frcp(ddn);
                    gd += -gdepth * (a.max_disp - a.min_disp) * rdd * rdd;
                }
                // smooth_loss(d / (mean + eps)) pullback
                float gs = -corr;
                auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
                auto ex = [&](int b) { float gi = 0.f; for (int c = 0; c < C; ++c) gi += fabsf(tg[c][b] - tg[c][b + 1]); if (C > 1) gi *= invC; return fexp(-gi); };
                auto ey = [&](int b) { float gi = 0.f; for (int c = 0; c < C; ++c) gi += fabsf(tg[c][b] - tg[c][b + VW]); if (C > 1) gi *= invC; return fexp(-gi); };
                if (px + 1 < W) gs += ax * sgn(d - dl[cb + 1]) * ex(cb);
                if (px >= 1) gs -= ax * sgn(dl[cb - 1] - d) * ex(cb - 1);
                if (py + 1 < H) gs += ay * sgn(d - dl[cb + VW]) * ey(cb);
                if (py >= 1) gs -= ay * sgn(dl[cb - VW] - d) * ey(cb - VW);
                if (dlo) {                                        // adjoint of NNlib.upsample_bilinear, row direction: the lanes of a wave hold different columns -> no conflicts
                    int h0, h1; float fh; up_src_r(py, dh, rh, h0, h1, fh);
                    const float g = gd + gs;
                    atomicAdd(crow + (h0 - ci_lo) * TW + lx, (1.f - fh) * g); atomicAdd(crow + (h1 - ci_lo) * TW + lx, fh * g);
                } else dd[(long)py * W + px] = gd + gs;
            }
        }
        if constexpr (C > 1) {
            const int row = tid >> 4;
#pragma unroll
            for (int k = 0; k < 24; ++k) { const float r = row16_sum(acc[k]); if ((tid & 15) == 0) red[row][k] += r; acc[k] = 0.f; }      // (lane 0 of a row is the only writer of red[row][*])
        }
        // ---- coarse scales: column direction on the few coarse rows, then flush (border rows / columns are shared with the neighbouring tiles -> float atomics)
        if (dlo) {
            __syncthreads();
            int i0, ci_hi, cj_hi; float f;
            up_src_r(min(ty0 + TH, H) - 1, dh, rh, i0, ci_hi, f); up_src_r(min(tx0 + TW, W) - 1, dw, rw, i0, cj_hi, f);
            const int nrow = ci_hi - ci_lo + 1, ncol = cj_hi - cj_lo + 1;
            for (int it = tid; it < nrow * TW; it += 256) {
                const int r = it / TW, p = it - r * TW;
                if (tx0 + p >= W) continue;
                int w0, w1; float fw; up_src_r(tx0 + p, dw, rw, w0, w1, fw);
                const float v = crow[it];
                atomicAdd(cacc + r * CW + w0 - cj_lo, (1.f - fw) * v); atomicAdd(cacc + r * CW + w1 - cj_lo, fw * v);
            }
            __syncthreads();
            for (int it = tid; it < nrow * ncol; it += 256) {
                const int r = it / ncol, c = it - r * ncol;
                const float v = cacc[r * CW + c];
                if (v != 0.f) atomicAdd(dlo + (long)(ci_lo + r) * dw + cj_lo + c, v);
            }
        }
    }
    // ---- block reduction: rows of 16 lanes by DPP, the 16 rows through LDS, one double atomic per value per block
    const int row = threadIdx.x >> 4;
    if constexpr (C == 1) {
#pragma unroll
        for (int k = 0; k < 24; ++k) { const float r = row16_sum(acc[k]); if ((threadIdx.x & 15) == 0) red[row][k] = r; }
    }
    { const float r = row16_sum(s_photo); if ((threadIdx.x & 15) == 0) red[row][24] = r; }
    __syncthreads();
    if (threadIdx.x < 25) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += red[r][threadIdx.x];
        if (o.blk) o.blk[((long)zi * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * 25 + threadIdx.x] = (double)v;
        else if (threadIdx.x == 24) atomicAdd(o.part + (long)zi * 4, (double)v);
        else if (v != 0.f) { const int s = threadIdx.x / 12, k = threadIdx.x % 12; atomicAdd(o.dRt_acc + ((long)s * a.N + n) * 12 + k, (double)v); }
    }
}
// deterministic mode: photometric sums and pose-gradient partials of loss_train_kernel in (scale, block) order
__global__ void loss_train_reduce_kernel(const double* __restrict__ blk, double* __restrict__ part, double* __restrict__ dRt_acc, int N, int nscales, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 24 * N) {
        const int k = i % 12, n = (i / 12) % N, s = i / (12 * N);
        double acc = 0;
        for (int sc = 0; sc < nscales; ++sc)
            for (int b = 0; b < nb; ++b) acc += blk[(((long)sc * N + n) * nb + b) * 25 + s * 12 + k];
        dRt_acc[((long)s * N + n) * 12 + k] = acc;
    } else if (i < 24 * N + N * nscales) {
        const int g = i - 24 * N;
        double acc = 0;
        for (int b = 0; b < nb; ++b) acc += blk[((long)g * nb + b) * 25 + 24];
        part[(long)g * 4] = acc;
    }
}
// launch_loss_smooth_sums fills slots 1..3 of part (zeroed by the caller), launch_loss_train (after it, same stream) slot 0 -- together what launch_loss_fwd gives; dd[sc] = [N][H][W] gradient wrt the full-resolution disparity of scale sc,
// or dlo[sc] = [N][dh][dw] (zeroed by the caller) its resize adjoint
void launch_loss_smooth_sums(const LossArgs& a, double* part, double* blk, hipStream_t s) {
    dim3 grid((a.W + 31) / 32, (a.H + 31) / 32, a.N);
    if (a.C == 1) hipLaunchKernelGGL(loss_smooth_sums_kernel<1>, grid, dim3(256), 0, s, a, part, blk);
    else hipLaunchKernelGGL(loss_smooth_sums_kernel<3>, grid, dim3(256), 0, s, a, part, blk);
    if (blk) hipLaunchKernelGGL(loss_smooth_reduce_kernel, dim3((a.N * a.nscales * 3 + 63) / 64), dim3(64), 0, s, blk, part, a.N, a.nscales, (int)(grid.x * grid.y));
}
void launch_loss_train(const LossArgs& a, double* part, float* const dd[MD2_MAX_SCALES], float* const dlo[MD2_MAX_SCALES], double* dRt_acc, double* blk, hipStream_t s) {
    if ((a.C != 1 && a.C != 3) || a.H < 2 || a.W < 2) { fprintf(stderr, "md2: loss_train: unsupported C=%d H=%d W=%d\n", a.C, a.H, a.W); abort(); }
    LossTrainOut o; o.part = part; o.dRt_acc = dRt_acc; o.blk = blk;
    for (int i = 0; i < MD2_MAX_SCALES; ++i) { o.dd[i] = dd[i]; o.dlo[i] = dlo[i]; }
    static const int tpb_env = md2_tune("MD2_LOSS_TPB", 0);
    const int ntx = (a.W + TW - 1) / TW, tpb = tpb_env > 0 ? tpb_env : (a.C == 1 ? 2 : 4);          // (tools/run_loss_tpb.sh, round 5: C = 1 0.433 / 0.427 / 0.429 / 0.489 ms for 1 / 2 / 3 / 4 tiles per workgroup at batch 32; C = 3 at 1024x320, batch 8: 1.231 / 1.203 / 1.210 / 1.183)
    const int th = a.C == 1 ? 32 : 16;
    dim3 grid((ntx + tpb - 1) / tpb, (a.H + th - 1) / th, a.N * a.nscales);
    if (a.C == 1) hipLaunchKernelGGL((loss_train_kernel<1, 32>), grid, dim3(256), 0, s, a, o, tpb, ntx);
    else hipLaunchKernelGGL((loss_train_kernel<3, 16>), grid, dim3(256), 0, s, a, o, tpb, ntx);
    if (blk) hipLaunchKernelGGL(loss_train_reduce_kernel, dim3((24 * a.N + a.N * a.nscales + 63) / 64), dim3(64), 0, s, blk, part, dRt_acc, a.N, a.nscales, (int)(grid.x * grid.y));
}

// =====================================================================================================================
// planar bilinear resize (align-corners) and its adjoint (gather form) — NNlib.upsample_bilinear(x; size), src/training.jl:45
// =====================================================================================================================
__global__ void upsample_planar_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int planes, int H, int W, int Ho, int Wo) {
    const long total = (long)planes * Ho * Wo;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int wo = i % Wo; const long t = i / Wo; const int ho = t % Ho; const int p = t / Ho;
        out[i] = disp_at(x + (long)p * H * W, H, W, Ho, Wo, ho, wo);
    }
}
void launch_upsample_planar_fwd(const float* x, float* out, int planes, int H, int W, int Ho, int Wo, hipStream_t s) {
    const long total = (long)planes * Ho * Wo;
    hipLaunchKernelGGL(upsample_planar_fwd_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, x, out, planes, H, W, Ho, Wo);
}
__global__ void upsample_planar_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int planes, int H, int W, int Ho, int Wo) {
    const long total = (long)planes * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = i % W; const long t = i / W; const int h = t % H; const int p = t / H;
        if (H == Ho && W == Wo) { dx[i] = dout[i]; continue; }
        const float rh = H > 1 ? (float)(Ho - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(Wo - 1) / (float)(W - 1) : 0.f;
        const int oh_lo = max(0, (int)floorf((h - 1) * rh) - 1), oh_hi = min(Ho - 1, (int)ceilf((h + 1) * rh) + 1);
        const int ow_lo = max(0, (int)floorf((w - 1) * rw) - 1), ow_hi = min(Wo - 1, (int)ceilf((w + 1) * rw) + 1);
        const float* g = dout + (long)p * Ho * Wo;
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            int h0, h1; float fh; up_src_f(oh, H, Ho, h0, h1, fh);
            const float ch = (h0 == h ? 1.f - fh : 0.f) + (h1 == h ? fh : 0.f);
            if (ch == 0.f) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                int w0, w1; float fw; up_src_f(ow, W, Wo, w0, w1, fw);
                const float cw = (w0 == w ? 1.f - fw : 0.f) + (w1 == w ? fw : 0.f);
                if (cw != 0.f) acc += ch * cw * g[(long)oh * Wo + ow];
            }
        }
        dx[i] = acc;
    }
}
void launch_upsample_planar_bwd(const float* dout, float* dx, int planes, int H, int W, int Ho, int Wo, hipStream_t s) {
    const long total = (long)planes * H * W;
    hipLaunchKernelGGL(upsample_planar_bwd_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, dout, dx, planes, H, W, Ho, Wo);
}

// =====================================================================================================================
// pose: hat / so3_exp_map / composeT (src/utils.jl:101-141,181-188) and pullbacks — one thread per (source, sample)
// =====================================================================================================================
__device__ inline void d_hat(const float* v, float* s) {
    for (int i = 0; i < 9; ++i) s[i] = 0.f;
    s[0 * 3 + 1] = v[2]; s[1 * 3 + 0] = -v[2]; s[0 * 3 + 2] = -v[1]; s[2 * 3 + 0] = v[1]; s[1 * 3 + 2] = v[0]; s[2 * 3 + 1] = -v[0];
}
__device__ inline void d_mat3mul(const float* A, const float* B, float* C) {
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) C[j * 3 + i] = A[0 * 3 + i] * B[j * 3 + 0] + A[1 * 3 + i] * B[j * 3 + 1] + A[2 * 3 + i] * B[j * 3 + 2];
}
__device__ inline void d_so3(const float* v, float* R) {
    float s[9], s2[9]; d_hat(v, s); d_mat3mul(s, s, s2);
    const float th = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float thi = 1.f / fmaxf(th, 1e-4f);
    const float f1 = thi * sinf(th), f2 = thi * thi * (1.f - cosf(th));
    for (int i = 0; i < 9; ++i) R[i] = f1 * s[i] + f2 * s2[i] + ((i % 4 == 0) ? 1.f : 0.f);
}
// theta == 0: the reference's sqrt pullback is NaN (README.md:47-51); 0 sub-gradient here (documented deviation)
__device__ inline void d_so3_bwd(const float* v, const float* g, float* dr) {
    float s[9], s2[9], st[9], a[9], b[9], dS[9]; d_hat(v, s); d_mat3mul(s, s, s2);
    const float th = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float thi = 1.f / fmaxf(th, 1e-4f);
    const float sn = sinf(th), cs = cosf(th), f1 = thi * sn, f2 = thi * thi * (1.f - cs);
    float df1 = 0.f, df2 = 0.f;
    for (int i = 0; i < 9; ++i) { df1 += g[i] * s[i]; df2 += g[i] * s2[i]; }
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) st[j * 3 + i] = s[i * 3 + j];
    d_mat3mul(g, st, a); d_mat3mul(st, g, b);
    for (int i = 0; i < 9; ++i) dS[i] = f1 * g[i] + f2 * (a[i] + b[i]);
    const float dthi = df1 * sn + df2 * 2.f * thi * (1.f - cs);
    float dth = df1 * thi * cs + df2 * thi * thi * sn;
    if (th > 1e-4f) dth += -dthi * thi * thi;
    const float hv[3] = {dS[1 * 3 + 2] - dS[2 * 3 + 1], -dS[0 * 3 + 2] + dS[2 * 3 + 0], dS[0 * 3 + 1] - dS[1 * 3 + 0]};
    for (int i = 0; i < 3; ++i) dr[i] = hv[i] + (th > 0.f ? dth * v[i] / th : 0.f);
}
__device__ inline void d_compose(const float* r, const float* t, int inv, float* R, float* to) {
    d_so3(r, R);
    if (inv) {
        float x;
        x = R[1]; R[1] = R[3]; R[3] = x; x = R[2]; R[2] = R[6]; R[6] = x; x = R[5]; R[5] = R[7]; R[7] = x;
        for (int i = 0; i < 3; ++i) to[i] = R[0 * 3 + i] * -t[0] + R[1 * 3 + i] * -t[1] + R[2 * 3 + i] * -t[2];
    } else
        for (int i = 0; i < 3; ++i) to[i] = t[i];
}
__device__ inline void d_compose_bwd(const float* r, const float* t, int inv, const float* dR, const float* dto, float* dr, float* dt) {
    float R[9], g[9]; d_so3(r, R);
    if (inv) {
        float gRt[9];
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) gRt[j * 3 + i] = dR[j * 3 + i] + dto[i] * -t[j];
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) g[j * 3 + i] = gRt[i * 3 + j];
        for (int i = 0; i < 3; ++i) dt[i] = -(R[0 * 3 + i] * dto[0] + R[1 * 3 + i] * dto[1] + R[2 * 3 + i] * dto[2]);
    } else {
        for (int i = 0; i < 9; ++i) g[i] = dR[i];
        for (int i = 0; i < 3; ++i) dt[i] = dto[i];
    }
    d_so3_bwd(r, g, dr);
}
__global__ void pose_compose_kernel(const float* __restrict__ rvec, const float* __restrict__ tvec, float* __restrict__ Rt, int N, int inv0, int inv1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= 2 * N) return;
    const int s = i / N;
    d_compose(rvec + i * 3, tvec + i * 3, s == 0 ? inv0 : inv1, Rt + (long)i * 12, Rt + (long)i * 12 + 9);
}
void launch_pose_compose(const float* rvec, const float* tvec, float* Rt, int N, int inv0, int inv1, hipStream_t s) {
    hipLaunchKernelGGL(pose_compose_kernel, dim3((2 * N + 63) / 64), dim3(64), 0, s, rvec, tvec, Rt, N, inv0, inv1);
}
__global__ void pose_compose_bwd_kernel(const float* __restrict__ rvec, const float* __restrict__ tvec, const double* __restrict__ dRt, float* __restrict__ drvec, float* __restrict__ dtvec, int N, int inv0, int inv1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= 2 * N) return;
    const int s = i / N;
    float g[12]; for (int k = 0; k < 12; ++k) g[k] = (float)dRt[(long)i * 12 + k];
    d_compose_bwd(rvec + i * 3, tvec + i * 3, s == 0 ? inv0 : inv1, g, g + 9, drvec + i * 3, dtvec + i * 3);
}
void launch_pose_compose_bwd(const float* rvec, const float* tvec, const double* dRt, float* drvec, float* dtvec, int N, int inv0, int inv1, hipStream_t s) {
    hipLaunchKernelGGL(pose_compose_bwd_kernel, dim3((2 * N + 63) / 64), dim3(64), 0, s, rvec, tvec, dRt, drvec, dtvec, N, inv0, inv1);
}

// =====================================================================================================================
// op-level kernels (reference layout) behind the stateless C entry points
// =====================================================================================================================
#define GRID1D(total) dim3((int)std::max<long>(1, std::min<long>(((total) + 255) / 256, 4096))), dim3(256)
#define FOR1D(i, total) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

__global__ void op_ssim_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int planes, int H, int W) {
    FOR1D(i, (long)planes * H * W) {
        const int w = i % W; const long t = i / W; const int h = t % H; const long p = t / H;
        const float* xp = x + p * H * W; const float* yp = y + p * H * W;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int o = refl_idx(h + dy, H) * W + refl_idx(w + dx, W);
                const float xv = xp[o], yv = yp[o];
                sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
            }
        SsimOut so; ssim_from_sums(sx, sy, sxx, syy, sxy, false, so);
        out[i] = so.val;
    }
}
// photometric for k predictions + min: preds [k][N][C][H][W]; out [N][H][W]; if k == 1 plain photometric
// prediction j of sample n starts at preds + j*pj + n*pn, the target of sample n at tgt + n*tn (floats): the automasking loss reads the frames of x in place
__global__ void op_prediction_kernel(const float* __restrict__ preds, int k, const float* __restrict__ tgt, float* __restrict__ out, int N, int C, int H, int W, long pj, long pn, long tn) {
    const long P = (long)H * W;
    FOR1D(i, (long)N * P) {
        const int w = i % W; const long t = i / W; const int h = t % H; const int n = t / H;
        float best = 0.f;
        for (int j = 0; j < k; ++j) {
            float ss = 0.f, l1 = 0.f;
            for (int c = 0; c < C; ++c) {
                const float* xp = preds + (long)j * pj + (long)n * pn + (long)c * P; const float* yp = tgt + (long)n * tn + (long)c * P;
                float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int o = refl_idx(h + dy, H) * W + refl_idx(w + dx, W);
                        const float xv = xp[o], yv = yp[o];
                        sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                    }
                SsimOut so; ssim_from_sums(sx, sy, sxx, syy, sxy, false, so);
                ss += so.val; l1 += fabsf(yp[(long)h * W + w] - xp[(long)h * W + w]);
            }
            const float pl = 0.85f * (ss / (float)C) + (1.f - 0.85f) * (l1 / (float)C);
            if (j == 0 || pl < best) best = pl;
        }
        out[i] = best;
    }
}
__global__ void op_apply_mask_kernel(const float* __restrict__ mask, const float* __restrict__ wl, float* __restrict__ out, long n) {      // _apply_mask, src/training.jl:17-19
    FOR1D(i, n) { const float m = mask[i], w = wl[i]; out[i] = w < m ? w : m; }                  // Julia `minimum` over (mask, warp_loss): the first of equal values, NaN-free inputs
}
__global__ void op_d2d_kernel(const float* __restrict__ d, float* __restrict__ out, long n, float min_disp, float max_disp) {
    FOR1D(i, n) out[i] = 1.f / (d[i] * (max_disp - min_disp) + min_disp);
}
__global__ void op_backproject_kernel(const float* __restrict__ depth, const float* __restrict__ invK, float* __restrict__ out, int N, int H, int W) {
    FOR1D(i, (long)N * H * W) {
        const int w = i % W; const int h = (i / W) % H;
        const float cx = (float)(w + 1), cy = (float)(h + 1);
        for (int k = 0; k < 3; ++k) out[i * 3 + k] = depth[i] * (invK[k] * cx + invK[3 + k] * cy + invK[6 + k] * 1.f);
    }
}
__global__ void op_project_kernel(const float* __restrict__ pts, const float* __restrict__ K, const float* __restrict__ R, const float* __restrict__ t, float* __restrict__ uv, int N, int H, int W) {
    const long P = (long)H * W;
    FOR1D(i, (long)N * P) {
        const int n = i / P; const float* Rn = R + n * 9; const float* tn = t + n * 3; const float* q = pts + i * 3;
        float r[3], c[3];
        for (int k = 0; k < 3; ++k) r[k] = (Rn[k] * q[0] + Rn[3 + k] * q[1] + Rn[6 + k] * q[2]) + tn[k];
        for (int k = 0; k < 3; ++k) c[k] = K[k] * r[0] + K[3 + k] * r[1] + K[6 + k] * r[2];
        const float denom = 1.f / (c[2] + 1e-7f);
        uv[i * 2 + 0] = (((c[0] * denom - 1.f) / (float)(W - 1)) - 0.5f) * 2.f;
        uv[i * 2 + 1] = (((c[1] * denom - 1.f) / (float)(H - 1)) - 0.5f) * 2.f;
    }
}
template <int BWD>
__global__ void op_grid_sample_kernel(const float* __restrict__ in, const float* __restrict__ grid, float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ dgrid,
                                      int N, int C, int H, int W, int Ho, int Wo, int border) {
    const long Po = (long)Ho * Wo, Pi = (long)H * W;
    FOR1D(i, (long)N * Po) {
        const int n = i / Po; const long p = i - (long)n * Po;
        const float gx = grid[i * 2], gy = grid[i * 2 + 1];
        float x = ((gx + 1.f) * 0.5f) * (float)(W - 1), y = ((gy + 1.f) * 0.5f) * (float)(H - 1);
        float mx = 0.5f * (float)(W - 1), my = 0.5f * (float)(H - 1);
        if (border) {
            if (!(x > 0.f)) { x = 0.f; mx = 0.f; } else if (x >= (float)(W - 1)) { x = (float)(W - 1); mx = 0.f; }
            if (!(y > 0.f)) { y = 0.f; my = 0.f; } else if (y >= (float)(H - 1)) { y = (float)(H - 1); my = 0.f; }
        }
        const float fx = floorf(x), fy = floorf(y);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = x - fx, wx0 = 1.f - wx1, wy1 = y - fy, wy0 = 1.f - wy1;
        const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
        float ggx = 0.f, ggy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* im = in + ((long)n * C + c) * Pi;
            const float v00 = (vy0 && vx0) ? im[(long)y0 * W + x0] : 0.f, v01 = (vy0 && vx1) ? im[(long)y0 * W + x1] : 0.f;
            const float v10 = (vy1 && vx0) ? im[(long)y1 * W + x0] : 0.f, v11 = (vy1 && vx1) ? im[(long)y1 * W + x1] : 0.f;
            if (!BWD) out[((long)n * C + c) * Po + p] = v00 * wy0 * wx0 + v01 * wy0 * wx1 + v10 * wy1 * wx0 + v11 * wy1 * wx1;
            else { const float g = dout[((long)n * C + c) * Po + p]; ggx += g * ((v01 - v00) * wy0 + (v11 - v10) * wy1); ggy += g * ((v10 - v00) * wx0 + (v11 - v01) * wx1); }
        }
        if (BWD) { dgrid[i * 2] = ggx * mx; dgrid[i * 2 + 1] = ggy * my; }
    }
}
__global__ void op_warp_kernel(const float* __restrict__ disp, const float* __restrict__ src, const float* __restrict__ K, const float* __restrict__ invK,
                               const float* __restrict__ R, const float* __restrict__ t, float* __restrict__ out, int N, int C, int H, int W, float min_disp, float max_disp) {
    const long P = (long)H * W;
    FOR1D(i, (long)N * P) {
        const int n = i / P; const long p = i - (long)n * P; const int py = p / W, px = p - (long)py * W;
        Pose12 T; for (int k = 0; k < 9; ++k) T.R[k] = R[n * 9 + k]; for (int k = 0; k < 3; ++k) T.t[k] = t[n * 3 + k];
        float Kl[9], iK[9]; for (int k = 0; k < 9; ++k) { Kl[k] = K[k]; iK[k] = invK[k]; }
        Proj pr; project_px(disp[i], py, px, Kl, iK, T, min_disp, max_disp, H, W, pr);
        Samp sm; sample_setup(pr.x, pr.y, H, W, sm);
        for (int c = 0; c < C; ++c) out[((long)n * C + c) * P + p] = sample_val(src + ((long)n * C + c) * P, sm);
    }
}
__global__ void op_hat_kernel(const float* __restrict__ r, float* __restrict__ S, int N) { FOR1D(i, N) d_hat(r + i * 3, S + i * 9); }
__global__ void op_so3_kernel(const float* __restrict__ r, float* __restrict__ R, int N) { FOR1D(i, N) d_so3(r + i * 3, R + i * 9); }
__global__ void op_so3_bwd_kernel(const float* __restrict__ r, const float* __restrict__ dR, float* __restrict__ dr, int N) { FOR1D(i, N) d_so3_bwd(r + i * 3, dR + i * 9, dr + i * 3); }
__global__ void op_compose_kernel(const float* __restrict__ r, const float* __restrict__ t, int inv, float* __restrict__ R, float* __restrict__ to, int N) {
    FOR1D(i, N) d_compose(r + i * 3, t + i * 3, inv, R + i * 9, to + i * 3);
}
// smooth_loss: block partial sums -> double atomics part[0]=Sx, part[1]=Sy ; finalised by op_smooth_final
__global__ void op_smooth_kernel(const float* __restrict__ d, const float* __restrict__ img, double* __restrict__ part, int N, int C, int H, int W) {
    const long P = (long)H * W;
    float sx = 0.f, sy = 0.f;
    FOR1D(i, (long)N * P) {
        const int n = i / P; const long p = i - (long)n * P; const int h = p / W, w = p - (long)h * W;
        if (w + 1 < W) { float gi = 0.f; for (int c = 0; c < C; ++c) { const float* im = img + ((long)n * C + c) * P; gi += fabsf(im[p] - im[p + 1]); } if (C > 1) gi /= (float)C; sx += fabsf(d[i] - d[i + 1]) * expf(-gi); }
        if (h + 1 < H) { float gi = 0.f; for (int c = 0; c < C; ++c) { const float* im = img + ((long)n * C + c) * P; gi += fabsf(im[p] - im[p + W]); } if (C > 1) gi /= (float)C; sy += fabsf(d[i] - d[i + W]) * expf(-gi); }
    }
    sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { atomicAdd(part, (double)sx); atomicAdd(part + 1, (double)sy); }
}
__global__ void op_smooth_final(const double* __restrict__ part, float* __restrict__ out, double cntx, double cnty) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = (float)(part[0] / cntx + part[1] / cnty); }

// ---- op-level pullbacks (reference layout): what a ChainRules `rrule` of the L2 functions calls when Zygote stays in charge of the step --------------------------
// d out / d x of SSIM (and of photometric_loss: scale_ssim = alpha / C, scale_l1 = (1 - alpha) / C), gather form: pixel p collects from every window q that holds one of
// its pad_reflect aliases u.  x / y: [planes][H][W]; dout: [planes / cd][H][W] (cd = 1: one map per plane; cd = C: one map per sample, shared by its channels);
// win / wj (optional): only pixels q with win[q] == wj pass their dout (the arg-min of prediction_loss)
__global__ void op_ssim_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dout, float* __restrict__ dx, int planes, int cd, int H, int W,
                                   float scale_ssim, float scale_l1, const float* __restrict__ win, float wj) {
    FOR1D(i, (long)planes * H * W) {
        const int w = i % W; const long t = i / W; const int h = t % H; const long pl = t / H;
        const float* xp = x + pl * H * W; const float* yp = y + pl * H * W;
        const long dbase = (pl / cd) * (long)H * W;
        const float xv = xp[(long)h * W + w], yv = yp[(long)h * W + w];
        int uys[3], uxs[3], ny = 0, nx = 0;
        uys[ny++] = h; if (h == 1) uys[ny++] = -1; if (h == H - 2) uys[ny++] = H;
        uxs[nx++] = w; if (w == 1) uxs[nx++] = -1; if (w == W - 2) uxs[nx++] = W;
        float g = 0.f;
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix)
                for (int dy = -1; dy <= 1; ++dy) {
                    const int qy = uys[iy] + dy; if (qy < 0 || qy >= H) continue;
                    for (int dxx = -1; dxx <= 1; ++dxx) {
                        const int qx = uxs[ix] + dxx; if (qx < 0 || qx >= W) continue;
                        const long qo = (long)qy * W + qx;
                        float d = dout[dbase + qo];
                        if (win && win[dbase + qo] != wj) d = 0.f;
                        if (d == 0.f) continue;
                        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
                        for (int a = -1; a <= 1; ++a)
                            for (int b = -1; b <= 1; ++b) {
                                const int o = refl_idx(qy + a, H) * W + refl_idx(qx + b, W);
                                const float xa = xp[o], ya = yp[o];
                                sx += xa; sy += ya; sxx += xa * xa; syy += ya * ya; sxy += xa * ya;
                            }
                        SsimOut so; ssim_from_sums(sx, sy, sxx, syy, sxy, true, so);
                        g += d * (1.f / 9.f) * (so.g_mx + 2.f * xv * so.g_pxx + yv * so.g_pxy);
                    }
                }
        float out = scale_ssim * g;
        if (scale_l1 != 0.f) {
            float d = dout[dbase + (long)h * W + w];
            if (win && win[dbase + (long)h * W + w] != wj) d = 0.f;
            const float df = yv - xv;
            out += scale_l1 * d * -(df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
        }
        dx[i] = out;
    }
}
// index (as float) of the first minimum over the k photometric losses of prediction_loss
__global__ void op_prediction_argmin_kernel(const float* __restrict__ preds, int k, const float* __restrict__ tgt, float* __restrict__ win, int N, int C, int H, int W, long pj, long pn, long tn) {
    const long P = (long)H * W;
    FOR1D(i, (long)N * P) {
        const int w = i % W; const long t = i / W; const int h = t % H; const int n = t / H;
        float best = 0.f; int bj = 0;
        for (int j = 0; j < k; ++j) {
            float ss = 0.f, l1 = 0.f;
            for (int c = 0; c < C; ++c) {
                const float* xp = preds + (long)j * pj + (long)n * pn + (long)c * P; const float* yp = tgt + (long)n * tn + (long)c * P;
                float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int o = refl_idx(h + dy, H) * W + refl_idx(w + dx, W);
                        const float xv = xp[o], yv = yp[o];
                        sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                    }
                SsimOut so; ssim_from_sums(sx, sy, sxx, syy, sxy, false, so);
                ss += so.val; l1 += fabsf(yp[(long)h * W + w] - xp[(long)h * W + w]);
            }
            const float pl = 0.85f * (ss / (float)C) + (1.f - 0.85f) * (l1 / (float)C);
            if (j == 0 || pl < best) { best = pl; bj = j; }
        }
        win[i] = (float)bj;
    }
}
// d smooth_loss / d disparity (the image is data): dl = upstream scalar (device)
__global__ void op_smooth_bwd_kernel(const float* __restrict__ d, const float* __restrict__ img, const float* __restrict__ dl, float* __restrict__ dd, int N, int C, int H, int W) {
    const long P = (long)H * W;
    const float gx = dl[0] / (float)((double)(W - 1) * H * N), gy = dl[0] / (float)((double)W * (H - 1) * N);
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    FOR1D(i, (long)N * P) {
        const int n = i / P; const long p = i - (long)n * P; const int h = p / W, w = p - (long)h * W;
        auto ex = [&](long q) { float gi = 0.f; for (int c = 0; c < C; ++c) { const float* im = img + ((long)n * C + c) * P; gi += fabsf(im[q] - im[q + 1]); } if (C > 1) gi /= (float)C; return expf(-gi); };
        auto ey = [&](long q) { float gi = 0.f; for (int c = 0; c < C; ++c) { const float* im = img + ((long)n * C + c) * P; gi += fabsf(im[q] - im[q + W]); } if (C > 1) gi /= (float)C; return expf(-gi); };
        float g = 0.f;
        if (w + 1 < W) g += gx * sgn(d[i] - d[i + 1]) * ex(p);
        if (w >= 1) g -= gx * sgn(d[i - 1] - d[i]) * ex(p - 1);
        if (h + 1 < H) g += gy * sgn(d[i] - d[i + W]) * ey(p);
        if (h >= 1) g -= gy * sgn(d[i - W] - d[i]) * ey(p - W);
        dd[i] = g;
    }
}
// pullback of the fused warp (md2_warp_fwd) wrt disparity, R (column-major 3x3) and t; the source image is data.  dR / dt: [N][9] / [N][3], zeroed by the caller, float atomics
__global__ __launch_bounds__(256) void op_warp_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ src, const float* __restrict__ K, const float* __restrict__ invK,
                                                         const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ dout, float* __restrict__ ddisp,
                                                         float* __restrict__ dR, float* __restrict__ dt, int C, int H, int W, float min_disp, float max_disp) {
    const long P = (long)H * W; const int n = blockIdx.y;
    Pose12 T; for (int k = 0; k < 9; ++k) T.R[k] = R[n * 9 + k]; for (int k = 0; k < 3; ++k) T.t[k] = t[n * 3 + k];
    float Kl[9], iK[9]; for (int k = 0; k < 9; ++k) { Kl[k] = K[k]; iK[k] = invK[k]; }
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        const int py = p / W, px = p - (long)py * W;
        const float d = disp[(long)n * P + p];
        Proj pr; project_px(d, py, px, Kl, iK, T, min_disp, max_disp, H, W, pr);
        Samp sm; sample_setup(pr.x, pr.y, H, W, sm);
        float ggx = 0.f, ggy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* im = src + ((long)n * C + c) * P; const float g = dout[((long)n * C + c) * P + p];
            const float v00 = im[sm.i00], v01 = sm.v01 ? im[sm.i01] : 0.f, v10 = sm.v10 ? im[sm.i10] : 0.f, v11 = sm.v11 ? im[sm.i11] : 0.f;
            ggx += g * ((v01 - v00) * sm.wy0 + (v11 - v10) * sm.wy1);
            ggy += g * ((v10 - v00) * sm.wx0 + (v11 - v01) * sm.wx1);
        }
        ggx *= pr.mx; ggy *= pr.my;
        const float gu = ggx * (2.f / (float)(W - 1)), gv = ggy * (2.f / (float)(H - 1));
        const float gc[3] = {gu * pr.denom, gv * pr.denom, -(gu * pr.c[0] + gv * pr.c[1]) * pr.denom * pr.denom};
        float gr[3], gP[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) gr[j] = Kl[j * 3 + 0] * gc[0] + Kl[j * 3 + 1] * gc[1] + Kl[j * 3 + 2] * gc[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gP[j] = T.R[j * 3 + 0] * gr[0] + T.R[j * 3 + 1] * gr[1] + T.R[j * 3 + 2] * gr[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[j * 3 + k] += gr[k] * pr.P[j];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[9 + k] += gr[k];
        const float cx = (float)(px + 1), cy = (float)(py + 1);
        float gdepth = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) gdepth += gP[k] * (iK[k] * cx + iK[3 + k] * cy + iK[6 + k]);
        const float ddn = d * (max_disp - min_disp) + min_disp, rdd = frcp(ddn);
        ddisp[(long)n * P + p] = -gdepth * (max_disp - min_disp) * rdd * rdd;
    }
    __shared__ float red[4][12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) red[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < 9) atomicAdd(dR + n * 9 + threadIdx.x, v); else atomicAdd(dt + n * 3 + threadIdx.x - 9, v);
    }
}

// automasking_loss (src/training.jl:9-11) on the triplet tensor x [N][3][C][H][W] in place: predictions = frames src0, src1, target = frame tgt
void launch_automask(const float* x, float* out, int N, int C, int H, int W, int src0, int src1, int tgt, hipStream_t s) {
    const long CP = (long)C * H * W;
    hipLaunchKernelGGL(op_prediction_kernel, GRID1D((long)N * H * W), 0, s, x + src0 * CP, 2, x + tgt * CP, out, N, C, H, W, (long)(src1 - src0) * CP, 3 * CP, 3 * CP);
}

}  // namespace md2

// ---- stateless C entry points (include/md2hip.h) ---------------------------------------------------------------------------
using namespace md2;
static inline hipStream_t S_(void* s) { return (hipStream_t)s; }
static inline int chk() { return hipGetLastError() == hipSuccess ? 0 : 1; }

extern "C" {
int32_t md2_ssim_fwd(const float* x, const float* y, float* out, int32_t planes, int32_t H, int32_t W, void* stream) {
    if (H < 2 || W < 2) return 2;
    hipLaunchKernelGGL(op_ssim_kernel, GRID1D((long)planes * H * W), 0, S_(stream), x, y, out, planes, H, W); return chk();
}
int32_t md2_photometric_loss_fwd(const float* pred, const float* target, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    if (H < 2 || W < 2) return 2;
    hipLaunchKernelGGL(op_prediction_kernel, GRID1D((long)N * H * W), 0, S_(stream), pred, 1, target, out, N, C, H, W, 0L, (long)C * H * W, (long)C * H * W); return chk();
}
int32_t md2_prediction_loss_fwd(const float* preds, int32_t k, const float* target, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    if (H < 2 || W < 2 || k < 1) return 2;
    hipLaunchKernelGGL(op_prediction_kernel, GRID1D((long)N * H * W), 0, S_(stream), preds, k, target, out, N, C, H, W, (long)N * C * H * W, (long)C * H * W, (long)C * H * W); return chk();
}
int32_t md2_apply_mask(const float* mask, const float* warp_loss, float* out, int64_t n, void* stream) {
    if (n < 1) return 2;
    hipLaunchKernelGGL(op_apply_mask_kernel, GRID1D((long)n), 0, S_(stream), mask, warp_loss, out, (long)n); return chk();
}
int32_t md2_ssim_bwd(const float* x, const float* y, const float* dout, float* dx, float* dy, int32_t planes, int32_t H, int32_t W, void* stream) {
    if (H < 2 || W < 2) return 2;
    if (dx) hipLaunchKernelGGL(op_ssim_bwd_kernel, GRID1D((long)planes * H * W), 0, S_(stream), x, y, dout, dx, planes, 1, H, W, 1.f, 0.f, (const float*)nullptr, 0.f);
    if (dy) hipLaunchKernelGGL(op_ssim_bwd_kernel, GRID1D((long)planes * H * W), 0, S_(stream), y, x, dout, dy, planes, 1, H, W, 1.f, 0.f, (const float*)nullptr, 0.f);      // SSIM is symmetric in its arguments
    return chk();
}
int32_t md2_photometric_loss_bwd(const float* pred, const float* target, const float* dout, float* dpred, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    if (H < 2 || W < 2) return 2;
    hipLaunchKernelGGL(op_ssim_bwd_kernel, GRID1D((long)N * C * H * W), 0, S_(stream), pred, target, dout, dpred, N * C, C, H, W, 0.85f / (float)C, (1.f - 0.85f) / (float)C, (const float*)nullptr, 0.f);
    return chk();
}
int32_t md2_prediction_loss_bwd(const float* preds, int32_t k, const float* target, const float* dout, float* dpreds, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    if (H < 2 || W < 2 || k < 1) return 2;
    float* win = nullptr;
    if (hipMallocAsync((void**)&win, (size_t)N * H * W * sizeof(float), S_(stream)) != hipSuccess) return 3;
    const long NC = (long)N * C * H * W;
    hipLaunchKernelGGL(op_prediction_argmin_kernel, GRID1D((long)N * H * W), 0, S_(stream), preds, k, target, win, N, C, H, W, NC, (long)C * H * W, (long)C * H * W);
    for (int j = 0; j < k; ++j)        // the gradient of the minimum goes to the first arg-min (Zygote `minimum`)
        hipLaunchKernelGGL(op_ssim_bwd_kernel, GRID1D(NC), 0, S_(stream), preds + j * NC, target, dout, dpreds + j * NC, N * C, C, H, W, 0.85f / (float)C, (1.f - 0.85f) / (float)C, win, (float)j);
    (void)hipFreeAsync(win, S_(stream));
    return chk();
}
int32_t md2_smooth_loss_bwd(const float* disp, const float* image, const float* dloss, float* ddisp, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    if (H < 2 || W < 2) return 2;
    hipLaunchKernelGGL(op_smooth_bwd_kernel, GRID1D((long)N * H * W), 0, S_(stream), disp, image, dloss, ddisp, N, C, H, W); return chk();
}
int32_t md2_warp_bwd(const float* disp, const float* src, const float* K, const float* invK, const float* R, const float* t, const float* dout, float* ddisp, float* dR, float* dt,
                     int32_t N, int32_t C, int32_t H, int32_t W, double min_depth, double max_depth, void* stream) {
    if (H < 2 || W < 2) return 2;
    (void)hipMemsetAsync(dR, 0, (size_t)N * 9 * sizeof(float), S_(stream)); (void)hipMemsetAsync(dt, 0, (size_t)N * 3 * sizeof(float), S_(stream));
    const int gx = (int)std::max<long>(1, std::min<long>(((long)H * W + 255) / 256, 512));
    hipLaunchKernelGGL(op_warp_bwd_kernel, dim3(gx, N), dim3(256), 0, S_(stream), disp, src, K, invK, R, t, dout, ddisp, dR, dt, C, H, W, (float)(1.0 / max_depth), (float)(1.0 / min_depth));
    return chk();
}
int32_t md2_alloc(void** ptr, int64_t bytes) { return (ptr && bytes >= 0 && hipMalloc(ptr, (size_t)bytes) == hipSuccess) ? 0 : 3; }
int32_t md2_free(void* ptr) { return hipFree(ptr) == hipSuccess ? 0 : 1; }
int32_t md2_disparity_to_depth(const float* d, float* out, int64_t n, double min_depth, double max_depth, void* stream) {
    hipLaunchKernelGGL(op_d2d_kernel, GRID1D(n), 0, S_(stream), d, out, (long)n, (float)(1.0 / max_depth), (float)(1.0 / min_depth)); return chk();
}
int32_t md2_backproject(const float* depth, const float* invK, float* out, int32_t N, int32_t H, int32_t W, void* stream) {
    hipLaunchKernelGGL(op_backproject_kernel, GRID1D((long)N * H * W), 0, S_(stream), depth, invK, out, N, H, W); return chk();
}
int32_t md2_project(const float* pts, const float* K, const float* R, const float* t, float* uv, int32_t N, int32_t H, int32_t W, void* stream) {
    hipLaunchKernelGGL(op_project_kernel, GRID1D((long)N * H * W), 0, S_(stream), pts, K, R, t, uv, N, H, W); return chk();
}
int32_t md2_grid_sample_fwd(const float* in, const float* grid, float* out, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t border, void* stream) {
    hipLaunchKernelGGL(op_grid_sample_kernel<0>, GRID1D((long)N * Ho * Wo), 0, S_(stream), in, grid, out, (const float*)nullptr, (float*)nullptr, N, C, H, W, Ho, Wo, border); return chk();
}
int32_t md2_grid_sample_bwd_grid(const float* in, const float* grid, const float* dout, float* dgrid, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t border, void* stream) {
    hipLaunchKernelGGL(op_grid_sample_kernel<1>, GRID1D((long)N * Ho * Wo), 0, S_(stream), in, grid, (float*)nullptr, dout, dgrid, N, C, H, W, Ho, Wo, border); return chk();
}
int32_t md2_warp_fwd(const float* disp, const float* src, const float* K, const float* invK, const float* R, const float* t, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                     double min_depth, double max_depth, void* stream) {
    hipLaunchKernelGGL(op_warp_kernel, GRID1D((long)N * H * W), 0, S_(stream), disp, src, K, invK, R, t, out, N, C, H, W, (float)(1.0 / max_depth), (float)(1.0 / min_depth)); return chk();
}
int32_t md2_upsample_bilinear_fwd(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, void* stream) { launch_upsample_planar_fwd(x, out, planes, H, W, Ho, Wo, S_(stream)); return chk(); }
int32_t md2_upsample_bilinear_bwd(const float* dout, float* dx, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, void* stream) { launch_upsample_planar_bwd(dout, dx, planes, H, W, Ho, Wo, S_(stream)); return chk(); }
int32_t md2_hat(const float* r, float* S, int32_t N, void* stream) { hipLaunchKernelGGL(op_hat_kernel, GRID1D(N), 0, S_(stream), r, S, N); return chk(); }
int32_t md2_so3_exp_map(const float* r, float* R, int32_t N, void* stream) { hipLaunchKernelGGL(op_so3_kernel, GRID1D(N), 0, S_(stream), r, R, N); return chk(); }
int32_t md2_so3_exp_map_bwd(const float* r, const float* dR, float* dr, int32_t N, void* stream) { hipLaunchKernelGGL(op_so3_bwd_kernel, GRID1D(N), 0, S_(stream), r, dR, dr, N); return chk(); }
int32_t md2_composeT(const float* r, const float* t, int32_t inv, float* R, float* to, int32_t N, void* stream) { hipLaunchKernelGGL(op_compose_kernel, GRID1D(N), 0, S_(stream), r, t, inv, R, to, N); return chk(); }
int32_t md2_smooth_loss(const float* disp, const float* image, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    double* part = nullptr;
    if (hipMallocAsync((void**)&part, 2 * sizeof(double), S_(stream)) != hipSuccess) return 3;
    (void)hipMemsetAsync(part, 0, 2 * sizeof(double), S_(stream));
    hipLaunchKernelGGL(op_smooth_kernel, GRID1D((long)N * H * W), 0, S_(stream), disp, image, part, N, C, H, W);
    hipLaunchKernelGGL(op_smooth_final, dim3(1), dim3(64), 0, S_(stream), part, out, (double)(W - 1) * H * N, (double)W * (H - 1) * N);
    (void)hipFreeAsync(part, S_(stream));
    return chk();
}
}
