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
#include "loss_device.h"

namespace md2 {

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

// launch_loss_smooth_sums fills slots 1..3 of part (zeroed by the caller), launch_loss_train (after it, same stream) slot 0 -- together what launch_loss_fwd gives; dd[sc] = [N][H][W] gradient wrt the full-resolution disparity of scale sc,
// or dlo[sc] = [N][dh][dw] (zeroed by the caller) its resize adjoint
void launch_loss_smooth_sums(const LossArgs& a, double* part, double* blk, hipStream_t s) {
    dim3 grid((a.W + 31) / 32, (a.H + 31) / 32, a.N);
    if (a.C == 1) hipLaunchKernelGGL(loss_smooth_sums_kernel<1>, grid, dim3(256), 0, s, a, part, blk);
    else hipLaunchKernelGGL(loss_smooth_sums_kernel<3>, grid, dim3(256), 0, s, a, part, blk);
    if (blk) hipLaunchKernelGGL(loss_smooth_reduce_kernel, dim3((a.N * a.nscales * 3 + 63) / 64), dim3(64), 0, s, blk, part, a.N, a.nscales, (int)(grid.x * grid.y));
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
