// loss_train.hip — the training form of the fused photometric loss (gfx950): loss value AND pullback of all four scales from ONE pass over the warp.
// Replaces the ~400 NNlib / broadcast kernels of src/training.jl:42-75 and their Zygote pullbacks; see loss.hip for the forward-only form,
// the smoothness sums that precede this kernel, and the op-level kernels.  A translation unit of its own since round 6: compiled with -fno-slp-vectorize (Makefile) --
// SLP packing cost loss_train_kernel<1, 32> ~300 v_mov and, through 32 more live registers, ~300 v_readlane of spilled scalars per tile (245 -> 213 VGPRs; the loss half
// 0.424 -> 0.403 ms at the headline shape, 1.19 -> 1.10 ms at configuration 5), while the packed forms are a gain in the bandwidth-bound kernels that stay in loss.hip.
#include "loss_device.h"

namespace md2 {

struct LossTrainOut { double* part; float* dd[MD2_MAX_SCALES]; float* dlo[MD2_MAX_SCALES]; double* dRt_acc; double* blk; };
// One block = `tpb` consecutive TW x TH tiles of one (scale, sample).  Per tile:
//  1. warped sources / target / disparity at the padded positions of tile + 2 -> LDS                           (the only global gathers of the forward)
//  2. window owners q in tile + 1, three per thread in a column (row sums of the five window moments shared between them):
//     photometric of both sources, arg-min [, automask], SSIM pullback coefficients -> LDS; owners inside the tile add to the loss sum
//  3. per tile pixel r, TH/8 per thread in a column: d loss / d warped_s[r] = box sum of the winning owners' coefficients (row sums shared
//     down the column; pad_reflect aliases are weights 2 on the affected row / column), then the pullback through grid_sample, Project,
//     Backproject and disparity_to_depth, plus the smoothness pullback.
// The 24 pose-gradient partials and the loss sum stay in registers across the block's tiles and are reduced once (through the value tiles' LDS, when the tiles are done).
#ifndef MD2_LOSS_HOISTED_PROJECTION
#define MD2_LOSS_HOISTED_PROJECTION 0      // 1: sampling coordinates from the hoisted (K R) P + K t (12 scalars per source, ~4 % faster) -- moves them by ~1e-5 px against
#endif                                     // the reference's K (R P + t), which on real texture flips a few more arg-min decisions than the oracle comparison likes
// Occupancy (round 6): <1, 32> runs at FOUR waves per SIMD (121 registers, 38 KB of LDS per workgroup), <3, 16> at THREE (163 registers, 52.7 KB).  Round 5 left both at
// two (245 registers each; <1, 32> went to three earlier this round, 149 registers, <3, 16> stood at 221).  What the registers had gone to was not the algorithm but the compiler's freedom: (a) hipcc interleaved the unrolled per-pixel
// chains of phase 3 (projection, sampling, pullback) -- one channel: a rolled loop with an explicitly rotated ring of owner-row sums; (b) it DELAYED arithmetic whose inputs
// were ready -- the SSIM coefficients of phase 2 and the per-channel box sums of phase 3 stayed alive as their ~3x larger sets of inputs until the consumer; an empty
// `asm volatile("" : "+v"(x))` on each finished value pins the computation where it is written (three channels: 221 -> 124 registers from the pins alone, no instruction
// added); (c) it hoisted per-thread tile indices out of the tile loop (an opaque copy of threadIdx.x per tile).  LDS: the column pass of the coarse scales' resize adjoint
// is a gather now (no second accumulator tile) and the 16 x 25 block-reduction scratch stands in the value tiles' place once they are done: 55,044 -> 52,724 bytes for
// three channels, which is what three workgroups per CU need -- 54,324 did NOT fit (measured: no change), the allocation granule of the 160 KB LDS is coarser than 512 bytes.
// Loss half at the headline shape 0.332 -> 0.294 ms, at configuration 5 1.090 -> 0.79 ms (tools/run_loss_variants.sh, profiles/r06_loss_variants.txt).
template <int C, int TH>
__global__ __launch_bounds__(256, C == 1 ? 4 : 3) void loss_train_kernel(LossArgs a, LossTrainOut o, int tpb, int ntx) {
    constexpr int OSH = 3;                       // window owners per thread in phase 2: (TW + 2) x ceil((TH + 2) / 3) owner triples -- 408 for <1, 32>, 204 for <3, 16> (one pass of the 256 threads)
    constexpr int VW = TW + 4, VH = TH + 4, OW = TW + 2, OH = TH + 2, SH = TH / 8, NSR = (OH + OSH - 1) / OSH;
    __shared__ float wv[2][C][VH * VW];          // warped sources at the padded positions of tile + 2
    __shared__ float tg[C][VH * VW];
    __shared__ float dl[VH * VW];
    __shared__ float coef[C][3][OH * OW];        // d photometric_{s*} / d (pool x, pool x^2, pool xy) at q in tile + 1, pre-scaled
    __shared__ signed char amin[OH * OW];        // winning source at q, -1: none (outside / masked)
    float (*red)[25] = reinterpret_cast<float (*)[25]>(&wv[0][0][0]);          // the 16 x 25 partials of the block reduction are needed once the tiles are done: in the value tiles' place
    constexpr int CH = TH / 2 + 2;
    __shared__ float crow[CH * TW];              // coarse scales, resize adjoint of the tile's d loss / d disparity: after the row direction [coarse row][tile column] ...
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
    for (int t = 0; t < tpb; ++t) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));              // keeps hipcc from hoisting the per-thread tile indices of every phase out of the tile loop (they lived in scratch: 56 spilled registers with three channels)
        const int tix = blockIdx.x * tpb + t;
        if (tix >= ntx) break;
        const int tx0 = tix * TW;
        if (t) __syncthreads();                  // the previous tile's phases 3 / 4 still read the LDS tiles
        int ci_lo = 0, cj_lo = 0;
        if (dlo) {
            int i1; float f;
            up_src_r(ty0, dh, rh, ci_lo, i1, f); up_src_r(tx0, dw, rw, cj_lo, i1, f);                 // first coarse row / column this tile touches
            for (int i = tid; i < CH * TW; i += 256) crow[i] = 0.f;
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
                        asm volatile("" : "+v"(gmx[oo][s][c]), "+v"(gpxx[oo][s][c]), "+v"(gpxy[oo][s][c]), "+v"(pl[oo][s]));      // finished values: computed HERE (see the header)
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
            if constexpr (C == 1) { rowsum(ly0, rs[0]); rowsum(ly0 + 1, rs[1]); }            // owner rows of pixel rows ly0-1 and ly0
            auto pixel = [&](const int i, const float (&ra)[2][C][3], const float (&rb)[2][C][3], const float (&rc)[2][C][3], const float (&gwc)[2][C]) {
                const int ly = ly0 + i, py = ty0 + ly;
                if (py >= H || px >= W) return;
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
                        if constexpr (C > 1) g = gwc[s][c];
                        else {
                            const float g0 = wym * ra[s][c][0] + rb[s][c][0] + wyp * rc[s][c][0];
                            const float g1 = wym * ra[s][c][1] + rb[s][c][1] + wyp * rc[s][c][1];
                            const float g2 = wym * ra[s][c][2] + rb[s][c][2] + wyp * rc[s][c][2];
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
            };
            if constexpr (C == 1) {
                // one channel: the thread's SH pixels as a ROLLED loop with an explicitly rotated ring of owner-row sums -- unrolled, hipcc interleaves the four pixels'
                // projection / sampling chains and needs ~200 registers for this phase alone
                float nogw[2][C] = {};
#pragma unroll 1
                for (int i = 0; i < SH; ++i) {
                    rowsum(ly0 + i + 2, rs[2]);
                    pixel(i, rs[0], rs[1], rs[2], nogw);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int k = 0; k < 3; ++k) { rs[0][s][0][k] = rs[1][s][0][k]; rs[1][s][0][k] = rs[2][s][0][k]; }
                }
            } else {
                // three channels: d loss / d warped_s[r][c] of the thread's SH = 2 pixels from the same four owner rows, one channel at a time (a ring of all channels' row
                // sums would be 54 registers), each finished value pinned; then the pixels one after the other
                float gwp[SH][2][C];
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
                            asm volatile("" : "+v"(gwp[i][s][c]));          // finished value: computed HERE (see the header)
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < SH; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    pixel(i, rs[0], rs[1], rs[2], gwp[i]);
                }
            }
        }
        // ---- coarse scales: column direction on the few coarse rows as a GATHER (round 6: each coarse (row, column) sums the tile columns that resize onto it, in column
        // order, and goes straight to global memory -- no second LDS accumulator, no LDS atomics, one barrier less), border rows / columns are shared with the
        // neighbouring tiles -> float atomics
        if (dlo) {
            __syncthreads();
            int i0, ci_hi, cj_hi; float f;
            up_src_r(min(ty0 + TH, H) - 1, dh, rh, i0, ci_hi, f); up_src_r(min(tx0 + TW, W) - 1, dw, rw, i0, cj_hi, f);
            const int nrow = ci_hi - ci_lo + 1, ncol = cj_hi - cj_lo + 1, pw = min(TW, W - tx0);
            const float irw = rw > 0.f ? 1.f / rw : 0.f;
            for (int it = tid; it < nrow * ncol; it += 256) {
                const int r = it / ncol, c = it - r * ncol, cj = cj_lo + c;
                // tile columns p whose source coordinate rw (tx0 + p) lies in [cj - 1, cj + 1): one column of margin either side, the exact test is below
                const int p_lo = rw > 0.f ? max((int)((float)(cj - 1) * irw) - tx0 - 1, 0) : 0, p_hi = rw > 0.f ? min((int)((float)(cj + 1) * irw) - tx0 + 2, pw) : pw;
                float v = 0.f;
                for (int p = p_lo; p < p_hi; ++p) {
                    int w0, w1; float fw; up_src_r(tx0 + p, dw, rw, w0, w1, fw);
                    const float x = crow[r * TW + p];
                    v += (w0 == cj ? (1.f - fw) * x : 0.f) + (w1 == cj ? fw * x : 0.f);
                }
                if (v != 0.f) atomicAdd(dlo + (long)(ci_lo + r) * dw + cj, v);
            }
        }
    }
    // ---- block reduction: rows of 16 lanes by DPP, the 16 rows through LDS, one double atomic per value per block
    const int row = threadIdx.x >> 4;
    __syncthreads();                             // the last tile's phase 3 reads the value tiles `red` stands in
#pragma unroll
    for (int k = 0; k < 24; ++k) { const float r = row16_sum(acc[k]); if ((threadIdx.x & 15) == 0) red[row][k] = r; }
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
// dd[sc] = [N][H][W] gradient wrt the full-resolution disparity of scale sc, or dlo[sc] = [N][dh][dw] (zeroed by the caller) its resize adjoint; runs after launch_loss_smooth_sums on the same stream
void launch_loss_train(const LossArgs& a, double* part, float* const dd[MD2_MAX_SCALES], float* const dlo[MD2_MAX_SCALES], double* dRt_acc, double* blk, hipStream_t s) {
    if ((a.C != 1 && a.C != 3) || a.H < 2 || a.W < 2) { fprintf(stderr, "md2: loss_train: unsupported C=%d H=%d W=%d\n", a.C, a.H, a.W); abort(); }
    LossTrainOut o; o.part = part; o.dRt_acc = dRt_acc; o.blk = blk;
    for (int i = 0; i < MD2_MAX_SCALES; ++i) { o.dd[i] = dd[i]; o.dlo[i] = dlo[i]; }
    static const int tpb_env = md2_tune("MD2_LOSS_TPB", 0);
    const int ntx = (a.W + TW - 1) / TW, tpb = tpb_env > 0 ? tpb_env : 2;          // (tools/run_loss_tpb.sh, round 6, at four / three waves per SIMD: C = 1 0.292 / 0.294 / 0.303 / 0.327 / 0.340 ms for 1 / 2 / 3 / 4 / 6 tiles per workgroup at batch 32; C = 3 at 1024x320, batch 8: 0.819 / 0.787 / 0.801 / 0.796 / 0.894)
    const int th = a.C == 1 ? 32 : 16;
    dim3 grid((ntx + tpb - 1) / tpb, (a.H + th - 1) / th, a.N * a.nscales);
    if (a.C == 1) hipLaunchKernelGGL((loss_train_kernel<1, 32>), grid, dim3(256), 0, s, a, o, tpb, ntx);
    else hipLaunchKernelGGL((loss_train_kernel<3, 16>), grid, dim3(256), 0, s, a, o, tpb, ntx);
    if (blk) hipLaunchKernelGGL(loss_train_reduce_kernel, dim3((24 * a.N + a.N * a.nscales + 63) / 64), dim3(64), 0, s, blk, part, dRt_acc, a.N, a.nscales, (int)(grid.x * grid.y));
}

}  // namespace md2
