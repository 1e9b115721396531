// The ResNet stem (7x7, stride 2, pad 3, C image channels -> 64; ResNet.jl trunk, call sites src/Monodepth.jl:117-118, src/model.jl:14) as
// DIRECT convolutions on the boundary tensor: the image patch of an output tile is staged in LDS once and the 49*C-long contraction
// gathers from it with stride-2 addressing.  Replaces im2col (a 327 MB [pixels][64] matrix written and read twice per step at B=32)
// + a K=64 GEMM + a K=pixels weight-gradient GEMM over that matrix.
//
//   forward : D[co][px]  = sum_k W[co][k] * P[k][px]          k = tap*C + c (the internal stem weight layout [64][Kpad], tap-major)
//   wgrad   : D[co][k]  += sum_px dz[px][co] * P[px][k]        GEMM-K = pixels, persistent blocks, fp32 atomics at the end
// both on v_mfma_f32_32x32x2_f32 (K = 2 per instruction: two k's / two pixels per lane half), operands read from LDS one float per lane:
//   patch   [c][row][72]  : 32 consecutive output columns read stride-2 floats -> 32 distinct even (or odd) banks per half-wave
//   weights [k][96]       : the two lane halves read rows k, k+1 -> 96-float rows put them on disjoint banks
//   dz      [px][64 ^ swz]: column index XOR 32 on odd pixels, same purpose
#include "md2_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace md2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int S_PC = 72;                       // patch row length in floats (>= 2*32 + 5)
constexpr int S_TW = 32;                       // output tile width
__device__ __forceinline__ int stem_koff(int kk, int C, int PR) {       // k = tap*C + c -> float offset of (c, kh, kw) in the patch
    const int tap = kk / C, c = kk - tap * C;
    const int kh = (tap * 37) >> 8, kw = tap - 7 * kh;                  // tap / 7 for tap < 64
    return (c * PR + kh) * S_PC + kw;
}
// patch of an output tile [y0, y0+TH) x [x0, x0+32): input rows 2*y0-3 .. +2*TH+4, columns 2*x0-3 .. +68; zero outside the image
template <int C, int PR>
__device__ __forceinline__ void stem_load_patch(float* Ps, const float* __restrict__ x, int n, int y0, int x0, int H, int W, int tid, int nthreads) {
    for (int f = tid; f < C * PR * S_PC; f += nthreads) {
        const int pc = f % S_PC, r = f / S_PC, pr = r % PR, c = r / PR;
        const int ih = 2 * y0 - 3 + pr, iw = 2 * x0 - 3 + pc;
        float v = 0.f;
        if (pc < 69 && ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[(((long)n * C + c) * H + ih) * W + iw];
        Ps[f] = v;
    }
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------
// forward: block = 8 x 32 output pixels of one image x all 64 channels; wave w owns tile rows 2w, 2w+1
// ------------------------------------------------------------------------------------------------------------
template <int C, bool BS = false>          // BS: z is a bf16 tensor
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, View z, int H, int W, int Kpad,
                                                       double* __restrict__ stats, int tiles_x, int tiles_y, int total_tiles) {
    constexpr int K = 49 * C, KP = (K + 1) & ~1, TH = 8, PR = 2 * TH + 5, NP = (C * PR * S_PC + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    float* Ps = ssm;                           // [C][PR][72]
    float* Ws = ssm + C * PR * S_PC;           // [KP][96]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
    for (int f = tid; f < 64 * KP; f += 256) {                         // once per (persistent) block: W[co][k] (row length Kpad) -> Ws[k][co]; k >= K rows are zero
        const int k = f >> 6, co = f & 63;                              // co fastest: conflict-free LDS writes, the 12.5 KB of weights stay in L1/L2
        Ws[k * 96 + co] = k < K ? w[(long)co * Kpad + k] : 0.f;
    }
    // this thread's patch slots: float f = tid + 256 i of [C][PR][72]; (c, row, column) never change, only the tile origin does
    int pslot[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int f = min(tid + 256 * i, C * PR * S_PC - 1), pc = f % S_PC, r = f / S_PC, pr = r % PR, c = r / PR;
        pslot[i] = (c << 16) | (pr << 8) | pc;
    }
    float pv[NP];
    const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const float* prow = Ps + (4 * wave) * S_PC + 2 * col;               // local output row 2*wave -> patch row 4*wave (+kh)
    for (int s = -1; s < my_tiles; ++s) {                               // s = -1: prologue, only issues tile 0's loads
        if (s >= 0) {
            __syncthreads();                                            // the previous tile's MFMAs are done reading the patch
#pragma unroll
            for (int i = 0; i < NP; ++i) if (tid + 256 * i < C * PR * S_PC) Ps[tid + 256 * i] = pv[i];
            __syncthreads();
        }
        if (s + 1 < my_tiles) {                                         // next tile's patch -> registers, in flight during this tile's MFMAs
            const int tile = blockIdx.x + (s + 1) * gridDim.x;
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
            const float* xn = x + (long)n * C * H * W;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int c = pslot[i] >> 16, pr = (pslot[i] >> 8) & 255, pc = pslot[i] & 255;
                const int ih = 2 * ty * TH - 3 + pr, iw = 2 * tx * S_TW - 3 + pc;
                const bool ok = pc < 69 && ih >= 0 && ih < H && iw >= 0 && iw < W;
                const float v = xn[((long)c * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)];
                pv[i] = ok ? v : 0.f;
            }
        }
        if (s < 0) continue;
        const int tile = blockIdx.x + s * gridDim.x;
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * S_TW;
        f32x16 acc[2][2];                      // [channel tile][tile row]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int k2 = 0; k2 < KP / 2; ++k2) {
            const int kk = 2 * k2 + half;
            const int ko = stem_koff(min(kk, K - 1), C, PR);            // a padded k has zero weights: any valid address
            const float a0 = Ws[kk * 96 + col], a1 = Ws[kk * 96 + 32 + col];
            const float b0 = prow[ko], b1 = prow[ko + 2 * S_PC];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        // epilogue: D col = pixel (lane&31), row = 8g + 4*half + e (register 4g+e) -> four consecutive channels = one 16-B NHWC store
        const int ox = x0 + col;
        float valid[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = y0 + 2 * wave + j;
            valid[j] = (oy < z.H && ox < z.W) ? 1.f : 0.f;
            if (valid[j] == 0.f) continue;
            const long zo = (long)n * z.bs + ((long)oy * z.W + ox) * z.ps;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    vst4<BS>(z.p, zo + i * 32 + 8 * g + 4 * half, v);
                }
        }
        // BatchNorm batch statistics of this tile (same halving butterfly as conv_igemm_kernel's epilogue; ragged-tile pixels are masked)
        if (stats) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[32];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = acc[i][0][r] * valid[0], q = acc[i][1][r] * valid[1];
                    v[2 * r] = p + q; v[2 * r + 1] = p * p + q * q;
                }
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    const bool up = (lane >> b) & 1;
#pragma unroll
                    for (int k = 0; k < (16 >> b); ++k) {
                        const float lo = v[2 * k], hi = v[2 * k + 1];
                        v[k] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, 1 << b);
                    }
                }
                const int r = col >> 1, co = i * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                atomicAdd(stats + (size_t)(tile % BN_STAT_SLOTS) * 128 + (col & 1) * 64 + co, (double)v[0]);
            }
        }
    }
}

void launch_stem_fwd(const float* x, int NL, int C, int H, int W, const float* w, int Kpad, const View& z, double* stats, hipStream_t s) {
    if ((C != 1 && C != 3) || z.C != 64 || z.H != H / 2 || z.W != W / 2 || z.N != NL || (H & 1) || (W & 1) || z.ps % 4 != 0) {
        fprintf(stderr, "md2: stem_fwd: unsupported shape C=%d %dx%d\n", C, H, W); abort();
    }
    const int tiles_x = (z.W + S_TW - 1) / S_TW, tiles_y = (z.H + 7) / 8;
    const int KP = (49 * C + 1) & ~1;
    const size_t lds = (size_t)(C * 21 * S_PC + KP * 96) * sizeof(float);
    const int total = tiles_x * tiles_y * NL;
    static int per_cu[2] = {0, 0};
    int& pc = per_cu[C == 3];
    const void* fn = C == 1 ? reinterpret_cast<const void*>(&stem_fwd_kernel<1>) : reinterpret_cast<const void*>(&stem_fwd_kernel<3>);
    if (!pc) {
        (void)hipFuncSetAttribute(C == 1 ? reinterpret_cast<const void*>(&stem_fwd_kernel<1, true>) : reinterpret_cast<const void*>(&stem_fwd_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn, 256, lds) != hipSuccess || pc < 1) pc = 1;
        pc = std::min(pc, 4);
    }
    const dim3 grid(std::min(total, 256 * pc));                        // persistent: the co-resident blocks
    if (z.bf) { if (C == 1) hipLaunchKernelGGL((stem_fwd_kernel<1, true>), grid, dim3(256), lds, s, x, w, z, H, W, Kpad, stats, tiles_x, tiles_y, total);
                else hipLaunchKernelGGL((stem_fwd_kernel<3, true>), grid, dim3(256), lds, s, x, w, z, H, W, Kpad, stats, tiles_x, tiles_y, total); }
    else if (C == 1) hipLaunchKernelGGL((stem_fwd_kernel<1>), grid, dim3(256), lds, s, x, w, z, H, W, Kpad, stats, tiles_x, tiles_y, total);
    else hipLaunchKernelGGL((stem_fwd_kernel<3>), grid, dim3(256), lds, s, x, w, z, H, W, Kpad, stats, tiles_x, tiles_y, total);
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient: persistent blocks walk 4 x 32 pixel tiles; wave w contracts tile row w (16 steps of 2 pixels); blockIdx.y selects 64 k's
// ------------------------------------------------------------------------------------------------------------
template <int C, bool FUSE, bool BS = false>          // BS: dz (and zbn) are bf16 tensors
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, View dz, float* __restrict__ dw, int H, int W, int Kpad,
                                                         int tiles_x, int tiles_y, int total_tiles, View zbn, const float* __restrict__ coef) {
    constexpr int K = 49 * C, TH = 4, PR = 2 * TH + 5, NP = (C * PR * S_PC + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    float* Ps = ssm;                           // [C][PR][72]
    float* Ys = ssm + C * PR * S_PC;           // [128 px][64], column ^ 32 on odd pixels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
    const int k0 = blockIdx.y * 64;
    // this lane's two k columns (one per k tile): patch offsets; k >= K lands in padding rows of dw that are never read -> masked at the end
    int ko[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) ko[t] = stem_koff(min(k0 + t * 32 + col, K - 1), C, PR);
    f32x16 acc[2][2];                          // [channel tile][k tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int pslot[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int f = min(tid + 256 * i, C * PR * S_PC - 1), pc = f % S_PC, r = f / S_PC, pr = r % PR, c = r / PR;
        pslot[i] = (c << 16) | (pr << 8) | pc;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    float pv[NP]; u32x4 yv[8]; unsigned yok = 0;                        // next tile's patch / dz in flight; bit i of yok: dz slot i is inside the image
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v zv[FUSE ? 8 : 1];                                              // FUSE: the BatchNorm input of the same positions; dz = cA * g + cB * z + cC (this thread's 4 channels never change)
    f32x4v cA = {0.f, 0.f, 0.f, 0.f}, cB = cA, cC = cA;
    if constexpr (FUSE) { const int q4 = (tid & 15) * 4; cA = *reinterpret_cast<const f32x4v*>(coef + q4); cB = *reinterpret_cast<const f32x4v*>(coef + 64 + q4); cC = *reinterpret_cast<const f32x4v*>(coef + 128 + q4); }
    const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const float* yrow = Ys + (wave * 32) * 64;
    const float* prow = Ps + (2 * wave) * S_PC;
    for (int s = -1; s < my_tiles; ++s) {                               // s = -1: prologue, only issues tile 0's loads
        if (s >= 0) {
            __syncthreads();                                            // the previous tile's MFMAs are done reading LDS
#pragma unroll
            for (int i = 0; i < NP; ++i) if (tid + 256 * i < C * PR * S_PC) Ps[tid + 256 * i] = pv[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {                               // dz slot f = tid + 256 i: pixel f/16, float4 f%16
                const int f = tid + 256 * i, px = f >> 4, q = f & 15;
                if constexpr (FUSE) {
                    const f32x4v g = __builtin_bit_cast(f32x4v, yv[i]);
                    yv[i] = __builtin_bit_cast(u32x4, cA * g + cB * zv[i] + cC);
                }
                *reinterpret_cast<u32x4*>(Ys + px * 64 + ((q * 4) ^ ((px & 1) * 32))) = yv[i] & (0u - ((yok >> i) & 1u));
            }
            __syncthreads();
        }
        if (s + 1 < my_tiles) {
            const int tile = blockIdx.x + (s + 1) * gridDim.x;
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
            const float* xn = x + (long)n * C * H * W;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int c = pslot[i] >> 16, pr = (pslot[i] >> 8) & 255, pc = pslot[i] & 255;
                const int ih = 2 * ty * TH - 3 + pr, iw = 2 * tx * S_TW - 3 + pc;
                const bool ok = pc < 69 && ih >= 0 && ih < H && iw >= 0 && iw < W;
                const float v = xn[((long)c * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)];
                pv[i] = ok ? v : 0.f;
            }
            const long dn = (long)n * dz.bs;
            yok = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = tid + 256 * i, px = f >> 4, q = f & 15, oy = ty * TH + (px >> 5), ox = tx * S_TW + (px & 31);
                yok |= ((oy < dz.H && ox < dz.W) ? 1u : 0u) << i;
                {
                    const float4 f = vld4<BS>(dz.p, dn + ((long)min(oy, dz.H - 1) * dz.W + min(ox, dz.W - 1)) * dz.ps + q * 4);
                    yv[i] = u32x4{__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w)};
                }
                if constexpr (FUSE) {
                    const float4 f = vld4<BS>(zbn.p, (long)n * zbn.bs + ((long)min(oy, dz.H - 1) * dz.W + min(ox, dz.W - 1)) * zbn.ps + q * 4);
                    zv[i] = f32x4v{f.x, f.y, f.z, f.w};
                }
            }
        }
        if (s < 0) continue;
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            const int px = 2 * st + half;                               // pixel of this lane half within the tile row
            const float a0 = yrow[px * 64 + (col ^ (half * 32))], a1 = yrow[px * 64 + ((32 + col) ^ (half * 32))];
            const float b0 = prow[ko[0] + 2 * px], b1 = prow[ko[1] + 2 * px];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // reduce the 4 waves through LDS ([4 tiles][16 regs][64 lanes] = 16 KB), then one atomic per weight and block
    float* red = ssm;
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float* dst = red + ((i * 2 + j) * 16 + e) * 64 + lane;
                        *dst = wv == 0 ? acc[i][j][e] : *dst + acc[i][j][e];
                    }
        }
    }
    __syncthreads();
    for (int f = tid; f < 4 * 16 * 64; f += 256) {
        const int ln = f & 63, e = (f >> 6) & 15, ij = f >> 10, i = ij >> 1, j = ij & 1;
        const int co = i * 32 + (e >> 2) * 8 + (ln >> 5) * 4 + (e & 3), k = k0 + j * 32 + (ln & 31);      // 32x32 D layout: row = co, col = k
        if (k < K) atomicAdd(dw + (long)co * Kpad + k, red[f]);
    }
}

void launch_stem_wgrad(const float* x, int NL, int C, int H, int W, const View& dz, float* dw, int Kpad, hipStream_t s, const View* zbn, const float* coef) {
    if ((C != 1 && C != 3) || dz.C != 64 || dz.H != H / 2 || dz.W != W / 2 || dz.N != NL || (H & 1) || (W & 1) || dz.ps % 4 != 0) {
        fprintf(stderr, "md2: stem_wgrad: unsupported shape C=%d %dx%d\n", C, H, W); abort();
    }
    const int tiles_x = (dz.W + S_TW - 1) / S_TW, tiles_y = (dz.H + 3) / 4, total = tiles_x * tiles_y * NL;
    const int kgroups = (49 * C + 63) / 64;
    const size_t lds = (size_t)std::max(C * 13 * S_PC + 128 * 64, 4 * 16 * 64) * sizeof(float);
    const bool fuse = zbn != nullptr;
    static int per_cu[4] = {0, 0, 0, 0};
    int& pc = per_cu[(C == 3) + 2 * fuse];
    if (!pc) {
        const void* fn = C == 1 ? (fuse ? reinterpret_cast<const void*>(&stem_wgrad_kernel<1, true>) : reinterpret_cast<const void*>(&stem_wgrad_kernel<1, false>))
                                : (fuse ? reinterpret_cast<const void*>(&stem_wgrad_kernel<3, true>) : reinterpret_cast<const void*>(&stem_wgrad_kernel<3, false>));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn, 256, lds) != hipSuccess || pc < 1) pc = 1;
        pc = std::min(pc, 3);
    }
    const int gx = g_deterministic ? 1 : std::max(1, std::min(total, 256 * pc / kgroups));       // persistent: the co-resident blocks, shared by the k groups
    const View zz = fuse ? *zbn : dz;
    if (fuse && (zz.C != 64 || zz.H != dz.H || zz.W != dz.W || zz.N != dz.N || zz.ps % 4 != 0)) { fprintf(stderr, "md2: stem_wgrad: BatchNorm input view mismatch\n"); abort(); }
    if (dz.bf != zz.bf) { fprintf(stderr, "md2: stem_wgrad: dz and the BatchNorm input share a storage type\n"); abort(); }
    if (dz.bf) {
        if (C == 1) { if (fuse) hipLaunchKernelGGL((stem_wgrad_kernel<1, true, true>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef);
                      else hipLaunchKernelGGL((stem_wgrad_kernel<1, false, true>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef); }
        else { if (fuse) hipLaunchKernelGGL((stem_wgrad_kernel<3, true, true>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef);
               else hipLaunchKernelGGL((stem_wgrad_kernel<3, false, true>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef); }
        return;
    }
    if (C == 1) { if (fuse) hipLaunchKernelGGL((stem_wgrad_kernel<1, true>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef);
                  else hipLaunchKernelGGL((stem_wgrad_kernel<1, false>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef); }
    else { if (fuse) hipLaunchKernelGGL((stem_wgrad_kernel<3, true>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef);
           else hipLaunchKernelGGL((stem_wgrad_kernel<3, false>), dim3(gx, kgroups), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef); }
}

}  // namespace md2
