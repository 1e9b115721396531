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
#include <type_traits>
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
// BFM (the bf16 modes, md2_config.precision >= 1): the products run on v_mfma_f32_32x32x16_bf16.  Patch and weights are held in LDS as bf16, and the contraction is
// re-ordered into k-blocks of 8 = one (channel, filter row) with kw = 0 .. 6 and a zero weight for kw = 7: a lane's operand is then 8 CONSECUTIVE patch elements (four dwords
// from an even column) and 16 bytes of its weight row -- 6 LDS reads per 4 MFMAs of 16 k instead of 4 per 4 MFMAs of 2 k, and 7 C / 2 MFMA steps instead of 49 C / 2.
template <int C> struct StemBf { static constexpr int NKB = 7 * C, STEPS = (NKB + 1) / 2, KS = STEPS * 16 + 8; };      // KS: bf16 per weight row in LDS, an odd number of 16-byte groups
template <int C, bool BS = false, bool BFM = false>          // BS: z is a bf16 tensor
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, View z, int H, int W, int Kpad,
                                                       double* __restrict__ stats, int tiles_x, int tiles_y, int total_tiles) {
    constexpr int K = 49 * C, KP = (K + 1) & ~1, TH = 8, PR = 2 * TH + 5, NP = (C * PR * S_PC + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    float* Ps = ssm;                           // [C][PR][72]
    float* Ws = ssm + C * PR * S_PC;           // [KP][96]
    // BFM: the image patch as TWO bf16 planes, hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits together): rounding the image itself to 8 bits puts a per-pixel error of
    // 2^-9 of the intensity on every tap, which is not small beside the local contrast the 7x7 filters respond to (measured: the step's gradient cosine against the fp32 step
    // fell from 0.87 to 0.65 at 416x128); rounding the WEIGHTS only shifts each output channel by a constant times the local mean, which the BatchNorm that follows removes
    unsigned short* Pb = reinterpret_cast<unsigned short*>(ssm);                 // [2][C][PR][72] bf16
    constexpr int PLANE = (C * PR * S_PC + 7) & ~7;
    unsigned short* Wb = Pb + 2 * PLANE;                                         // [64][KS] bf16, k = 8 * (c * 7 + kh) + kw
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
    if constexpr (BFM) {
        constexpr int KS = StemBf<C>::KS;
        for (int f = tid; f < 64 * KS; f += 256) {
            const int co = f / KS, kk = f - co * KS, kb = kk >> 3, kw = kk & 7, c = kb / 7, kh = kb - 7 * c;
            const float v = (kb < StemBf<C>::NKB && kw < 7) ? w[(long)co * Kpad + (kh * 7 + kw) * C + c] : 0.f;
            const __bf16 hv = (__bf16)v; Wb[f] = __builtin_bit_cast(unsigned short, hv);
        }
    } else
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
            for (int i = 0; i < NP; ++i) if (tid + 256 * i < C * PR * S_PC) {
                if constexpr (BFM) {
                    const __bf16 hv = (__bf16)pv[i]; const __bf16 lv = (__bf16)(pv[i] - (float)hv);
                    Pb[tid + 256 * i] = __builtin_bit_cast(unsigned short, hv); Pb[PLANE + tid + 256 * i] = __builtin_bit_cast(unsigned short, lv);
                } else Ps[tid + 256 * i] = pv[i];
            }
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
        if constexpr (BFM) {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            constexpr int KS = StemBf<C>::KS;
            const unsigned* Pd = reinterpret_cast<const unsigned*>(Pb);
#pragma unroll
            for (int st = 0; st < StemBf<C>::STEPS; ++st) {
                const int kb = min(2 * st + half, StemBf<C>::NKB - 1), c = (kb * 37) >> 8, kh = kb - 7 * c;       // (a k-block past the last has zero weights: any valid patch row)
                const int po = ((c * PR + kh + 4 * wave) * S_PC) / 2 + col;                                          // dword index of patch element (row, 2 col)
                const uint4 a0 = *reinterpret_cast<const uint4*>(Wb + col * KS + 16 * st + 8 * half), a1 = *reinterpret_cast<const uint4*>(Wb + (32 + col) * KS + 16 * st + 8 * half);
#pragma unroll
                for (int pl = 1; pl >= 0; --pl) {                                                                     // small plane first
                    const unsigned* Pp = Pd + pl * (PLANE / 2);
                    const uint4 b0 = make_uint4(Pp[po], Pp[po + 1], Pp[po + 2], Pp[po + 3]), b1 = make_uint4(Pp[po + S_PC], Pp[po + S_PC + 1], Pp[po + S_PC + 2], Pp[po + S_PC + 3]);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc[1][1], 0, 0, 0);
                }
            }
        } else
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

template <int C, bool BS, bool BFM>
static void launch_stem_fwd_t(const float* x, int NL, int H, int W, const float* w, int Kpad, const View& z, double* stats, hipStream_t s) {
    const int tiles_x = (z.W + S_TW - 1) / S_TW, tiles_y = (z.H + 7) / 8;
    const int KP = (49 * C + 1) & ~1;
    const size_t lds = BFM ? (size_t)(2 * ((C * 21 * S_PC + 7) & ~7) + 64 * StemBf<C>::KS) * sizeof(unsigned short) : (size_t)(C * 21 * S_PC + KP * 96) * sizeof(float);
    const int total = tiles_x * tiles_y * NL;
    static int pc = 0;                                                  // (one per instantiation)
    const void* fn = reinterpret_cast<const void*>(&stem_fwd_kernel<C, BS, BFM>);
    if (!pc) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn, 256, lds) != hipSuccess || pc < 1) pc = 1;
        pc = std::min(pc, 4);
    }
    const dim3 grid(std::min(total, 256 * pc));                        // persistent: the co-resident blocks
    hipLaunchKernelGGL((stem_fwd_kernel<C, BS, BFM>), grid, dim3(256), lds, s, x, w, z, H, W, Kpad, stats, tiles_x, tiles_y, total);
}
void launch_stem_fwd(const float* x, int NL, int C, int H, int W, const float* w, int Kpad, const View& z, double* stats, hipStream_t s, int bf16_products) {
    if ((C != 1 && C != 3) || z.C != 64 || z.H != H / 2 || z.W != W / 2 || z.N != NL || (H & 1) || (W & 1) || z.ps % 4 != 0) {
        fprintf(stderr, "md2: stem_fwd: unsupported shape C=%d %dx%d\n", C, H, W); abort();
    }
    static const bool no_bfm = md2_tune_set("MD2_STEM_FP32");
    const bool bfm = bf16_products && !no_bfm;
#define STEM_FWD(C_, BS_, BFM_) launch_stem_fwd_t<C_, BS_, BFM_>(x, NL, H, W, w, Kpad, z, stats, s)
    if (C == 1) { if (z.bf) { if (bfm) STEM_FWD(1, true, true); else STEM_FWD(1, true, false); } else { if (bfm) STEM_FWD(1, false, true); else STEM_FWD(1, false, false); } }
    else        { if (z.bf) { if (bfm) STEM_FWD(3, true, true); else STEM_FWD(3, true, false); } else { if (bfm) STEM_FWD(3, false, true); else STEM_FWD(3, false, false); } }
#undef STEM_FWD
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

// The same weight gradient on v_mfma_f32_32x32x16_bf16 (the bf16 modes).  GEMM-K = 16 pixels per MFMA: the dz tile is kept CHANNEL-major in LDS as bf16 ([64 co][4 x 32 px],
// transposed on the way in: a thread loads the same 4 channels of two neighbouring pixels and writes 4 dwords), so a lane's dz operand is one ds_read_b128; the patch
// operand is 8 stride-2 bf16 gathers.  The matrix work per tile drops 16x, so ONE block owns all 49 C columns (2 / 5 k tiles of 32) instead of re-loading dz and the patch
// per 64-column group.
template <int C, bool FUSE, bool BS>
__global__ __launch_bounds__(256, C == 1 ? 2 : 1) void stem_wgrad_bf16_kernel(const float* __restrict__ x, View dz, float* __restrict__ dw, int H, int W, int Kpad,
                                                              int tiles_x, int tiles_y, int total_tiles, View zbn, const float* __restrict__ coef) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int K = 49 * C, TH = 4, PR = 2 * TH + 5, NP = (C * PR * S_PC + 255) / 256, KT = (K + 31) / 32;
    constexpr int YCH = (128 / 8 + 1) * 16;                                     // bytes per channel row of the dz tile
    extern __shared__ __attribute__((aligned(16))) char sbm[];
    unsigned short* Pb = reinterpret_cast<unsigned short*>(sbm);                 // [C][PR][72] bf16
    char* Ys = sbm + ((C * PR * S_PC * 2 + 15) & ~15);                          // [64][YCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
    int ko[KT];                                                                  // this lane's k column of every k tile: element offset of its tap in the patch, at this wave's tile row
#pragma unroll
    for (int t = 0; t < KT; ++t) ko[t] = stem_koff(min(t * 32 + col, K - 1), C, PR) + (2 * wave) * S_PC;
    f32x16 acc[2][KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int pslot[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int f = min(tid + 256 * i, C * PR * S_PC - 1), pc = f % S_PC, r = f / S_PC, pr = r % PR, c = r / PR;
        pslot[i] = (c << 16) | (pr << 8) | pc;
    }
    // dz slots: (pixel pair pp = slot / 16, channel quad q = slot % 16 = tid % 16): 64 x 16 = 1024 -> 4 per thread, two pixels each
    typedef typename std::conditional<BS, uint2, float4>::type LT;              // a loaded channel quad as it sits in memory (bf16 tensors: 8 bytes, widened when it is committed)
    auto ldq = [](const float* base, long off) -> LT { if constexpr (BS) return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off); else return *reinterpret_cast<const float4*>(base + off); };
    auto wide = [](const LT& v) -> float4 { if constexpr (BS) return bf16x4_unpack(v); else return v; };
    float pv[NP]; LT yv[4][2]; LT zv[FUSE ? 4 : 1][2]; unsigned yok = 0;
    const int q = tid & 15;
    float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA, cC = cA;
    if constexpr (FUSE) { cA = *reinterpret_cast<const float4*>(coef + q * 4); cB = *reinterpret_cast<const float4*>(coef + 64 + q * 4); cC = *reinterpret_cast<const float4*>(coef + 128 + q * 4); }
    const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int s = -1; s < my_tiles; ++s) {
        if (s >= 0) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NP; ++i) if (tid + 256 * i < C * PR * S_PC) { const __bf16 hv = (__bf16)pv[i]; Pb[tid + 256 * i] = __builtin_bit_cast(unsigned short, hv); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pp = (tid + 256 * i) >> 4;
                float4 g[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    g[j] = wide(yv[i][j]);
                    if constexpr (FUSE) { const float4 zz = wide(zv[i][j]); g[j] = make_float4(cA.x * g[j].x + cB.x * zz.x + cC.x, cA.y * g[j].y + cB.y * zz.y + cC.y, cA.z * g[j].z + cB.z * zz.z + cC.z, cA.w * g[j].w + cB.w * zz.w + cC.w); }
                    if (!((yok >> (2 * i + j)) & 1u)) g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *reinterpret_cast<unsigned*>(Ys + wb_addr(4 * q + 0, 2 * pp, YCH)) = bf16x2_pack(g[0].x, g[1].x);
                *reinterpret_cast<unsigned*>(Ys + wb_addr(4 * q + 1, 2 * pp, YCH)) = bf16x2_pack(g[0].y, g[1].y);
                *reinterpret_cast<unsigned*>(Ys + wb_addr(4 * q + 2, 2 * pp, YCH)) = bf16x2_pack(g[0].z, g[1].z);
                *reinterpret_cast<unsigned*>(Ys + wb_addr(4 * q + 3, 2 * pp, YCH)) = bf16x2_pack(g[0].w, g[1].w);
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
            for (int i = 0; i < 4; ++i) {
                const int pp = (tid + 256 * i) >> 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int px = 2 * pp + j, oy = ty * TH + (px >> 5), ox = tx * S_TW + (px & 31);
                    yok |= ((oy < dz.H && ox < dz.W) ? 1u : 0u) << (2 * i + j);
                    const long po = ((long)min(oy, dz.H - 1) * dz.W + min(ox, dz.W - 1));
                    yv[i][j] = ldq(dz.p, dn + po * dz.ps + q * 4);
                    if constexpr (FUSE) zv[i][j] = ldq(zbn.p, (long)n * zbn.bs + po * zbn.ps + q * 4);
                }
            }
        }
        if (s < 0) continue;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int pxl = 16 * st + 8 * half;                             // first of this lane half's 8 pixels within the tile row
            const uint4 a0 = *reinterpret_cast<const uint4*>(Ys + wb_addr(col, wave * 32 + pxl, YCH)), a1 = *reinterpret_cast<const uint4*>(Ys + wb_addr(32 + col, wave * 32 + pxl, YCH));
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const unsigned short* pk = Pb + ko[t] + 2 * pxl;
                const uint4 b = make_uint4((unsigned)pk[0] | ((unsigned)pk[2] << 16), (unsigned)pk[4] | ((unsigned)pk[6] << 16), (unsigned)pk[8] | ((unsigned)pk[10] << 16), (unsigned)pk[12] | ((unsigned)pk[14] << 16));
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b), acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b), acc[1][t], 0, 0, 0);
            }
        }
    }
    // reduce the 4 waves through LDS ([2 KT tiles][16 regs][64 lanes]), then one atomic per weight and block
    float* red = reinterpret_cast<float*>(sbm);
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float* dst = red + ((i * KT + j) * 16 + e) * 64 + lane;
                        *dst = wv == 0 ? acc[i][j][e] : *dst + acc[i][j][e];
                    }
        }
    }
    __syncthreads();
    for (int f = tid; f < 2 * KT * 16 * 64; f += 256) {
        const int ln = f & 63, e = (f >> 6) & 15, ij = f >> 10, i = ij / KT, j = ij - i * KT;
        const int co = i * 32 + (e >> 2) * 8 + (ln >> 5) * 4 + (e & 3), k = j * 32 + (ln & 31);      // 32x32 D layout: row = co, col = k
        if (k < K) atomicAdd(dw + (long)co * Kpad + k, red[f]);
    }
}
template <int C, bool FUSE, bool BS>
static void launch_stem_wgrad_bf16_t(const float* x, int NL, int H, int W, const View& dz, float* dw, int Kpad, hipStream_t s, const View& zz, const float* coef) {
    constexpr int KT = (49 * C + 31) / 32;
    const int tiles_x = (dz.W + S_TW - 1) / S_TW, tiles_y = (dz.H + 3) / 4, total = tiles_x * tiles_y * NL;
    const size_t lds = std::max<size_t>(((C * 13 * S_PC * 2 + 15) & ~15) + 64 * ((128 / 8 + 1) * 16), (size_t)2 * KT * 16 * 64 * sizeof(float));
    static int pc = 0;
    const void* fn = reinterpret_cast<const void*>(&stem_wgrad_bf16_kernel<C, FUSE, BS>);
    if (!pc) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn, 256, lds) != hipSuccess || pc < 1) pc = 1;
        pc = std::min(pc, 3);
    }
    const int gx = g_deterministic ? 1 : std::max(1, std::min(total, 256 * pc));
    hipLaunchKernelGGL((stem_wgrad_bf16_kernel<C, FUSE, BS>), dim3(gx), dim3(256), lds, s, x, dz, dw, H, W, Kpad, tiles_x, tiles_y, total, zz, coef);
}

void launch_stem_wgrad(const float* x, int NL, int C, int H, int W, const View& dz, float* dw, int Kpad, hipStream_t s, const View* zbn, const float* coef, int bf16_products) {
    if ((C != 1 && C != 3) || dz.C != 64 || dz.H != H / 2 || dz.W != W / 2 || dz.N != NL || (H & 1) || (W & 1) || dz.ps % 4 != 0) {
        fprintf(stderr, "md2: stem_wgrad: unsupported shape C=%d %dx%d\n", C, H, W); abort();
    }
    static const bool no_bfm = md2_tune_set("MD2_STEM_FP32");
    if (bf16_products && !no_bfm) {
        const View zb = zbn ? *zbn : dz;
        if (zbn && (zb.C != 64 || zb.H != dz.H || zb.W != dz.W || zb.N != dz.N || zb.ps % 4 != 0 || zb.bf != dz.bf)) { fprintf(stderr, "md2: stem_wgrad: BatchNorm input view mismatch\n"); abort(); }
#define STEM_WG(C_, F_, B_) launch_stem_wgrad_bf16_t<C_, F_, B_>(x, NL, H, W, dz, dw, Kpad, s, zb, coef)
        if (C == 1) { if (zbn) { if (dz.bf) STEM_WG(1, true, true); else STEM_WG(1, true, false); } else { if (dz.bf) STEM_WG(1, false, true); else STEM_WG(1, false, false); } }
        else        { if (zbn) { if (dz.bf) STEM_WG(3, true, true); else STEM_WG(3, true, false); } else { if (dz.bf) STEM_WG(3, false, true); else STEM_WG(3, false, false); } }
#undef STEM_WG
        return;
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
