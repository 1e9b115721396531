// layers.hip — the HBM-bound layer kernels around the MFMA convolutions (gfx950).
// Replaces: cuDNN BatchNorm/MaxPool, NNlibCUDA upsample_bilinear, NNlib pad_reflect pullback, Flux ADAM
// broadcasts, GPUArrays mapreduce (SURVEY.md §2a k2,k3,k5,k7,k8,k10,k18,k20).  All kernels are NHWC views,
// 16-B per lane along the channel axis, grid-stride over pixels.
#include "md2_internal.h"
#include <algorithm>

namespace md2 {

// one instantiation per storage type of the activation views (View::bf): fp32 or bf16 elements
#define LAUNCH_BS(bf, kern, ...) do { if (bf) hipLaunchKernelGGL((kern<true>), __VA_ARGS__); else hipLaunchKernelGGL((kern<false>), __VA_ARGS__); } while (0)
#define LAUNCH_BSP(bf, kern, pre, ...) do { if (bf) hipLaunchKernelGGL((kern<pre, true>), __VA_ARGS__); else hipLaunchKernelGGL((kern<pre, false>), __VA_ARGS__); } while (0)
#define MD2_COMMA ,
static inline int grid_for(long work, int per_block, int cap = 2048) { return (int)std::max<long>(1, std::min<long>((work + per_block - 1) / per_block, cap)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// a lane's run of consecutive channels: 4 fp32 (16 B) or, on bf16 tensors, 8 (16 B as well: the row-indexed kernels below are instruction-bound at 8-byte accesses)
template <bool BS> struct LaneN { static constexpr int N = BS ? 8 : 4; };
template <bool BS> __device__ __forceinline__ void vldn(const float* base, long off, float (&v)[LaneN<BS>::N]) {
    if constexpr (BS) {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + off);
        const float4 a = bf16x4_unpack(make_uint2(u.x, u.y)), b = bf16x4_unpack(make_uint2(u.z, u.w));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else { const float4 a = *reinterpret_cast<const float4*>(base + off); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
}
template <bool BS> __device__ __forceinline__ void vstn(float* base, long off, const float (&v)[LaneN<BS>::N]) {
    if constexpr (BS) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + off) = make_uint4(bf16x2_pack(v[0], v[1]), bf16x2_pack(v[2], v[3]), bf16x2_pack(v[4], v[5]), bf16x2_pack(v[6], v[7]));
    else *reinterpret_cast<float4*>(base + off) = make_float4(v[0], v[1], v[2], v[3]);
}
template <int N> __device__ __forceinline__ void ldparams(const float* p, float (&v)[N]) {
#pragma unroll
    for (int e = 0; e < N; e += 4) { const float4 a = *reinterpret_cast<const float4*>(p + e); v[e] = a.x; v[e + 1] = a.y; v[e + 2] = a.z; v[e + 3] = a.w; }
}
static inline bool dense_view_h(const View& v) { return v.ps == v.C && v.bs == (long)v.H * v.W * v.C; }
static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ---- row-indexed element kernels ------------------------------------------------------------------------------------
// The pointwise / stencil kernels below used to decode a flat 64-bit element index per element (i % Q, i / Q, p / HW, r / W: four 64-bit divisions, ~400 vector
// instructions -- more than the 16 + 16 bytes of traffic per lane they serve, so they ran at the vector rate, not at HBM's).  Now blockIdx.x owns RPB consecutive image
// rows (n, h: block-uniform, scalar arithmetic), blockIdx.y a 256-wide chunk of the row's W x (C/4) float4 elements, and a thread keeps its (w, channel quad) for life:
// one 32-bit division per thread, everything per row is uniform.
struct RowGrid { dim3 grid; int rpb; };
static inline RowGrid row_grid(long rows, int rowlen, int want_blocks = 8192, int min_rpb = 1) {          // min_rpb: rows a thread must amortise its per-thread set-up over
    const int chunks = (rowlen + 255) / 256;
    int rpb = (int)std::max<long>(min_rpb, (rows * chunks + want_blocks - 1) / want_blocks);
    rpb = std::min(rpb, 16);
    RowGrid g; g.rpb = rpb; g.grid = dim3((unsigned)((rows + rpb - 1) / rpb), (unsigned)chunks, 1);
    return g;
}

// ---- layout conversion for the op-level entry points --------------------------------------------------------------
template <bool BS>
__global__ void nchw_to_view_kernel(const float* __restrict__ src, View v) {
    const long total = (long)v.N * v.H * v.W * v.C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % v.C; const long r = i / v.C; const long hw = r % ((long)v.H * v.W); const int n = r / ((long)v.H * v.W);
        vst1<BS>(v.p, (long)n * v.bs + hw * v.ps + c, src[((long)n * v.C + c) * v.H * v.W + hw]);
    }
}
template <bool BS>
__global__ void view_to_nchw_kernel(View v, float* __restrict__ dst) {
    const long total = (long)v.N * v.H * v.W * v.C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % ((long)v.H * v.W); const long r = i / ((long)v.H * v.W); const int c = r % v.C; const int n = r / v.C;
        dst[i] = vld1<BS>(v.p, (long)n * v.bs + hw * v.ps + c);
    }
}
void launch_nchw_to_view(const float* src, const View& v, hipStream_t s) {
    LAUNCH_BS(v.bf, nchw_to_view_kernel, dim3(grid_for((long)v.N * v.H * v.W * v.C, 256, 4096)), dim3(256), 0, s, src, v);
}
void launch_nhwc_to_nchw(const float*, const View& v, float* dst, hipStream_t s) {
    LAUNCH_BS(v.bf, view_to_nchw_kernel, dim3(grid_for((long)v.N * v.H * v.W * v.C, 256, 4096)), dim3(256), 0, s, v, dst);
}
void launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) { launch_nchw_to_view(src, make_view(dst, N, H, W, C), s); }

// ---- per-channel statistics ---------------------------------------------------------------------------------------
// block = 256 threads = RG row-groups x (C/4) channel quads (C <= 512 -> RG >= 2).  Each thread strides over pixels,
// LDS-reduces over row groups, then one double atomic per channel per block.
template <int MODE, bool FLAT, bool BS>   // 0: sum z, sum z^2 ; 1: sum g, sum g*xhat  (g = dy * act'(yact)); FLAT: dense NHWC views, pixel p at p*C
__global__ __launch_bounds__(256) void colstats_kernel(View z, View dy, View yact, int has_yact, int act, const float* __restrict__ stats, double* __restrict__ acc, int rows_per_block) {
    const int Qall = z.C / 4;                 // channel quads; more than 256 of them (C = 2048, ResNet-50) are split over blockIdx.y
    const int qoff = blockIdx.y * 256, Q = min(Qall - qoff, 256);
    const int RG = 256 / Q;                   // row groups
    const int q = qoff + threadIdx.x % Q, rg = threadIdx.x / Q;
    __shared__ float red[2 * 4 * 256];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const long P = z.pixels(); const int HW = z.H * z.W;
    const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(p0 + rows_per_block, P);
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f), istd = mean;
    if (MODE == 1) { mean = ld4(stats + q * 4); istd = ld4(stats + z.C + q * 4); }
    if (rg < RG)
#pragma unroll 2
        for (long p = p0 + rg; p < p1; p += RG) {
            const int n = FLAT ? 0 : (int)(p / HW); const long r = FLAT ? p : p - (long)n * HW;      // dense: bs == HW*ps, so (0, p) addresses pixel p
            const float4 v = vld4<BS>(z.p, voff(z, n, r) + q * 4);
            if (MODE == 0) {
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
            } else {
                float4 g = vld4<BS>(dy.p, voff(dy, n, r) + q * 4);
                if (has_yact) { const float4 o = vld4<BS>(yact.p, voff(yact, n, r) + q * 4); g.x *= act_deriv(o.x, act); g.y *= act_deriv(o.y, act); g.z *= act_deriv(o.z, act); g.w *= act_deriv(o.w, act); }
                s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                s2.x += g.x * (v.x - mean.x) * istd.x; s2.y += g.y * (v.y - mean.y) * istd.y; s2.z += g.z * (v.z - mean.z) * istd.z; s2.w += g.w * (v.w - mean.w) * istd.w;
            }
        }
    float* r1 = red; float* r2 = red + 4 * 256;
    r1[0 * 256 + threadIdx.x] = s1.x; r1[1 * 256 + threadIdx.x] = s1.y; r1[2 * 256 + threadIdx.x] = s1.z; r1[3 * 256 + threadIdx.x] = s1.w;
    r2[0 * 256 + threadIdx.x] = s2.x; r2[1 * 256 + threadIdx.x] = s2.y; r2[2 * 256 + threadIdx.x] = s2.z; r2[3 * 256 + threadIdx.x] = s2.w;
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * 4 * Q; c += 256) {                 // this block's 4*Q channels
        const int which = c / (4 * Q), ch = c % (4 * Q), qq = ch / 4, e = ch % 4;
        const float* rr = which ? r2 : r1;
        float s = 0.f;
        for (int g = 0; g < RG; ++g) s += rr[e * 256 + g * Q + qq];
        atomicAdd(acc + which * z.C + qoff * 4 + ch, (double)s);
    }
}
static void colstats_launch(int mode, const View& z, const View* dy, const View* yact, int act, const float* stats, double* acc, hipStream_t s) {
    const int Qh = z.C / 4;
    if (z.C % 4 != 0 || (Qh <= 256 ? 256 % Qh != 0 : Qh % 256 != 0)) { fprintf(stderr, "md2: colstats: unsupported C=%d\n", z.C); abort(); }
    const long P = z.pixels();
    // (a small map -- the deepest stage at batch 4 is 624 pixels -- is a latency chain of its few workgroups' row loops: shorter ones there, 20 -> 8 us)
    const int rpb_min = P < 2048 ? 16 : 64;
    int rpb = 2048; while (rpb > rpb_min && (P + rpb - 1) / rpb < 1024) rpb >>= 1;
    if (g_deterministic) rpb = (int)std::min<long>(P, 1 << 30);          // one workgroup per channel group: one atomic per channel
    const dim3 grid((unsigned)((P + rpb - 1) / rpb), (unsigned)((Qh + 255) / 256));
    View d = dy ? *dy : z, ya = yact ? *yact : z;
    const bool flat = dense_view_h(z) && dense_view_h(d) && dense_view_h(ya);
    if (mode == 0) { if (flat) LAUNCH_BSP(z.bf, colstats_kernel, 0 MD2_COMMA true, grid, dim3(256), 0, s, z, d, ya, 0, 0, stats, acc, rpb); else LAUNCH_BSP(z.bf, colstats_kernel, 0 MD2_COMMA false, grid, dim3(256), 0, s, z, d, ya, 0, 0, stats, acc, rpb); }
    else { if (flat) LAUNCH_BSP(z.bf, colstats_kernel, 1 MD2_COMMA true, grid, dim3(256), 0, s, z, d, ya, yact ? 1 : 0, act, stats, acc, rpb); else LAUNCH_BSP(z.bf, colstats_kernel, 1 MD2_COMMA false, grid, dim3(256), 0, s, z, d, ya, yact ? 1 : 0, act, stats, acc, rpb); }
}
void launch_colstats(const View& z, double* acc, hipStream_t s) { colstats_launch(0, z, nullptr, nullptr, 0, nullptr, acc, s); }
void launch_bn_bwd_reduce(const View& dy, const View* yact, int act, const View& z, const float* stats, double* acc, hipStream_t s) {
    colstats_launch(1, z, &dy, yact, act, stats, acc, s);
}

// running (optional, [mean[C] | var[C]]): Flux BatchNorm's tracked statistics, mu = (1 - mtm) mu + mtm mean, sigma^2 = (1 - mtm) sigma^2 + mtm * m/(m-1) * var
// (Flux `_track_stats!`, momentum 0.1, unbiased correction of the batch variance)
// one channel's batch statistics from its double accumulators: what bn_finalize_kernel writes, shared with the consumers that finalise in their own prologue (BnFin)
struct BnChan { float mean, invstd; double mean_d, var_d; };
__device__ __forceinline__ BnChan bn_finalize_channel(const double* __restrict__ acc, int C, int c, double count, int slots) {
    double s1 = 0, s2 = 0;
    for (int k = 0; k < slots; ++k) { s1 += acc[(size_t)k * 2 * C + c]; s2 += acc[(size_t)k * 2 * C + C + c]; }
    BnChan r;
    r.mean_d = s1 / count;
    r.var_d = s2 / count - r.mean_d * r.mean_d;        // biased variance (Flux BatchNorm train mode)
    if (r.var_d < 0) r.var_d = 0;
    r.mean = (float)r.mean_d;
    r.invstd = (float)(1.0 / sqrt(r.var_d + 1e-5));
    return r;
}
// stats = mean | invstd | scale | shift (y = z * scale + shift == gamma * (z - mean) * invstd + beta: what a consumer's loader applies on the fly); running: Flux `_track_stats!`
__device__ __forceinline__ void bn_store_channel(const BnChan& r, float* __restrict__ stats, int C, int c, double count, float* __restrict__ running, float momentum,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta) {
    stats[c] = r.mean; stats[C + c] = r.invstd;
    if (gamma) { const float sc = gamma[c] * r.invstd; stats[2 * C + c] = sc; stats[3 * C + c] = fmaf(-r.mean, sc, beta[c]); }
    if (running) {
        const float corr = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
        running[c] = (1.f - momentum) * running[c] + momentum * (float)r.mean_d;
        running[C + c] = (1.f - momentum) * running[C + c] + momentum * corr * (float)r.var_d;
    }
}
__global__ void bn_finalize_kernel(const double* __restrict__ acc, float* __restrict__ stats, int C, double count, int slots, float* __restrict__ running, float momentum,
                                   const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    bn_store_channel(bn_finalize_channel(acc, C, c, count, slots), stats, C, c, count, running, momentum, gamma, beta);
}
// Consumer-side finalisation (round 6): a BatchNorm whose FIRST reader is an elementwise kernel (bn_apply, bn_relu_maxpool) has no bn_finalize launch of its own -- every
// workgroup of that reader derives mean | invstd of the channels from the accumulators into LDS (slots x 2 L2-resident doubles per channel), workgroup 0 also writes the
// stats block (and the running statistics) for the later readers: the pullback kernels and the weight-gradient loaders.  Same arithmetic as bn_finalize_kernel.
constexpr int BN_FIN_MAXC = 1024;
__device__ __forceinline__ void bn_fin_prologue(const BnFin& f, int C, const float* __restrict__ gamma, const float* __restrict__ beta, bool writer, float* __restrict__ sm /*[2C]*/) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const BnChan r = bn_finalize_channel(f.acc, C, c, f.count, f.slots);
        sm[c] = r.mean; sm[C + c] = r.invstd;
        if (writer) bn_store_channel(r, f.stats, C, c, f.count, f.running, f.momentum, gamma, beta);
    }
}
void launch_bn_finalize(const double* acc, float* stats, int C, double count, int slots, float* running, float momentum, hipStream_t s, const float* gamma, const float* beta) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, acc, stats, C, count, slots, running, momentum, gamma, beta);
}
// Flux.testmode!: normalise with the tracked statistics, (x - mu) / sqrt(sigma^2 + eps)
__global__ void bn_stats_from_running_kernel(const float* __restrict__ running, float* __restrict__ stats, int C, const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats[c] = running[c];
    stats[C + c] = 1.f / sqrtf(running[C + c] + 1e-5f);
    if (gamma) { const float sc = gamma[c] * stats[C + c]; stats[2 * C + c] = sc; stats[3 * C + c] = beta[c] - stats[c] * sc; }
}
void launch_bn_stats_from_running(const float* running, float* stats, int C, hipStream_t s, const float* gamma, const float* beta) {
    hipLaunchKernelGGL(bn_stats_from_running_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running, stats, C, gamma, beta);
}

// FAST: the channel-quad count Q is a power of two (it divides the grid stride, so a thread keeps ONE quad for its whole life and the
// per-channel constants are loaded once) and every view is dense NHWC (element i of the [pixels][Q] float4 array is at float offset 4 i:
// no per-element 64-bit divisions).  Every BatchNorm of the encoder qualifies; the generic path serves strided views.
// rstats / rgamma / rbeta (optional): the residual is itself a BatchNorm output that is not materialised -- the 1x1 shortcut branch of a down-sampling block:
// y = act(bn(z) + bn_r(res)) in ONE pass instead of bn_r -> buffer -> add
template <bool FAST, bool BS>
__global__ __launch_bounds__(256) void bn_apply_kernel(View z, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       View res, int has_res, int act, View y, const float* __restrict__ rstats, const float* __restrict__ rgamma, const float* __restrict__ rbeta,
                                                       BnFin fin, BnFin rfin) {
    const int Q = z.C / 4; const long total = z.pixels() * Q; const int HW = z.H * z.W;
    if (FAST) {          // a lane keeps one run of N consecutive channels (16 bytes) for life
        constexpr int N = LaneN<BS>::N;
        const int QN = z.C / N; const long totalN = z.pixels() * QN;
        const int q = (int)((blockIdx.x * (long)blockDim.x + threadIdx.x) & (QN - 1));
        float m[N], is[N], g[N], b[N], rm[N], ris[N], rg[N], rb[N];
        __shared__ float sfin[4 * BN_FIN_MAXC];          // mean | invstd of this BatchNorm, then of the residual's (fin / rfin: statistics finalised here, launcher: C <= BN_FIN_MAXC)
        if (fin.acc || rfin.acc) {
            if (fin.acc) bn_fin_prologue(fin, z.C, gamma, beta, blockIdx.x == 0, sfin);
            if (rfin.acc) bn_fin_prologue(rfin, z.C, rgamma, rbeta, blockIdx.x == 0, sfin + 2 * z.C);
            __syncthreads();
        }
        if (fin.acc) { ldparams<N>(sfin + q * N, m); ldparams<N>(sfin + z.C + q * N, is); } else { ldparams<N>(stats + q * N, m); ldparams<N>(stats + z.C + q * N, is); }
        ldparams<N>(gamma + q * N, g); ldparams<N>(beta + q * N, b);
        if (rstats) {
            if (rfin.acc) { ldparams<N>(sfin + 2 * z.C + q * N, rm); ldparams<N>(sfin + 3 * z.C + q * N, ris); } else { ldparams<N>(rstats + q * N, rm); ldparams<N>(rstats + z.C + q * N, ris); }
            ldparams<N>(rgamma + q * N, rg); ldparams<N>(rbeta + q * N, rb);
        }
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < totalN; i += (long)gridDim.x * blockDim.x) {
            float v[N], o[N];
            vldn<BS>(z.p, i * N, v);
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = fmaf(g[e], (v[e] - m[e]) * is[e], b[e]);          // (explicit: with identity statistics this IS the convolution loaders' fmaf(z, scale, shift), whatever -ffp-contract says)
            if (has_res) {
                float rr[N]; vldn<BS>(res.p, i * N, rr);
                if (rstats) {
#pragma unroll
                    for (int e = 0; e < N; ++e) rr[e] = rg[e] * ((rr[e] - rm[e]) * ris[e]) + rb[e];
                }
#pragma unroll
                for (int e = 0; e < N; ++e) o[e] += rr[e];
            }
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = act_apply(o[e], act);
            vstn<BS>(y.p, i * N, o);
        }
        return;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = i % Q; const long p = i / Q; const int n = p / HW; const long r = p - (long)n * HW;
        const float4 v = vld4<BS>(z.p, voff(z, n, r) + q * 4), m = ld4(stats + q * 4), is = ld4(stats + z.C + q * 4), g = ld4(gamma + q * 4), b = ld4(beta + q * 4);
        float4 o = make_float4(fmaf(g.x, (v.x - m.x) * is.x, b.x), fmaf(g.y, (v.y - m.y) * is.y, b.y), fmaf(g.z, (v.z - m.z) * is.z, b.z), fmaf(g.w, (v.w - m.w) * is.w, b.w));
        if (has_res) {
            float4 rr = vld4<BS>(res.p, voff(res, n, r) + q * 4);
            if (rstats) {
                const float4 rm = ld4(rstats + q * 4), ris = ld4(rstats + z.C + q * 4), rg = ld4(rgamma + q * 4), rb = ld4(rbeta + q * 4);
                rr = make_float4(rg.x * ((rr.x - rm.x) * ris.x) + rb.x, rg.y * ((rr.y - rm.y) * ris.y) + rb.y, rg.z * ((rr.z - rm.z) * ris.z) + rb.z, rg.w * ((rr.w - rm.w) * ris.w) + rb.w);
            }
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        o.x = act_apply(o.x, act); o.y = act_apply(o.y, act); o.z = act_apply(o.z, act); o.w = act_apply(o.w, act);
        vst4<BS>(y.p, voff(y, n, r) + q * 4, o);
    }
}
void launch_bn_apply(const View& z, const float* stats, const float* gamma, const float* beta, const View* residual, int act, const View& y, hipStream_t s,
                     const float* rstats, const float* rgamma, const float* rbeta, const BnFin* fin, const BnFin* rfin) {
    const int ln = z.bf ? 8 : 4;                  // channels per lane on the dense path
    const bool fast = z.C % ln == 0 && pow2(z.C / ln) && z.C / ln <= 256 && dense_view_h(z) && dense_view_h(y) && (!residual || dense_view_h(*residual)) && y.bf == z.bf && (!residual || residual->bf == z.bf);
    BnFin f{}, rf{};
    if (fin && (fin->stats != stats || (rfin && rfin->stats != rstats))) { fprintf(stderr, "md2: bn_apply: the statistics finalised here must be the ones applied\n"); abort(); }
    if (fin || rfin) {
        if (fast && z.C <= BN_FIN_MAXC && z.C % 4 == 0) { if (fin) f = *fin; if (rfin) rf = *rfin; }          // finalised in the kernel's prologue
        else {                                                                                                  // shapes off the dense path keep the separate launches
            if (fin) launch_bn_finalize(fin->acc, fin->stats, z.C, fin->count, fin->slots, fin->running, fin->momentum, s, gamma, beta);
            if (rfin) launch_bn_finalize(rfin->acc, rfin->stats, z.C, rfin->count, rfin->slots, rfin->running, rfin->momentum, s, rgamma, rbeta);
        }
    }
    // (finalising blocks re-derive the statistics: fewer, longer-lived blocks, as in bn_bwd_apply)
    const dim3 grid(grid_for(z.pixels() * (z.C / (fast ? ln : 4)), 256, (f.acc || rf.acc) ? 2048 : 4096));
    if (fast) LAUNCH_BSP(z.bf, bn_apply_kernel, true, grid, dim3(256), 0, s, z, stats, gamma, beta, residual ? *residual : z, residual ? 1 : 0, act, y, rstats, rgamma, rbeta, f, rf);
    else LAUNCH_BSP(z.bf, bn_apply_kernel, false, grid, dim3(256), 0, s, z, stats, gamma, beta, residual ? *residual : z, residual ? 1 : 0, act, y, rstats, rgamma, rbeta, f, rf);
}

// sums -> per-channel coefficients, once per BatchNorm (it used to be redone by every thread of the apply kernel): coef = m1 | m2 = sum g / count | sum g*xhat / count;
// also d gamma += sum g*xhat, d beta += sum g.  acc: [slots][sum g | sum g*xhat][C] (a producer's epilogue spreads its atomics over BN_STAT_SLOTS)
// With stats / gamma given, coef[2C..5C) also holds the pullback as an affine map of (g, z): dz = A g + B z + C0, for consumers that form dz in their loader
// (the stem's weight gradient) instead of reading a materialised dz.
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ acc, int slots, int C, double count, float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       const float* __restrict__ stats, const float* __restrict__ gamma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < slots; ++k) { s1 += acc[(size_t)k * 2 * C + c]; s2 += acc[(size_t)k * 2 * C + C + c]; }
    const float m1 = (float)(s1 / count), m2 = (float)(s2 / count);
    coef[c] = m1; coef[C + c] = m2;
    dbeta[c] += (float)s1; dgamma[c] += (float)s2;
    if (stats) {
        const float mean = stats[c], is = stats[C + c], k = gamma[c] * is;
        coef[2 * C + c] = k; coef[3 * C + c] = -k * is * m2; coef[4 * C + c] = k * (mean * is * m2 - m1);
    }
}
void launch_bn_bwd_finalize(const double* acc, int slots, int C, double count, float* coef, float* dgamma, float* dbeta, const float* stats, const float* gamma, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, acc, slots, C, count, coef, dgamma, dbeta, stats, gamma);
}
// FAST also folds bn_bwd_finalize_kernel in: the first Q threads of every block sum the 8 slots of their channel quad (64 L2-resident doubles) into LDS, block 0
// adds d gamma / d beta -- 19 launches of ~5 us fewer per step, which is what a batch-4 step is made of.  coef keeps receiving m1 | m2 (block 0) for callers that read it.
template <bool FAST, bool BS>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(View dy, View yact, int has_yact, int act, View z, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           float* __restrict__ coef, View dz, const double* __restrict__ acc, int slots, double count,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int C = z.C, Q = C / 4; const long total = z.pixels() * Q; const int HW = z.H * z.W;
    if (FAST) {                                 // one run of N channels (16 bytes) per thread: the per-channel constants leave the loop
        constexpr int N = LaneN<BS>::N;
        __shared__ float4 sm1[256], sm2[256];
        if (threadIdx.x < Q) {                  // (per channel quad, whatever N is)
            const int c0 = threadIdx.x * 4;
            double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
            for (int k = 0; k < slots; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += acc[(size_t)k * 2 * C + c0 + e]; s2[e] += acc[(size_t)k * 2 * C + C + c0 + e]; }
            const float4 a1 = make_float4((float)(s1[0] / count), (float)(s1[1] / count), (float)(s1[2] / count), (float)(s1[3] / count));
            const float4 a2 = make_float4((float)(s2[0] / count), (float)(s2[1] / count), (float)(s2[2] / count), (float)(s2[3] / count));
            sm1[threadIdx.x] = a1; sm2[threadIdx.x] = a2;
            if (blockIdx.x == 0) {
                st4(coef + c0, a1); st4(coef + C + c0, a2);
#pragma unroll
                for (int e = 0; e < 4; ++e) { dbeta[c0 + e] += (float)s1[e]; dgamma[c0 + e] += (float)s2[e]; }
            }
        }
        __syncthreads();
        const int QN = C / N; const long totalN = z.pixels() * QN;
        const int q = (int)((blockIdx.x * (long)blockDim.x + threadIdx.x) & (QN - 1));
        float m[N], is[N], ga[N], m1[N], m2[N];
        ldparams<N>(stats + q * N, m); ldparams<N>(stats + C + q * N, is); ldparams<N>(gamma + q * N, ga);
#pragma unroll
        for (int e = 0; e < N; e += 4) { const float4 a = sm1[(q * N + e) >> 2], b = sm2[(q * N + e) >> 2]; m1[e] = a.x; m1[e + 1] = a.y; m1[e + 2] = a.z; m1[e + 3] = a.w; m2[e] = b.x; m2[e + 1] = b.y; m2[e + 2] = b.z; m2[e + 3] = b.w; }
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < totalN; i += (long)gridDim.x * blockDim.x) {
            float g[N], v[N], o[N];
            vldn<BS>(dy.p, i * N, g);
            if (has_yact) {
                float ya[N]; vldn<BS>(yact.p, i * N, ya);
#pragma unroll
                for (int e = 0; e < N; ++e) g[e] *= act_deriv(ya[e], act);
            }
            vldn<BS>(z.p, i * N, v);
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = ga[e] * is[e] * (g[e] - m1[e] - (v[e] - m[e]) * is[e] * m2[e]);
            vstn<BS>(dz.p, i * N, o);
        }
        return;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = i % Q; const long p = i / Q; const int n = p / HW; const long r = p - (long)n * HW;
        float4 g = vld4<BS>(dy.p, voff(dy, n, r) + q * 4);
        if (has_yact) { const float4 o = vld4<BS>(yact.p, voff(yact, n, r) + q * 4); g.x *= act_deriv(o.x, act); g.y *= act_deriv(o.y, act); g.z *= act_deriv(o.z, act); g.w *= act_deriv(o.w, act); }
        const float4 v = vld4<BS>(z.p, voff(z, n, r) + q * 4), m = ld4(stats + q * 4), is = ld4(stats + C + q * 4), ga = ld4(gamma + q * 4), m1 = ld4(coef + q * 4), m2 = ld4(coef + C + q * 4);
        float4 o;
        o.x = ga.x * is.x * (g.x - m1.x - (v.x - m.x) * is.x * m2.x);
        o.y = ga.y * is.y * (g.y - m1.y - (v.y - m.y) * is.y * m2.y);
        o.z = ga.z * is.z * (g.z - m1.z - (v.z - m.z) * is.z * m2.z);
        o.w = ga.w * is.w * (g.w - m1.w - (v.w - m.w) * is.w * m2.w);
        vst4<BS>(dz.p, voff(dz, n, r) + q * 4, o);
    }
}
void launch_bn_bwd_apply(const View& dy, const View* yact, int act, const View& z, const float* stats, const float* gamma, const double* acc, double count,
                         const View& dz, float* dgamma, float* dbeta, hipStream_t s, int slots, float* coef) {
    const int ln = z.bf ? 8 : 4;                  // channels per lane on the dense path
    const bool fast = z.C % ln == 0 && pow2(z.C / ln) && z.C / 4 <= 256 && dense_view_h(z) && dense_view_h(dy) && dense_view_h(dz) && (!yact || dense_view_h(*yact)) && dy.bf == z.bf && dz.bf == z.bf && (!yact || yact->bf == z.bf);
    // (fast: every block re-derives its coefficients -- fewer, longer-lived blocks than the 4096 of the plain element kernels)
    const dim3 grid(grid_for(z.pixels() * (z.C / (fast ? ln : 4)), 256, fast ? 2048 : 4096));
    if (fast) LAUNCH_BSP(z.bf, bn_bwd_apply_kernel, true, grid, dim3(256), 0, s, dy, yact ? *yact : dy, yact ? 1 : 0, act, z, stats, gamma, coef, dz, acc, slots, count, dgamma, dbeta);
    else {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((z.C + 255) / 256), dim3(256), 0, s, acc, slots, z.C, count, coef, dgamma, dbeta, (const float*)nullptr, (const float*)nullptr);
        LAUNCH_BSP(z.bf, bn_bwd_apply_kernel, false, grid, dim3(256), 0, s, dy, yact ? *yact : dy, yact ? 1 : 0, act, z, stats, gamma, coef, dz, acc, slots, count, dgamma, dbeta);
    }
}

// ---- MaxPool((3,3); pad=1, stride=2): first maximum in (kh,kw) scan order wins; the argmax tap (0..8) is kept ------------
template <bool BS>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(View x, View y, uint8_t* __restrict__ arg) {
    const int Q = x.C / 4; const long total = y.pixels() * Q; const int HWo = y.H * y.W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = i % Q; const long p = i / Q; const int n = p / HWo; const int r = p - (long)n * HWo; const int oh = r / y.W, ow = r - oh * y.W;
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}; uint8_t bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh; if (ih < 0 || ih >= x.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw; if (iw < 0 || iw >= x.W) continue;
                const float4 v = vld4<BS>(x.p, voff(x, n, (long)ih * x.W + iw) + q * 4);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) if (vv[e] > best[e]) { best[e] = vv[e]; bi[e] = kh * 3 + kw; }
            }
        }
        vst4<BS>(y.p, voff(y, n, r) + q * 4, make_float4(best[0], best[1], best[2], best[3]));
        *reinterpret_cast<uchar4*>(arg + (p * x.C + q * 4)) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}
void launch_maxpool_fwd(const View& x, const View& y, uint8_t* arg, hipStream_t s) {
    LAUNCH_BS(x.bf, maxpool_fwd_kernel, dim3(grid_for(y.pixels() * (x.C / 4), 256, 4096)), dim3(256), 0, s, x, y, arg);
}
// Stem tail fused: a = relu(bn(z)) is never read back -- one pass computes the 3x3 / stride-2 max-pool of a on the fly from z, and writes a itself only for the
// frames the depth decoder's skip connection needs (n % L == tgt; all frames when L == 1).  Replaces bn_apply (read z, write a) + maxpool_fwd (read a).
template <bool BS>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(View z, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              View a, int L, int tgt, View y, uint8_t* __restrict__ arg, int rpb, BnFin fin) {
    constexpr int N = LaneN<BS>::N;
    const int Q = z.C / N, idx = blockIdx.y * 256 + threadIdx.x;
    __shared__ float sfin[2 * 256];          // (fin: the stem BatchNorm's statistics are finalised here, launcher: C <= 256)
    if (fin.acc) { bn_fin_prologue(fin, z.C, gamma, beta, blockIdx.x == 0 && blockIdx.y == 0, sfin); __syncthreads(); }
    if (idx >= y.W * Q) return;
    const int ow = idx / Q, q = idx - ow * Q;
    float m[N], is[N], g[N], b[N];
    if (fin.acc) { ldparams<N>(sfin + q * N, m); ldparams<N>(sfin + z.C + q * N, is); } else { ldparams<N>(stats + q * N, m); ldparams<N>(stats + z.C + q * N, is); }
    ldparams<N>(gamma + q * N, g); ldparams<N>(beta + q * N, b);
    const int rows = y.N * y.H, row_end = min(rows, (int)(blockIdx.x + 1) * rpb);
    for (int row = blockIdx.x * rpb; row < row_end; ++row) {
        const int n = row / y.H, oh = row - n * y.H;
        const bool wa = (L == 1) || (n % L) == tgt;
        float best[N]; uint8_t bi[N];
#pragma unroll
        for (int e = 0; e < N; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh; if (ih < 0 || ih >= z.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw; if (iw < 0 || iw >= z.W) continue;
                float vv[N];
                vldn<BS>(z.p, voff(z, n, (long)ih * z.W + iw) + q * N, vv);
#pragma unroll
                for (int e = 0; e < N; ++e) { vv[e] = g[e] * ((vv[e] - m[e]) * is[e]) + b[e]; vv[e] = vv[e] > 0.f ? vv[e] : 0.f; }
                if constexpr (BS) {          // the pooled maximum is taken over the values the activation tensor holds (rounded), so that it equals the stored a at its arg-max
#pragma unroll
                    for (int e = 0; e < N; e += 2) { const unsigned u = bf16x2_pack(vv[e], vv[e + 1]); vv[e] = __uint_as_float(u << 16); vv[e + 1] = __uint_as_float(u & 0xFFFF0000u); }
                }
#pragma unroll
                for (int e = 0; e < N; ++e) if (vv[e] > best[e]) { best[e] = vv[e]; bi[e] = kh * 3 + kw; }
                if (wa && kh >= 1 && kw >= 1) vstn<BS>(a.p, voff(a, n, (long)ih * z.W + iw) + q * N, vv);   // this output owns the 2x2 block (2oh.., 2ow..)
            }
        }
        const long r = (long)oh * y.W + ow;
        vstn<BS>(y.p, voff(y, n, r) + q * N, best);
        uint8_t* ap = arg + (((long)n * y.H * y.W + r) * z.C + q * N);
        *reinterpret_cast<uchar4*>(ap) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
        if constexpr (N == 8) *reinterpret_cast<uchar4*>(ap + 4) = make_uchar4(bi[4], bi[5], bi[6], bi[7]);
    }
}
void launch_bn_relu_maxpool(const View& z, const float* stats, const float* gamma, const float* beta, const View& a, int L, int tgt, const View& y, uint8_t* arg, hipStream_t s, const BnFin* fin) {
    const int ln = z.bf ? 8 : 4;                  // channels per lane
    if ((z.H & 1) || (z.W & 1) || y.H != z.H / 2 || y.W != z.W / 2 || z.C % ln != 0 || (256 % (z.C / ln)) != 0 || a.bf != z.bf || y.bf != z.bf) { fprintf(stderr, "md2: bn_relu_maxpool: unsupported shape\n"); abort(); }
    BnFin f{};
    if (fin && fin->stats != stats) { fprintf(stderr, "md2: bn_relu_maxpool: the statistics finalised here must be the ones applied\n"); abort(); }
    if (fin) { if (z.C <= 256) f = *fin; else launch_bn_finalize(fin->acc, fin->stats, z.C, fin->count, fin->slots, fin->running, fin->momentum, s, gamma, beta); }
    const RowGrid rg = row_grid((long)y.N * y.H, y.W * (z.C / ln), f.acc ? 4096 : 8192);
    LAUNCH_BS(z.bf, bn_relu_maxpool_kernel, rg.grid, dim3(256), 0, s, z, stats, gamma, beta, a, L, tgt, y, arg, rg.rpb, f);
}

// gather form: input pixel (ih,iw) is covered by outputs oh in {(ih+1)/2 - (0|1)}...; no atomics, deterministic
// yrelu: the pooled OUTPUT of a max-pool over post-ReLU activations; the ReLU derivative of the winning input equals (pooled value > 0),
// so the mask is read from the 4x smaller pooled tensor instead of the activations themselves
// bz / bstat / bacc (optional): dx is the gradient of a BatchNorm output whose input is bz -- the kernel also accumulates that BatchNorm's pullback sums
// (sum g, sum g*xhat per channel; needs 256 % (C/4) == 0 so that a thread keeps one channel quad) and the separate reduction pass over (g, z) goes away
template <bool BS>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(View dy, const uint8_t* __restrict__ arg, View dx, int accumulate, View yrelu, int has_yrelu,
                                                          View bz, const float* __restrict__ bstat, double* __restrict__ bacc, int acc_frame, int rpb) {
    constexpr int N = LaneN<BS>::N;
    const int Q = dx.C / N, idx = blockIdx.y * 256 + threadIdx.x;
    const bool live = idx < dx.W * Q;
    const int iw = live ? idx / Q : 0, q = live ? idx - iw * Q : 0;
    float s1[N], s2[N], mu[N], is[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { s1[e] = 0.f; s2[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; }
    if (bacc) { ldparams<N>(bstat + q * N, mu); ldparams<N>(bstat + dx.C + q * N, is); }
    // the (at most two) pooled columns whose 3x3 window covers input column iw, with the window tap kw each sees it through: fixed for the thread's life
    int ows[2], kws[2], nw = 0;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) { const int u = iw + 1 - kw; if (u >= 0 && !(u & 1) && (u >> 1) < dy.W) { ows[nw] = u >> 1; kws[nw] = kw; ++nw; } }
    const int rows = dx.N * dx.H, row_end = min(rows, (int)(blockIdx.x + 1) * rpb);
    const long HWo = (long)dy.H * dy.W;
    if (live)
    for (int row = blockIdx.x * rpb; row < row_end; ++row) {
        const int n = row / dx.H, ih = row - n * dx.H;
        float g[N];
#pragma unroll
        for (int e = 0; e < N; ++e) g[e] = 0.f;
        for (int kh = 0; kh < 3; ++kh) {
            const int t = ih + 1 - kh; if (t < 0 || (t & 1)) continue; const int oh = t >> 1; if (oh >= dy.H) continue;          // (block-uniform)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b >= nw) break;
                const long ro = (long)oh * dy.W + ows[b];
                const uint8_t* ap = arg + (((long)n * HWo + ro) * dx.C + q * N);
                uint8_t av[N];
                { const uchar4 a0 = *reinterpret_cast<const uchar4*>(ap); av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w;
                  if constexpr (N == 8) { const uchar4 a1 = *reinterpret_cast<const uchar4*>(ap + 4); av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w; } }
                float d[N];
                vldn<BS>(dy.p, voff(dy, n, ro) + q * N, d);
                if (has_yrelu) {
                    float o[N]; vldn<BS>(yrelu.p, voff(yrelu, n, ro) + q * N, o);
#pragma unroll
                    for (int e = 0; e < N; ++e) d[e] *= act_deriv(o[e], ACT_RELU);
                }
                const int tap = kh * 3 + kws[b];
#pragma unroll
                for (int e = 0; e < N; ++e) if (av[e] == tap) g[e] += d[e];
            }
        }
        const long r = (long)ih * dx.W + iw;
        const long dpo = voff(dx, n, r) + q * N;
        if (accumulate == 1 || (accumulate == 2 && (n % 3) == acc_frame)) {      // earlier contributions are already masked
            float o[N]; vldn<BS>(dx.p, dpo, o);
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] += o[e];
        }
        if constexpr (BS) {                                                      // the sums below see the stored (rounded) gradient
#pragma unroll
            for (int e = 0; e < N; e += 2) { const unsigned u = bf16x2_pack(g[e], g[e + 1]); g[e] = __uint_as_float(u << 16); g[e + 1] = __uint_as_float(u & 0xFFFF0000u); }
        }
        vstn<BS>(dx.p, dpo, g);
        if (bacc) {
            float z[N]; vldn<BS>(bz.p, voff(bz, n, r) + q * N, z);
#pragma unroll
            for (int e = 0; e < N; ++e) { s1[e] += g[e]; s2[e] += g[e] * ((z[e] - mu[e]) * is[e]); }
        }
    }
    if (bacc) {                                   // block reduction over the 256 / Q threads that share a channel group (256 % Q == 0: thread t holds group t % Q), then one double atomic per channel and block
        __shared__ float red[2 * N * 256];
#pragma unroll
        for (int e = 0; e < N; ++e) { red[e * 256 + threadIdx.x] = s1[e]; red[(N + e) * 256 + threadIdx.x] = s2[e]; }
        __syncthreads();
        const int C = dx.C, RG = 256 / Q;
        const int slot = (blockIdx.x + blockIdx.y) % BN_STAT_SLOTS;
        for (int c = threadIdx.x; c < 2 * C; c += 256) {
            const int which = c / C, ch = c % C, qq = ch / N, e = ch % N;
            float t = 0.f;
            for (int gI = 0; gI < RG; ++gI) t += red[(which * N + e) * 256 + gI * Q + qq];
            atomicAdd(bacc + (size_t)slot * 2 * C + which * C + ch, (double)t);
        }
    }
}
void launch_maxpool_bwd(const View& dy, const uint8_t* arg, const View& dx, int accumulate, const View* yrelu, hipStream_t s, const View* bz, const float* bstat, double* bacc, int acc_frame) {
    const int Q = dx.C / (dx.bf ? 8 : 4);         // channel groups: a lane's 16 bytes
    if (dx.C % (dx.bf ? 8 : 4) != 0 || dy.bf != dx.bf || (yrelu && yrelu->bf != dx.bf) || (bz && bz->bf != dx.bf)) { fprintf(stderr, "md2: maxpool_bwd: unsupported views\n"); abort(); }
    if (bacc && (256 % Q != 0 || !bz)) { fprintf(stderr, "md2: maxpool_bwd: the fused BatchNorm sums need 256 %% (channel groups) == 0\n"); abort(); }
    const RowGrid rg = row_grid((long)dx.N * dx.H, dx.W * Q);          // (with the fused sums every block ends in 2 C double atomics: ~8192 blocks measured best, B=32)
    LAUNCH_BS(dx.bf, maxpool_bwd_kernel, rg.grid, dim3(256), 0, s, dy, arg, dx, accumulate, yrelu ? *yrelu : dy, yrelu ? 1 : 0,
                       bz ? *bz : dx, bstat, bacc, acc_frame, rg.rpb);
}

// ---- NNlib.upsample_bilinear(x, (2,2)) (src/depth_decoder.jl:18-19), align-corners mapping --------------------------------
__device__ __forceinline__ void up_src(int o, int in, int out, int& i0, int& i1, float& w1) {
    const float ratio = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float s = ratio * (float)o;
    i0 = min((int)floorf(s), in - 1);
    i1 = min(i0 + 1, in - 1);
    w1 = s - (float)i0;
}
// sk (optional): the encoder skip connection that shares y's channel axis behind x's channels -- cat(upsample(x), skip) of src/depth_decoder.jl:60-61 in ONE pass
// (y is the whole concat buffer then: C = x.C + sk.C; the skip quads of a row are plain copies)
template <bool BS>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(View x, View y, int rpb, View sk, int has_sk) {
    constexpr int N = LaneN<BS>::N;
    const int Qx = x.C / N, Q = has_sk ? Qx + sk.C / N : Qx, idx = blockIdx.y * 256 + threadIdx.x;
    if (idx >= y.W * Q) return;
    const int ow = idx / Q, q = idx - ow * Q;
    const int rows = y.N * y.H, row_end = min(rows, (int)(blockIdx.x + 1) * rpb);
    if (q >= Qx) {
        for (int row = blockIdx.x * rpb; row < row_end; ++row) {
            const int n = row / y.H, oh = row - n * y.H;
            float v[N]; vldn<BS>(sk.p, voff(sk, n, (long)oh * y.W + ow) + (q - Qx) * N, v);
            vstn<BS>(y.p, voff(y, n, (long)oh * y.W + ow) + q * N, v);
        }
        return;
    }
    int w0, w1; float fw; up_src(ow, x.W, y.W, w0, w1, fw);          // the column taps are the thread's for life
    const float gw = 1.f - fw;
    for (int row = blockIdx.x * rpb; row < row_end; ++row) {
        const int n = row / y.H, oh = row - n * y.H;
        int h0, h1; float fh; up_src(oh, x.H, y.H, h0, h1, fh);      // the row taps are block-uniform
        float a[N], b[N], c[N], d[N], o[N];
        vldn<BS>(x.p, voff(x, n, (long)h0 * x.W + w0) + q * N, a); vldn<BS>(x.p, voff(x, n, (long)h0 * x.W + w1) + q * N, b);
        vldn<BS>(x.p, voff(x, n, (long)h1 * x.W + w0) + q * N, c); vldn<BS>(x.p, voff(x, n, (long)h1 * x.W + w1) + q * N, d);
        const float gh = 1.f - fh;
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = gh * (gw * a[e] + fw * b[e]) + fh * (gw * c[e] + fw * d[e]);
        vstn<BS>(y.p, voff(y, n, (long)oh * y.W + ow) + q * N, o);
    }
}
void launch_upsample2x_fwd(const View& x, const View& y, hipStream_t s, const View* skip) {
    const int ln = x.bf ? 8 : 4;
    if (x.C % ln != 0 || y.bf != x.bf) { fprintf(stderr, "md2: upsample2x_fwd: unsupported views\n"); abort(); }
    if (skip && (skip->C % ln != 0 || skip->bf != x.bf || y.C != x.C + skip->C || skip->H != y.H || skip->W != y.W || skip->N != y.N)) { fprintf(stderr, "md2: upsample2x_fwd: the skip connection does not fit the concat buffer\n"); abort(); }
    const RowGrid rg = row_grid((long)y.N * y.H, y.W * ((x.C + (skip ? skip->C : 0)) / ln));
    LAUNCH_BS(x.bf, upsample2x_fwd_kernel, rg.grid, dim3(256), 0, s, x, y, rg.rpb, skip ? *skip : x, skip ? 1 : 0);
}
// exact adjoint in gather form: every candidate output row/col recomputes the forward's (i0,i1,w1)
// candidates of one axis: outputs o with a source index in (i-1, i+1), i.e. o in [(i-1) r, (i+1) r], r = (out-1)/(in-1); their adjoint weights (i0 == i ? 1 - f : 0) + (i1 == i ? f : 0).
// (a candidate just outside the open interval has weight 0 or a rounding-sized one: no +-1 safety margin.)  At most UP_MAXC candidates carry weight for ratios <= 2.
constexpr int UP_MAXC = 6;
__device__ __forceinline__ int up_adjoint_taps(int i, int in, int out, int (&os)[UP_MAXC], float (&cs)[UP_MAXC]) {
    const float r = in > 1 ? (float)(out - 1) / (float)(in - 1) : 0.f;
    const int lo = max(0, (int)floorf((i - 1) * r)), hi = min(out - 1, (int)ceilf((i + 1) * r));
    // slot k <-> candidate lo + k (static indexing keeps the arrays in registers); zero-weight candidates stay in the list with weight 0
    int n = 0;
#pragma unroll
    for (int k = 0; k < UP_MAXC; ++k) {
        const int o = lo + k;
        int i0, i1; float f; up_src(min(o, out - 1), in, out, i0, i1, f);
        const float c = o <= hi ? (i0 == i ? 1.f - f : 0.f) + (i1 == i ? f : 0.f) : 0.f;
        os[k] = min(o, out - 1); cs[k] = c;
        if (o <= hi) n = k + 1;
    }
    return n;
}
// the same taps compacted to FOUR slots: for ratios <= 2 the outputs with a non-zero weight span at most four consecutive indices, starting at candidate 0, 1 or 2 of the window
// above (checked exhaustively for in = 2 .. 1024, out = 2 in); unused slots carry weight 0 and a valid (clamped) index, so a consumer can issue all its loads without a branch
__device__ __forceinline__ void up_adjoint_taps4(int i, int in, int out, int (&os)[4], float (&cs)[4]) {
    int o6[UP_MAXC]; float c6[UP_MAXC];
    (void)up_adjoint_taps(i, in, out, o6, c6);
    const int s0 = c6[0] != 0.f ? 0 : (c6[1] != 0.f ? 1 : 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        os[k] = s0 == 0 ? o6[k] : (s0 == 1 ? o6[k + 1] : o6[k + 2]);
        cs[k] = s0 == 0 ? c6[k] : (s0 == 1 ? c6[k + 1] : c6[k + 2]);
    }
}
template <bool BS>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(View dy, View dx, View xact, int has_xact, int act, int rpb) {
    const int Q = dx.C / 4, idx = blockIdx.y * 256 + threadIdx.x;
    if (idx >= dx.W * Q) return;
    const int w = idx / Q, q = idx - w * Q;
    int ows[UP_MAXC]; float cws[UP_MAXC];
    const int nw = up_adjoint_taps(w, dx.W, dy.W, ows, cws);          // column taps: once per thread
    const int rows = dx.N * dx.H, row_end = min(rows, (int)(blockIdx.x + 1) * rpb);
    for (int row = blockIdx.x * rpb; row < row_end; ++row) {
        const int n = row / dx.H, h = row - n * dx.H;
        int ohs[UP_MAXC]; float chs[UP_MAXC];
        const int nh = up_adjoint_taps(h, dx.H, dy.H, ohs, chs);      // row taps: block-uniform
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < UP_MAXC; ++a) {
            if (a >= nh) break;
#pragma unroll
            for (int b = 0; b < UP_MAXC; ++b) {
                if (b >= nw) break;
                const float4 d = vld4<BS>(dy.p, voff(dy, n, (long)ohs[a] * dy.W + ows[b]) + q * 4);
                const float c = chs[a] * cws[b];
                g.x += c * d.x; g.y += c * d.y; g.z += c * d.z; g.w += c * d.w;
            }
        }
        const long r = (long)h * dx.W + w;
        if (has_xact) { const float4 o = vld4<BS>(xact.p, voff(xact, n, r) + q * 4); g.x *= act_deriv(o.x, act); g.y *= act_deriv(o.y, act); g.z *= act_deriv(o.z, act); g.w *= act_deriv(o.w, act); }
        vst4<BS>(dx.p, voff(dx, n, r) + q * 4, g);
    }
}
void launch_upsample2x_bwd(const View& dy, const View& dx, const View* xact, int act, hipStream_t s) {
    if (dy.H > 2 * dx.H || dy.W > 2 * dx.W) { fprintf(stderr, "md2: upsample2x_bwd: ratio above 2\n"); abort(); }
    const RowGrid rg = row_grid((long)dx.N * dx.H, dx.W * (dx.C / 4));
    LAUNCH_BS(dx.bf, upsample2x_bwd_kernel, rg.grid, dim3(256), 0, s, dy, dx, xact ? *xact : dx, xact ? 1 : 0, act, rg.rpb);
}

// fold_reflect followed by the adjoint of the bilinear x2 (the two passes between a DecoderBlock's second convolution and its first on the way back, src/depth_decoder.jl:18-19,5):
//   dx[h][w] = act'(xact) * sum over fine pixels (oh, ow) of the x2 adjoint weights * fold(sp)[oh][ow],   fold(sp)[oh][ow] = sum of the padded positions that mirror onto it.
// One gather pass over the padded gradient instead of writing and re-reading the folded full-resolution tensor (109 + 109 + 109 MB for the last branch at batch 32).
template <bool BS>
__device__ __forceinline__ void fold_up2_bwd_body(const View& sp, const View& dx, const View& xact, int has_xact, int act, int rpb, int bx, int by) {
    constexpr int N = LaneN<BS>::N;
    const int Q = dx.C / N, idx = by * 256 + threadIdx.x;
    if (idx >= dx.W * Q) return;
    const int w = idx / Q, q = idx - w * Q;
    const int Hf = sp.H - 2, Wf = sp.W - 2;         // the fine (unpadded) map
    int ows[4]; float cws[4];
    up_adjoint_taps4(w, dx.W, Wf, ows, cws);
    // a second padded column folds onto fine columns 1 and Wf - 2 (pad_reflect mirrors WITHOUT the edge): rare, handled after the branch-free 4 x 4 gather
    bool colx = false;
#pragma unroll
    for (int b = 0; b < 4; ++b) colx = colx || (cws[b] != 0.f && (ows[b] == 1 || ows[b] == Wf - 2));
    const int rows = dx.N * dx.H, row_end = min(rows, (bx + 1) * rpb);
    for (int row = bx * rpb; row < row_end; ++row) {
        const int n = row / dx.H, h = row - n * dx.H;
        int ohs[4]; float chs[4];
        up_adjoint_taps4(h, dx.H, Hf, ohs, chs);      // block-uniform
        float g[N];
#pragma unroll
        for (int e = 0; e < N; ++e) g[e] = 0.f;
        // 16 independent gathers, all in flight together (unused slots: weight 0, clamped index)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float d[N]; vldn<BS>(sp.p, voff(sp, n, (long)(ohs[a] + 1) * sp.W + ows[b] + 1) + q * N, d);
                const float c = chs[a] * cws[b];
#pragma unroll
                for (int e = 0; e < N; ++e) g[e] += c * d[e];
            }
        bool rowx = false;
#pragma unroll
        for (int a = 0; a < 4; ++a) rowx = rowx || (chs[a] != 0.f && (ohs[a] == 1 || ohs[a] == Hf - 2));
        if (rowx || colx) {                            // the mirrored frame: extra padded rows / columns (and the corner where both meet)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oy2 = chs[a] == 0.f ? -1 : (ohs[a] == 1 ? 0 : (ohs[a] == Hf - 2 ? Hf + 1 : -1));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int ox2 = cws[b] == 0.f ? -1 : (ows[b] == 1 ? 0 : (ows[b] == Wf - 2 ? Wf + 1 : -1));
                    const float c = chs[a] * cws[b];
                    float d[N];
                    if (oy2 >= 0) { vldn<BS>(sp.p, voff(sp, n, (long)oy2 * sp.W + ows[b] + 1) + q * N, d);
#pragma unroll
                        for (int e = 0; e < N; ++e) g[e] += c * d[e]; }
                    if (ox2 >= 0) { vldn<BS>(sp.p, voff(sp, n, (long)(ohs[a] + 1) * sp.W + ox2) + q * N, d);
#pragma unroll
                        for (int e = 0; e < N; ++e) g[e] += c * d[e]; }
                    if (oy2 >= 0 && ox2 >= 0) { vldn<BS>(sp.p, voff(sp, n, (long)oy2 * sp.W + ox2) + q * N, d);
#pragma unroll
                        for (int e = 0; e < N; ++e) g[e] += c * d[e]; }
                }
            }
        }
        const long r = (long)h * dx.W + w;
        if (has_xact) {
            float o[N]; vldn<BS>(xact.p, voff(xact, n, r) + q * N, o);
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] *= act_deriv(o[e], act);
        }
        vstn<BS>(dx.p, voff(dx, n, r) + q * N, g);
    }
}
template <bool BS>
__global__ __launch_bounds__(256) void fold_up2_bwd_kernel(View sp, View dx, View xact, int has_xact, int act, int rpb) {
    fold_up2_bwd_body<BS>(sp, dx, xact, has_xact, act, rpb, (int)blockIdx.x, (int)blockIdx.y);
}
void launch_fold_up2_bwd(const View& src_pad, const View& dx, const View* xact, int act, hipStream_t s) {
    const int Hf = src_pad.H - 2, Wf = src_pad.W - 2;
    // (maps of at least 4 x 4: on a 3-wide map both frame columns mirror onto column 1, which the one-extra-position bookkeeping above does not cover)
    if (Hf < 4 || Wf < 4 || Hf > 2 * dx.H || Wf > 2 * dx.W || src_pad.C < dx.C) { fprintf(stderr, "md2: fold_up2_bwd: unsupported shape\n"); abort(); }
    if (dx.C % (dx.bf ? 8 : 4) != 0 || src_pad.bf != dx.bf || (xact && xact->bf != dx.bf)) { fprintf(stderr, "md2: fold_up2_bwd: unsupported views\n"); abort(); }
    const RowGrid rg = row_grid((long)dx.N * dx.H, dx.W * (dx.C / (dx.bf ? 8 : 4)));          // (more rows per thread were slower: the kernel lives on parallel gathers, 0.224 -> 0.306 ms at four rows)
    LAUNCH_BS(dx.bf, fold_up2_bwd_kernel, rg.grid, dim3(256), 0, s, src_pad, dx, xact ? *xact : dx, xact ? 1 : 0, act, rg.rpb);
}

template <bool BS>
__device__ __forceinline__ void fold_reflect_body(const View& sp, const View& dst, int accumulate, const View& actsrc, int has_act, int act, int rpb, int bx, int by) {
    constexpr int N = LaneN<BS>::N;
    const int Q = dst.C / N, H = dst.H, W = dst.W, idx = by * 256 + threadIdx.x;
    if (idx >= W * Q) return;
    const int w = idx / Q, q = idx - w * Q;
    // padded-domain columns / rows that fold onto column w / row h (pad_reflect mirrors WITHOUT the edge: -1 <-> 1, W <-> W - 2): the position itself, then the two
    // frame positions where they apply -- as scalars selected by the unrolled loop index (indexed local arrays went to scratch in the bf16 instantiation)
    constexpr int NONE = -0x40000000;
    const int ux1 = w == 1 ? -1 : NONE, ux2 = w == W - 2 ? W : NONE;
    const int rows = dst.N * H, row_end = min(rows, (bx + 1) * rpb);
    for (int row = bx * rpb; row < row_end; ++row) {
        const int n = row / H, h = row - n * H;
        const int uy1 = h == 1 ? -1 : NONE, uy2 = h == H - 2 ? H : NONE;
        float g[N];
#pragma unroll
        for (int e = 0; e < N; ++e) g[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int uy = a == 0 ? h : (a == 1 ? uy1 : uy2);
            if (uy == NONE) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int ux = b == 0 ? w : (b == 1 ? ux1 : ux2);
                if (ux == NONE) continue;
                float d[N]; vldn<BS>(sp.p, voff(sp, n, (long)(uy + 1) * sp.W + (ux + 1)) + q * N, d);
#pragma unroll
                for (int e = 0; e < N; ++e) g[e] += d[e];
            }
        }
        const long r = (long)h * W + w;
        const long dpo = voff(dst, n, r) + q * N;
        if (accumulate) {
            float o[N]; vldn<BS>(dst.p, dpo, o);
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] += o[e];
        }
        if (has_act) {
            float o[N]; vldn<BS>(actsrc.p, voff(actsrc, n, r) + q * N, o);
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] *= act_deriv(o[e], act);
        }
        vstn<BS>(dst.p, dpo, g);
    }
}
template <bool BS>
__global__ __launch_bounds__(256) void fold_reflect_kernel(View sp, View dst, int accumulate, View actsrc, int has_act, int act, int rpb) {
    fold_reflect_body<BS>(sp, dst, accumulate, actsrc, has_act, act, rpb, (int)blockIdx.x, (int)blockIdx.y);
}
// The two adjoints that split a DecoderBlock's padded concat gradient (src/depth_decoder.jl:60-61 backwards) in ONE launch: workgroups [0, nbx_up) gather the upsampled
// half's channels into the coarse map (fold_up2_bwd), the rest fold the skip connection's channels onto the encoder feature's gradient (fold_reflect) -- they read
// disjoint channel slices of the same tensor and write different ones, so nothing orders them; one launch instead of two back-to-back ones.
template <bool BS>
__global__ __launch_bounds__(256) void fold_cat_bwd_kernel(View sp_up, View dx, View xact, int has_xact, int act_up, int rpb_up, int nbx_up,
                                                           View sp_sk, View dst, int accumulate, View actsrc, int has_act, int act_sk, int rpb_sk) {
    if ((int)blockIdx.x < nbx_up) fold_up2_bwd_body<BS>(sp_up, dx, xact, has_xact, act_up, rpb_up, (int)blockIdx.x, (int)blockIdx.y);
    else fold_reflect_body<BS>(sp_sk, dst, accumulate, actsrc, has_act, act_sk, rpb_sk, (int)blockIdx.x - nbx_up, (int)blockIdx.y);
}
void launch_fold_cat_bwd(const View& pad_up, const View& dx, const View* xact, int act_up, const View& pad_sk, const View& dst, int accumulate, const View* actsrc, int act_sk, hipStream_t s) {
    const int Hf = pad_up.H - 2, Wf = pad_up.W - 2, ln = dx.bf ? 8 : 4;
    if (Hf < 4 || Wf < 4 || Hf > 2 * dx.H || Wf > 2 * dx.W || pad_up.C < dx.C) { fprintf(stderr, "md2: fold_cat_bwd: unsupported shape (upsampled half)\n"); abort(); }
    if (pad_sk.H != dst.H + 2 || pad_sk.W != dst.W + 2 || pad_sk.C < dst.C) { fprintf(stderr, "md2: fold_cat_bwd: shape mismatch (skip half)\n"); abort(); }
    if (dx.C % ln != 0 || dst.C % ln != 0 || pad_up.bf != dx.bf || pad_sk.bf != dx.bf || dst.bf != dx.bf || (xact && xact->bf != dx.bf) || (actsrc && actsrc->bf != dx.bf)) { fprintf(stderr, "md2: fold_cat_bwd: unsupported views\n"); abort(); }
    const RowGrid ru = row_grid((long)dx.N * dx.H, dx.W * (dx.C / ln)), rs = row_grid((long)dst.N * dst.H, dst.W * (dst.C / ln));
    const dim3 grid(ru.grid.x + rs.grid.x, std::max(ru.grid.y, rs.grid.y), 1);          // (a body returns at once for a chunk index beyond its own row length)
    LAUNCH_BS(dx.bf, fold_cat_bwd_kernel, grid, dim3(256), 0, s, pad_up, dx, xact ? *xact : dx, xact ? 1 : 0, act_up, ru.rpb, (int)ru.grid.x,
              pad_sk, dst, accumulate, actsrc ? *actsrc : dst, actsrc ? 1 : 0, act_sk, rs.rpb);
}
void launch_fold_reflect(const View& src_pad, const View& dst, int accumulate, const View* actsrc, int act, hipStream_t s) {
    if (src_pad.H != dst.H + 2 || src_pad.W != dst.W + 2 || src_pad.C < dst.C) { fprintf(stderr, "md2: fold_reflect: shape mismatch\n"); abort(); }
    if (dst.C % (dst.bf ? 8 : 4) != 0 || src_pad.bf != dst.bf || (actsrc && actsrc->bf != dst.bf)) { fprintf(stderr, "md2: fold_reflect: unsupported views\n"); abort(); }
    const RowGrid rg = row_grid((long)dst.N * dst.H, dst.W * (dst.C / (dst.bf ? 8 : 4)));
    LAUNCH_BS(dst.bf, fold_reflect_kernel, rg.grid, dim3(256), 0, s, src_pad, dst, accumulate, actsrc ? *actsrc : dst, actsrc ? 1 : 0, act, rg.rpb);
}

template <bool BS>
__global__ __launch_bounds__(256) void axpy_view_kernel(View src, View dst, float a, int accumulate, View actsrc, int has_act, int act, int rpb) {
    constexpr int N = LaneN<BS>::N;
    const int Q = dst.C / N, idx = blockIdx.y * 256 + threadIdx.x;
    if (idx >= dst.W * Q) return;
    const int w = idx / Q, q = idx - w * Q;
    const int rows = dst.N * dst.H, row_end = min(rows, (int)(blockIdx.x + 1) * rpb);
    for (int row = blockIdx.x * rpb; row < row_end; ++row) {
        const int n = row / dst.H, h = row - n * dst.H;
        const long r = (long)h * dst.W + w;
        float g[N]; vldn<BS>(src.p, voff(src, n, r) + q * N, g);
#pragma unroll
        for (int e = 0; e < N; ++e) g[e] *= a;
        if (has_act) {
            float o[N]; vldn<BS>(actsrc.p, voff(actsrc, n, r) + q * N, o);
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] *= act_deriv(o[e], act);
        }
        const long dpo = voff(dst, n, r) + q * N;
        if (accumulate) {
            float o[N]; vldn<BS>(dst.p, dpo, o);
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] += o[e];
        }
        vstn<BS>(dst.p, dpo, g);
    }
}
void launch_axpy_view(const View& src, const View& dst, float a, int accumulate, const View* actsrc, int act, hipStream_t s) {
    if (dst.C % (dst.bf ? 8 : 4) != 0 || src.bf != dst.bf || (actsrc && actsrc->bf != dst.bf)) { fprintf(stderr, "md2: axpy_view: unsupported views\n"); abort(); }
    const RowGrid rg = row_grid((long)dst.N * dst.H, dst.W * (dst.C / (dst.bf ? 8 : 4)));
    LAUNCH_BS(dst.bf, axpy_view_kernel, rg.grid, dim3(256), 0, s, src, dst, a, accumulate, actsrc ? *actsrc : dst, actsrc ? 1 : 0, act, rg.rpb);
}

// ---- bias gradient: out[c] += sum over pixels -----------------------------------------------------------------------------
template <bool BS>
__global__ __launch_bounds__(256) void colsum_kernel(View v, float* __restrict__ out, int rows_per_block) {
    const int Q = v.C / 4, RG = 256 / Q;
    const int q = threadIdx.x % Q, rg = threadIdx.x / Q;
    __shared__ float red[4 * 256];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
    const long P = v.pixels(); const int HW = v.H * v.W;
    const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(p0 + rows_per_block, P);
    if (rg < RG)
        for (long p = p0 + rg; p < p1; p += RG) {
            const int n = p / HW; const long r = p - (long)n * HW;
            const float4 x = vld4<BS>(v.p, voff(v, n, r) + q * 4);
            s1.x += x.x; s1.y += x.y; s1.z += x.z; s1.w += x.w;
        }
    red[0 * 256 + threadIdx.x] = s1.x; red[1 * 256 + threadIdx.x] = s1.y; red[2 * 256 + threadIdx.x] = s1.z; red[3 * 256 + threadIdx.x] = s1.w;
    __syncthreads();
    for (int c = threadIdx.x; c < v.C; c += 256) {
        const int qq = c / 4, e = c % 4; float s = 0.f;
        for (int g = 0; g < RG; ++g) s += red[e * 256 + g * Q + qq];
        atomicAdd(out + c, s);
    }
}
void launch_colsum(const View& v, float* out, hipStream_t s) {
    if (v.C % 4 != 0 || v.C > 1024 || 256 % (v.C / 4) != 0) { fprintf(stderr, "md2: colsum: unsupported C=%d\n", v.C); abort(); }
    const long P = v.pixels();
    int rpb = 2048; while (rpb > 64 && (P + rpb - 1) / rpb < 512) rpb >>= 1;
    if (g_deterministic) rpb = (int)std::min<long>(P, 1 << 30);
    LAUNCH_BS(v.bf, colsum_kernel, dim3((int)((P + rpb - 1) / rpb)), dim3(256), 0, s, v, out, rpb);
}

// ---- pose squeezer bookkeeping (src/model.jl:30-35, src/pose_decoder.jl:23-25): the 1x1 squeezer runs ONCE over all 3N frames (sq3 [3N][hw][256]); the pose decoder's input
//      for source s is cat(frame fr[s][0], frame fr[s][1]) with (fr) = ((0,1),(1,2)): sqcat [2][N][hw][512].  Forward = gather, pullback = scatter-add (frame 1 is used twice).
struct PoseFrames { int f[4]; };              // (first, second) frame of the pair of source 0, of source 1
template <bool BS>
__global__ __launch_bounds__(256) void pose_pairs_fwd_kernel(const float* __restrict__ sq3, float* __restrict__ sqcat, int N, int hw, PoseFrames pf) {
    const long total = (long)2 * N * hw * 128;                        // float4 slots of sqcat
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = i % 128; const long r = i / 128; const int px = r % hw; const long sn = r / hw; const int n = sn % N, s = sn / N;
        const int j = c4 >> 6, fr = pf.f[2 * s + j];                  // default frames (0,1) for source 0, (1,2) for source 1
        vst4<BS>(sqcat, i * 4, vld4<BS>(sq3, (((long)(n * 3 + fr) * hw + px) * 64 + (c4 & 63)) * 4));
    }
}
template <bool BS>
__global__ __launch_bounds__(256) void pose_pairs_bwd_kernel(const float* __restrict__ dsqcat, float* __restrict__ dsq3, int N, int hw, PoseFrames pf) {
    const long total = (long)3 * N * hw * 64;                          // float4 slots of dsq3
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = i % 64; const long r = i / 64; const int px = r % hw; const long nf = r / hw; const int fr = nf % 3, n = nf / 3;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);                    // a frame collects every use of it (the target frame is used by both pairs)
#pragma unroll
        for (int sj = 0; sj < 4; ++sj)
            if (pf.f[sj] == fr) { const float4 t = vld4<BS>(dsqcat, ((((long)((sj >> 1) * N + n) * hw + px) * 128) + (sj & 1) * 64 + c4) * 4); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
        vst4<BS>(dsq3, i * 4, g);
    }
}
void launch_pose_pairs_fwd(const float* sq3, float* sqcat, int N, int hw, const int fr[4], hipStream_t s, int bf) { PoseFrames pf{{fr[0], fr[1], fr[2], fr[3]}}; LAUNCH_BS(bf, pose_pairs_fwd_kernel, dim3(grid_for((long)2 * N * hw * 128, 256, 2048)), dim3(256), 0, s, sq3, sqcat, N, hw, pf); }
void launch_pose_pairs_bwd(const float* dsqcat, float* dsq3, int N, int hw, const int fr[4], hipStream_t s, int bf) { PoseFrames pf{{fr[0], fr[1], fr[2], fr[3]}}; LAUNCH_BS(bf, pose_pairs_bwd_kernel, dim3(grid_for((long)3 * N * hw * 64, 256, 2048)), dim3(256), 0, s, dsqcat, dsq3, N, hw, pf); }

// ---- pose head tail: mean over (W,H) * 0.01 (src/pose_decoder.jl:29-31); y view [B][h*w][>=6 channels] -------------------------
template <bool BS>
__global__ void pose_mean_fwd_kernel(View y, float* __restrict__ rvec, float* __restrict__ tvec) {
    const int b = blockIdx.x, c = threadIdx.x >> 5, l = threadIdx.x & 31;   // 6 channels x 32 lanes
    if (c >= 6) return;
    const int HW = y.H * y.W; float s = 0.f;
    for (int r = l; r < HW; r += 32) s += vld1<BS>(y.p, (long)b * y.bs + (long)r * y.ps + c);
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l == 0) { const float v = 0.01f * (s / (float)HW); if (c < 3) rvec[b * 3 + c] = v; else tvec[b * 3 + c - 3] = v; }
}
void launch_pose_mean_fwd(const View& y, float* rvec, float* tvec, hipStream_t s) { LAUNCH_BS(y.bf, pose_mean_fwd_kernel, dim3(y.N), dim3(192), 0, s, y, rvec, tvec); }
template <bool BS>
__global__ void pose_mean_bwd_kernel(const float* __restrict__ drvec, const float* __restrict__ dtvec, View dy) {
    const int HW = dy.H * dy.W; const long total = (long)dy.N * HW * dy.C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % dy.C; const long pr = i / dy.C; const int r = pr % HW; const int b = pr / HW;
        float g = 0.f;
        if (c < 3) g = 0.01f * drvec[b * 3 + c] / (float)HW; else if (c < 6) g = 0.01f * dtvec[b * 3 + c - 3] / (float)HW;
        vst1<BS>(dy.p, (long)b * dy.bs + (long)r * dy.ps + c, g);
    }
}
void launch_pose_mean_bwd(const float* drvec, const float* dtvec, const View& dy, hipStream_t s) {
    LAUNCH_BS(dy.bf, pose_mean_bwd_kernel, dim3(grid_for((long)dy.N * dy.H * dy.W * dy.C, 256, 1024)), dim3(256), 0, s, drvec, dtvec, dy);
}

// ---- Flux ADAM (beta 0.9/0.999, eps 1e-8): one fused pass over the flat parameter buffer ------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n4, float c1, float c2, float lr, float gs) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 P = ld4(p + i * 4), G = ld4(g + i * 4), M = ld4(m + i * 4), V = ld4(v + i * 4);
        float* pp = &P.x; float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gg[e] * gs;
            mm[e] = 0.9f * mm[e] + (1.f - 0.9f) * ge;
            vv[e] = 0.999f * vv[e] + (1.f - 0.999f) * ge * ge;
            pp[e] -= mm[e] / c1 / (sqrtf(vv[e] / c2) + 1e-8f) * lr;
        }
        st4(p + i * 4, P); st4(m + i * 4, M); st4(v + i * 4, V);
    }
}
__global__ void adam_tail_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n, float c1, float c2, float lr, float gs) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x; if (i >= n) return;
    const float ge = g[i] * gs;
    m[i] = 0.9f * m[i] + (1.f - 0.9f) * ge; v[i] = 0.999f * v[i] + (1.f - 0.999f) * ge * ge;
    p[i] -= m[i] / c1 / (sqrtf(v[i] / c2) + 1e-8f) * lr;
}
void launch_adam(float* p, const float* g, float* m, float* v, long n, float c1, float c2, float lr, float gscale, hipStream_t s) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const long n4 = aligned ? n / 4 : 0;
    if (n4 > 0) hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4, 256, 4096)), dim3(256), 0, s, p, g, m, v, n4, c1, c2, lr, gscale);
    const long rem = n - 4 * n4;                        // tail (or everything, for unaligned / tiny arrays such as a 3-vector pose)
    if (rem > 0) hipLaunchKernelGGL(adam_tail_kernel, dim3(grid_for(rem, 256, 4096)), dim3(256), 0, s, p + 4 * n4, g + 4 * n4, m + 4 * n4, v + 4 * n4, rem, c1, c2, lr, gscale);
}
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) vst4<true>(reinterpret_cast<float*>(dst), i * 4, ld4(src + i * 4));
}
void launch_cast_bf16(const float* src, uint16_t* dst, long n, hipStream_t s) {
    if (n & 3) { fprintf(stderr, "md2: cast_bf16: length must be a multiple of 4\n"); abort(); }
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4, 256, 2048)), dim3(256), 0, s, src, dst, n / 4);
}
__global__ void fill_kernel(float* p, long n, float v) { for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v; }
void launch_fill(float* p, long n, float v, hipStream_t s) { hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, p, n, v); }

template <bool BS>
__global__ __launch_bounds__(256) void fill_view_kernel(View v, float val) {
    const int Q = v.C / 4; const long total = v.pixels() * Q; const int HW = v.H * v.W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = i % Q; const long p = i / Q; const int n = p / HW; const long r = p - (long)n * HW;
        vst4<BS>(v.p, voff(v, n, r) + q * 4, make_float4(val, val, val, val));
    }
}
void launch_fill_view(const View& v, float val, hipStream_t s) { LAUNCH_BS(v.bf, fill_view_kernel, dim3(grid_for(v.pixels() * (v.C / 4), 256, 4096)), dim3(256), 0, s, v, val); }
template <bool BS>
__global__ __launch_bounds__(256) void bias_act_view_kernel(View y, const float* __restrict__ bias, int act) {
    const int Q = y.C / 4; const long total = y.pixels() * Q; const int HW = y.H * y.W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = i % Q; const long p = i / Q; const int n = p / HW; const long r = p - (long)n * HW;
        const long ypo = voff(y, n, r) + q * 4;
        float4 v = vld4<BS>(y.p, ypo);
        if (bias) { const float4 b = ld4(bias + q * 4); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
        vst4<BS>(y.p, ypo, v);
    }
}
void launch_bias_act_view(const View& y, const float* bias, int act, hipStream_t s) {
    LAUNCH_BS(y.bf, bias_act_view_kernel, dim3(grid_for(y.pixels() * (y.C / 4), 256, 4096)), dim3(256), 0, s, y, bias, act);
}

}  // namespace md2
