// md2_internal.h — shared declarations of the gfx950 kernels behind include/md2hip.h.
// Internal activation layout is NHWC fp32 with explicit batch / pixel strides (a "view"), so that
// frame slicing (src/model.jl:16,32-34) and channel concat (src/depth_decoder.jl:19, src/pose_decoder.jl:24)
// cost nothing: producers write straight into channel slices / strided batches.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

namespace md2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MD2_DETERMINISTIC=1 (read by md2_create) or md2_set_deterministic(): every cross-workgroup reduction is done in a fixed order -- no split-K,
// one workgroup per weight-gradient tile, BatchNorm / bias sums by a single workgroup per channel group, per-block loss partials summed
// sequentially -- so two runs on the same inputs are bit-identical.  Much slower; exists to bisect parity failures.
extern int g_deterministic;
// md2_set_winograd(): the 3x3 / stride-1 layers take the fused Winograd F(2x2,3x3) kernel when the launch has at least g_wino_min_blocks workgroups (MD2_NO_WINO=1 /
// MD2_WINO_MIN_BLOCKS override the defaults at load time); off = the direct implicit GEMM everywhere
extern int g_wino_on, g_wino_min_blocks, g_wino_min_blocks_default;
// md2_set_conv_products(): how the fp32 products of the convolutions are formed.  0 = the fp32 MFMA (v_mfma_f32_32x32x2_f32); 3 = every fp32 operand as the exact sum of three
// bf16 pieces (hi + mid + lo = all 24 mantissa bits), six v_mfma_f32_32x32x16_bf16 per product group with fp32 accumulation -- the same accuracy as the fp32 MFMA on the
// bf16 matrix pipe; 1 (default) = per launch whichever of those two FULL-ACCURACY forms is faster at the launch's shape (wino_resolve_pieces / wgrad_resolve_pieces below);
// 2 = two pieces (hi + lo, ~16-bit operand mantissas): REDUCED precision, opt-in only.  g_wino_pieces: Winograd forward / input-gradient kernels (the mode the Winograd
// operands U were written under is recorded per handle when they are refreshed; the resolved layout travels in ConvArgs::wino_pieces); g_wgrad_pieces: weight-gradient kernels.
extern int g_wino_pieces, g_wgrad_pieces;
extern int g_wino_cb2;           // conv_winograd.hip: 1 = the fp32-MFMA Winograd kernel takes two 32-channel output blocks per workgroup where Cout % 64 == 0
extern int g_wino_wide;          // conv_winograd.hip: 0 (default); 1 = split launches with Cout % 64 == 0 take the 64-channel workgroup (conv_wino3w_kernel): md2_set_winograd(on + 8, .) sets it, any other md2_set_winograd call clears it
extern int g_wino_bf16;          // conv_winograd.hip: 0 (default); 1 = bf16 tensors take the Winograd kernel conv_wino3b_kernel: md2_set_winograd(on + 32, .) sets it, any other md2_set_winograd call clears it
extern int g_wino_persist;       // conv_winograd.hip: 0 (default); 1 = split launches with more items than resident workgroups take the persistent kernel (conv_wino3p_kernel): md2_set_winograd(on + 16, .) sets it, any other md2_set_winograd call clears it
// load-time defaults of the two (MD2_WINO_PIECES / MD2_WGRAD_PIECES override): every product of the default step carries full fp32 operand accuracy
#define MD2_DEFAULT_WINO_PIECES 1
#define MD2_DEFAULT_WGRAD_PIECES 1
// mode 1 resolved for one launch (in-model HIP-event timings at batch 32 / 4, tools/run_wgrad_pieces_compare.sh, round 4):
//   forward / input gradient (GEMM-K = the launch's input channels): the exact split wins from 96 channels up (0.131 -> 0.117 ms at 128 -> 128, 0.148 -> 0.125 at 512 -> 512);
//   short reductions (16 .. 64 channels) spend their time outside the chunk loop, where the split kernel's 16-channel chunks and piece registers only cost.
//   Used by the weight transform (device) and by the launcher (host): both derive the layout of a layer's U from the same (mode, channels).
__host__ __device__ inline int wino_resolve_pieces(int mode, int cin) { return mode == 1 ? (cin >= 96 ? 3 : 0) : mode; }
//   Winograd-domain weight gradient: the split kernel's 32 x 32-channel workgroups fill the chip on small launches (batch 4: 0.050 -> 0.037 ms at 256 -> 256) but repeat the
//   transforms of a tile for twice as many channel blocks as the fp32 kernel's 64 x 32 ones (batch 32: 0.200 -> 0.295 ms at 512 -> 512): by work, M x Cout x Cin.
inline int wgrad_resolve_pieces_wino(int mode, long M, int Cout, int Cin) { return mode == 1 ? ((double)M * Cout * Cin < 3e8 ? 3 : 0) : mode; }
//   per-tap weight gradient (1x1 and stride-2 layers): the exact split is ahead at every size measured (0.146 -> 0.101 ms, 0.046 -> 0.033 ms)
inline int wgrad_resolve_pieces_tap(int mode) { return mode == 1 ? 3 : mode; }
//   direct implicit GEMM, forward / input gradient (1x1 and stride-2 layers, the stride-2 input gradients' residue classes, 3x3 launches below the Winograd threshold; round 5):
//   the exact split is ahead at every shape measured, tools/run_igemm_split_bench.sh -- 0.127 -> 0.107 ms on the 64 -> 128 stride-2 3x3 at batch 32, 0.147 -> 0.114 / 0.143 -> 0.113 ms
//   on ResNet-50's 512 -> 128 / 1024 -> 256 bottleneck 1x1s at 640x192, 0.028 -> 0.025 on the 64-channel 1x1 shortcut; 64-channel-wide tiles only (Cout % 64 == 0)
inline int igemm_resolve_split(int mode, int taps, int cin, int cout) { (void)taps; (void)cin; return (cout % 64 == 0 && mode != 0) ? 1 : 0; }

// Launch-policy constants.  The product library compiles them in: the only environment variables it reads are the documented load-time defaults of the md2_set_* calls
// (MD2_DETERMINISTIC, MD2_ONE_STREAM, MD2_NO_WINO / MD2_WINO_MODE / MD2_WINO_MIN_BLOCKS, MD2_WINO_PIECES / MD2_WGRAD_PIECES; include/md2hip.h).  The measuring tools
// (tools/conv_bench*.cpp, built with -DMD2_TUNE_ENV -- or -DMD2_ABLATE, which also adds the kernels' timing switches) read the same names from the environment, which is how
// the values below were chosen.
#if defined(MD2_TUNE_ENV) || defined(MD2_ABLATE)
inline long md2_tune(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
inline bool md2_tune_set(const char* name) { return getenv(name) != nullptr; }
#else
constexpr long md2_tune(const char*, long dflt) { return dflt; }
constexpr bool md2_tune_set(const char*) { return false; }
#endif

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SIGMOID = 3 };
enum PadMode : int { PAD_ZERO = 0, PAD_REFLECT = 1 };

// NHWC view: element (n,h,w,c) at p[n*bs + (h*W + w)*ps + c]
// bf: the elements are bf16 (2 bytes; md2_config.precision = 2, "bf16 activation storage"): p is then the BYTE address of element 0 typed as float* only nominally, every
// offset stays in ELEMENTS and the kernels go through vld4 / vst4 below.  0 = fp32 elements (everything else in the library).
struct View {
    float* p = nullptr; int N = 0, H = 0, W = 0, C = 0; long bs = 0; int ps = 0; int bf = 0;
    __host__ __device__ long pixels() const { return (long)N * H * W; }
};
inline View make_view(float* p, int N, int H, int W, int C, int bf = 0) { View v; v.p = p; v.N = N; v.H = H; v.W = W; v.C = C; v.ps = C; v.bs = (long)H * W * C; v.bf = bf; return v; }
inline float* view_at(const View& v, long elem) { return reinterpret_cast<float*>(reinterpret_cast<char*>(v.p) + elem * (v.bf ? 2 : 4)); }
inline View chan_slice(const View& v, int c0, int c) { View r = v; r.p = view_at(v, c0); r.C = c; return r; }
// frame l of a [N*L] batch (batch index n*L + l)
inline View frame_view(const View& v, int L, int l) { View r = v; r.p = view_at(v, (long)l * v.bs); r.N = v.N / L; r.bs = v.bs * L; return r; }

// ---- element access of an activation tensor, fp32 or bf16 storage (BS).  `base` is the tensor's address, `off` an ELEMENT offset (a multiple of 4 for the quads);
//      bf16 stores round to nearest even (v_cvt_pk_bf16_f32), loads are exact (a bf16 is the upper half of the fp32 with the same value)
#ifdef __HIPCC__
__device__ __forceinline__ float4 bf16x4_unpack(uint2 u) { return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)); }
__device__ __forceinline__ unsigned bf16x2_pack(float a, float b) { typedef __bf16 b2 __attribute__((ext_vector_type(2))); b2 h; h[0] = (__bf16)a; h[1] = (__bf16)b; return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ uint2 bf16x4_pack(float4 v) { return make_uint2(bf16x2_pack(v.x, v.y), bf16x2_pack(v.z, v.w)); }
template <bool BS> __device__ __forceinline__ float4 vld4(const float* base, long off) {
    if constexpr (BS) return bf16x4_unpack(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off));
    else return *reinterpret_cast<const float4*>(base + off);
}
template <bool BS> __device__ __forceinline__ void vst4(float* base, long off, float4 v) {
    if constexpr (BS) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = bf16x4_pack(v);
    else *reinterpret_cast<float4*>(base + off) = v;
}
template <bool BS> __device__ __forceinline__ float vld1(const float* base, long off) {
    if constexpr (BS) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[off] << 16);
    else return base[off];
}
template <bool BS> __device__ __forceinline__ void vst1(float* base, long off, float v) {
    if constexpr (BS) { const __bf16 h = (__bf16)v; reinterpret_cast<unsigned short*>(base)[off] = __builtin_bit_cast(unsigned short, h); }
    else base[off] = v;
}
__device__ __forceinline__ long voff(const View& v, int n, long r) { return (long)n * v.bs + r * v.ps; }
// byte offset of (channel c, pixel p) in a CHANNEL-major bf16 LDS tile whose channel rows are `chs` bytes (an odd number of 16-byte groups of 8 pixels): the group index is
// XORed with 2 * bit 3 of the channel, which spreads the dword writes of a wave's 8 channel quads over four bank octets and keeps 8 consecutive channels' 16-byte reads on 8 groups
__device__ __forceinline__ int wb_addr(int c, int p, int chs) { return c * chs + (((p >> 3) ^ (((c >> 3) & 1) << 1)) << 4) + ((p & 7) << 1); }
#endif

__host__ __device__ inline int refl_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__device__ inline float act_apply(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_ELU) return v > 0.f ? v : expm1f(v);
    if (act == ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
// derivative of the activation expressed through its OUTPUT value o
__device__ inline float act_deriv(float o, int act) {
    if (act == ACT_RELU) return o > 0.f ? 1.f : 0.f;
    if (act == ACT_ELU) return o > 0.f ? 1.f : o + 1.f;
    if (act == ACT_SIGMOID) return o * (1.f - o);
    return 1.f;
}

// ---- implicit-GEMM convolution (conv_igemm.hip) -------------------------------------------------
// y[n,oh,ow,co] = epi( sum_{kh,kw,ci} w[co][(kh*KW+kw)*Cin+ci] * x[n, ih, iw, ci] ),
//   t = o*stride - pad + k ; dil>1: taps with t % dil != 0 contribute 0 and ih = t/dil (transposed conv, used for
//   the input-gradient of strided convs) ; PAD_ZERO: out-of-range -> 0 ; PAD_REFLECT: mirrored.
//   epi: (+bias) -> act -> (* act_deriv(dact_src)) -> (+ old y if accumulate)
constexpr int BN_STAT_SLOTS = 8;
struct ConvArgs {
    const float* x; long x_bs; int x_ps; int H, W, Cin;
    const float* w;
    float* y; long y_bs; int y_ps; int Ho, Wo, Cout;
    int N, KH, KW, stride, pad, dil, padmode;
    const float* bias; int act; int accumulate;               // accumulate: 0 overwrite, 1 add to y, 2 add only where (batch index % 3) == acc_frame (the target frames)
    int acc_frame;                                            // frame of the triplet that already holds a contribution in accumulate mode 2 (0-based target frame)
    const float* dact_src; long da_bs; int da_ps; int dact;   // optional: multiply by act_deriv(dact_src[n,oh,ow,co])
    const float* res; long res_bs; int res_ps;                // optional residual gradient added BEFORE the act_deriv factor: y = (conv + res) * act'(dact_src) [+ y]
    int M;                                                    // N*Hl*Wl
    // output lattice: this launch produces pixels (oy0 + i*ostep, ox0 + j*ostep), i < Hl, j < Wl (dense: 0,0,1,Ho,Wo)
    int oy0, ox0, ostep, Hl, Wl;
    // tap lattice: taps kh = kh0 + a*khs (a < nkh), kw = kw0 + b*kws (b < nkw)   (dense: 0,1,KH / 0,1,KW)
    int kh0, khs, nkh, kw0, kws, nkw;
    int ksplit;                   // set by the launcher: >1 = the channel chunks are split over gridDim.y; the partial sums go to ws[ksplit][M][Cout] and a second pass
                                  // (splitk_reduce_kernel: + bias, activation, accumulate, BatchNorm statistics) writes y -- or, without a workspace, fp32 atomics into y
    float* ws; long ws_cap;       // optional split-K workspace of the calling stream (floats); null = atomics
    // optional training-BatchNorm statistics of the OUTPUT (no bias / activation): stats[slot][0..Cout) += sum_pixels y, stats[slot][Cout..2Cout) += sum y^2,
    // slot = blockIdx.x % BN_STAT_SLOTS spreads the double atomics; fused into the 32x32 tiles' epilogue, a colstats launch (slot 0) otherwise
    double* stats;
    // bf16 mixed precision (md2_config.precision = 1): the operands of the 32x32 MFMA tiles are rounded to bf16 on their way from LDS to the
    // matrix pipe (v_cvt_pk_bf16_f32 + v_mfma_f32_32x32x16_bf16); accumulation, storage, weights and every other kernel stay fp32
    // 2 (md2_config.precision = 2, bf16 activation storage): x / y / res / dact_src / bz are bf16 tensors and `w` a bf16 copy of the weights; a K-step is then 32 channels
    // (the same 64 bytes per pixel row), two v_mfma_f32_32x32x16_bf16 (or one 16x16x32) per tile pair; fp32 accumulation, bias, statistics and split-K workspace
    int bf16;
    // ---- BatchNorm folded into the neighbouring convolutions (the activation a = relu(bn(z)) between two encoder convs is never materialised) ----
    // input transform: the gathered operand is relu(x * in_scale[ci] + in_shift[ci]) (zero padding applies AFTER it); null = the raw tensor
    const float* in_scale; const float* in_shift;
    // the activation derivative is evaluated at dact_src * dact_scale[co] + dact_shift[co] (dact_src is then the producer's raw conv output z)
    const float* dact_scale; const float* dact_shift;
    // BatchNorm-pullback sums of THIS launch's final output g (after residual / act' / accumulate), fused into the 32x32 tiles' epilogue (a column pass otherwise):
    //   bst[slot][0..Cout) += sum_px g ; bst[slot][Cout..2Cout) += sum_px g * xhat,  xhat = (bz - mean) * invstd with bstat = mean | invstd.
    // A second (bstB, bzB, bstatB) serves a block output that feeds two BatchNorms (main branch and 1x1 shortcut).
    double* bst; const float* bz; long bz_bs; int bz_ps; const float* bstat;
    double* bstB; const float* bzB; long bzB_bs; int bzB_ps; const float* bstatB;
    // stride-2 input gradient split into residue classes: accumulate mode of class (0,0) if >= 0 (the 1x1 shortcut's input gradient wrote exactly those pixels first);
    // skip_empty: residue classes without a tap are left untouched instead of refusing a non-accumulating launch
    int accumulate_c00, skip_empty;
    // dense launch: output pixels on the lattice (oy % acc_lat == 0 && ox % acc_lat == 0) ADD to y whatever `accumulate` says (a strided 1x1 shortcut's input gradient
    // wrote exactly those pixels first); 0 = off
    int acc_lat;
    // Winograd F(2x2,3x3) operand of this layer (conv_winograd.hip): U = G g G^T of `w` in MFMA fragment order, refreshed once per step; null = direct convolution only
    const float* wino_u;
    int split3;                   // fp32 tensors (bf16 == 0): 1 = the direct implicit GEMM forms its products from the exact three-bf16-piece split of both operands (BF = 3; 64-channel-wide 32 x 32 tiles only), see igemm_resolve_split
    int wino_pieces;              // layout wino_u was written in: 0 = fp32 fragments, 2 / 3 = bf16 pieces per operand (split kernels); set by whoever refreshed U
#ifdef MD2_ABLATE
    int ablate, force_tile;       // tools/conv_bench only
    unsigned long long* dbg;      // tools/conv_bench only: per-wave s_memtime stamps of conv_wino3s_kernel ([workgroup][wave][8]: 7 stamps + the hardware id), abl bit 512
#endif
};
void launch_conv_igemm(const ConvArgs& a, hipStream_t s);
double conv_igemm_flops(const ConvArgs& a);
double conv_igemm_executed_flops(const ConvArgs& a);      // what the matrix cores really multiply (Winograd launches: 16/36 of the algorithmic count)
// fused Winograd F(2x2,3x3) for the 3x3 / stride-1 layers (conv_winograd.hip); launch_conv_igemm routes to it when a.wino_u is set and the shape qualifies
bool conv_wino_eligible(const ConvArgs& a);
void launch_conv_wino(const ConvArgs& a, hipStream_t s);
size_t conv_wino_u_floats(int Cout, int Cin);
void launch_wino_weights(const float* w /*[Cout][9][Cin]*/, float* U, int Cout, int Cin, int pieces, hipStream_t s);
long wino_weight_blocks(int Cout, int Cin);
// desc (device): {w offset, U offset, Cout, Cin, blocks} per layer
void launch_wino_weights_batched(const float* wbase, float* Ubase, const long* desc_dev, int nlayers, int total_blocks, int pieces, hipStream_t s);

// dw[co][(kh*KW+kw)*Cin+ci] += sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, ih, iw, ci]   (atomic accumulation, split over pixels)
struct WgradArgs {
    const float* x; long x_bs; int x_ps; int H, W, Cin;
    const float* dy; long dy_bs; int dy_ps; int Ho, Wo, Cout;
    float* dw;
    int N, KH, KW, stride, pad, padmode;
    int M;
    int bf16;                     // same switch for the all-taps weight gradient (conv3x3_wgrad_taps32_kernel)
    int overlapped;               // 1: the launch runs on the weight-gradient stream beside the pullback's dependency chain -- split-K'd up to 256 workgroups instead of the chip's 512 (round 6)
    int pieces_mode1;             // md2_set_conv_products' weight-gradient mode + 1 as the CALLER snapshot it (a handle: at the start of its pullback, so that a concurrent
                                  // md2_set_conv_products cannot mix product forms inside one step); 0 = not given: the process-wide g_wgrad_pieces at launch time
    float* db;                    // optional bias gradient: db[co] += sum_{n,oh,ow} dy (fused into the all-taps kernels, a colsum launch otherwise)
    const float* in_scale; const float* in_shift;      // x operand = relu(x * in_scale[ci] + in_shift[ci]) applied on the way into LDS (all-taps kernel only); null = raw
#ifdef MD2_ABLATE
    int ablate;
#endif
};
void launch_conv_wgrad(const WgradArgs& a, hipStream_t s);
bool conv_wgrad_supports_in_transform(int H, int W, int Cin, int Cout);
// weight gradient in the Winograd domain (conv_winograd.hip); launch_conv_wgrad routes the 3x3 / stride-1 layers with Cout % 64 == 0, Cin % 32 == 0 to it
bool conv_wino_wgrad_eligible(const WgradArgs& a);
void launch_conv_wino_wgrad(const WgradArgs& a, hipStream_t s);

// weight transforms between the three layouts
//   flux   : [Cout][Cin][KH][KW], true convolution (kernel flipped)           (boundary)
//   fwd    : [CoutP][KH*KW][CinP] cross-correlation taps                       (internal parameter storage)
//   dgrad  : [CinP][KH*KW flipped][CoutP]                                       (per-step transposed copy)
void launch_weight_to_dgrad(const float* w_fwd, float* w_dg, int CoutP, int CinP, int taps, hipStream_t s);
// every layer's transposed copy in one LDS-tiled launch that also zeroes `zero[0 .. nzero)` (the flat gradient buffer); desc5 = {offset, CoutP, CinP, taps, first block} per layer
long weight_prepare_blocks(int CoutP, int CinP, int taps);
void launch_weight_prepare(const float* w_fwd, float* w_dg, const long* desc5_dev, int nlayers, int nblocks_w, float* zero, long nzero, hipStream_t s);

// ---- 3x3 reflect-pad conv with ONE output channel (disparity heads, src/depth_decoder.jl:43) ------
void launch_head_fwd(const View& x, const float* w /*[9][Cin]*/, const float* b, const View& y /*C=1, sigmoid*/, hipStream_t s);
// dz = dy * y*(1-y) computed on the fly.  dx = (accumulate? dx : 0) + reflect-adjoint conv ; dw += ; db +=
// x_act != ACT_NONE: x is the activation OUTPUT of its producer and this head is its only consumer -> dx is multiplied by act'(x)
void launch_head_bwd(const View& x, const float* w, const View& y, const View& dy, const View& dx, int accumulate, int x_act, float* dw, float* db, hipStream_t s);

// ---- layer kernels (layers.hip) ----------------------------------------------------------------
// ---- ResNet stem, 7x7 stride 2 pad 3 on the boundary tensor x [NL][C][H][W] (stem.hip); w / dw: [64][Kpad], k = (kh*7+kw)*C + c
// bf16_products (the bf16 modes): operands rounded to bf16 in LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulation
void launch_stem_fwd(const float* x, int NL, int C, int H, int W, const float* w, int Kpad, const View& z, double* bn_stats /*[BN_STAT_SLOTS][128] or null*/, hipStream_t s, int bf16_products = 0);
// zbn / coef (optional): dz is not materialised -- the view passed as dz holds g (the gradient of the BatchNorm OUTPUT), zbn the BatchNorm input, and the kernel forms
// dz = coef[0..64) g + coef[64..128) z + coef[128..192) per channel on the way into LDS
void launch_stem_wgrad(const float* x, int NL, int C, int H, int W, const View& dz, float* dw, int Kpad, hipStream_t s, const View* zbn = nullptr, const float* coef = nullptr, int bf16_products = 0);
void launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s);
void launch_nhwc_to_nchw(const float* src, const View& v, float* dst, hipStream_t s);
void launch_nchw_to_view(const float* src, const View& v, hipStream_t s);
// per-channel sum / sum of squares of a view -> double[2*C] (must be zeroed by caller)
void launch_colstats(const View& z, double* acc, hipStream_t s);
// mean/invstd from acc; also zeroes nothing.  stats: float[2*C] = mean, invstd
// stats: float[4*C] = mean | invstd | scale | shift (the last two only when gamma / beta are given: y = z * scale + shift, what a consumer's loader applies)
void launch_bn_finalize(const double* acc /*[slots][2C]*/, float* stats, int C, double count, int slots, float* running /* [2C] or null */, float momentum, hipStream_t s,
                        const float* gamma = nullptr, const float* beta = nullptr);
void launch_bn_stats_from_running(const float* running, float* stats, int C, hipStream_t s, const float* gamma = nullptr, const float* beta = nullptr);
// Batch statistics still in their double accumulators: an elementwise kernel that is the FIRST reader of a BatchNorm finalises them in its own prologue (every workgroup
// into LDS, workgroup 0 also writes `stats` and the running statistics for the later readers) -- the arithmetic of bn_finalize_kernel without its launch.
// `stats` must be the array the same call applies.
struct BnFin { const double* acc = nullptr; double count = 0; int slots = 0; float* stats = nullptr; float* running = nullptr; float momentum = 0.f; };
void launch_bn_relu_maxpool(const View& z, const float* stats, const float* gamma, const float* beta, const View& a, int L, int tgt, const View& y, uint8_t* arg, hipStream_t s,
                            const BnFin* fin = nullptr);
// y = act(gamma*(z-mean)*invstd + beta + residual)
void launch_bn_apply(const View& z, const float* stats, const float* gamma, const float* beta, const View* residual, int act, const View& y, hipStream_t s,
                     const float* rstats = nullptr, const float* rgamma = nullptr, const float* rbeta = nullptr,      // r*: the residual is a raw conv output, BatchNorm'ed on the fly
                     const BnFin* fin = nullptr, const BnFin* rfin = nullptr);                                        // fin / rfin: finalise stats / rstats here (no bn_finalize launch)
// g = dy * act_deriv(yact) ; acc[0..C) += sum g ; acc[C..2C) += sum g*xhat          (acc double, zeroed by caller)
void launch_bn_bwd_reduce(const View& dy, const View* yact, int act, const View& z, const float* stats, double* acc, hipStream_t s);
// dz = gamma*invstd*(g - s1/M - xhat*s2/M) ; dgamma += s2, dbeta += s1 (written by block 0)
void launch_bn_bwd_apply(const View& dy, const View* yact, int act, const View& z, const float* stats, const float* gamma, const double* acc /* [slots][2C] */,
                         double count, const View& dz, float* dgamma, float* dbeta, hipStream_t s, int slots, float* coef /* float[2C] scratch: m1 | m2 */);
void launch_maxpool_fwd(const View& x, const View& y, uint8_t* arg, hipStream_t s);
// dx (+)= scatter of dy by arg, gather form; optional multiply by relu'(xact).  accumulate as in ConvArgs (0 / 1 / 2)
void launch_maxpool_bwd(const View& dy, const uint8_t* arg, const View& dx, int accumulate, const View* yrelu /* pooled output of a post-ReLU input, or null */, hipStream_t s,
                        const View* bz = nullptr, const float* bstat = nullptr, double* bacc = nullptr, int acc_frame = 1);       // optional: BatchNorm-pullback sums of dx w.r.t. BatchNorm input bz
// sums -> coef = m1 | m2 [| A | B | C0 with stats / gamma: dz = A g + B z + C0], d gamma / d beta accumulated
void launch_bn_bwd_finalize(const double* acc, int slots, int C, double count, float* coef, float* dgamma, float* dbeta, const float* stats, const float* gamma, hipStream_t s);
void launch_upsample2x_fwd(const View& x, const View& y, hipStream_t s, const View* skip = nullptr);      // skip: y is the concat buffer, its channels behind x's receive a copy of *skip
// dx = adjoint(up2)(dy) * act_deriv(xact)
void launch_upsample2x_bwd(const View& dy, const View& dx, const View* xact, int act, hipStream_t s);
// fold_reflect + upsample2x_bwd in one gather pass: dx = adjoint(up2)(fold(src_padded)) * act_deriv(xact)     (fine map at least 4 x 4, ratio <= 2)
void launch_fold_up2_bwd(const View& src_pad, const View& dx, const View* xact, int act, hipStream_t s);
// both halves of a DecoderBlock's padded concat gradient in one launch: fold_up2_bwd on the upsampled channels, fold_reflect on the skip connection's
void launch_fold_cat_bwd(const View& pad_up, const View& dx, const View* xact, int act_up, const View& pad_sk, const View& dst, int accumulate, const View* actsrc, int act_sk, hipStream_t s);
// reflect-pad adjoint: dst = ((accumulate ? dst : 0) + fold(src_padded)) * act_deriv(actsrc)
void launch_fold_reflect(const View& src_pad, const View& dst, int accumulate, const View* actsrc, int act, hipStream_t s);
// dst = (accumulate ? dst : 0) + a*src (optionally * act_deriv(actsrc))
void launch_axpy_view(const View& src, const View& dst, float a, int accumulate, const View* actsrc, int act, hipStream_t s);
// out[c] (+)= sum over pixels of v[.,c]   (float atomics)
void launch_colsum(const View& v, float* out, hipStream_t s);
// pose squeezer: sq3 [3N][hw][256] (all frames, squeezed once) <-> sqcat [2][N][hw][512] (the temporally ordered pairs of both sources); the pullback adds frame 1's two uses
// fr[4] = the two (first, second) frame pairs, temporally ordered: {a0, b0, a1, b1} (0-based; src/model.jl:30-35)
void launch_pose_pairs_fwd(const float* sq3, float* sqcat, int N, int hw, const int fr[4], hipStream_t s, int bf = 0);      // bf: the tensors hold bf16 elements
void launch_pose_pairs_bwd(const float* dsqcat, float* dsq3, int N, int hw, const int fr[4], hipStream_t s, int bf = 0);
// pose head: y [B][HW][16] -> rvec/tvec [B][3] = 0.01 * mean ; and its adjoint
void launch_pose_mean_fwd(const View& y, float* rvec, float* tvec, hipStream_t s);
void launch_pose_mean_bwd(const float* drvec, const float* dtvec, const View& dy, hipStream_t s);
void launch_adam(float* p, const float* g, float* m, float* v, long n, float c1, float c2, float lr, float gscale, hipStream_t s);
void launch_fill(float* p, long n, float v, hipStream_t s);
// dst[i] = bf16(src[i]) (round to nearest even): the per-step bf16 copies of the weights when activations are stored in bf16
void launch_cast_bf16(const float* src, uint16_t* dst, long n, hipStream_t s);
void launch_fill_view(const View& v, float val, hipStream_t s);
// y = act(y + bias[c])   (second pass of a split-K convolution)
void launch_bias_act_view(const View& y, const float* bias, int act, hipStream_t s);

// ---- loss kernels (loss.hip) --------------------------------------------------------------------
#define MD2_MAX_SCALES 5     // DepthDecoder scale_levels has at most 5 entries (src/depth_decoder.jl:27-29)
struct LossArgs {
    const float* x;            // [N][3][C][H][W]
    const float* disp[MD2_MAX_SCALES];      // [N][h_i][w_i]
    int dh[MD2_MAX_SCALES], dw[MD2_MAX_SCALES];
    const float* Rt;           // [2][N][12] : R (col-major 9) + t (3), after composeT
    const float* auto_loss;    // [N][H][W] or null
    float K[9], invK[9];       // column-major
    int N, C, H, W, nscales;
    float min_disp, max_disp;  // 1/max_depth, 1/min_depth
    float smooth_w[MD2_MAX_SCALES];         // disparity_smoothness * scale_i
    int tgt, src0, src1;       // 0-based frame indices
    int normalize;             // 1: smooth_loss(d / (mean(d)+1e-7)) (train_loss, src/training.jl:64-67); 0: smooth_loss(d) (slow_depth, src/simple_depth.jl:39)
};
// per-scale partial sums -> part[(scale*N + n)*4 + {photo, sum d, Sx, Sy}] (double atomics; zeroed by caller)
// blk (deterministic mode, else null): per-block partials [nscales*N][blocks][4] summed in order by launch_loss_part_reduce instead of atomics
void launch_loss_fwd(const LossArgs& a, double* part, double* blk, hipStream_t s);
int loss_blocks_per_image(const LossArgs& a);
// totals[(scale*N+n)*4+..] float copy + scalar loss
void launch_loss_finalize(const LossArgs& a, const double* part, float* totals, float* loss, hipStream_t s);
// training: smoothness sums, then loss value + pullback from one pass over the warp.  part / dRt_acc zeroed by the caller; dd[sc] = [N][H][W] gradient wrt the
// full-resolution disparity of scale sc, or -- where dlo[sc] is not null -- dlo[sc] = [N][dh][dw] (zeroed by the caller) receives the adjoint of the bilinear resize
// of that gradient instead (float atomics); blk as above (needs >= max(N*tiles32*12, nscales*N*blocks*25) doubles)
void launch_loss_smooth_sums(const LossArgs& a, double* part, double* blk, hipStream_t s);      // first: part slots 1..3 (sum d, Sx, Sy per scale and sample)
void launch_loss_train(const LossArgs& a, double* part, float* const dd[MD2_MAX_SCALES], float* const dlo[MD2_MAX_SCALES], double* dRt_acc, double* blk, hipStream_t s);
void launch_automask(const float* x, float* out, int N, int C, int H, int W, int src0, int src1, int tgt, hipStream_t s);
// planar bilinear (align-corners) upsample adjoint, gather form: [planes][Ho][Wo] -> [planes][H][W]
void launch_upsample_planar_bwd(const float* dout, float* dx, int planes, int H, int W, int Ho, int Wo, hipStream_t s);
void launch_upsample_planar_fwd(const float* x, float* out, int planes, int H, int W, int Ho, int Wo, hipStream_t s);
// composeT for both sources: rvec,tvec [2][N][3] -> Rt [2][N][12] ; adjoint dRt(double) -> drvec,dtvec
void launch_pose_compose(const float* rvec, const float* tvec, float* Rt, int N, int inv0, int inv1, hipStream_t s);
void launch_pose_compose_bwd(const float* rvec, const float* tvec, const double* dRt, float* drvec, float* dtvec, int N, int inv0, int inv1, hipStream_t s);

struct Timer;   // profile.hpp

}  // namespace md2
