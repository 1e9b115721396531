// conv_winograd.hip — fused Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 (CDNA4).
//
// Serves every 3x3 / stride-1 convolution with Cout % 32 == 0 (forward AND input gradient -- the latter is the same kernel on dY with the
// transposed / flipped weights): the encoder's BasicBlock convs (ResNet.jl via src/model.jl:14), DecoderBlock (src/depth_decoder.jl:13-14) and the
// PoseDecoder's 3x3 layers (src/pose_decoder.jl:15-19).  The reference gets the same algorithmic cut from cuDNN's algorithm choice.
//
// Why: the direct implicit GEMM (conv_igemm.hip) sits at ~0.62 of the fp32 MFMA peak with a measured ceiling of ~0.75; these layers are 85-90 % of the
// step's FLOPs and F(2x2,3x3) needs 16 instead of 36 multiplies per 2x2 output tile (2.25x fewer MFMA cycles).  Nothing transformed ever touches HBM
// except the weights (once per step, 16/9 of their size).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d: 4x4 input patch of a 2x2 output tile, g: 3x3 taps (cross-correlation form)
//
// Two kernels: conv_wino3_kernel (64 tiles x 32 channels per block, further down) is the one the training step takes; conv_wino_kernel (first version, described here)
// stays for shapes the other cannot stage (tile rows shorter than 4 tiles) and as the pinned alternative of the tests.
// Block = 256 threads = 4 waves; block tile = 32 Winograd tiles (128 output pixels, flattened over (n, ty, tx) so that 13x4 maps pack like 104x32 ones)
// x 32*NS output channels.  Wave i owns Winograd row i: the four positions (i, j), j = 0..3, i.e. acc[j][s] = U_(i,j)[32 couts s][K] . V_(i,j)[K][32 tiles].
//   * input: the raw 4x4 patches of the 32 tiles for a 16-channel chunk are staged in LDS ([16 patch slots][32 tiles][16 ch], 32 KB, double buffered);
//     BatchNorm + ReLU of the producer (XF) and the zero-padding mask are applied on the way in, exactly as in the implicit GEMM.  Wave i reads only the two
//     patch rows its B^T row needs (ds_read_b128, XOR-swizzled 16-B chunks: conflict-free) and forms V_(i,0..3) in registers directly in MFMA B-operand
//     layout (lane = tile, k-half): 32 VALU instructions per 8 channels against 32 MFMAs of 64 cycles.
//   * weights: U = G g G^T is precomputed once per step (wino_weights_kernel) in MFMA A-operand FRAGMENT order, so a wave streams its operands with
//     fully coalesced 1-KB global_load_dwordx4 straight into registers -- every U element is used by exactly one wave of the block: LDS would add nothing.
//   * output: each wave applies the column half of A^T . A in registers (4 positions -> 2), the row half goes through LDS ([4][2][32 tiles][couts], 68 KB
//     reusing the stages); then 16 lanes own the 64 channels of one pixel (256-B rows to HBM) and run the same epilogue contract as conv_igemm_kernel:
//     bias, ELU / ReLU, residual, activation-derivative mask (incl. BatchNorm-folded), accumulate modes, BatchNorm statistics / pullback sums.
#include "md2_internal.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

#include <type_traits>
namespace md2 {

// exact floor(m / d) for 0 <= m < 2^24, d >= 1 from the float reciprocal with a +-1 correction: ~8 VALU instructions instead of the ~40 of a 32-bit integer division.  The
// per-workgroup prologue / epilogue of the kernels below held 21 of those per thread -- about a fifth of a workgroup's fixed cost (tools/run_wino3_ablate.sh: the skeleton
// without loads, MFMAs and transforms is 23-35 % of a launch).
__device__ __forceinline__ int fdiv24(int m, int d) {
    int q = (int)((float)m * __builtin_amdgcn_rcpf((float)d));
    const int r = m - q * d;
    q += (r >= d) ? 1 : 0; q -= (r < 0) ? 1 : 0;
    return q;
}


typedef float f32x16 __attribute__((ext_vector_type(16)));

// pieces per fp32 operand of the split Winograd path (g_wino_pieces, md2_set_conv_products): 0 = off (fp32 MFMAs), 3 = hi + mid + lo (exact), 2 = hi + lo (~16-bit operand
// mantissas: reduced precision, opt-in).  A launch never reads the global: the layout its operand U was written in travels in ConvArgs::wino_pieces.

__device__ __forceinline__ int wino_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int WT = 32;                       // Winograd tiles per block
constexpr int W_STAGE = 16 * WT * 16;        // floats per LDS stage (32 KB)

// split mode (conv_wino3s_kernel): the same (G g G^T)[i][j] as two bf16 pieces, u = hi + lo, in the A-operand layout of v_mfma_f32_32x32x16_bf16 --
// [Cout/32][4 (i)][Cin/16][4 (j)][np pieces][64 lanes][8]: lane l, element e = (co = 32 ct + (l & 31), ci = 16 c16 + 8 (l >> 5) + e).  np = 2: the byte count of the fp32 fragments,
// np = 3 (hi + mid + lo: exact): 1.5 x, which is what conv_wino_u_floats reserves.
__device__ __forceinline__ void wino_store_split(float* __restrict__ U, int ct, int i, int j, int Cin, int ci, int co_lo, const float4& u, int np) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    b4 hi, mid, lo;          // np == 2: u = hi + lo (+ 2^-17 |u|); np == 3: u = hi + mid + lo (+ 2^-25 |u|: every bit of the fp32 value); np == 4 (bf16 tensors, round 6): ONE piece, u rounded to bf16
    hi[0] = (__bf16)u.x; hi[1] = (__bf16)u.y; hi[2] = (__bf16)u.z; hi[3] = (__bf16)u.w;
    const float r0 = u.x - (float)hi[0], r1 = u.y - (float)hi[1], r2 = u.z - (float)hi[2], r3 = u.w - (float)hi[3];
    mid[0] = (__bf16)r0; mid[1] = (__bf16)r1; mid[2] = (__bf16)r2; mid[3] = (__bf16)r3;
    lo[0] = (__bf16)(r0 - (float)mid[0]); lo[1] = (__bf16)(r1 - (float)mid[1]); lo[2] = (__bf16)(r2 - (float)mid[2]); lo[3] = (__bf16)(r3 - (float)mid[3]);
    const int c16 = ci >> 4, g = (ci >> 3) & 1, e0 = ci & 4;
    const int npe = np == 4 ? 1 : np;          // pieces stored per position
    __bf16* base = reinterpret_cast<__bf16*>(U) + ((((long)(ct * 4 + i) * (Cin >> 4) + c16) * 4 + j) * npe) * 512 + (co_lo + 32 * g) * 8 + e0;
    *reinterpret_cast<b4*>(base) = hi;
    if (npe >= 2) *reinterpret_cast<b4*>(base + 512) = mid;
    if (npe == 3) *reinterpret_cast<b4*>(base + 1024) = lo;
}
// ---- weights: w [Cout][9][Cin] (cross-correlation taps, tap = kh*3 + kw) -> U fragments [Cout/32][4 (i)][Cin/8][4 (j)][64 lanes][4]:
//      lane l, element e of fragment (ct, i, c8, j) = (G g G^T)[i][j] of (co = 32 ct + (l & 31), ci = 8 c8 + 4 (l >> 5) + e)
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int split) {
    const int gid = blockIdx.x * 256 + threadIdx.x;          // (ci quad, co): co fastest within 32 so that the 16-B stores of a half-wave are contiguous
    const int co_lo = gid & 31, rest = gid >> 5;
    const int nq = Cin >> 2;
    const int cq = rest % nq, ct = rest / nq;
    if (ct >= (Cout >> 5)) return;
    const int co = ct * 32 + co_lo, ci = cq * 4;
    split = wino_resolve_pieces(split, Cin);          // (`split` arrives as the mode: automatic = by the layer's reduction length)
    float4 g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = *reinterpret_cast<const float4*>(w + ((long)co * 9 + t) * Cin + ci);
    // Gg: rows i of G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
    float4 gg[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 a = g[c], b = g[3 + c], d = g[6 + c];
        gg[0][c] = a; gg[3][c] = d;
        gg[1][c] = make_float4(0.5f * (a.x + b.x + d.x), 0.5f * (a.y + b.y + d.y), 0.5f * (a.z + b.z + d.z), 0.5f * (a.w + b.w + d.w));
        gg[2][c] = make_float4(0.5f * (a.x - b.x + d.x), 0.5f * (a.y - b.y + d.y), 0.5f * (a.z - b.z + d.z), 0.5f * (a.w - b.w + d.w));
    }
    const int c8 = ci >> 3, h = (ci >> 2) & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = gg[i][0], b = gg[i][1], d = gg[i][2];
        float4 u[4];
        u[0] = a; u[3] = d;
        u[1] = make_float4(0.5f * (a.x + b.x + d.x), 0.5f * (a.y + b.y + d.y), 0.5f * (a.z + b.z + d.z), 0.5f * (a.w + b.w + d.w));
        u[2] = make_float4(0.5f * (a.x - b.x + d.x), 0.5f * (a.y - b.y + d.y), 0.5f * (a.z - b.z + d.z), 0.5f * (a.w - b.w + d.w));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (split) wino_store_split(U, ct, i, j, Cin, ci, co_lo, u[j], split);
            else *reinterpret_cast<float4*>(U + ((((long)(ct * 4 + i) * (Cin >> 3) + c8) * 4 + j) * 64 + (co_lo + 32 * h)) * 4) = u[j];
        }
    }
}
void launch_wino_weights(const float* w, float* U, int Cout, int Cin, int pieces, hipStream_t s) {
    const long n = (long)(Cout / 32) * (Cin / 4) * 32;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, U, Cout, Cin, pieces);
}
// all eligible layers of a handle in one launch: desc = {w offset, U offset, Cout, Cin} per layer, block ranges by prefix sums in `first`
__global__ __launch_bounds__(256) void wino_weights_batched_kernel(const float* __restrict__ wbase, float* __restrict__ Ubase, const long* __restrict__ desc, int nlayers, int split) {
    int b = blockIdx.x, l = 0;
    for (; l < nlayers; ++l) { const long nb = desc[l * 5 + 4]; if (b < nb) break; b -= (int)nb; }
    if (l >= nlayers) return;
    const float* w = wbase + desc[l * 5 + 0]; float* U = Ubase + desc[l * 5 + 1];
    const int Cout = (int)desc[l * 5 + 2], Cin = (int)desc[l * 5 + 3];
    const int gid = b * 256 + threadIdx.x;
    const int co_lo = gid & 31, rest = gid >> 5;
    const int nq = Cin >> 2;
    const int cq = rest % nq, ct = rest / nq;
    if (ct >= (Cout >> 5)) return;
    const int co = ct * 32 + co_lo, ci = cq * 4;
    split = wino_resolve_pieces(split, Cin);          // (`split` arrives as the mode: automatic = by the layer's reduction length)
    float4 g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = *reinterpret_cast<const float4*>(w + ((long)co * 9 + t) * Cin + ci);
    float4 gg[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 a = g[c], bb = g[3 + c], d = g[6 + c];
        gg[0][c] = a; gg[3][c] = d;
        gg[1][c] = make_float4(0.5f * (a.x + bb.x + d.x), 0.5f * (a.y + bb.y + d.y), 0.5f * (a.z + bb.z + d.z), 0.5f * (a.w + bb.w + d.w));
        gg[2][c] = make_float4(0.5f * (a.x - bb.x + d.x), 0.5f * (a.y - bb.y + d.y), 0.5f * (a.z - bb.z + d.z), 0.5f * (a.w - bb.w + d.w));
    }
    const int c8 = ci >> 3, h = (ci >> 2) & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = gg[i][0], bb = gg[i][1], d = gg[i][2];
        float4 u[4];
        u[0] = a; u[3] = d;
        u[1] = make_float4(0.5f * (a.x + bb.x + d.x), 0.5f * (a.y + bb.y + d.y), 0.5f * (a.z + bb.z + d.z), 0.5f * (a.w + bb.w + d.w));
        u[2] = make_float4(0.5f * (a.x - bb.x + d.x), 0.5f * (a.y - bb.y + d.y), 0.5f * (a.z - bb.z + d.z), 0.5f * (a.w - bb.w + d.w));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (split) wino_store_split(U, ct, i, j, Cin, ci, co_lo, u[j], split);
            else *reinterpret_cast<float4*>(U + ((((long)(ct * 4 + i) * (Cin >> 3) + c8) * 4 + j) * 64 + (co_lo + 32 * h)) * 4) = u[j];
        }
    }
}
void launch_wino_weights_batched(const float* wbase, float* Ubase, const long* desc_dev, int nlayers, int total_blocks, int pieces, hipStream_t s) {
    if (nlayers <= 0 || total_blocks <= 0) return;
    hipLaunchKernelGGL(wino_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, s, wbase, Ubase, desc_dev, nlayers, pieces);
}
long wino_weight_blocks(int Cout, int Cin) { return ((long)(Cout / 32) * (Cin / 4) * 32 + 255) / 256; }

// ---------------------------------------------------------------------------------------------------------------------
template <int NS, int XF, int BST>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(ConvArgs a, int TH, int TW, int Mt, int ntp) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int COB = 32 * NS;                 // output channels per block
    constexpr int F = 8 * NS;                    // U fragments per 16-channel chunk and wave
    constexpr int R = 2 * NS;                    // fragment ring: loads run two j-steps (8 NS MFMAs) ahead of their use
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = wino_xcd_remap(blockIdx.x, gridDim.x);
    // pixel tile fastest inside an XCD's contiguous id range: the blocks running together on one XCD share a cout tile, so its U slice (2 MB at Cin = 512) stays in that L2
    const int tile_p = bid % ntp, tile_c = bid / ntp;
    const int m0 = tile_p * WT, c0 = tile_c * COB;
    const int THW = TH * TW;

    // ---- loader constants: thread = (tile t, 16-B chunk q, column pair w2); 8 patch slots ab = w2 + 2 jj : row a = jj >> 1, column b = w2 + 2 (jj & 1)
    const int lt = (tid >> 2) & 31, q = tid & 3, w2 = tid >> 7;
    unsigned l_off[8], l_mask = 0;
    {
        const int m = m0 + lt;
        const bool tv = m < Mt;
        const int mm = tv ? m : 0;
        const int n = fdiv24(mm, THW), r = mm - n * THW, ty = fdiv24(r, TW), tx = r - ty * TW;
        const unsigned base = (unsigned)((long)n * a.x_bs) + (unsigned)(q * 4);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            int ih = 2 * ty - a.pad + (jj >> 1), iw = 2 * tx - a.pad + w2 + 2 * (jj & 1);
            bool ok = tv;
            if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
            else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            l_mask |= ok ? (1u << jj) : 0u;
            l_off[jj] = ok ? (base + (unsigned)((ih * a.W + iw) * a.x_ps)) * 4u : (unsigned)(q * 16);       // masked slots read a valid dummy address
        }
    }
    const int l_dst = ((w2 * 32 + lt) * 16) + ((q ^ ((lt >> 2) & 3)) << 2);        // + jj * 2 * 32 * 16 floats
    const char* xb = reinterpret_cast<const char*>(a.x);
    uint4 ra[8];
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto gload = [&](int c) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ra[jj] = *reinterpret_cast<const uint4*>(xb + (l_off[jj] + (unsigned)(c * 64)));
        if constexpr (XF != 0) { xsc = *reinterpret_cast<const float4*>(a.in_scale + c * 16 + q * 4); xsh = *reinterpret_cast<const float4*>(a.in_shift + c * 16 + q * 4); }
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const unsigned mk = 0u - ((l_mask >> jj) & 1u);
            uint4 v = ra[jj];
            if constexpr (XF != 0) {
                v.x = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), xsc.x, xsh.x), 0.f)); v.y = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), xsc.y, xsh.y), 0.f));
                v.z = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), xsc.z, xsh.z), 0.f)); v.w = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), xsc.w, xsh.w), 0.f));
            }
            v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;
            *reinterpret_cast<uint4*>(wsm + so + l_dst + jj * (2 * 32 * 16)) = v;
        }
    };

    // ---- consumer constants: wave i = Winograd row; B^T row i combines patch rows (r1, r2): d[r1] + sg * d[r2]
    const int wi = wave;
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const int ct = lane & 31, ch = lane >> 5;
    const int p0 = ((ch ^ ((ct >> 2) & 3)) << 2);                       // swizzled 16-B chunk of k-half ch, first 8 channels of the chunk; second 8: ^ 8 floats
    const int rd1 = (r1 * 4 * 32 + ct) * 16, rd2 = (r2 * 4 * 32 + ct) * 16;

    f32x16 acc[4][NS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][s][e] = 0.f;

    // U streams of this wave: fragment (c8, j) of cout sub-tile s at ub[s] + (c8 * 4 + j) * 256 floats
    const int nchunks = a.Cin >> 4;
    const int nfr = nchunks * 8;                    // fragments per stream
    const float* ub[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) ub[s] = a.wino_u + ((long)((tile_c * NS + s) * 4 + wi) * (a.Cin >> 3)) * 4 * 256 + lane * 4;
    float4 ring[R];
    // ring slot g % R holds fragment step g of the chunk: (fragment index fi = g / NS within the chunk, sub-tile s = g % NS)
#pragma unroll
    for (int g = 0; g < R; ++g) ring[g] = *reinterpret_cast<const float4*>(ub[g % NS] + (long)min(g / NS, nfr - 1) * 256);

#ifdef MD2_ABLATE
    const int abl = a.ablate;                    // tools/conv_bench: 1 = no patch loads after the first chunk, 2 = no U refills, 4 = no MFMAs, 8 = no V transform reads
#else
    constexpr int abl = 0;
#endif
    gload(0); lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int so = (c & 1) * W_STAGE;
        if (c + 1 < nchunks && !(abl & 1)) gload(c + 1);
        float4 V[4];
#pragma unroll
        for (int g = 0; g < F; ++g) {
            const int fi = g / NS, s = g % NS, c8l = fi >> 2, j = fi & 3;
            if (s == 0 && j == 0 && !(abl & 8 && c > 0)) {
                // V_(i, 0..3) for this lane's tile and k-half: row combine, then the B^T column combine
                const int pp = c8l ? (p0 ^ 8) : p0;
                float4 rr[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 d1 = *reinterpret_cast<const float4*>(wsm + so + rd1 + b * (32 * 16) + pp);
                    const float4 d2 = *reinterpret_cast<const float4*>(wsm + so + rd2 + b * (32 * 16) + pp);
                    rr[b] = make_float4(fmaf(sg, d2.x, d1.x), fmaf(sg, d2.y, d1.y), fmaf(sg, d2.z, d1.z), fmaf(sg, d2.w, d1.w));
                }
                V[0] = make_float4(rr[0].x - rr[2].x, rr[0].y - rr[2].y, rr[0].z - rr[2].z, rr[0].w - rr[2].w);
                V[1] = make_float4(rr[1].x + rr[2].x, rr[1].y + rr[2].y, rr[1].z + rr[2].z, rr[1].w + rr[2].w);
                V[2] = make_float4(rr[2].x - rr[1].x, rr[2].y - rr[1].y, rr[2].z - rr[1].z, rr[2].w - rr[1].w);
                V[3] = make_float4(rr[1].x - rr[3].x, rr[1].y - rr[3].y, rr[1].z - rr[3].z, rr[1].w - rr[3].w);
            }
            const float4 u = ring[g % R];
            if (!(abl & 4)) {
            acc[j][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(u.x, V[j].x, acc[j][s], 0, 0, 0);
            acc[j][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(u.y, V[j].y, acc[j][s], 0, 0, 0);
            acc[j][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(u.z, V[j].z, acc[j][s], 0, 0, 0);
            acc[j][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(u.w, V[j].w, acc[j][s], 0, 0, 0);
            } else asm volatile("" :: "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "v"(V[j].x));
            // refill this ring slot with fragment step g + R (of this or the next chunk; past the end: the last fragment again, never used)
            const int gn = g + R, fn = c * 8 + gn / NS;
            if (!(abl & 2)) ring[g % R] = *reinterpret_cast<const float4*>(ub[gn % NS] + (long)min(fn, nfr - 1) * 256);
        }
        if (c + 1 < nchunks && !(abl & 1)) lstore(W_STAGE - so);
        __syncthreads();
    }

    // ---- output transform.  Column half in registers: R_b = sum_j M[i][j] A^T[b][j]  (A^T = [1 1 1 0; 0 1 -1 -1])
    constexpr int RLD = COB + 4;
    float* Rl = wsm;                             // [4 (i)][2 (b)][32 tiles][RLD]
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o0, o1;
            o0.x = acc[0][s][4 * g + 0] + acc[1][s][4 * g + 0] + acc[2][s][4 * g + 0]; o1.x = acc[1][s][4 * g + 0] - acc[2][s][4 * g + 0] - acc[3][s][4 * g + 0];
            o0.y = acc[0][s][4 * g + 1] + acc[1][s][4 * g + 1] + acc[2][s][4 * g + 1]; o1.y = acc[1][s][4 * g + 1] - acc[2][s][4 * g + 1] - acc[3][s][4 * g + 1];
            o0.z = acc[0][s][4 * g + 2] + acc[1][s][4 * g + 2] + acc[2][s][4 * g + 2]; o1.z = acc[1][s][4 * g + 2] - acc[2][s][4 * g + 2] - acc[3][s][4 * g + 2];
            o0.w = acc[0][s][4 * g + 3] + acc[1][s][4 * g + 3] + acc[2][s][4 * g + 3]; o1.w = acc[1][s][4 * g + 3] - acc[2][s][4 * g + 3] - acc[3][s][4 * g + 3];
            const int col = s * 32 + 8 * g + 4 * ch;
            *reinterpret_cast<float4*>(Rl + ((wi * 2 + 0) * 32 + ct) * RLD + col) = o0;
            *reinterpret_cast<float4*>(Rl + ((wi * 2 + 1) * 32 + ct) * RLD + col) = o1;
        }
    __syncthreads();
    // ---- row half + epilogue: QN lanes own the COB channels of one pixel
    constexpr int QN = COB / 4, RP = 256 / QN;
    const int q4 = (tid % QN) * 4, rr0 = tid / QN, co = c0 + q4;
    const bool want_st = a.stats != nullptr, want_b = BST != 0 && a.bst != nullptr;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, s3 = s1;
    float4 bias4 = s1, dsc = s1, dsh = s1, muA = s1, isA = s1, muB = s1, isB = s1;
    if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + co);
    if (a.dact_scale) { dsc = *reinterpret_cast<const float4*>(a.dact_scale + co); dsh = *reinterpret_cast<const float4*>(a.dact_shift + co); }
    if constexpr (BST != 0) if (want_b) {
        muA = *reinterpret_cast<const float4*>(a.bstat + co); isA = *reinterpret_cast<const float4*>(a.bstat + a.Cout + co);
        if (a.bstB) { muB = *reinterpret_cast<const float4*>(a.bstatB + co); isB = *reinterpret_cast<const float4*>(a.bstatB + a.Cout + co); }
    }
#pragma unroll 1
    for (int p = 0; p < WT / RP; ++p) {
        const int t = p * RP + rr0, m = m0 + t;
        if (m >= Mt) continue;
        const int n = fdiv24(m, THW), r = m - n * THW, ty = fdiv24(r, TW), tx = r - ty * TW;
        float4 Rv[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) Rv[i][b] = *reinterpret_cast<const float4*>(Rl + ((i * 2 + b) * 32 + t) * RLD + q4);
#pragma unroll
        for (int oa = 0; oa < 2; ++oa)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int oy = 2 * ty + oa, ox = 2 * tx + ob;
                if (oy >= a.Ho || ox >= a.Wo) continue;
                float4 v;
                if (oa == 0) { v.x = Rv[0][ob].x + Rv[1][ob].x + Rv[2][ob].x; v.y = Rv[0][ob].y + Rv[1][ob].y + Rv[2][ob].y; v.z = Rv[0][ob].z + Rv[1][ob].z + Rv[2][ob].z; v.w = Rv[0][ob].w + Rv[1][ob].w + Rv[2][ob].w; }
                else { v.x = Rv[1][ob].x - Rv[2][ob].x - Rv[3][ob].x; v.y = Rv[1][ob].y - Rv[2][ob].y - Rv[3][ob].y; v.z = Rv[1][ob].z - Rv[2][ob].z - Rv[3][ob].z; v.w = Rv[1][ob].w - Rv[2][ob].w - Rv[3][ob].w; }
                const long pix = (long)oy * a.Wo + ox;
                if (a.res) { const float4 tt = *reinterpret_cast<const float4*>(a.res + (long)n * a.res_bs + pix * a.res_ps + co); v.x += tt.x; v.y += tt.y; v.z += tt.z; v.w += tt.w; }
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                if (a.act != ACT_NONE) { v.x = act_apply(v.x, a.act); v.y = act_apply(v.y, a.act); v.z = act_apply(v.z, a.act); v.w = act_apply(v.w, a.act); }
                if (a.dact_src) {
                    float4 d = *reinterpret_cast<const float4*>(a.dact_src + (long)n * a.da_bs + pix * a.da_ps + co);
                    if (a.dact_scale) { d.x = fmaf(d.x, dsc.x, dsh.x); d.y = fmaf(d.y, dsc.y, dsh.y); d.z = fmaf(d.z, dsc.z, dsh.z); d.w = fmaf(d.w, dsc.w, dsh.w); }
                    v.x *= act_deriv(d.x, a.dact); v.y *= act_deriv(d.y, a.dact); v.z *= act_deriv(d.z, a.dact); v.w *= act_deriv(d.w, a.dact);
                }
                float* yp = a.y + (long)n * a.y_bs + pix * a.y_ps + co;
                if (a.accumulate == 1 || (a.accumulate == 2 && (n % 3) == a.acc_frame)) { const float4 o = *reinterpret_cast<const float4*>(yp); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *reinterpret_cast<float4*>(yp) = v;
                if (want_st) { s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w; s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w; }
                if constexpr (BST != 0) if (want_b) {
                    const float4 z = *reinterpret_cast<const float4*>(a.bz + (long)n * a.bz_bs + pix * a.bz_ps + co);
                    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                    s2.x += v.x * ((z.x - muA.x) * isA.x); s2.y += v.y * ((z.y - muA.y) * isA.y); s2.z += v.z * ((z.z - muA.z) * isA.z); s2.w += v.w * ((z.w - muA.w) * isA.w);
                    if (a.bstB) {
                        const float4 zb = *reinterpret_cast<const float4*>(a.bzB + (long)n * a.bzB_bs + pix * a.bzB_ps + co);
                        s3.x += v.x * ((zb.x - muB.x) * isB.x); s3.y += v.y * ((zb.y - muB.y) * isB.y); s3.z += v.z * ((zb.z - muB.z) * isB.z); s3.w += v.w * ((zb.w - muB.w) * isB.w);
                    }
                }
            }
    }
    if (!(want_st || want_b)) return;                  // block-uniform
    // column sums: the RP threads that share a channel quad meet in LDS; thread (k, channel) then adds the block's total with one double atomic
    __syncthreads();
    float* Rs = wsm;                                   // [3][4][256]
    Rs[0 * 256 + tid] = s1.x; Rs[1 * 256 + tid] = s1.y; Rs[2 * 256 + tid] = s1.z; Rs[3 * 256 + tid] = s1.w;
    Rs[1024 + 0 * 256 + tid] = s2.x; Rs[1024 + 1 * 256 + tid] = s2.y; Rs[1024 + 2 * 256 + tid] = s2.z; Rs[1024 + 3 * 256 + tid] = s2.w;
    Rs[2048 + 0 * 256 + tid] = s3.x; Rs[2048 + 1 * 256 + tid] = s3.y; Rs[2048 + 2 * 256 + tid] = s3.z; Rs[2048 + 3 * 256 + tid] = s3.w;
    __syncthreads();
    if (tid < 3 * COB) {
        const int k = tid / COB, chn = tid % COB, qq = chn >> 2, e = chn & 3;
        float sum = 0.f;
#pragma unroll 4
        for (int g = 0; g < RP; ++g) sum += Rs[k * 1024 + e * 256 + g * QN + qq];
        const size_t slot = (size_t)(blockIdx.x % BN_STAT_SLOTS) * 2 * a.Cout;
        if (want_st) { if (k < 2) atomicAdd(a.stats + slot + k * a.Cout + c0 + chn, (double)sum); }
        else if (k == 0) { atomicAdd(a.bst + slot + c0 + chn, (double)sum); if (a.bstB) atomicAdd(a.bstB + slot + c0 + chn, (double)sum); }
        else if (k == 1) atomicAdd(a.bst + slot + a.Cout + c0 + chn, (double)sum);
        else if (a.bstB) atomicAdd(a.bstB + slot + a.Cout + c0 + chn, (double)sum);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// What the first kernel is bound by (tools/conv_bench ablations, 96 x 32x104, 512 -> 64 channels): MFMAs alone 0.68 ms, operand loads alone 0.88 ms -- the U fragments
// 0.65 of it at 4 KB in flight per wave, the 16 patch slots per tile the rest -- i.e. L2 -> CU operand traffic and its latency, not the matrix pipe.  The kernel below
// moves less per multiply.  Shared pieces first: the geometry of row-segment patch staging and the epilogue.
//   Patches: horizontally adjacent tiles share two of their four columns.  A block's 64 consecutive (flattened) tiles are cut into row segments; a segment of L tiles stages
//   4 rows x (2 L + 2) pixels once instead of 16 L pixel slots (~1.9x fewer gathered bytes and LDS); slot = segment base + row * (2 L + 2) + column.
// Measured and rejected on the way (same shape; kept out of the source): 512 threads per 64 x 64 block with the U fragments shared through an LDS double buffer (1.02 ms: one
// lockstep workgroup per CU overlaps nothing -- its skeleton without MFMAs and loads is 0.28 ms; filled by global_load_lds it was slower still, because hipcc drains
// vmcnt(0) before the first ds_read of every chunk, an LDS-DMA may alias any LDS read); one wave per SIMD with 512 registers and 256 accumulators (64 x 64 per wave,
// 5.1 ms: nothing hides that wave's own LDS / transform latency).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int W2_T = 64;
struct Wino2Geo { int TH, TW, Mt, ntp, nslots; unsigned xbytes; };      // nslots: patch slots allocated per stage (host: worst case over the blocks), multiple of 16; xbytes: byte span of the input view (buffer descriptor range)
// Patch loads through a buffer descriptor (raw buffer, stride 0, num_records = the view's byte span): an out-of-range voffset returns zeros, so the loader's zero padding costs no
// instruction -- inactive and padded items carry W_OOB as their offset instead of a mask that four ANDs per 16-byte item would apply at the commit (24 of the ~330 vector instructions
// per chunk and thread of the split kernel; the channel-chunk offset travels in soffset, which is outside the range check).  Kernels whose loader transforms what it loads (XF:
// relu(x * scale + shift) of a padding zero is not zero) keep the mask.
constexpr unsigned W_OOB = 0xFFFFFFF0u;
typedef unsigned wino_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 wino_buffer_load16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    const wino_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// BS: y, res, dact_src, bz, bzB are bf16 tensors (element offsets unchanged; md2_internal.h vld4 / vst4)
template <int NTHR, int COB, int BST, bool BS = false>
__device__ __forceinline__ void wino_final_epilogue(const ConvArgs& a, float* __restrict__ wsm, const int m0, const int ntile, const int Mt, const int THW, const int TW, const int c0, const int tid) {
    // R[i][b][tile][COB + 4] holds the column-transformed sums; QN lanes own the COB channels of one pixel.  Same contract as conv_igemm_kernel's epilogue.
    constexpr int RLD = COB + 4, QN = COB / 4, RP = NTHR / QN;
    const float* Rl = wsm;
    const int q4 = (tid % QN) * 4, rr0 = tid / QN, co = c0 + q4;
    const bool want_st = a.stats != nullptr, want_b = BST != 0 && a.bst != nullptr;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, s3 = s1;
    float4 bias4 = s1, dsc = s1, dsh = s1, muA = s1, isA = s1, muB = s1, isB = s1;
    if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + co);
    if (a.dact_scale) { dsc = *reinterpret_cast<const float4*>(a.dact_scale + co); dsh = *reinterpret_cast<const float4*>(a.dact_shift + co); }
    if constexpr (BST != 0) if (want_b) {
        muA = *reinterpret_cast<const float4*>(a.bstat + co); isA = *reinterpret_cast<const float4*>(a.bstat + a.Cout + co);
        if (a.bstB) { muB = *reinterpret_cast<const float4*>(a.bstatB + co); isB = *reinterpret_cast<const float4*>(a.bstatB + a.Cout + co); }
    }
#pragma unroll 1
    for (int p = 0; p < ntile / RP; ++p) {
        const int t = p * RP + rr0, m = m0 + t;
        if (m >= Mt) continue;
        const int n = fdiv24(m, THW), r = m - n * THW, ty = fdiv24(r, TW), tx = r - ty * TW;
        float4 Rv[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) Rv[i][b] = *reinterpret_cast<const float4*>(Rl + ((i * 2 + b) * ntile + t) * RLD + q4);
#pragma unroll
        for (int oa = 0; oa < 2; ++oa)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int oy = 2 * ty + oa, ox = 2 * tx + ob;
                if (oy >= a.Ho || ox >= a.Wo) continue;
                float4 v;
                if (oa == 0) { v.x = Rv[0][ob].x + Rv[1][ob].x + Rv[2][ob].x; v.y = Rv[0][ob].y + Rv[1][ob].y + Rv[2][ob].y; v.z = Rv[0][ob].z + Rv[1][ob].z + Rv[2][ob].z; v.w = Rv[0][ob].w + Rv[1][ob].w + Rv[2][ob].w; }
                else { v.x = Rv[1][ob].x - Rv[2][ob].x - Rv[3][ob].x; v.y = Rv[1][ob].y - Rv[2][ob].y - Rv[3][ob].y; v.z = Rv[1][ob].z - Rv[2][ob].z - Rv[3][ob].z; v.w = Rv[1][ob].w - Rv[2][ob].w - Rv[3][ob].w; }
                const long pix = (long)oy * a.Wo + ox;
                if (a.res) { const float4 tt = vld4<BS>(a.res, (long)n * a.res_bs + pix * a.res_ps + co); v.x += tt.x; v.y += tt.y; v.z += tt.z; v.w += tt.w; }
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                if (a.act != ACT_NONE) { v.x = act_apply(v.x, a.act); v.y = act_apply(v.y, a.act); v.z = act_apply(v.z, a.act); v.w = act_apply(v.w, a.act); }
                if (a.dact_src) {
                    float4 d = vld4<BS>(a.dact_src, (long)n * a.da_bs + pix * a.da_ps + co);
                    if (a.dact_scale) { d.x = fmaf(d.x, dsc.x, dsh.x); d.y = fmaf(d.y, dsc.y, dsh.y); d.z = fmaf(d.z, dsc.z, dsh.z); d.w = fmaf(d.w, dsc.w, dsh.w); }
                    v.x *= act_deriv(d.x, a.dact); v.y *= act_deriv(d.y, a.dact); v.z *= act_deriv(d.z, a.dact); v.w *= act_deriv(d.w, a.dact);
                }
                const long yo = (long)n * a.y_bs + pix * a.y_ps + co;
                if (a.accumulate == 1 || (a.accumulate == 2 && (n % 3) == a.acc_frame)) { const float4 o = vld4<BS>(a.y, yo); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                vst4<BS>(a.y, yo, v);
                if (want_st) { s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w; s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w; }
                if constexpr (BST != 0) if (want_b) {
                    const float4 z = vld4<BS>(a.bz, (long)n * a.bz_bs + pix * a.bz_ps + co);
                    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                    s2.x += v.x * ((z.x - muA.x) * isA.x); s2.y += v.y * ((z.y - muA.y) * isA.y); s2.z += v.z * ((z.z - muA.z) * isA.z); s2.w += v.w * ((z.w - muA.w) * isA.w);
                    if (a.bstB) {
                        const float4 zb = vld4<BS>(a.bzB, (long)n * a.bzB_bs + pix * a.bzB_ps + co);
                        s3.x += v.x * ((zb.x - muB.x) * isB.x); s3.y += v.y * ((zb.y - muB.y) * isB.y); s3.z += v.z * ((zb.z - muB.z) * isB.z); s3.w += v.w * ((zb.w - muB.w) * isB.w);
                    }
                }
            }
    }
    if (!(want_st || want_b)) return;                  // block-uniform
    __syncthreads();
    float* Rs = wsm;                                   // [3][4][NTHR]
    Rs[0 * NTHR + tid] = s1.x; Rs[1 * NTHR + tid] = s1.y; Rs[2 * NTHR + tid] = s1.z; Rs[3 * NTHR + tid] = s1.w;
    Rs[4 * NTHR + 0 * NTHR + tid] = s2.x; Rs[4 * NTHR + 1 * NTHR + tid] = s2.y; Rs[4 * NTHR + 2 * NTHR + tid] = s2.z; Rs[4 * NTHR + 3 * NTHR + tid] = s2.w;
    Rs[8 * NTHR + 0 * NTHR + tid] = s3.x; Rs[8 * NTHR + 1 * NTHR + tid] = s3.y; Rs[8 * NTHR + 2 * NTHR + tid] = s3.z; Rs[8 * NTHR + 3 * NTHR + tid] = s3.w;
    __syncthreads();
    if (tid < 3 * COB) {
        const int k = tid / COB, chn = tid % COB, qq = chn >> 2, e = chn & 3;
        float sum = 0.f;
#pragma unroll 4
        for (int g = 0; g < RP; ++g) sum += Rs[k * 4 * NTHR + e * NTHR + g * QN + qq];
        const size_t slot = (size_t)(blockIdx.x % BN_STAT_SLOTS) * 2 * a.Cout;
        if (want_st) { if (k < 2) atomicAdd(a.stats + slot + k * a.Cout + c0 + chn, (double)sum); }
        else if (k == 0) { atomicAdd(a.bst + slot + c0 + chn, (double)sum); if (a.bstB) atomicAdd(a.bstB + slot + c0 + chn, (double)sum); }
        else if (k == 1) atomicAdd(a.bst + slot + a.Cout + c0 + chn, (double)sum);
        else if (a.bstB) atomicAdd(a.bstB + slot + a.Cout + c0 + chn, (double)sum);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The kernel the step uses: 256 threads (wave = Winograd row i), TWO independent workgroups per CU, block tile = 64 tiles x 32 output channels, 8-channel chunks:
//   * a wave contracts BOTH 32-tile halves against the same U fragment (8 MFMAs per 1-KB fragment: half the U traffic per multiply of conv_wino_kernel), and the
//     fragment is private to the wave, so it goes global -> registers with no LDS round trip.  Four fragment registers refilled IN PLACE right behind a fragment's last use
//     in the chunk: three MFMA groups, the commit, the barrier and a transform phase of prefetch distance for 16 registers, and every U load is older than the patch loads
//     issued after it, so its counted wait never waits for them;
//   * row-segment patch staging (each staged pixel serves two tiles), branch-free loader (conditional stores make hipcc sink every prefetch load into the branch of its
//     store), commit pinned behind the chunk's MFMAs with sched_barrier + an empty asm on the loaded registers (instruction selection otherwise floats the padding masks,
//     and with them the loads' wait, to the top of the block).
// In-model at batch 32 (tools/run_layer_compare.sh): 0.215 -> 0.153 ms on the 64-channel layers, 0.217 -> 0.141 (128), 0.211 -> 0.160 (256), 0.218 -> 0.188 (512).
// ---------------------------------------------------------------------------------------------------------------------
// NH = 32-tile halves per workgroup.  NH = 2 (64 tiles, two workgroups per CU) halves the U traffic per multiply; NH = 1 (32 tiles, 64 accumulator registers: THREE workgroups per
// CU, 768 resident on the chip) is for launches with too few 64-tile workgroups to fill 512 slots twice -- the 1/16- and 1/32-scale layers left a third of the wave slots empty
// (SQ_WAVE_CYCLES of profiles/r03_wino3_pmc.md: 68 % of the slot-time occupied) -- and gives the scheduler a third wave per SIMD to cover the waits of the other two.
// CB (round 5) = 32-channel output blocks per workgroup.  CB = 2 (with NH = 1: 32 tiles x 64 channels, 128 accumulator registers, two workgroups per CU): the input transform of a
// chunk feeds the MFMAs of both channel blocks -- for the fp32 MFMA, which never co-executes with vector instructions, every transform instruction saved is matrix-pipe time --
// and a workgroup pulls 44 instead of 2 x 28 KB of operands per chunk into its CU.
template <int NLI, int XF, int BST, int NH = 2, int CB = 1>
__global__ __launch_bounds__(256, (NH == 1 && CB == 1) ? 3 : 2) void conv_wino3_kernel(ConvArgs a, Wino2Geo geo) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int COB = 32 * CB, WT = 32 * NH, NA = NH * CB;          // NA accumulator sets per position: (tile half th, channel block cb) -> th * CB + cb
    constexpr int PST = 128 * NLI * 8;               // floats per patch stage: every (slot, half) item of the loader has a real slot (those past geo.nslots are never read)
    const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;
    const int TW = geo.TW, TH = geo.TH, Mt = geo.Mt, ntp = geo.ntp;
    const int bid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_p = bid % ntp, tile_c = bid / ntp;
    const int m0 = tile_p * WT, c0 = tile_c * COB;
    const int THW = TH * TW;
    const int rr_first = m0 / TW, tx0 = m0 - rr_first * TW;
    const int L0 = min(TW - tx0, WT);
    const int S0 = 4 * (2 * L0 + 2), Lf = min(TW, WT), S = 4 * (2 * Lf + 2);
    const int n_rows_total = a.N * TH;

    // ---- loader: item f = tid + 256 li = (slot f >> 1, 16-B half f & 1); LDS position = slot * 8 + swizzled half, i.e. l_dst0 + li * 1024 floats
    // (round 5) geometry once per SLOT: the two 16-byte halves of a slot share everything but their channel offset -- thread t works out slots t + 256 q, the table goes through
    // the second pipeline stage's LDS (nobody writes it before the end of chunk 0, a barrier later), a thread picks up its NLI slots from there
    unsigned l_off[NLI]; unsigned l_ok = 0;
    const int h2 = tid & 1;
    const int l_dst0 = (tid >> 1) * 8 + ((h2 ^ ((tid >> 4) & 1)) << 2);
    {
        unsigned* tab = reinterpret_cast<unsigned*>(wsm) + PST;
        constexpr int NSL = 128 * NLI;
#pragma unroll
        for (int q = 0; q < (NSL + 255) / 256; ++q) {
            const int slot = tid + 256 * q;
            if (slot < NSL) {
                int k, r, WP, txa, Lk;
                if (slot < S0) { k = 0; r = slot; WP = 2 * L0 + 2; txa = tx0; Lk = L0; }
                else { const int s2 = slot - S0; k = 1 + fdiv24(s2, S); r = s2 - (k - 1) * S; txa = 0; Lk = max(min(TW, WT - L0 - (k - 1) * TW), 0); WP = 2 * Lk + 2; }          // (rows are as wide as the segment: the last one is short)
                const int pa = fdiv24(r, WP), pc = r - pa * WP;
                const int rr = rr_first + k;
                const bool act = slot < geo.nslots && pa < 4 && Lk > 0 && pc < 2 * Lk + 2 && rr < n_rows_total;
                const int rrc = act ? rr : 0;
                const int n = fdiv24(rrc, TH), ty = rrc - n * TH;
                int ih = 2 * ty - a.pad + pa, iw = 2 * txa - a.pad + pc;
                bool ok = act;
                if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
                else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                tab[slot] = ok ? ((unsigned)((long)n * a.x_bs) + (unsigned)((ih * a.W + iw) * a.x_ps)) * 4u : 0xFFFFFFFFu;
            }
        }
        __syncthreads();
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            const unsigned v = tab[(tid >> 1) + 128 * li];
            const bool ok = v != 0xFFFFFFFFu;
            l_ok |= ok ? (1u << li) : 0u;
            l_off[li] = ok ? v + (unsigned)(h2 * 16) : (XF != 0 ? (unsigned)(h2 * 16) : W_OOB);
        }
    }
    const char* xb = reinterpret_cast<const char*>(a.x);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, geo.xbytes, 0x00020000);
    uint4 ra[NLI];
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto gload = [&](int c) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            if constexpr (XF != 0) ra[li] = *reinterpret_cast<const uint4*>(xb + (l_off[li] + (unsigned)(c * 32)));
            else ra[li] = wino_buffer_load16(xrs, l_off[li], c * 32);
        }
        if constexpr (XF != 0) { xsc = *reinterpret_cast<const float4*>(a.in_scale + c * 8 + h2 * 4); xsh = *reinterpret_cast<const float4*>(a.in_shift + c * 8 + h2 * 4); }
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            const unsigned mk = 0u - ((l_ok >> li) & 1u);
            uint4 v = ra[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));          // pins the mask / transform arithmetic (and the loads' wait) at the commit point
            if constexpr (XF != 0) {
                v.x = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), xsc.x, xsh.x), 0.f)); v.y = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), xsc.y, xsh.y), 0.f));
                v.z = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), xsc.z, xsh.z), 0.f)); v.w = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), xsc.w, xsh.w), 0.f));
            }
            if constexpr (XF != 0) { v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk; }          // (XF == 0: out-of-range items came back as zeros from the buffer load)
            *reinterpret_cast<uint4*>(wsm + so + l_dst0 + li * 1024) = v;
        }
    };

    // ---- consumer: lane = (tile within a 32-tile half, k-half); B^T row i combines patch rows (r1, r2)
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const int tl = lane & 31, ch = lane >> 5;
    int c_off1[NH][4], c_off2[NH][4];
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        const int ct = th * 32 + tl;
        const int u = tx0 + ct, k = fdiv24(u, TW);
        const int lc = (k == 0) ? ct : (u - k * TW);
        const int base = (k == 0) ? 0 : S0 + (k - 1) * S, WP = (k == 0) ? 2 * L0 + 2 : 2 * min(TW, WT - L0 - (k - 1) * TW) + 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sl1 = base + r1 * WP + 2 * lc + b, sl2 = base + r2 * WP + 2 * lc + b;
            c_off1[th][b] = sl1 * 8 + ((ch ^ ((sl1 >> 3) & 1)) << 2);
            c_off2[th][b] = sl2 * 8 + ((ch ^ ((sl2 >> 3) & 1)) << 2);
#ifdef MD2_ABLATE
            if (a.ablate & 16) { c_off1[th][b] = tl * 4 + ch * 128 + th * 256 + b * 512; c_off2[th][b] = c_off1[th][b] + 2048; }      // timing experiment: conflict-free (wrong) read addresses
#endif
        }
    }

    f32x16 acc[4][NA];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NA; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][t][e] = 0.f;

    // U stream of this wave: fragment (c8, j) of cout sub-tile tile_c at ub + (c8 * 4 + j) * 256 floats
    const int nchunks = a.Cin >> 3;
    const float* ub = a.wino_u + ((long)(tile_c * CB * 4 + wi) * nchunks) * 4 * 256 + lane * 4;          // channel block cb of the workgroup: + cb * ubs
    const long ubs = (long)4 * nchunks * 4 * 256;
    f32x4 uf[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) uf[cb][j] = *reinterpret_cast<const f32x4*>(ub + cb * ubs + j * 256);

#ifdef MD2_ABLATE
    const int abl = a.ablate;
#else
    constexpr int abl = 0;
#endif
    gload(0); lstore(0);
    __syncthreads();
    // (round 5) the chunk loop is unrolled by two with the stage offset a compile-time constant: LDS read addresses are base registers + an immediate
    auto chunk = [&](const int c, auto so_tag) {
        constexpr int so = decltype(so_tag)::value;
        const int cn = min(c + 1, nchunks - 1);          // past the end: the last chunk again, into the stage nobody reads any more (keeps the loop body branch-free)
        const float* un = ub + (long)cn * 1024;
        if (!(abl & 1)) gload(cn);
        __builtin_amdgcn_sched_barrier(0);
#define W3_MFMA(th_, j_, uf_) do { if (!(abl & 4)) { \
                acc[j_][th_] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf_[0], V[j_].x, acc[j_][th_], 0, 0, 0); \
                acc[j_][th_] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf_[1], V[j_].y, acc[j_][th_], 0, 0, 0); \
                acc[j_][th_] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf_[2], V[j_].z, acc[j_][th_], 0, 0, 0); \
                acc[j_][th_] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf_[3], V[j_].w, acc[j_][th_], 0, 0, 0); \
            } else asm volatile("" :: "v"(uf_[0]), "v"(uf_[1]), "v"(uf_[2]), "v"(uf_[3]), "v"(V[j_].x)); } while (0)
#pragma unroll
        for (int th = 0; th < NH; ++th) {
            float4 V[4];
#ifdef MD2_ABLATE
            if (abl & 32) { V[0] = V[1] = V[2] = V[3] = make_float4(sg, sg, sg, sg); } else          // timing experiment: no LDS reads, no input transform
#endif
            {
                float4 rr[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 d1 = *reinterpret_cast<const float4*>(wsm + so + c_off1[th][b]);
                    const float4 d2 = *reinterpret_cast<const float4*>(wsm + so + c_off2[th][b]);
                    rr[b] = make_float4(fmaf(sg, d2.x, d1.x), fmaf(sg, d2.y, d1.y), fmaf(sg, d2.z, d1.z), fmaf(sg, d2.w, d1.w));
                }
                V[0] = make_float4(rr[0].x - rr[2].x, rr[0].y - rr[2].y, rr[0].z - rr[2].z, rr[0].w - rr[2].w);
                V[1] = make_float4(rr[1].x + rr[2].x, rr[1].y + rr[2].y, rr[1].z + rr[2].z, rr[1].w + rr[2].w);
                V[2] = make_float4(rr[2].x - rr[1].x, rr[2].y - rr[1].y, rr[2].z - rr[1].z, rr[2].w - rr[1].w);
                V[3] = make_float4(rr[1].x - rr[3].x, rr[1].y - rr[3].y, rr[1].z - rr[3].z, rr[1].w - rr[3].w);
            }
            // in-place refill right behind a fragment's last use in this chunk (tile half 1): the next use is three MFMA groups, the commit, the barrier and a transform
            // phase later -- for every j alike
            // (the sched_barriers keep hipcc from collecting the four loads behind the last MFMA)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                W3_MFMA(th * CB + cb, 0, uf[cb][0]); if (th == NH - 1 && !(abl & 2)) { uf[cb][0] = *reinterpret_cast<const f32x4*>(un + cb * ubs); __builtin_amdgcn_sched_barrier(0); }
                W3_MFMA(th * CB + cb, 1, uf[cb][1]); if (th == NH - 1 && !(abl & 2)) { uf[cb][1] = *reinterpret_cast<const f32x4*>(un + cb * ubs + 256); __builtin_amdgcn_sched_barrier(0); }
                W3_MFMA(th * CB + cb, 2, uf[cb][2]); if (th == NH - 1 && !(abl & 2)) { uf[cb][2] = *reinterpret_cast<const f32x4*>(un + cb * ubs + 512); __builtin_amdgcn_sched_barrier(0); }
                W3_MFMA(th * CB + cb, 3, uf[cb][3]); if (th == NH - 1 && !(abl & 2)) { uf[cb][3] = *reinterpret_cast<const f32x4*>(un + cb * ubs + 768); if (cb + 1 < CB) __builtin_amdgcn_sched_barrier(0); }
            }
        }
#undef W3_MFMA
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 1)) lstore(PST - so);
        if (!(abl & 8)) __syncthreads();
    };
    {
        int c = 0;
        for (; c + 1 < nchunks; c += 2) { chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, PST>{}); }
        if (c < nchunks) chunk(c, std::integral_constant<int, 0>{});
    }

#ifdef MD2_ABLATE
    if (abl & 64) { if (acc[0][0][0] + acc[1][NA - 1][3] + acc[2][0][5] + acc[3][NA - 1][7] == 123.456f) a.y[0] = 1.f; return; }          // timing experiment: no output transform / epilogue
#endif
    // ---- output transform: column half in registers, row half through LDS (R[i][b][64 tiles][COB + 4])
    constexpr int RLD = COB + 4;
    float* Rl = wsm;
#pragma unroll
    for (int tc = 0; tc < NA; ++tc)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int th = tc / CB, cb = tc % CB;
            float4 o0, o1;
            o0.x = acc[0][tc][4 * g + 0] + acc[1][tc][4 * g + 0] + acc[2][tc][4 * g + 0]; o1.x = acc[1][tc][4 * g + 0] - acc[2][tc][4 * g + 0] - acc[3][tc][4 * g + 0];
            o0.y = acc[0][tc][4 * g + 1] + acc[1][tc][4 * g + 1] + acc[2][tc][4 * g + 1]; o1.y = acc[1][tc][4 * g + 1] - acc[2][tc][4 * g + 1] - acc[3][tc][4 * g + 1];
            o0.z = acc[0][tc][4 * g + 2] + acc[1][tc][4 * g + 2] + acc[2][tc][4 * g + 2]; o1.z = acc[1][tc][4 * g + 2] - acc[2][tc][4 * g + 2] - acc[3][tc][4 * g + 2];
            o0.w = acc[0][tc][4 * g + 3] + acc[1][tc][4 * g + 3] + acc[2][tc][4 * g + 3]; o1.w = acc[1][tc][4 * g + 3] - acc[2][tc][4 * g + 3] - acc[3][tc][4 * g + 3];
            const int col = 32 * cb + 8 * g + 4 * ch, ct = th * 32 + tl;
            *reinterpret_cast<float4*>(Rl + ((wi * 2 + 0) * WT + ct) * RLD + col) = o0;
            *reinterpret_cast<float4*>(Rl + ((wi * 2 + 1) * WT + ct) * RLD + col) = o1;
        }
    __syncthreads();
    wino_final_epilogue<256, COB, BST>(a, wsm, m0, WT, Mt, THW, TW, c0, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same convolution with the multiplications on the bf16 matrix pipe at fp32 accuracy ("split" mode, md2_set_winograd(on = 4)): every fp32 operand is the exact sum of
// bf16 pieces, u = u_hi + u_lo (+ 2^-17 |u|), v = v_hi + v_lo, and  u v = u_hi v_hi + u_hi v_lo + u_lo v_hi  up to 2^-16 |u v| -- three v_mfma_f32_32x32x16_bf16 (fp32
// accumulation, products of bf16 pairs are exact in fp32) instead of eight v_mfma_f32_32x32x2_f32 per 16 channels: 96 instead of 512 matrix-pipe cycles.  More important
// than the cycle count: profiles/r03_wino3_pmc.md shows SQ_VALU_MFMA_COEXEC_CYCLES = 0 for the fp32 MFMA -- it shares its issue with every vector instruction of the SIMD --
// whereas the bf16 MFMA is an XDL op that runs beside the other wave's transforms, masks and address arithmetic.  The pieces of V are formed in registers right after the input
// transform (v_cvt_pk_bf16_f32 + one subtraction per value); the pieces of U are written by the weight transform (same bytes as the fp32 U: 2 x bf16).
// Workgroup = 32 tiles x 32 channels, 16-channel chunks staged as two 8-channel planes with the layout of conv_wino3_kernel (lane = tile, k-group = plane).
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
// Splitting fp32 values into bf16 pieces, two values per register.  Rounding to bf16 is v_cvt_pk_bf16_f32 (two lanes at once); the residual x - piece is ONE instruction as well:
// a bf16 dot product with the constants (-1, 0) / (0, -1) and x as the addend (v_dot2c_f32_bf16; exact -- checked bit for bit against shift / mask + subtract on 10^6 values).
// Everything is written with builtins, not inline asm: the result of a dot instruction needs wait states before its first use, and hipcc inserts them only between
// instructions it knows (an inline-asm v_cvt_pk right behind the dot read garbage).
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wbf16x2 cvt_pk_bf16(float a, float b) { wbf16x2 r; r[0] = (__bf16)a; r[1] = (__bf16)b; return r; }
// (the constant pair goes through an empty asm so that it stays a register operand: folded into the instruction as the inline constant -1.0 it selects the wrong half)
__device__ __forceinline__ wbf16x2 bf16_pair_const(unsigned bits) { asm("" : "+v"(bits)); return __builtin_bit_cast(wbf16x2, bits); }
__device__ __forceinline__ float bf16_resid_lo(float x, wbf16x2 pk) { return __builtin_amdgcn_fdot2_f32_bf16(pk, bf16_pair_const(0x0000bf80u), x, false); }
__device__ __forceinline__ float bf16_resid_hi(float x, wbf16x2 pk) { return __builtin_amdgcn_fdot2_f32_bf16(pk, bf16_pair_const(0xbf800000u), x, false); }
// (a, b) = hi + lo up to 2^-17: four vector instructions per pair
__device__ __forceinline__ void split_pk(float a, float b, unsigned& hi, unsigned& lo) {
    const wbf16x2 h = cvt_pk_bf16(a, b), l = cvt_pk_bf16(bf16_resid_lo(a, h), bf16_resid_hi(b, h));
    hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split_bf16(const float4& v, unsigned (&hi)[2], unsigned (&lo)[2]) {
    split_pk(v.x, v.y, hi[0], lo[0]); split_pk(v.z, v.w, hi[1], lo[1]);
}
// (a, b) = hi + mid + lo up to 2^-25: every bit of the fp32 values; seven vector instructions per pair
__device__ __forceinline__ void split3_pk(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    const wbf16x2 h = cvt_pk_bf16(a, b);
    const float ra = bf16_resid_lo(a, h), rb = bf16_resid_hi(b, h);
    const wbf16x2 m = cvt_pk_bf16(ra, rb), l = cvt_pk_bf16(bf16_resid_lo(ra, m), bf16_resid_hi(rb, m));
    hi = __builtin_bit_cast(unsigned, h); mid = __builtin_bit_cast(unsigned, m); lo = __builtin_bit_cast(unsigned, l);
}
// NP = pieces per operand: 2 (hi + lo: three MFMAs per product group, ~17-bit mantissas) or 3 (hi + mid + lo: six MFMAs -- hh, hm, mh, hl, lh, mm; the dropped terms are
// below 2^-24 of the product -- i.e. fp32 accuracy on the bf16 pipe, still 192 instead of 512 matrix-pipe cycles per 16 channels, and co-executing with the vector work)
template <int XF, int BST, int NP>
__global__ __launch_bounds__(256, 2) void conv_wino3s_kernel(ConvArgs a, Wino2Geo geo) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
#ifdef MD2_ABLATE
#define W3S_STAMP(k_) do { if (a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
    W3S_STAMP(0);
    if (a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#else
#define W3S_STAMP(k_) do { } while (0)
#endif
    constexpr int COB = 32, WT = 32, NLI = 6;
    constexpr int PL = 128 * 3 * 8;                  // floats per 8-channel plane (384 slots), two planes per stage
    constexpr int PST = 2 * PL;
    const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;
    const int TW = geo.TW, TH = geo.TH, Mt = geo.Mt, ntp = geo.ntp;
    const int bid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_p = bid % ntp, tile_c = bid / ntp;
    const int m0 = tile_p * WT, c0 = tile_c * COB;
    const int THW = TH * TW;
    const int rr_first = m0 / TW, tx0 = m0 - rr_first * TW;
    const int L0 = min(TW - tx0, WT);
    const int S0 = 4 * (2 * L0 + 2), Lf = min(TW, WT), S = 4 * (2 * Lf + 2);
    const int n_rows_total = a.N * TH;

    // U stream of this wave: chunk c16, position j: piece q at ub + ((c16 * 4 + j) * NP + q) * 512 bf16 (64 lanes x 8).  The first chunk's fragments are requested before
    // anything else (round 5): they depend on the workgroup id alone, and behind the barrier of the loader's geometry table their latency was the head of every workgroup's life
    const int nchunks = a.Cin >> 4;
    const __bf16* ub = reinterpret_cast<const __bf16*>(a.wino_u) + ((long)(tile_c * 4 + wi) * nchunks) * 4 * NP * 512 + lane * 8;
    wbf16x8 u[4][NP];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < NP; ++q) u[j][q] = *reinterpret_cast<const wbf16x8*>(ub + (j * NP + q) * 512);

    // ---- loader: item f = tid + 256 li = (slot f >> 2, 16-B piece f & 3 of the 16 channels); piece >> 1 = plane, piece & 1 = half inside the plane's 8-float slot
    // Geometry once per SLOT (round 5): the four 16-byte pieces of a slot share everything but their channel offset, so thread t works out the byte offset of slots t and
    // t + 256 (or the "nothing to load" marker), the table goes through the second pipeline stage's LDS (nobody writes it before the end of chunk 0, a barrier later) and a thread
    // picks up its six slots from there -- 2 instead of 6 evaluations of the segment / reflection / bounds arithmetic per thread (~400 of the ~750 prologue vector instructions).
    unsigned l_off[NLI]; unsigned l_ok = 0;
    const int piece = tid & 3, pl = piece >> 1, h2 = piece & 1;
    const int l_dst0 = pl * PL + (tid >> 2) * 8 + ((h2 ^ ((tid >> 5) & 1)) << 2);          // + li * 512 floats (64 slots)
    {
        unsigned* tab = reinterpret_cast<unsigned*>(wsm) + PST;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int slot = tid + 256 * q;
            if (slot < 384) {
                int k, r, WP, txa, Lk;
                if (slot < S0) { k = 0; r = slot; WP = 2 * L0 + 2; txa = tx0; Lk = L0; }
                else { const int s2 = slot - S0; k = 1 + fdiv24(s2, S); r = s2 - (k - 1) * S; txa = 0; Lk = max(min(TW, WT - L0 - (k - 1) * TW), 0); WP = 2 * Lk + 2; }
                const int pa = fdiv24(r, WP), pc = r - pa * WP;
                const int rr = rr_first + k;
                const bool act = slot < geo.nslots && pa < 4 && Lk > 0 && pc < 2 * Lk + 2 && rr < n_rows_total;
                const int rrc = act ? rr : 0;
                const int n = fdiv24(rrc, TH), ty = rrc - n * TH;
                int ih = 2 * ty - a.pad + pa, iw = 2 * txa - a.pad + pc;
                bool ok = act;
                if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
                else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                tab[slot] = ok ? ((unsigned)((long)n * a.x_bs) + (unsigned)((ih * a.W + iw) * a.x_ps)) * 4u : 0xFFFFFFFFu;          // (valid offsets are multiples of 16 below 2^32 - 16)
            }
        }
        __syncthreads();
        W3S_STAMP(1);
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            const unsigned v = tab[(tid >> 2) + 64 * li];
            const bool ok = v != 0xFFFFFFFFu;
            l_ok |= ok ? (1u << li) : 0u;
            l_off[li] = ok ? v + (unsigned)(piece * 16) : (XF != 0 ? (unsigned)(piece * 16) : W_OOB);
        }
    }
    const char* xb = reinterpret_cast<const char*>(a.x);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, geo.xbytes, 0x00020000);
    uint4 ra[NLI];
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto gload = [&](int c) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            if constexpr (XF != 0) ra[li] = *reinterpret_cast<const uint4*>(xb + (l_off[li] + (unsigned)(c * 64)));
            else ra[li] = wino_buffer_load16(xrs, l_off[li], c * 64);
        }
        if constexpr (XF != 0) { xsc = *reinterpret_cast<const float4*>(a.in_scale + c * 16 + piece * 4); xsh = *reinterpret_cast<const float4*>(a.in_shift + c * 16 + piece * 4); }
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            const unsigned mk = 0u - ((l_ok >> li) & 1u);
            uint4 v = ra[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            if constexpr (XF != 0) {
                v.x = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), xsc.x, xsh.x), 0.f)); v.y = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), xsc.y, xsh.y), 0.f));
                v.z = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), xsc.z, xsh.z), 0.f)); v.w = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), xsc.w, xsh.w), 0.f));
            }
            if constexpr (XF != 0) { v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk; }
            *reinterpret_cast<uint4*>(wsm + so + l_dst0 + li * 512) = v;
        }
    };

    // ---- consumer: lane = (tile, k-group g = plane); B^T row i combines patch rows (r1, r2); the lane's 8 channels are the plane's two 16-B pieces
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const int tl = lane & 31, g = lane >> 5;
    int c_off1[4], c_off2[4];          // piece 0 of the slot; piece 1 is at (offset ^ 4)
    {
        const int u = tx0 + tl, k = fdiv24(u, TW);
        const int lc = (k == 0) ? tl : (u - k * TW);
        const int base = (k == 0) ? 0 : S0 + (k - 1) * S, WP = (k == 0) ? 2 * L0 + 2 : 2 * min(TW, WT - L0 - (k - 1) * TW) + 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sl1 = base + r1 * WP + 2 * lc + b, sl2 = base + r2 * WP + 2 * lc + b;
            c_off1[b] = g * PL + sl1 * 8 + (((sl1 >> 3) & 1) << 2);
            c_off2[b] = g * PL + sl2 * 8 + (((sl2 >> 3) & 1) << 2);
        }
    }

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    gload(0); lstore(0);
    __syncthreads();
    W3S_STAMP(2);
    // the chunk loop is unrolled by two with the stage offset a compile-time constant (round 5): the 16 LDS read addresses of a chunk are then base registers + an immediate
    // instead of 16 v_lshl_add per chunk
    auto chunk = [&](const int c, auto so_tag) {
        constexpr int so = decltype(so_tag)::value;
        const int cn = min(c + 1, nchunks - 1);
        const __bf16* un = ub + (long)cn * (4 * NP * 512);
        gload(cn);
        __builtin_amdgcn_sched_barrier(0);
        // B^T row combination of the lane's 8 channels (two 16-B pieces), kept for the four positions
        float4 rr[4][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 d1 = *reinterpret_cast<const float4*>(wsm + so + (c_off1[b] ^ (p << 2)));
                const float4 d2 = *reinterpret_cast<const float4*>(wsm + so + (c_off2[b] ^ (p << 2)));
                rr[b][p] = make_float4(fmaf(sg, d2.x, d1.x), fmaf(sg, d2.y, d1.y), fmaf(sg, d2.z, d1.z), fmaf(sg, d2.w, d1.w));
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // V_(i,j) of the 8 channels, split into NP bf16 pieces right here (the pieces of one position live only across its MFMAs)
            unsigned vp[NP][4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 x0 = rr[j == 0 ? 0 : (j == 2 ? 2 : 1)][p], x1 = rr[j == 0 ? 2 : (j == 1 ? 2 : (j == 2 ? 1 : 3))][p];
                const float sj = (j == 1) ? 1.f : -1.f;
                const float4 v = make_float4(fmaf(sj, x1.x, x0.x), fmaf(sj, x1.y, x0.y), fmaf(sj, x1.z, x0.z), fmaf(sj, x1.w, x0.w));
                if constexpr (NP == 2) { split_pk(v.x, v.y, vp[0][2 * p], vp[1][2 * p]); split_pk(v.z, v.w, vp[0][2 * p + 1], vp[1][2 * p + 1]); }
                else { split3_pk(v.x, v.y, vp[0][2 * p], vp[1][2 * p], vp[2][2 * p]); split3_pk(v.z, v.w, vp[0][2 * p + 1], vp[1][2 * p + 1], vp[2][2 * p + 1]); }
            }
            wbf16x8 vb[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) vb[q] = __builtin_bit_cast(wbf16x8, (wu32x4){vp[q][0], vp[q][1], vp[q][2], vp[q][3]});
            // small terms first
            if constexpr (NP == 3) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][2], vb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][0], vb[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][1], vb[1], acc[j], 0, 0, 0);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][1], vb[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][0], vb[1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][0], vb[0], acc[j], 0, 0, 0);
            // refill in place right behind the position's last use, as in conv_wino3_kernel
#pragma unroll
            for (int q = 0; q < NP; ++q) u[j][q] = *reinterpret_cast<const wbf16x8*>(un + (j * NP + q) * 512);
            __builtin_amdgcn_sched_barrier(0);
        }
        lstore(PST - so);
        __syncthreads();
    };
    {
        int c = 0;
        for (; c + 1 < nchunks; c += 2) { chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, PST>{}); }
        if (c < nchunks) chunk(c, std::integral_constant<int, 0>{});
    }
    W3S_STAMP(3);

    // ---- output transform: column half in registers, row half through LDS (R[i][b][32 tiles][COB + 4]) -- as conv_wino3_kernel
    constexpr int RLD = COB + 4;
    float* Rl = wsm;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        float4 o0, o1;
        o0.x = acc[0][4 * gq + 0] + acc[1][4 * gq + 0] + acc[2][4 * gq + 0]; o1.x = acc[1][4 * gq + 0] - acc[2][4 * gq + 0] - acc[3][4 * gq + 0];
        o0.y = acc[0][4 * gq + 1] + acc[1][4 * gq + 1] + acc[2][4 * gq + 1]; o1.y = acc[1][4 * gq + 1] - acc[2][4 * gq + 1] - acc[3][4 * gq + 1];
        o0.z = acc[0][4 * gq + 2] + acc[1][4 * gq + 2] + acc[2][4 * gq + 2]; o1.z = acc[1][4 * gq + 2] - acc[2][4 * gq + 2] - acc[3][4 * gq + 2];
        o0.w = acc[0][4 * gq + 3] + acc[1][4 * gq + 3] + acc[2][4 * gq + 3]; o1.w = acc[1][4 * gq + 3] - acc[2][4 * gq + 3] - acc[3][4 * gq + 3];
        const int col = 8 * gq + 4 * g;
        *reinterpret_cast<float4*>(Rl + ((wi * 2 + 0) * WT + tl) * RLD + col) = o0;
        *reinterpret_cast<float4*>(Rl + ((wi * 2 + 1) * WT + tl) * RLD + col) = o1;
    }
    __syncthreads();
    W3S_STAMP(4);
    wino_final_epilogue<256, COB, BST>(a, wsm, m0, WT, Mt, THW, TW, c0, tid);
    W3S_STAMP(5);
}
#undef W3S_STAMP

// ---------------------------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) on bf16 TENSORS (md2_config.precision = 2; round 6, the review's "missing": those tensors took the 9-tap direct implicit GEMM on every 3x3 / stride-1
// layer).  conv_wino3s_kernel with one bf16 piece per operand: x, y and the epilogue's tensors are bf16 in HBM; a loader item is four channels = 8 bytes, widened (exactly) to
// fp32 on its way into the unchanged LDS patch layout; the input transform runs in fp32 and V is rounded to bf16 ONCE; U = G g G^T is written as one bf16 piece by the weight
// transform from the fp32 master weights (ConvArgs::wino_pieces = 4); one v_mfma_f32_32x32x16_bf16 per position and 16-channel chunk, fp32 accumulation, output transform and
// epilogue in fp32, one rounding on the store.  Against the direct kernel on bf16 tensors the products differ by V's and U's roundings (the direct kernel multiplies the stored
// bf16 values exactly): the layer tests bound it against the fp32 oracle on bf16-rounded inputs like every other kernel of that mode.
// ---------------------------------------------------------------------------------------------------------------------
template <int BST>
__global__ __launch_bounds__(256, 3) void conv_wino3b_kernel(ConvArgs a, Wino2Geo geo) {
    constexpr int NP = 1;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int COB = 32, WT = 32, NLI = 3;          // loader items per thread: (slot, 8-channel plane) = 16 bytes of bf16
    constexpr int PL = 128 * 3 * 8;                  // floats per 8-channel plane (384 slots), two planes per stage
    constexpr int PST = 2 * PL;
    const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;
    const int TW = geo.TW, TH = geo.TH, Mt = geo.Mt, ntp = geo.ntp;
    const int bid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_p = bid % ntp, tile_c = bid / ntp;
    const int m0 = tile_p * WT, c0 = tile_c * COB;
    const int THW = TH * TW;
    const int rr_first = m0 / TW, tx0 = m0 - rr_first * TW;
    const int L0 = min(TW - tx0, WT);
    const int S0 = 4 * (2 * L0 + 2), Lf = min(TW, WT), S = 4 * (2 * Lf + 2);
    const int n_rows_total = a.N * TH;

    // U stream of this wave: chunk c16, position j: piece q at ub + ((c16 * 4 + j) * NP + q) * 512 bf16 (64 lanes x 8).  The first chunk's fragments are requested before
    // anything else (round 5): they depend on the workgroup id alone, and behind the barrier of the loader's geometry table their latency was the head of every workgroup's life
    const int nchunks = a.Cin >> 4;
    const __bf16* ub = reinterpret_cast<const __bf16*>(a.wino_u) + ((long)(tile_c * 4 + wi) * nchunks) * 4 * NP * 512 + lane * 8;
    wbf16x8 u[4][NP];          // one bf16 piece per position
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < NP; ++q) u[j][q] = *reinterpret_cast<const wbf16x8*>(ub + (j * NP + q) * 512);

    // ---- loader: item f = tid + 256 li = (slot f >> 1, plane f & 1): the plane's eight bf16 channels = ONE 16-byte load, widened into the plane's two 16-byte LDS pieces.
    // Geometry once per slot through the table, as conv_wino3s_kernel.
    unsigned l_off[NLI];
    const int pl = tid & 1, sw = ((tid >> 4) & 1) << 2;          // slot (tid >> 1) + 128 li: bit 3 of the slot is bit 4 of tid
    const int l_dst0 = pl * PL + (tid >> 1) * 8;          // + li * 1024 floats (128 slots); channels 0..3 at + sw, 4..7 at + (sw ^ 4)
    {
        unsigned* tab = reinterpret_cast<unsigned*>(wsm) + PST;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int slot = tid + 256 * q;
            if (slot < 384) {
                int k, r, WP, txa, Lk;
                if (slot < S0) { k = 0; r = slot; WP = 2 * L0 + 2; txa = tx0; Lk = L0; }
                else { const int s2 = slot - S0; k = 1 + fdiv24(s2, S); r = s2 - (k - 1) * S; txa = 0; Lk = max(min(TW, WT - L0 - (k - 1) * TW), 0); WP = 2 * Lk + 2; }
                const int pa = fdiv24(r, WP), pc = r - pa * WP;
                const int rr = rr_first + k;
                const bool act = slot < geo.nslots && pa < 4 && Lk > 0 && pc < 2 * Lk + 2 && rr < n_rows_total;
                const int rrc = act ? rr : 0;
                const int n = fdiv24(rrc, TH), ty = rrc - n * TH;
                int ih = 2 * ty - a.pad + pa, iw = 2 * txa - a.pad + pc;
                bool ok = act;
                if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
                else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                tab[slot] = ok ? ((unsigned)((long)n * a.x_bs) + (unsigned)((ih * a.W + iw) * a.x_ps)) * 2u : 0xFFFFFFFFu;          // (bf16 elements: valid offsets are multiples of 8 below 2^32 - 16)
            }
        }
        __syncthreads();
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            const unsigned v = tab[(tid >> 1) + 128 * li];
            l_off[li] = v != 0xFFFFFFFFu ? v + (unsigned)(pl * 16) : W_OOB;          // the plane's eight channels: 16 bytes
        }
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, geo.xbytes, 0x00020000);
    uint4 ra[NLI];          // eight bf16 channels per item
    auto gload = [&](int c) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) ra[li] = wino_buffer_load16(xrs, l_off[li], c * 32);          // (a chunk's 16 channels are 32 bytes of a pixel)
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            uint4 r = ra[li];
            asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
            // bf16 -> fp32: exact; padding arrives as zeros from the range check
            *reinterpret_cast<uint4*>(wsm + so + l_dst0 + li * 1024 + sw) = make_uint4(r.x << 16, r.x & 0xFFFF0000u, r.y << 16, r.y & 0xFFFF0000u);
            *reinterpret_cast<uint4*>(wsm + so + l_dst0 + li * 1024 + (sw ^ 4)) = make_uint4(r.z << 16, r.z & 0xFFFF0000u, r.w << 16, r.w & 0xFFFF0000u);
        }
    };

    // ---- consumer: lane = (tile, k-group g = plane); B^T row i combines patch rows (r1, r2); the lane's 8 channels are the plane's two 16-B pieces
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const int tl = lane & 31, g = lane >> 5;
    int c_off1[4], c_off2[4];          // piece 0 of the slot; piece 1 is at (offset ^ 4)
    {
        const int u = tx0 + tl, k = fdiv24(u, TW);
        const int lc = (k == 0) ? tl : (u - k * TW);
        const int base = (k == 0) ? 0 : S0 + (k - 1) * S, WP = (k == 0) ? 2 * L0 + 2 : 2 * min(TW, WT - L0 - (k - 1) * TW) + 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sl1 = base + r1 * WP + 2 * lc + b, sl2 = base + r2 * WP + 2 * lc + b;
            c_off1[b] = g * PL + sl1 * 8 + (((sl1 >> 3) & 1) << 2);
            c_off2[b] = g * PL + sl2 * 8 + (((sl2 >> 3) & 1) << 2);
        }
    }

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    gload(0); lstore(0);
    __syncthreads();
    // the chunk loop is unrolled by two with the stage offset a compile-time constant (round 5): the 16 LDS read addresses of a chunk are then base registers + an immediate
    // instead of 16 v_lshl_add per chunk
    auto chunk = [&](const int c, auto so_tag) {
        constexpr int so = decltype(so_tag)::value;
        const int cn = min(c + 1, nchunks - 1);
        const __bf16* un = ub + (long)cn * (4 * NP * 512);
        gload(cn);
        __builtin_amdgcn_sched_barrier(0);
        // B^T row combination of the lane's 8 channels (two 16-B pieces), kept for the four positions
        float4 rr[4][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 d1 = *reinterpret_cast<const float4*>(wsm + so + (c_off1[b] ^ (p << 2)));
                const float4 d2 = *reinterpret_cast<const float4*>(wsm + so + (c_off2[b] ^ (p << 2)));
                rr[b][p] = make_float4(fmaf(sg, d2.x, d1.x), fmaf(sg, d2.y, d1.y), fmaf(sg, d2.z, d1.z), fmaf(sg, d2.w, d1.w));
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // V_(i,j) of the 8 channels, rounded to bf16 once (two values per v_cvt_pk_bf16_f32)
            unsigned vp[4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 x0 = rr[j == 0 ? 0 : (j == 2 ? 2 : 1)][p], x1 = rr[j == 0 ? 2 : (j == 1 ? 2 : (j == 2 ? 1 : 3))][p];
                const float sj = (j == 1) ? 1.f : -1.f;
                const float4 v = make_float4(fmaf(sj, x1.x, x0.x), fmaf(sj, x1.y, x0.y), fmaf(sj, x1.z, x0.z), fmaf(sj, x1.w, x0.w));
                vp[2 * p] = __builtin_bit_cast(unsigned, cvt_pk_bf16(v.x, v.y)); vp[2 * p + 1] = __builtin_bit_cast(unsigned, cvt_pk_bf16(v.z, v.w));
            }
            const wbf16x8 vb = __builtin_bit_cast(wbf16x8, (wu32x4){vp[0], vp[1], vp[2], vp[3]});
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][0], vb, acc[j], 0, 0, 0);
            // refill in place right behind the position's last use, as in conv_wino3_kernel
#pragma unroll
            for (int q = 0; q < NP; ++q) u[j][q] = *reinterpret_cast<const wbf16x8*>(un + (j * NP + q) * 512);
            __builtin_amdgcn_sched_barrier(0);
        }
        lstore(PST - so);
        __syncthreads();
    };
    {
        int c = 0;
        for (; c + 1 < nchunks; c += 2) { chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, PST>{}); }
        if (c < nchunks) chunk(c, std::integral_constant<int, 0>{});
    }

    // ---- output transform: column half in registers, row half through LDS (R[i][b][32 tiles][COB + 4]) -- as conv_wino3_kernel
    constexpr int RLD = COB + 4;
    float* Rl = wsm;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        float4 o0, o1;
        o0.x = acc[0][4 * gq + 0] + acc[1][4 * gq + 0] + acc[2][4 * gq + 0]; o1.x = acc[1][4 * gq + 0] - acc[2][4 * gq + 0] - acc[3][4 * gq + 0];
        o0.y = acc[0][4 * gq + 1] + acc[1][4 * gq + 1] + acc[2][4 * gq + 1]; o1.y = acc[1][4 * gq + 1] - acc[2][4 * gq + 1] - acc[3][4 * gq + 1];
        o0.z = acc[0][4 * gq + 2] + acc[1][4 * gq + 2] + acc[2][4 * gq + 2]; o1.z = acc[1][4 * gq + 2] - acc[2][4 * gq + 2] - acc[3][4 * gq + 2];
        o0.w = acc[0][4 * gq + 3] + acc[1][4 * gq + 3] + acc[2][4 * gq + 3]; o1.w = acc[1][4 * gq + 3] - acc[2][4 * gq + 3] - acc[3][4 * gq + 3];
        const int col = 8 * gq + 4 * g;
        *reinterpret_cast<float4*>(Rl + ((wi * 2 + 0) * WT + tl) * RLD + col) = o0;
        *reinterpret_cast<float4*>(Rl + ((wi * 2 + 1) * WT + tl) * RLD + col) = o1;
    }
    __syncthreads();
    wino_final_epilogue<256, COB, BST, true>(a, wsm, m0, WT, Mt, THW, TW, c0, tid);
}



// ---------------------------------------------------------------------------------------------------------------------
// Persistent (tile-looping) form of conv_wino3s_kernel (round 6).  Round 5's ablation put a third of a launch of that kernel into per-workgroup fixed cost -- 10.5 k ticks of
// every workgroup's life outside the chunk loop: first U fragments, the loader's geometry through LDS, the first patch loads, their commit, then at the end the output
// transform and the epilogue -- a latency chain that the second workgroup of the CU only partly covers.  Here the grid is the RESIDENT set (two workgroups per CU, 512) and a
// workgroup walks its share of (tile group, 32-channel block) items; the head of item k + 1 rides inside the loop of item k:
//   * chunk n - 2 of an item works out the NEXT item's loader geometry into a table of its own in LDS (published by that chunk's barrier);
//   * the LAST chunk prefetches, exactly where every chunk prefetches its successor, chunk 0 of the next item: patch loads from the next item's offsets, U refills from
//     the next item's fragment stream, the commit into stage 0 -- so the next loop starts with its operands in place (the chunk count is even, stage 0 is the free one);
//   * the output transform and the statistics scratch live behind stage 0 (stage 1 + 12 KB: 62 KB of LDS per workgroup), so the epilogue of item k does not touch what the
//     last chunk staged; one barrier separates it from the first commit of item k + 1 into stage 1.
// Items are dealt XCD by XCD in contiguous ranges (blocks b and b + 8 share an XCD): neighbouring items -- same U, adjacent tiles -- run at the same time on one L2, as the
// hardware's dispatch order did for the one-item workgroups.  No input transform (XF): those launches keep conv_wino3s_kernel.  Same U layout, loader, LDS patch layout,
// arithmetic order and epilogue contract: results are bit-identical to conv_wino3s_kernel's (tests/test_gpu_layers.py).
// ---------------------------------------------------------------------------------------------------------------------
template <int BST, int NP>
__global__ __launch_bounds__(256, 2) void conv_wino3p_kernel(ConvArgs a, Wino2Geo geo, int nitems) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int COB = 32, WT = 32, NLI = 6;
    constexpr int PL = 128 * 3 * 8;                  // floats per 8-channel plane (384 slots), two planes per stage
    constexpr int PST = 2 * PL;
    constexpr int R_OFF = PST;                       // output transform R[8][32][36] (9216 floats = stage 1 + PL) and the statistics scratch
    constexpr int TAB_OFF = 2 * PST + PL;            // loader geometry of the next item: 384 words
    const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;
    const int TW = geo.TW, TH = geo.TH, Mt = geo.Mt, ntp = geo.ntp;
    const int THW = TH * TW;
    const int Lf = min(TW, WT), S = 4 * (2 * Lf + 2);
    const int n_rows_total = a.N * TH;
    const int nchunks = a.Cin >> 4;                  // even, >= 4 (launcher)

    // ---- this workgroup's items: local index it, it + nl, ... of its XCD's contiguous range [it_lo, it_lo + it_cnt)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, nl = (nwg - xcd + 7) >> 3;
    const int q8 = nitems >> 3, r8 = nitems & 7;
    const int it_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, it_cnt = q8 + (xcd < r8 ? 1 : 0);
    int it = blockIdx.x >> 3;
    if (it >= it_cnt) return;                        // (workgroup-uniform, ahead of every barrier; does not happen for grid <= nitems)
    struct Item { int m0, c0, tile_c, rr_first, tx0, L0, S0; };
    auto decode = [&](int idl) {
        const int id = it_lo + idl;
        Item t; t.tile_c = id / ntp; const int tile_p = id - t.tile_c * ntp;
        t.m0 = tile_p * WT; t.c0 = t.tile_c * COB;
        t.rr_first = t.m0 / TW; t.tx0 = t.m0 - t.rr_first * TW;
        t.L0 = min(TW - t.tx0, WT); t.S0 = 4 * (2 * t.L0 + 2);
        return t;
    };
    const long ustride = (long)4 * NP * 512;         // bf16 elements per chunk of a wave's U stream
    auto ustream = [&](const Item& t) { return reinterpret_cast<const __bf16*>(a.wino_u) + ((long)(t.tile_c * 4 + wi) * nchunks) * ustride + lane * 8; };

    Item cur = decode(it);
    const __bf16* ub = ustream(cur);
    wbf16x8 u[4][NP];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < NP; ++q) u[j][q] = *reinterpret_cast<const wbf16x8*>(ub + (j * NP + q) * 512);

    // ---- loader (as conv_wino3s_kernel): item f = tid + 256 li = (slot f >> 2, 16-B piece f & 3); geometry once per slot through the table
    unsigned l_off[NLI];
    const int piece = tid & 3, pl = piece >> 1, h2 = piece & 1;
    const int l_dst0 = pl * PL + (tid >> 2) * 8 + ((h2 ^ ((tid >> 5) & 1)) << 2);
    unsigned* tab = reinterpret_cast<unsigned*>(wsm) + TAB_OFF;
    // (fresh(): the thread index behind an empty asm -- what the per-item helpers derive from it cannot be hoisted out of the item loop, where it would sit in registers across
    //  every chunk; with the hoisting the kernel needed 20 registers more than conv_wino3s_kernel and its BST form spilled)
    auto fresh = [&](int v) { asm volatile("" : "+v"(v)); return v; };
    auto fill_tab = [&](const Item& t, const bool valid) {
        const int tid = fresh(threadIdx.x);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int slot = tid + 256 * q;
            if (slot < 384) {
                int k, r, WP, txa, Lk;
                if (slot < t.S0) { k = 0; r = slot; WP = 2 * t.L0 + 2; txa = t.tx0; Lk = t.L0; }
                else { const int s2 = slot - t.S0; k = 1 + fdiv24(s2, S); r = s2 - (k - 1) * S; txa = 0; Lk = max(min(TW, WT - t.L0 - (k - 1) * TW), 0); WP = 2 * Lk + 2; }
                const int pa = fdiv24(r, WP), pc = r - pa * WP;
                const int rr = t.rr_first + k;
                const bool act = valid && slot < geo.nslots && pa < 4 && Lk > 0 && pc < 2 * Lk + 2 && rr < n_rows_total;
                const int rrc = act ? rr : 0;
                const int n = fdiv24(rrc, TH), ty = rrc - n * TH;
                int ih = 2 * ty - a.pad + pa, iw = 2 * txa - a.pad + pc;
                bool ok = act;
                if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
                else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                tab[slot] = ok ? ((unsigned)((long)n * a.x_bs) + (unsigned)((ih * a.W + iw) * a.x_ps)) * 4u : 0xFFFFFFFFu;
            }
        }
    };
    auto read_tab = [&]() {
        const int tid = fresh(threadIdx.x);
        const int piece = tid & 3;
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            const unsigned v = tab[(tid >> 2) + 64 * li];
            l_off[li] = v != 0xFFFFFFFFu ? v + (unsigned)(piece * 16) : W_OOB;
        }
    };
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, geo.xbytes, 0x00020000);
    uint4 ra[NLI];
    auto gload = [&](int c) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) ra[li] = wino_buffer_load16(xrs, l_off[li], c * 64);
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            uint4 v = ra[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            *reinterpret_cast<uint4*>(wsm + so + l_dst0 + li * 512) = v;
        }
    };

    // ---- consumer (as conv_wino3s_kernel)
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const int tl = lane & 31, g = lane >> 5;
    int c_off1[4], c_off2[4];
    auto set_consumer = [&](const Item& t) {
        const int lane = fresh(threadIdx.x) & 63, tl = lane & 31, g = lane >> 5;
        const int uu = t.tx0 + tl, k = fdiv24(uu, TW);
        const int lc = (k == 0) ? tl : (uu - k * TW);
        const int base = (k == 0) ? 0 : t.S0 + (k - 1) * S, WP = (k == 0) ? 2 * t.L0 + 2 : 2 * min(TW, WT - t.L0 - (k - 1) * TW) + 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sl1 = base + r1 * WP + 2 * lc + b, sl2 = base + r2 * WP + 2 * lc + b;
            c_off1[b] = g * PL + sl1 * 8 + (((sl1 >> 3) & 1) << 2);
            c_off2[b] = g * PL + sl2 * 8 + (((sl2 >> 3) & 1) << 2);
        }
    };

    fill_tab(cur, true);
    __syncthreads();
    read_tab();
    set_consumer(cur);
    gload(0); lstore(0);
    __syncthreads();

    f32x16 acc[4];
    // one chunk: prefetch the patches of (cn, with the offsets l_off holds at that moment) and the U fragments at `un`, contract the chunk staged at `so`, commit the prefetch
    // into the other stage.  The body is conv_wino3s_kernel's.
    auto chunk = [&](const int cn, const __bf16* un, auto so_tag) {
        constexpr int so = decltype(so_tag)::value;
        gload(cn);
        __builtin_amdgcn_sched_barrier(0);
        float4 rr[4][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                // (the offsets reach this loop through the item switch: without the hint hipcc no longer sees their 16-byte alignment and reads with 64 ds_read2_b32 per chunk)
                const float4 d1 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(wsm + so + (c_off1[b] ^ (p << 2)), 16));
                const float4 d2 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(wsm + so + (c_off2[b] ^ (p << 2)), 16));
                rr[b][p] = make_float4(fmaf(sg, d2.x, d1.x), fmaf(sg, d2.y, d1.y), fmaf(sg, d2.z, d1.z), fmaf(sg, d2.w, d1.w));
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned vp[NP][4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 x0 = rr[j == 0 ? 0 : (j == 2 ? 2 : 1)][p], x1 = rr[j == 0 ? 2 : (j == 1 ? 2 : (j == 2 ? 1 : 3))][p];
                const float sj = (j == 1) ? 1.f : -1.f;
                const float4 v = make_float4(fmaf(sj, x1.x, x0.x), fmaf(sj, x1.y, x0.y), fmaf(sj, x1.z, x0.z), fmaf(sj, x1.w, x0.w));
                if constexpr (NP == 2) { split_pk(v.x, v.y, vp[0][2 * p], vp[1][2 * p]); split_pk(v.z, v.w, vp[0][2 * p + 1], vp[1][2 * p + 1]); }
                else { split3_pk(v.x, v.y, vp[0][2 * p], vp[1][2 * p], vp[2][2 * p]); split3_pk(v.z, v.w, vp[0][2 * p + 1], vp[1][2 * p + 1], vp[2][2 * p + 1]); }
            }
            wbf16x8 vb[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) vb[q] = __builtin_bit_cast(wbf16x8, (wu32x4){vp[q][0], vp[q][1], vp[q][2], vp[q][3]});
            if constexpr (NP == 3) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][2], vb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][0], vb[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][1], vb[1], acc[j], 0, 0, 0);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][1], vb[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][0], vb[1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[j][0], vb[0], acc[j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NP; ++q) u[j][q] = *reinterpret_cast<const wbf16x8*>(un + (j * NP + q) * 512);
            __builtin_amdgcn_sched_barrier(0);
        }
        lstore(PST - so);
        __syncthreads();
    };

    while (true) {
        const int itn = it + nl;
        const bool has_next = itn < it_cnt;
        const Item nx = decode(has_next ? itn : it);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        int c = 0;
        for (; c + 2 < nchunks; c += 2) {
            chunk(c + 1, ub + (long)(c + 1) * ustride, std::integral_constant<int, 0>{});
            chunk(c + 2, ub + (long)(c + 2) * ustride, std::integral_constant<int, PST>{});
        }
        // the last two chunks: the next item's geometry goes into the table during chunk n - 2 (that chunk's barrier publishes it), and chunk n - 1 prefetches the next
        // item's chunk 0 -- or, behind the workgroup's last item, out-of-range offsets (zeros, no traffic) and its own U stream once more
        fill_tab(nx, has_next);
        chunk(c + 1, ub + (long)(c + 1) * ustride, std::integral_constant<int, 0>{});
        read_tab();
        ub = ustream(nx);                              // (this item's stream has been read to its end)
        chunk(0, ub, std::integral_constant<int, PST>{});

        // ---- output transform: column half in registers, row half through LDS (R[i][b][32 tiles][COB + 4]) behind stage 0
        constexpr int RLD = COB + 4;
        float* Rl = wsm + R_OFF;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float4 o0, o1;
            o0.x = acc[0][4 * gq + 0] + acc[1][4 * gq + 0] + acc[2][4 * gq + 0]; o1.x = acc[1][4 * gq + 0] - acc[2][4 * gq + 0] - acc[3][4 * gq + 0];
            o0.y = acc[0][4 * gq + 1] + acc[1][4 * gq + 1] + acc[2][4 * gq + 1]; o1.y = acc[1][4 * gq + 1] - acc[2][4 * gq + 1] - acc[3][4 * gq + 1];
            o0.z = acc[0][4 * gq + 2] + acc[1][4 * gq + 2] + acc[2][4 * gq + 2]; o1.z = acc[1][4 * gq + 2] - acc[2][4 * gq + 2] - acc[3][4 * gq + 2];
            o0.w = acc[0][4 * gq + 3] + acc[1][4 * gq + 3] + acc[2][4 * gq + 3]; o1.w = acc[1][4 * gq + 3] - acc[2][4 * gq + 3] - acc[3][4 * gq + 3];
            const int col = 8 * gq + 4 * g;
            *reinterpret_cast<float4*>(Rl + ((wi * 2 + 0) * WT + tl) * RLD + col) = o0;
            *reinterpret_cast<float4*>(Rl + ((wi * 2 + 1) * WT + tl) * RLD + col) = o1;
        }
        __syncthreads();
        wino_final_epilogue<256, COB, BST>(a, Rl, cur.m0, WT, Mt, THW, TW, cur.c0, fresh(threadIdx.x));
        if (!has_next) break;
        it = itn; cur = nx;
        set_consumer(cur);
        __syncthreads();          // R and the statistics scratch are free before chunk 0 of the next item commits into stage 1
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The split kernel with 64 output channels per workgroup (round 5).  conv_wino3s_kernel redoes the input transform and the three-piece split of the same V in every one of
// its Cout / 32 workgroups per tile block: 12-14.6 vector instructions per MFMA, matrix pipe 0.20 busy (profiles/r04_mfma_pmc.md).  Here a workgroup is 512 threads = 8 waves,
// wave = (Winograd row i, position pair jp): it forms V_(i, 2 jp) and V_(i, 2 jp + 1) -- three of the four row-combined patch columns, 12 instead of 32 ds_read_b128 per chunk --
// splits them once and contracts the pieces against the U fragments of BOTH 32-channel halves (24 MFMAs per wave and chunk for half the transform / split work, every LDS
// read and every loader instruction amortised over twice the products).  The accumulators stay at 64 registers per lane (2 positions x 2 channel halves), the U ring at 48,
// so the kernel sits near 200 registers where a four-wave form of the same tile (128 accumulators) does not close its budget.  One workgroup per CU
// (two waves per SIMD, as before).  The two position pairs of a row meet in the output transform: the jp = 0 waves store their column-half partial sums, the jp = 1 waves add
// theirs in place, then the common epilogue.  Same U layout, loader, LDS patch layout and epilogue contract as conv_wino3s_kernel.
// ---------------------------------------------------------------------------------------------------------------------
template <int XF, int BST, int NP>
__global__ __launch_bounds__(512, 2) void conv_wino3w_kernel(ConvArgs a, Wino2Geo geo) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int COB = 64, WT = 32, NLI = 3;
    constexpr int PL = 128 * 3 * 8;                  // floats per 8-channel plane (384 slots), two planes per stage
    constexpr int PST = 2 * PL;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wi = wv & 3, jp = wv >> 2;
    const int TW = geo.TW, TH = geo.TH, Mt = geo.Mt, ntp = geo.ntp;
    const int bid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_p = bid % ntp, tile_c = bid / ntp;
    const int m0 = tile_p * WT, c0 = tile_c * COB;
    const int THW = TH * TW;
    const int rr_first = m0 / TW, tx0 = m0 - rr_first * TW;
    const int L0 = min(TW - tx0, WT);
    const int S0 = 4 * (2 * L0 + 2), Lf = min(TW, WT), S = 4 * (2 * Lf + 2);
    const int n_rows_total = a.N * TH;

    // ---- loader: item f = tid + 512 li = (slot f >> 2, 16-B piece f & 3 of the 16 channels); piece >> 1 = plane, piece & 1 = half inside the plane's 8-float slot
    unsigned l_off[NLI]; unsigned l_ok = 0;
    const int piece = tid & 3, pl = piece >> 1, h2 = piece & 1;
    const int l_dst0 = pl * PL + (tid >> 2) * 8 + ((h2 ^ ((tid >> 5) & 1)) << 2);          // + li * 1024 floats (128 slots)
#pragma unroll
    for (int li = 0; li < NLI; ++li) {
        const int slot = (tid + 512 * li) >> 2;
        int k, r, WP, txa, Lk;
        if (slot < S0) { k = 0; r = slot; WP = 2 * L0 + 2; txa = tx0; Lk = L0; }
        else { const int s2 = slot - S0; k = 1 + fdiv24(s2, S); r = s2 - (k - 1) * S; txa = 0; Lk = max(min(TW, WT - L0 - (k - 1) * TW), 0); WP = 2 * Lk + 2; }
        const int pa = fdiv24(r, WP), pc = r - pa * WP;
        const int rr = rr_first + k;
        const bool act = slot < geo.nslots && pa < 4 && Lk > 0 && pc < 2 * Lk + 2 && rr < n_rows_total;
        const int rrc = act ? rr : 0;
        const int n = fdiv24(rrc, TH), ty = rrc - n * TH;
        int ih = 2 * ty - a.pad + pa, iw = 2 * txa - a.pad + pc;
        bool ok = act;
        if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
        else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        l_ok |= ok ? (1u << li) : 0u;
        l_off[li] = ok ? ((unsigned)((long)n * a.x_bs) + (unsigned)((ih * a.W + iw) * a.x_ps) + (unsigned)(piece * 4)) * 4u : (XF != 0 ? (unsigned)(piece * 16) : W_OOB);
    }
    const char* xb = reinterpret_cast<const char*>(a.x);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, geo.xbytes, 0x00020000);
    uint4 ra[NLI];
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto gload = [&](int c) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            if constexpr (XF != 0) ra[li] = *reinterpret_cast<const uint4*>(xb + (l_off[li] + (unsigned)(c * 64)));
            else ra[li] = wino_buffer_load16(xrs, l_off[li], c * 64);
        }
        if constexpr (XF != 0) { xsc = *reinterpret_cast<const float4*>(a.in_scale + c * 16 + piece * 4); xsh = *reinterpret_cast<const float4*>(a.in_shift + c * 16 + piece * 4); }
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int li = 0; li < NLI; ++li) {
            const unsigned mk = 0u - ((l_ok >> li) & 1u);
            uint4 v = ra[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            if constexpr (XF != 0) {
                v.x = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), xsc.x, xsh.x), 0.f)); v.y = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), xsc.y, xsh.y), 0.f));
                v.z = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), xsc.z, xsh.z), 0.f)); v.w = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), xsc.w, xsh.w), 0.f));
            }
            if constexpr (XF != 0) { v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk; }
            *reinterpret_cast<uint4*>(wsm + so + l_dst0 + li * 1024) = v;
        }
    };

    // ---- consumer: lane = (tile, k-group g = plane); B^T row i combines patch rows (r1, r2).  The wave's two positions need three of the four row-combined columns:
    //      jp = 0: V_0 = c0 - c2, V_1 = c1 + c2;   jp = 1: V_2 = c2 - c1, V_3 = c1 - c3.   With (P, Q, R) = (c0, c2, c1) / (c2, c1, c3): V_a = P - Q, V_b = Q + s R, s = +1 / -1.
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const float sR = jp ? -1.f : 1.f;
    const int tl = lane & 31, g = lane >> 5;
    int c_off1[3], c_off2[3];          // piece 0 of the slot; piece 1 is at (offset ^ 4)
    {
        const int u = tx0 + tl, k = fdiv24(u, TW);
        const int lc = (k == 0) ? tl : (u - k * TW);
        const int base = (k == 0) ? 0 : S0 + (k - 1) * S, WP = (k == 0) ? 2 * L0 + 2 : 2 * min(TW, WT - L0 - (k - 1) * TW) + 2;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int col = jp ? (b == 0 ? 2 : (b == 1 ? 1 : 3)) : (b == 0 ? 0 : (b == 1 ? 2 : 1));
            const int sl1 = base + r1 * WP + 2 * lc + col, sl2 = base + r2 * WP + 2 * lc + col;
            c_off1[b] = g * PL + sl1 * 8 + (((sl1 >> 3) & 1) << 2);
            c_off2[b] = g * PL + sl2 * 8 + (((sl2 >> 3) & 1) << 2);
        }
    }

    f32x16 acc[2][2];          // [position of the pair][32-channel half]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jj][cb][e] = 0.f;

    // U stream of this wave: channel half cb, chunk c16, position j = 2 jp + jj, piece q at ub[cb] + ((c16 * 4 + j) * NP + q) * 512 bf16 (64 lanes x 8)
    const int nchunks = a.Cin >> 4;
    const __bf16* ub[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) ub[cb] = reinterpret_cast<const __bf16*>(a.wino_u) + ((long)((tile_c * 2 + cb) * 4 + wi) * nchunks) * 4 * NP * 512 + (2 * jp) * NP * 512 + lane * 8;
    wbf16x8 u[2][2][NP];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < NP; ++q) u[jj][cb][q] = *reinterpret_cast<const wbf16x8*>(ub[cb] + (jj * NP + q) * 512);
#ifdef MD2_ABLATE
    if (a.ablate & 128) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < NP; ++q) u[jj][cb][q] = __builtin_bit_cast(wbf16x8, (wu32x4){(unsigned)lane, 1u, 2u, 3u});
    }
#endif

#ifdef MD2_ABLATE
    const int abl = a.ablate;          // timing experiments (tools/conv_bench): 1 no patch loads / commits, 2 no U refills, 4 no MFMAs, 8 no barrier, 32 no LDS reads / transform / split
#else
    constexpr int abl = 0;
#endif
    if (!(abl & 128)) { gload(0); lstore(0); }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int so = (c & 1) * PST;
        const int cn = min(c + 1, nchunks - 1);
        const long uno = (long)cn * (4 * NP * 512);
        if (!(abl & 1)) gload(cn);
        __builtin_amdgcn_sched_barrier(0);
        // B^T row combination of the lane's 8 channels (two 16-B pieces) for the three columns P, Q, R
        float4 rr[3][2];
#ifdef MD2_ABLATE
        if (abl & 32) { for (int b = 0; b < 3; ++b) rr[b][0] = rr[b][1] = make_float4(sg, sg, sg, sg); } else
#endif
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float4 d1 = *reinterpret_cast<const float4*>(wsm + so + (c_off1[b] ^ (p << 2)));
                const float4 d2 = *reinterpret_cast<const float4*>(wsm + so + (c_off2[b] ^ (p << 2)));
                rr[b][p] = make_float4(fmaf(sg, d2.x, d1.x), fmaf(sg, d2.y, d1.y), fmaf(sg, d2.z, d1.z), fmaf(sg, d2.w, d1.w));
            }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            unsigned vp[NP][4];
#ifdef MD2_ABLATE
            if (abl & 32) {
#pragma unroll
                for (int q = 0; q < NP; ++q) { vp[q][0] = vp[q][1] = vp[q][2] = vp[q][3] = 0x3f803f80u + (unsigned)jj; }
            } else
#endif
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float4 v;
                if (jj == 0) v = make_float4(rr[0][p].x - rr[1][p].x, rr[0][p].y - rr[1][p].y, rr[0][p].z - rr[1][p].z, rr[0][p].w - rr[1][p].w);
                else v = make_float4(fmaf(sR, rr[2][p].x, rr[1][p].x), fmaf(sR, rr[2][p].y, rr[1][p].y), fmaf(sR, rr[2][p].z, rr[1][p].z), fmaf(sR, rr[2][p].w, rr[1][p].w));
                if constexpr (NP == 2) { split_pk(v.x, v.y, vp[0][2 * p], vp[1][2 * p]); split_pk(v.z, v.w, vp[0][2 * p + 1], vp[1][2 * p + 1]); }
                else { split3_pk(v.x, v.y, vp[0][2 * p], vp[1][2 * p], vp[2][2 * p]); split3_pk(v.z, v.w, vp[0][2 * p + 1], vp[1][2 * p + 1], vp[2][2 * p + 1]); }
            }
            wbf16x8 vb[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) vb[q] = __builtin_bit_cast(wbf16x8, (wu32x4){vp[q][0], vp[q][1], vp[q][2], vp[q][3]});
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                // small terms first
                if (!(abl & 4)) {
                if constexpr (NP == 3) {
                    acc[jj][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[jj][cb][2], vb[0], acc[jj][cb], 0, 0, 0);
                    acc[jj][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[jj][cb][0], vb[2], acc[jj][cb], 0, 0, 0);
                    acc[jj][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[jj][cb][1], vb[1], acc[jj][cb], 0, 0, 0);
                }
                acc[jj][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[jj][cb][1], vb[0], acc[jj][cb], 0, 0, 0);
                acc[jj][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[jj][cb][0], vb[1], acc[jj][cb], 0, 0, 0);
                acc[jj][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[jj][cb][0], vb[0], acc[jj][cb], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < NP; ++q) asm volatile("" :: "v"(u[jj][cb][q]), "v"(vb[q]));
                }
                // refill in place right behind the fragment set's last use, as in conv_wino3_kernel
                if (!(abl & 2)) {
#pragma unroll
                for (int q = 0; q < NP; ++q) u[jj][cb][q] = *reinterpret_cast<const wbf16x8*>(ub[cb] + uno + (jj * NP + q) * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(abl & 1)) lstore(PST - so);
        if (!(abl & 8)) __syncthreads();
    }

#ifdef MD2_ABLATE
    if (abl & 64) { if (acc[0][0][0] + acc[1][1][3] == 123.456f) a.y[0] = 1.f; return; }          // timing experiment: no output transform / epilogue
    if (abl & 256) {          // timing experiment: plain stores of the accumulators instead of the output transform + epilogue (wrong values, the same bytes)
        float* yp = a.y + ((long)bid * 512 + tid) * 16;          // (32 tiles x 4 pixels x 64 channels = 512 threads x 16 floats: the workgroup's own output bytes)
        f32x16 t = acc[0][0];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] += acc[0][1][e] + acc[1][0][e] + acc[1][1][e];
        if ((long)(bid + 1) * 512 * 16 <= (long)a.N * a.Ho * a.Wo * a.Cout) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<float4*>(yp + gq * 4) = make_float4(t[4 * gq], t[4 * gq + 1], t[4 * gq + 2], t[4 * gq + 3]);
        }
        return;
    }
#endif
    // ---- output transform: column half in registers -- o0 = M0 + M1 + M2, o1 = M1 - M2 - M3, the jp = 0 waves hold M0, M1 and store (M0 + M1, M1), the jp = 1 waves hold
    //      M2, M3 and add (M2, -M2 - M3) in place -- row half through LDS (R[i][b][32 tiles][COB + 4]), as conv_wino3_kernel
    constexpr int RLD = COB + 4;
    float* Rl = wsm;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (jp == pass) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float4 o0, o1;
                    const f32x16& A0 = acc[0][cb]; const f32x16& A1 = acc[1][cb];
                    if (pass == 0) {
                        o0 = make_float4(A0[4 * gq + 0] + A1[4 * gq + 0], A0[4 * gq + 1] + A1[4 * gq + 1], A0[4 * gq + 2] + A1[4 * gq + 2], A0[4 * gq + 3] + A1[4 * gq + 3]);
                        o1 = make_float4(A1[4 * gq + 0], A1[4 * gq + 1], A1[4 * gq + 2], A1[4 * gq + 3]);
                    } else {
                        o0 = make_float4(A0[4 * gq + 0], A0[4 * gq + 1], A0[4 * gq + 2], A0[4 * gq + 3]);
                        o1 = make_float4(-A0[4 * gq + 0] - A1[4 * gq + 0], -A0[4 * gq + 1] - A1[4 * gq + 1], -A0[4 * gq + 2] - A1[4 * gq + 2], -A0[4 * gq + 3] - A1[4 * gq + 3]);
                    }
                    const int col = 32 * cb + 8 * gq + 4 * g;
                    float4* p0 = reinterpret_cast<float4*>(Rl + ((wi * 2 + 0) * WT + tl) * RLD + col);
                    float4* p1 = reinterpret_cast<float4*>(Rl + ((wi * 2 + 1) * WT + tl) * RLD + col);
                    if (pass == 1) { const float4 q0 = *p0, q1 = *p1; o0.x += q0.x; o0.y += q0.y; o0.z += q0.z; o0.w += q0.w; o1.x += q1.x; o1.y += q1.y; o1.z += q1.z; o1.w += q1.w; }
                    *p0 = o0; *p1 = o1;
                }
        }
        __syncthreads();
    }
    wino_final_epilogue<512, COB, BST>(a, wsm, m0, WT, Mt, THW, TW, c0, tid);
}

// patch slots a block can need at tile-row width TW: worst case over the position of its first tile
static int wino2_slots(int TW, int T = W2_T) {
    const int Lf = std::min(TW, T), S = 4 * (2 * Lf + 2);
    int worst = 0;
    for (int tx0 = 0; tx0 < TW; ++tx0) {
        const int L0 = std::min(TW - tx0, T);
        const int rest = T - L0, nfull = rest / TW, Lr = rest - nfull * TW;          // whole rows, then a short last segment with rows as wide as it needs
        worst = std::max(worst, 4 * (2 * L0 + 2) + nfull * S + (Lr ? 4 * (2 * Lr + 2) : 0));
    }
    return (worst + 15) / 16 * 16;
}
// third kernel: Cout % 32 == 0, Cin % 8 == 0; 64 tiles per workgroup with up to 640 patch slots per stage (tile rows of at least 4 tiles), or 32 tiles with up to 384
static int g_wino_nh = md2_tune("MD2_WINO_NH", 0);          // 0: by workgroup count, 1 / 2: pin the 32- / 64-tile workgroup (tools/conv_bench)
static bool wino3_eligible(const ConvArgs& a, Wino2Geo& g, int& nh) {
    static const bool off = md2_tune_set("MD2_NO_WINO3");
    if (off || g_wino_on == 2 || a.Cout % 32 != 0 || a.Cin % 8 != 0) return false;
    g.TH = (a.Ho + 1) / 2; g.TW = (a.Wo + 1) / 2; g.Mt = a.N * g.TH * g.TW;
    const int s64 = wino2_slots(g.TW, 64), s32 = wino2_slots(g.TW, 32);
    const long b64 = (long)((g.Mt + 63) / 64) * (a.Cout / 32), b32 = (long)((g.Mt + 31) / 32) * (a.Cout / 32);
    const bool ok64 = s64 <= 128 * 5, ok32 = s32 <= 128 * 3;
    if (!ok64 && !ok32) return false;
    // 32-tile workgroups by default: measured on every 3x3 / stride-1 shape of the step at batch 32 (tools/run_wino_nh_bench.sh) they are 4-18 % faster on the encoder stages
    // (0.159 -> 0.140 ms at 256 channels, 0.185 -> 0.152 at 512), 5-12 % on most decoder layers and within +-4 % on the rest; MD2_WINO_SMALL_BLOCKS=n keeps the 64-tile
    // workgroup for launches with at least n of them
    static const long small = md2_tune("MD2_WINO_SMALL_BLOCKS", (1L << 40));
    nh = g_wino_nh == 1 ? 1 : (g_wino_nh == 2 ? 2 : (b64 < small ? 1 : 2));
    if (a.wino_pieces) { if (!ok32 || a.Cin % 16 != 0) return false; nh = 1; }          // split mode: conv_wino3s_kernel (32 tiles, 16-channel chunks) or the direct convolution
    if (nh == 1 && !ok32) nh = 2;
    if (nh == 2 && !ok64) nh = 1;
    const int T = 32 * nh;
    g.ntp = (g.Mt + T - 1) / T; g.nslots = nh == 1 ? s32 : s64;
    g.xbytes = (unsigned)std::min<long>((a.bf16 == 2 ? 2 : 4) * ((long)a.N * a.x_bs + (long)a.H * a.W * a.x_ps), (long)W_OOB - 16);          // (launch_conv_wino checks the span is below 2^32)
    // a launch below the threshold keeps the direct kernel and its split-K -- except short reductions (Cin <= 128), where a workgroup's life is mostly fixed cost either way and
    // 0.6 x the threshold already wins (batch 4, encoder layer 2: 156 workgroups, 0.051 -> 0.040 ms in the model)
    return g_wino_on == 3 || b64 >= g_wino_min_blocks || (a.Cin <= 128 && b64 * 5 >= (long)g_wino_min_blocks * 3);
}
template <int NLI, int NH, int CB = 1>
static void launch_wino3_t(const ConvArgs& a, const Wino2Geo& g, hipStream_t s) {
    constexpr size_t lds_pipe = (size_t)2 * 128 * NLI * 8 * sizeof(float), lds_epi = (size_t)8 * (32 * NH) * (32 * CB + 4) * sizeof(float), lds_st = (size_t)12 * 256 * sizeof(float);
    constexpr size_t lds = std::max(std::max(lds_pipe, lds_epi), lds_st);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3_kernel<NLI, 0, 0, NH, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3_kernel<NLI, 1, 0, NH, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3_kernel<NLI, 0, 1, NH, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = g.ntp * (a.Cout / (32 * CB));
    if (a.in_scale) hipLaunchKernelGGL((conv_wino3_kernel<NLI, 1, 0, NH, CB>), dim3(grid), dim3(256), lds, s, a, g);
    else if (a.bst) hipLaunchKernelGGL((conv_wino3_kernel<NLI, 0, 1, NH, CB>), dim3(grid), dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((conv_wino3_kernel<NLI, 0, 0, NH, CB>), dim3(grid), dim3(256), lds, s, a, g);
}
template <int NP>
static void launch_wino3s_t(const ConvArgs& a, const Wino2Geo& g, hipStream_t s) {
    // two stages of two 8-channel planes (48 KB) > the epilogue's 36 KB; MD2_WINO_LDS_PAD (tuning builds only): extra bytes, to time the kernel at ONE workgroup per CU (tools/run_wino_occupancy_probe.sh)
    static const size_t lds = (size_t)2 * 2 * 128 * 3 * 8 * sizeof(float) + (size_t)md2_tune("MD2_WINO_LDS_PAD", 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3s_kernel<0, 0, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3s_kernel<1, 0, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3s_kernel<0, 1, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = g.ntp * (a.Cout / 32);
    if (a.in_scale) hipLaunchKernelGGL((conv_wino3s_kernel<1, 0, NP>), dim3(grid), dim3(256), lds, s, a, g);
    else if (a.bst) hipLaunchKernelGGL((conv_wino3s_kernel<0, 1, NP>), dim3(grid), dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((conv_wino3s_kernel<0, 0, NP>), dim3(grid), dim3(256), lds, s, a, g);
}
template <int NP>
static void launch_wino3w_t(const ConvArgs& a, const Wino2Geo& g, hipStream_t s) {
    constexpr size_t lds = (size_t)8 * 32 * (64 + 4) * sizeof(float);          // the output transform's R[i][b][32 tiles][64 + 4] (68 KB) > the two pipeline stages (48 KB)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3w_kernel<0, 0, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3w_kernel<1, 0, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3w_kernel<0, 1, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = g.ntp * (a.Cout / 64);
    if (a.in_scale) hipLaunchKernelGGL((conv_wino3w_kernel<1, 0, NP>), dim3(grid), dim3(512), lds, s, a, g);
    else if (a.bst) hipLaunchKernelGGL((conv_wino3w_kernel<0, 1, NP>), dim3(grid), dim3(512), lds, s, a, g);
    else hipLaunchKernelGGL((conv_wino3w_kernel<0, 0, NP>), dim3(grid), dim3(512), lds, s, a, g);
}
#ifndef MD2_WINO_CB2_DEFAULT
#define MD2_WINO_CB2_DEFAULT 1
#endif
int g_wino_cb2 = MD2_WINO_CB2_DEFAULT;           // fp32-MFMA Winograd launches with Cout % 64 == 0: 32 tiles x 64 channels per workgroup (conv_wino3_kernel<., ., ., 1, 2>); 0: 32 x 32 (tools/conv_bench: CB1)
int g_wino_wide = 0;          // 1 (md2_set_winograd(on + 8, .)): split launches with Cout % 64 == 0 take the 64-channel workgroup (conv_wino3w_kernel: faster alone, slower in the step -- DESIGN.md section 4); 0: the 32-channel one everywhere
// persistent form (opt-in): launches with more items than the 512 resident workgroups, an even chunk count (every layer of the networks here) and no input transform
int g_wino_bf16 = 0;             // md2_set_winograd(on + 32, .) sets it: bf16 tensors (precision 2) take conv_wino3b_kernel on their 3x3 / stride-1 layers.  Measured: 1.7 - 4 % of a step, paid with V's and U's roundings -- the gradient of the 1024x320 step moves from cosine 0.83 to 0.79 against the fp32 oracle -- so opt-in
int g_wino_persist = 0;          // md2_set_winograd(on + 16, .) sets it: measured equal to the one-item workgroups launch by launch and 1 % behind them in the step, so opt-in
template <int NP>
static void launch_wino3p_t(const ConvArgs& a, const Wino2Geo& g, int nitems, hipStream_t s) {
    constexpr size_t lds = (size_t)(2 * 2 * 128 * 3 * 8 + 128 * 3 * 8 + 384) * sizeof(float);          // two stages, the output transform's overhang, the geometry table: 62 976 B
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3p_kernel<0, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3p_kernel<1, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = std::min(nitems, 512);
    if (a.bst) hipLaunchKernelGGL((conv_wino3p_kernel<1, NP>), dim3(grid), dim3(256), lds, s, a, g, nitems);
    else hipLaunchKernelGGL((conv_wino3p_kernel<0, NP>), dim3(grid), dim3(256), lds, s, a, g, nitems);
}
static void launch_wino3s(const ConvArgs& a, const Wino2Geo& g, hipStream_t s) {
    const long nitems = (long)g.ntp * (a.Cout / 32);
    const int nchunks = a.Cin >> 4;
    if (g_wino_wide && a.Cout % 64 == 0) { if (a.wino_pieces == 3) launch_wino3w_t<3>(a, g, s); else launch_wino3w_t<2>(a, g, s); }
    else if (g_wino_persist && !a.in_scale && nitems > 512 && nitems < (1L << 24) && nchunks >= 4 && (nchunks & 1) == 0) { if (a.wino_pieces == 3) launch_wino3p_t<3>(a, g, (int)nitems, s); else launch_wino3p_t<2>(a, g, (int)nitems, s); }
    else if (a.wino_pieces == 3) launch_wino3s_t<3>(a, g, s); else launch_wino3s_t<2>(a, g, s);
}
static void launch_wino3b(const ConvArgs& a, const Wino2Geo& g, hipStream_t s) {          // bf16 tensors, U as one bf16 piece (wino_pieces == 4)
    constexpr size_t lds = (size_t)2 * 2 * 128 * 3 * 8 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3b_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3b_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = g.ntp * (a.Cout / 32);
    if (a.bst) hipLaunchKernelGGL((conv_wino3b_kernel<1>), dim3(grid), dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((conv_wino3b_kernel<0>), dim3(grid), dim3(256), lds, s, a, g);
}
static void launch_wino3(const ConvArgs& a, const Wino2Geo& g, int nh, hipStream_t s) {
    if (a.bf16 == 2) launch_wino3b(a, g, s);
    else if (a.wino_pieces) launch_wino3s(a, g, s);
    // (two channel blocks per workgroup where that still leaves >= 2 rounds of the 512 resident workgroups: 0.155 -> 0.142 ms on the 64 -> 64 layers at batch 32, 0.167 -> 0.150 with the
    //  BatchNorm transform in the loader; launches with fewer workgroups lose more to the emptier last round than the shared transform saves -- tools/run_wino_cb_bench.sh)
    else if (nh == 1) { if (g_wino_cb2 && a.Cout % 64 == 0 && (long)g.ntp * (a.Cout / 64) >= 1024) launch_wino3_t<3, 1, 2>(a, g, s); else launch_wino3_t<3, 1>(a, g, s); }
    else launch_wino3_t<5, 2>(a, g, s);
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient in the Winograd domain (cuDNN's algorithm choice for Flux's conv filter pullback, src/depth_decoder.jl:13-14 / ResNet.jl):
//   dU_(i,j)[co][ci] = sum over tiles  dM_(i,j)[tile][co] * V_(i,j)[tile][ci],   dM = A dY A^T (2x2 output-gradient tile -> 4x4),  V = B^T d B (as in the forward kernel),
//   dw = G^T dU G  (4x4 -> the 3x3 taps).  16 multiplies per tile and channel pair instead of 36.
// GEMM-K = tiles.  Block = 256 threads (wave = Winograd row i), 64 output channels x 32 input channels x all 16 positions, a contiguous range of K batches; a batch = up to 16
// tiles of ONE tile row, so its patch is a single row segment (4 rows x (2 L + 2) pixels) and its geometry is scalar work.  Per k-step (two tiles, MFMA k = lane half):
//   A operand: lane (channel pair 2r, 2r+1 ; tile) reads the tile's 4 dy pixels as ds_read_b64, forms dM_(i,0..3) for both channels -> two 32-row MFMA tiles (even / odd
//              channels), B operand: lane (channel r ; tile) reads 2 patch rows x 4 columns as ds_read_b32 and forms V_(i,0..3); 8 MFMAs per k-step and wave.
// The block's 16 x 64 x 32 sums leave through LDS (G^T . G: column half in registers, row half across the four waves) as one fp32 atomic per weight; the block column with
// input-channel tile 0 also produces the bias gradient (sum of dy).  Two workgroups per CU (68 KB of LDS, 128 accumulator registers).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WG_TB = 16, WG_WP = 2 * WG_TB + 2, WG_XSLOTS = 160;          // tiles per batch, patch row stride (slots), slots allocated (>= 4 * WG_WP = 136: every loader item has a real slot)
constexpr int WG_XF = WG_XSLOTS * 32, WG_DYF = WG_TB * 4 * 64, WG_STAGE = WG_XF + WG_DYF;      // floats per stage: x [slot][32 ci] | dy [tile][4 px][64 co]
// BPR batches per tile row, nb batches in all, bpb batches per block.  Tile rows of at most 8 tiles (the 1/32-scale maps: 7 tiles at 416 wide) would fill a 16-tile batch
// -- the MFMA's K -- to less than half: there a batch holds RB consecutive tile rows (flattened over (n, ty)) as RB row segments of WPS = 2 TW + 2 patch columns each
// (RB = 1, WPS = WG_WP: one segment, the layout of the wide maps); nrows = N * TH.
struct WinoWgGeo { int TH, TW, BPR, nb, bpb, nco, nci, RB, WPS, nrows; unsigned xbytes, dybytes; };          // (byte spans of the two views: buffer-descriptor ranges, see W_OOB)

template <int XF, int MR>          // MR = 1: batches of several short tile rows (WinoWgGeo::RB > 1); MR = 0 folds RB = 1 / WPS = WG_WP at compile time: the single-segment code of the wide maps
__global__ __launch_bounds__(256, 2) void conv_wino_wgrad_kernel(WgradArgs a, WinoWgGeo geo) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;
    const int bid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_co = bid % geo.nco, tile_ci = (bid / geo.nco) % geo.nci, ks = bid / (geo.nco * geo.nci);
    const int co0 = tile_co * 64, ci0 = tile_ci * 32;
    const int b_lo = ks * geo.bpb, b_hi = min(b_lo + geo.bpb, geo.nb);
    const int TW = geo.TW, TH = geo.TH;

    // ---- loader items (fixed per thread; the pixel they address changes with the batch)
    //   x : item f = tid + 256 li (li < 5) = (slot f >> 3 = row pa * 34 + column pc, 16-B piece f & 7 of the 32 channels)
    //   dy: item f = tid + 256 li (li < 4) = (pixel slot f >> 4 = tile * 4 + (oa * 2 + ob), 16-B piece f & 15 of the 64 channels)
    const int RB = MR ? geo.RB : 1, WPS = MR ? geo.WPS : WG_WP;          // row segments per batch, patch columns (slots) per segment row
    int x_pa[5], x_pc[5], x_sg[5];
#pragma unroll
    for (int li = 0; li < 5; ++li) { const int slot = (tid + 256 * li) >> 3; x_sg[li] = slot / (4 * WPS); const int r = slot - x_sg[li] * 4 * WPS; x_pa[li] = r / WPS; x_pc[li] = r - x_pa[li] * WPS; }
    // dy items: tile t of the batch = segment t / TW (several tile rows per batch), tile column t % TW
    int d_sg[4], d_tx[4];
#pragma unroll
    for (int li = 0; li < 4; ++li) { const int t = (tid + 256 * li) >> 6; d_sg[li] = RB > 1 ? t / TW : 0; d_tx[li] = t - d_sg[li] * TW; }
    const int xq = tid & 7, dq = tid & 15;
    // single-segment batches on zero-padded layers (every encoder layer; round 5): a batch's geometry is (image n, tile row ty, first tile tx0) -- block-uniform scalars that
    // advance incrementally from batch to batch -- plus per-thread constants, so an item's address is one add and its validity three compares instead of ~25 vector
    // instructions per item and two scalar integer divisions per batch
    const bool fastgeo = !MR && a.padmode != PAD_REFLECT;
    int x_c[5], x_dpa[5], x_dpc[5], d_c[4];
#pragma unroll
    for (int li = 0; li < 5; ++li) { x_dpa[li] = x_pa[li] - a.pad; x_dpc[li] = x_pc[li] - a.pad; x_c[li] = (x_dpa[li] * a.W + x_dpc[li]) * a.x_ps + ci0 + xq * 4; if (x_sg[li] != 0 || x_pa[li] >= 4) x_dpa[li] = -(1 << 20); }
#pragma unroll
    for (int li = 0; li < 4; ++li) { const int ps = (tid + 256 * li) >> 4; d_c[li] = (((ps >> 1) & 1) * a.Wo + 2 * (ps >> 2) + (ps & 1)) * a.dy_ps + co0 + dq * 4; }
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (XF != 0) { xsc = *reinterpret_cast<const float4*>(a.in_scale + ci0 + xq * 4); xsh = *reinterpret_cast<const float4*>(a.in_shift + ci0 + xq * 4); }
    uint4 rx[5], rd[4]; unsigned okx = 0, okd = 0;
    const char* xb = reinterpret_cast<const char*>(a.x);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, geo.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, geo.dybytes, 0x00020000);
    auto gload_fast = [&](int n_u, int ty_u, int bq) {              // (fastgeo) batch (image, tile row, 16-tile group of the row) -> registers (+ validity bits)
        const int tx0 = bq * WG_TB, L = min(WG_TB, TW - tx0);
        okx = 0; okd = 0;
        const int ih0 = 2 * ty_u, iw0 = 2 * tx0;
        const unsigned ubx = (unsigned)((long)n_u * a.x_bs) + (unsigned)((ih0 * a.W + iw0) * a.x_ps), ubd = (unsigned)((long)n_u * a.dy_bs) + (unsigned)((ih0 * a.Wo + iw0) * a.dy_ps);
#pragma unroll
        for (int li = 0; li < 5; ++li) {
            const bool ok = (unsigned)(ih0 + x_dpa[li]) < (unsigned)a.H && (unsigned)(iw0 + x_dpc[li]) < (unsigned)a.W && x_pc[li] < 2 * L + 2;
            const unsigned off = ok ? (ubx + (unsigned)x_c[li]) * 4u : (XF != 0 ? (unsigned)(xq * 16) : W_OOB);
            if constexpr (XF != 0) rx[li] = *reinterpret_cast<const uint4*>(xb + off); else rx[li] = wino_buffer_load16(xrs, off, 0);
            okx |= ok ? (1u << li) : 0u;
        }
#pragma unroll
        for (int li = 0; li < 4; ++li) {
            const int ps = (tid + 256 * li) >> 4;
            const bool ok = (ps >> 2) < L && ih0 + ((ps >> 1) & 1) < a.Ho && iw0 + 2 * (ps >> 2) + (ps & 1) < a.Wo;
            rd[li] = wino_buffer_load16(drs, ok ? (ubd + (unsigned)d_c[li]) * 4u : W_OOB, 0);
            okd |= ok ? (1u << li) : 0u;
        }
    };
    auto gload = [&](int b) {              // batch b -> registers (+ validity bits); b is block-uniform
        const int rr0 = RB > 1 ? b * RB : b / geo.BPR, bq = RB > 1 ? 0 : b - rr0 * geo.BPR;
        const int tx0 = bq * WG_TB, L = min(WG_TB, TW - tx0);
        okx = 0; okd = 0;
        const int n_u = rr0 / TH, ty_u = rr0 - n_u * TH;          // single-segment batches: the image and tile row are block-uniform (scalar arithmetic)
#pragma unroll
        for (int li = 0; li < 5; ++li) {
            int n = n_u, ty = ty_u; bool ok = x_sg[li] == 0;
            if (RB > 1) { const int rr = rr0 + x_sg[li], rrc = min(rr, geo.nrows - 1); n = fdiv24(rrc, TH); ty = rrc - n * TH; ok = x_sg[li] < RB && rr < geo.nrows; }
            int ih = 2 * ty - a.pad + x_pa[li], iw = 2 * tx0 - a.pad + x_pc[li];
            ok = ok && x_pa[li] < 4 && x_pc[li] < 2 * L + 2;
            if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
            else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            const unsigned off = ok ? ((unsigned)((long)n * a.x_bs) + (unsigned)((ih * a.W + iw) * a.x_ps) + (unsigned)(ci0 + xq * 4)) * 4u : (XF != 0 ? (unsigned)(xq * 16) : W_OOB);
            if constexpr (XF != 0) rx[li] = *reinterpret_cast<const uint4*>(xb + off); else rx[li] = wino_buffer_load16(xrs, off, 0);
            okx |= ok ? (1u << li) : 0u;
        }
#pragma unroll
        for (int li = 0; li < 4; ++li) {
            const int ps = (tid + 256 * li) >> 4;
            int n = n_u, ty = ty_u; bool ok = true;
            if (RB > 1) { const int rr = rr0 + d_sg[li], rrc = min(rr, geo.nrows - 1); n = fdiv24(rrc, TH); ty = rrc - n * TH; ok = d_sg[li] < RB && rr < geo.nrows; }
            const int oy = 2 * ty + ((ps >> 1) & 1), ox = 2 * (tx0 + d_tx[li]) + (ps & 1);
            ok = ok && d_tx[li] < L && oy < a.Ho && ox < a.Wo;
            const unsigned off = ok ? ((unsigned)((long)n * a.dy_bs) + (unsigned)((oy * a.Wo + ox) * a.dy_ps) + (unsigned)(co0 + dq * 4)) * 4u : W_OOB;
            rd[li] = wino_buffer_load16(drs, off, 0);
            okd |= ok ? (1u << li) : 0u;
        }
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int li = 0; li < 5; ++li) {
            const unsigned mk = 0u - ((okx >> li) & 1u);
            uint4 v = rx[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            if constexpr (XF != 0) {
                v.x = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), xsc.x, xsh.x), 0.f)); v.y = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), xsc.y, xsh.y), 0.f));
                v.z = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), xsc.z, xsh.z), 0.f)); v.w = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), xsc.w, xsh.w), 0.f));
            }
            if constexpr (XF != 0) { v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk; }          // (XF == 0: zeros from the range-checked buffer load)
            *reinterpret_cast<uint4*>(wsm + so + ((tid + 256 * li) >> 3) * 32 + xq * 4) = v;
        }
#pragma unroll
        for (int li = 0; li < 4; ++li) {
            uint4 v = rd[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            *reinterpret_cast<uint4*>(wsm + so + WG_XF + ((tid + 256 * li) >> 4) * 64 + dq * 4) = v;
        }
    };

    // ---- consumer constants: wave i.  A (dM) row combine: r_b = ca * dY[0][b] + cb * dY[1][b]; B (V) row combine: d[r1] + sg * d[r2]
    const float ca = (wi == 3) ? 0.f : 1.f, cb = (wi == 0) ? 0.f : (wi == 1 ? 1.f : -1.f);
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const int lr = lane & 31, lh = lane >> 5;
    const int xo1 = r1 * WPS * 32 + lr, xo2 = r2 * WPS * 32 + lr;          // + the tile's slot offset (segment * 4 WPS + 2 tile column) * 32 + column b * 32
    const int dyo = WG_XF + (lh * 4) * 64 + 2 * lr;                                                  // + k-step * 8 pixel rows + px * 64

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][m][e] = 0.f;
    float2 bsum = make_float2(0.f, 0.f);
    const bool do_bias = a.db && tile_ci == 0 && wi == 0;

    if (b_lo < b_hi) {
        // (fastgeo) geometry of the batch in flight: (image, tile row, 16-tile group), advanced by one batch per iteration; cq = the group of the batch being contracted
        int gq = 0, gty = 0, gn = 0, cq = 0;
        if (fastgeo) { const int rr0 = b_lo / geo.BPR; gq = b_lo - rr0 * geo.BPR; gn = rr0 / TH; gty = rr0 - gn * TH; cq = gq; gload_fast(gn, gty, gq); } else gload(b_lo);
        lstore(0);
        __syncthreads();
        for (int b = b_lo; b < b_hi; ++b) {
            const int so = ((b - b_lo) & 1) * WG_STAGE;
            const int bn = min(b + 1, b_hi - 1);
            if (fastgeo) {
                cq = gq;
                if (b + 1 < b_hi) { if (++gq == geo.BPR) { gq = 0; if (++gty == TH) { gty = 0; ++gn; } } }          // (past the end: the last batch again, into the stage nobody reads)
                gload_fast(gn, gty, gq);
            } else gload(bn);
            __builtin_amdgcn_sched_barrier(0);
            const int bq = RB > 1 ? 0 : (fastgeo ? cq : b % geo.BPR), L = min(WG_TB, TW - bq * WG_TB);
            const int ntile = RB > 1 ? min(RB, geo.nrows - b * RB) * TW : L, nk = (ntile + 1) >> 1;
            auto kstep = [&](const int k) {
                const int t = 2 * k + lh, seg = RB > 1 ? (t >= TW ? fdiv24(t, TW) : 0) : 0;          // this lane's tile of the k-step (tiles past the batch were committed as zeros)
                const float* xs = wsm + so + (seg * 4 * WPS + 2 * (t - seg * TW)) * 32;
                const float* ds = wsm + so + dyo + k * (8 * 64);
                const float2 y00 = *reinterpret_cast<const float2*>(ds), y01 = *reinterpret_cast<const float2*>(ds + 64), y10 = *reinterpret_cast<const float2*>(ds + 128), y11 = *reinterpret_cast<const float2*>(ds + 192);
                float xv1[4], xv2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { xv1[c] = xs[xo1 + c * 32]; xv2[c] = xs[xo2 + c * 32]; }
                if (do_bias) { bsum.x += (y00.x + y01.x) + (y10.x + y11.x); bsum.y += (y00.y + y01.y) + (y10.y + y11.y); }          // (wave-uniform: one wave of one block column)
                // dM_(i, 0..3) of both channels
                const float p0x = ca * y00.x + cb * y10.x, p1x = ca * y01.x + cb * y11.x, p0y = ca * y00.y + cb * y10.y, p1y = ca * y01.y + cb * y11.y;
                const float mx[4] = {p0x, p0x + p1x, p0x - p1x, -p1x}, my[4] = {p0y, p0y + p1y, p0y - p1y, -p1y};
                // V_(i, 0..3)
                float rr[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) rr[c] = fmaf(sg, xv2[c], xv1[c]);
                const float V[4] = {rr[0] - rr[2], rr[1] + rr[2], rr[2] - rr[1], rr[1] - rr[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(mx[j], V[j], acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(my[j], V[j], acc[j][1], 0, 0, 0);
                }
            };
            for (int k = 0; k < nk; ++k) kstep(k);
            __builtin_amdgcn_sched_barrier(0);
            lstore(WG_STAGE - so);
            __syncthreads();
        }
    }

    // ---- dw = G^T dU G.  Column half in registers: T_q = sum_j dU_(i,j) G[j][q], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; row half across the waves through LDS
    //      (T[i][q][32 rows][33]), one pass per channel parity m; row r of pass m is output channel co0 + 2 r + m
    float* T = wsm;
    constexpr int TLD = 33;
    const int ci = tid & 31, rw = tid >> 5;                  // combine: thread -> input channel ci, rows rw + 8 k
    for (int m = 0; m < 2; ++m) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float a0 = acc[0][m][e], a1 = acc[1][m][e], a2 = acc[2][m][e], a3 = acc[3][m][e];
            const int row = 8 * (e >> 2) + 4 * lh + (e & 3);
            T[((wi * 3 + 0) * 32 + row) * TLD + lr] = a0 + 0.5f * (a1 + a2);
            T[((wi * 3 + 1) * 32 + row) * TLD + lr] = 0.5f * (a1 - a2);
            T[((wi * 3 + 2) * 32 + row) * TLD + lr] = 0.5f * (a1 + a2) + a3;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = rw + 8 * k;
            float* dst = a.dw + ((long)(co0 + 2 * row + m) * 9) * a.Cin + ci0 + ci;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float t0 = T[((0 * 3 + q) * 32 + row) * TLD + ci], t1 = T[((1 * 3 + q) * 32 + row) * TLD + ci], t2 = T[((2 * 3 + q) * 32 + row) * TLD + ci], t3 = T[((3 * 3 + q) * 32 + row) * TLD + ci];
                atomicAdd(dst + (0 * 3 + q) * a.Cin, t0 + 0.5f * (t1 + t2));
                atomicAdd(dst + (1 * 3 + q) * a.Cin, 0.5f * (t1 - t2));
                atomicAdd(dst + (2 * 3 + q) * a.Cin, 0.5f * (t1 + t2) + t3);
            }
        }
    }
    // bias gradient: wave 0 of the block column with input-channel tile 0 (every wave read the same dy)
    if (do_bias) {
        bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32);
        if (lh == 0) { atomicAdd(a.db + co0 + 2 * lr, bsum.x); atomicAdd(a.db + co0 + 2 * lr + 1, bsum.y); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same weight gradient with the products on the bf16 matrix pipe ("split": dM and V as bf16 pieces of the fp32 values, v_mfma_f32_32x32x16_bf16 with fp32 accumulation --
// see conv_wino3s_kernel for the arithmetic).  NP = 3 (hi + mid + lo, six MFMAs per product) carries every mantissa bit of both operands: fp32-MFMA accuracy, and it is what
// g_wgrad_pieces = 3 selects; NP = 2 (hi + lo, three MFMAs) rounds the operands to ~16 bits -- reduced precision, opt-in only (md2_set_conv_products(-1, 2)).
// The MFMA's K = 16 is a batch's 16 tiles: lane = (channel, tile group of 8).  Block = 32 output x 32 input channels (64 accumulator registers: the 64-channel block of the
// fp32 kernel spilled once the piece vectors joined its 128 accumulators), LDS stage = x [160 slots][32 ci] | dy [16 tiles][4 px][32 co].
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WS_DYF = WG_TB * 4 * 32, WS_STAGE = WG_XF + WS_DYF;
// NP = pieces per operand: 3 = hi + mid + lo, six MFMAs per product (lh, hl, mm, mh, hm, hh: the dropped terms are below 2^-24 of the product) -- every bit of both fp32
// operands enters, i.e. the accuracy of the fp32 MFMA kernel above; 2 = hi + lo, three MFMAs (~16-bit operand mantissas: reduced precision, opt-in)
template <int XF, int NP, int MR>
__global__ __launch_bounds__(256, 2) void conv_wino_wgrad_s_kernel(WgradArgs a, WinoWgGeo geo) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;
    const int bid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_co = bid % geo.nco, tile_ci = (bid / geo.nco) % geo.nci, ks = bid / (geo.nco * geo.nci);
    const int co0 = tile_co * 32, ci0 = tile_ci * 32;
    const int b_lo = ks * geo.bpb, b_hi = min(b_lo + geo.bpb, geo.nb);
    const int TW = geo.TW, TH = geo.TH;

    // ---- loader items: x as in conv_wino_wgrad_kernel (5 per thread); dy: item f = tid + 256 li (li < 2) = (pixel slot f >> 3 = tile * 4 + (oa * 2 + ob), 16-B piece f & 7 of 32 channels)
    const int RB = MR ? geo.RB : 1, WPS = MR ? geo.WPS : WG_WP;          // row segments per batch, patch columns (slots) per segment row (WinoWgGeo; MR as in conv_wino_wgrad_kernel)
    int x_pa[5], x_pc[5], x_sg[5];
#pragma unroll
    for (int li = 0; li < 5; ++li) { const int slot = (tid + 256 * li) >> 3; x_sg[li] = slot / (4 * WPS); const int r = slot - x_sg[li] * 4 * WPS; x_pa[li] = r / WPS; x_pc[li] = r - x_pa[li] * WPS; }
    int d_sg[2], d_tx[2];
#pragma unroll
    for (int li = 0; li < 2; ++li) { const int t = (tid + 256 * li) >> 5; d_sg[li] = RB > 1 ? t / TW : 0; d_tx[li] = t - d_sg[li] * TW; }
    const int xq = tid & 7;
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (XF != 0) { xsc = *reinterpret_cast<const float4*>(a.in_scale + ci0 + xq * 4); xsh = *reinterpret_cast<const float4*>(a.in_shift + ci0 + xq * 4); }
    uint4 rx[5], rd[2]; unsigned okx = 0, okd = 0;
    const char* xb = reinterpret_cast<const char*>(a.x);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, geo.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, geo.dybytes, 0x00020000);
    auto gload = [&](int b) {
        const int rr0 = RB > 1 ? b * RB : b / geo.BPR, bq = RB > 1 ? 0 : b - rr0 * geo.BPR;
        const int tx0 = bq * WG_TB, L = min(WG_TB, TW - tx0);
        okx = 0; okd = 0;
        const int n_u = rr0 / TH, ty_u = rr0 - n_u * TH;
#pragma unroll
        for (int li = 0; li < 5; ++li) {
            int n = n_u, ty = ty_u; bool ok = x_sg[li] == 0;
            if (RB > 1) { const int rr = rr0 + x_sg[li], rrc = min(rr, geo.nrows - 1); n = fdiv24(rrc, TH); ty = rrc - n * TH; ok = x_sg[li] < RB && rr < geo.nrows; }
            int ih = 2 * ty - a.pad + x_pa[li], iw = 2 * tx0 - a.pad + x_pc[li];
            ok = ok && x_pa[li] < 4 && x_pc[li] < 2 * L + 2;
            if (a.padmode == PAD_REFLECT) { ih = refl_idx(max(min(ih, a.H), -1), a.H); iw = refl_idx(max(min(iw, a.W), -1), a.W); }
            else ok = ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            const unsigned off = ok ? ((unsigned)((long)n * a.x_bs) + (unsigned)((ih * a.W + iw) * a.x_ps) + (unsigned)(ci0 + xq * 4)) * 4u : (XF != 0 ? (unsigned)(xq * 16) : W_OOB);
            if constexpr (XF != 0) rx[li] = *reinterpret_cast<const uint4*>(xb + off); else rx[li] = wino_buffer_load16(xrs, off, 0);
            okx |= ok ? (1u << li) : 0u;
        }
#pragma unroll
        for (int li = 0; li < 2; ++li) {
            const int ps = (tid + 256 * li) >> 3;
            int n = n_u, ty = ty_u; bool ok = true;
            if (RB > 1) { const int rr = rr0 + d_sg[li], rrc = min(rr, geo.nrows - 1); n = fdiv24(rrc, TH); ty = rrc - n * TH; ok = d_sg[li] < RB && rr < geo.nrows; }
            const int oy = 2 * ty + ((ps >> 1) & 1), ox = 2 * (tx0 + d_tx[li]) + (ps & 1);
            ok = ok && d_tx[li] < L && oy < a.Ho && ox < a.Wo;
            const unsigned off = ok ? ((unsigned)((long)n * a.dy_bs) + (unsigned)((oy * a.Wo + ox) * a.dy_ps) + (unsigned)(co0 + xq * 4)) * 4u : W_OOB;
            rd[li] = wino_buffer_load16(drs, off, 0);
            okd |= ok ? (1u << li) : 0u;
        }
    };
    auto lstore = [&](int so) {
#pragma unroll
        for (int li = 0; li < 5; ++li) {
            const unsigned mk = 0u - ((okx >> li) & 1u);
            uint4 v = rx[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            if constexpr (XF != 0) {
                v.x = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), xsc.x, xsh.x), 0.f)); v.y = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), xsc.y, xsh.y), 0.f));
                v.z = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), xsc.z, xsh.z), 0.f)); v.w = __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), xsc.w, xsh.w), 0.f));
            }
            if constexpr (XF != 0) { v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk; }          // (XF == 0: zeros from the range-checked buffer load)
            *reinterpret_cast<uint4*>(wsm + so + ((tid + 256 * li) >> 3) * 32 + xq * 4) = v;
        }
#pragma unroll
        for (int li = 0; li < 2; ++li) {
            uint4 v = rd[li];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            *reinterpret_cast<uint4*>(wsm + so + WG_XF + ((tid + 256 * li) >> 3) * 32 + xq * 4) = v;
        }
    };

    // ---- consumer: wave i; lane = (channel lr, tile group g): tiles 8 g .. 8 g + 7 of the batch
    const float ca = (wi == 3) ? 0.f : 1.f, cb = (wi == 0) ? 0.f : (wi == 1 ? 1.f : -1.f);
    const int r1 = (wi == 0) ? 0 : (wi == 2 ? 2 : 1), r2 = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = (wi == 1) ? 1.f : -1.f;
    const int lr = lane & 31, g8 = (lane >> 5) * 8;
    const int xo1 = r1 * WPS * 32 + lr, xo2 = r2 * WPS * 32 + lr;          // + the tile's slot offset xoff(t) + column b * 32
    // slot offset (in floats) of tile t = g8 + i of the batch: segment t / TW, tile column t % TW (one segment: 2 t slots)
    // (row-packed batches hold RB * TW <= 16 tiles in at most 160 patch slots: a tile index past them -- the MFMA's K is always 16 -- must not be turned into a slot, it would
    //  point past the patch region and, for the second stage, past the workgroup's LDS; its dy is zero, any staged patch will do: the last real tile's)
    auto xoff = [&](const int i) { const int t = RB > 1 ? min(g8 + i, RB * TW - 1) : g8 + i, seg = (RB > 1 && t >= TW) ? fdiv24(t, TW) : 0; return (seg * 4 * WPS + 2 * (t - seg * TW)) * 32; };
    const int dyo = WG_XF + (g8 * 4) * 32 + lr;                                                      // + tile t * 4 pixel rows + px * 32
    typedef wbf16x8 b8_t;
    auto cat8 = [](const unsigned (&v)[4]) { return __builtin_bit_cast(b8_t, (wu32x4){v[0], v[1], v[2], v[3]}); };

    f32x16 P[6];          // P00, P01, P02, P11, P12, P13
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) P[j][e] = 0.f;
    float bsum = 0.f;
    const bool do_bias = a.db && tile_ci == 0 && wi == 0;

    if (b_lo < b_hi) {
        gload(b_lo); lstore(0);
        __syncthreads();
        for (int b = b_lo; b < b_hi; ++b) {
            const int so = ((b - b_lo) & 1) * WS_STAGE;
            const int bn = min(b + 1, b_hi - 1);
            gload(bn);
            __builtin_amdgcn_sched_barrier(0);
            // (tiles past the end of a short last batch were committed as zeros: they contribute nothing)
            // The sums are bilinear in (p, r) -- the row-combined dy columns p_0, p_1 and patch columns r_0 .. r_3 of this wave's Winograd row:
            //   dM_(i,.) = (p0, p0 + p1, p0 - p1, -p1),  V_(i,.) = (r0 - r2, r1 + r2, r2 - r1, r1 - r3)
            // so instead of forming (and splitting) the four dM and four V columns, the SIX products P_ab = sum_t p_a[t] r_b[t] that occur are accumulated and combined once per
            // workgroup (below): 6 instead of 8 values to split per tile and no column transforms -- a third fewer vector instructions in a loop that is bound by them.
            if (do_bias) {          // one wave of one block column: a loop of its own, or hipcc if-converts the 32 additions into every wave's batch loop
                float t = 0.f;
#pragma unroll
                for (int q = 0; q < 32; ++q) t += wsm[so + dyo + q * 32];
                bsum += t;
            }
            // small terms first (a = dy column pieces, b = patch column pieces; piece index 0 = hi, NP - 1 = lo)
            auto prod = [&](f32x16& acc, const unsigned (&ap)[NP][4], const unsigned (&bp)[NP][4]) {
                const b8_t ah = cat8(ap[0]), al = cat8(ap[NP - 1]), bh = cat8(bp[0]), bl = cat8(bp[NP - 1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                if constexpr (NP == 3) {
                    const b8_t am = cat8(ap[1]), bm = cat8(bp[1]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            };
            if constexpr (NP == 2) {
                unsigned pp[2][NP][4], rp[4][NP][4];          // [column][piece: hi, lo][tile pair]: two bf16 (tiles 2 tp, 2 tp + 1) per register
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    float R[2][4], Pv[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float* xs = wsm + so + xoff(2 * tp + u);
#pragma unroll
                        for (int c = 0; c < 4; ++c) R[u][c] = fmaf(sg, xs[xo2 + c * 32], xs[xo1 + c * 32]);
                        const float* ds = wsm + so + dyo + (2 * tp + u) * (4 * 32);
                        const float y00 = ds[0], y01 = ds[32], y10 = ds[64], y11 = ds[96];
                        Pv[u][0] = ca * y00 + cb * y10; Pv[u][1] = ca * y01 + cb * y11;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) split_pk(R[0][c], R[1][c], rp[c][0][tp], rp[c][1][tp]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) split_pk(Pv[0][q], Pv[1][q], pp[q][0][tp], pp[q][1][tp]);
                    __builtin_amdgcn_sched_barrier(0);          // (keeps hipcc from hoisting all 96 LDS reads of the batch: registers)
                }
                prod(P[0], pp[0], rp[0]); prod(P[1], pp[0], rp[1]); prod(P[2], pp[0], rp[2]);          // P00, P01, P02
                prod(P[3], pp[1], rp[1]); prod(P[4], pp[1], rp[2]); prod(P[5], pp[1], rp[3]);          // P11, P12, P13
            } else {
                // three pieces: the dy columns first (24 registers of pieces), then one patch column at a time (12), each followed by the products it takes part in -- all six
                // columns' pieces at once (72 registers beside 96 accumulators) spilt
                unsigned pp[2][NP][4];
                int xoffs[8];
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) xoffs[i8] = xoff(i8);
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    float Pv[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float* ds = wsm + so + dyo + (2 * tp + u) * (4 * 32);
                        const float y00 = ds[0], y01 = ds[32], y10 = ds[64], y11 = ds[96];
                        Pv[u][0] = ca * y00 + cb * y10; Pv[u][1] = ca * y01 + cb * y11;
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) split3_pk(Pv[0][q], Pv[1][q], pp[q][0][tp], pp[q][1][tp], pp[q][2][tp]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned rc[NP][4];
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        const float* x0 = wsm + so + xoffs[2 * tp] + c * 32;
                        const float* x1 = wsm + so + xoffs[2 * tp + 1] + c * 32;
                        const float R0 = fmaf(sg, x0[xo2], x0[xo1]), R1 = fmaf(sg, x1[xo2], x1[xo1]);
                        split3_pk(R0, R1, rc[0][tp], rc[1][tp], rc[2][tp]);
                    }
                    if (c == 0) prod(P[0], pp[0], rc);                                          // P00
                    if (c == 1) { prod(P[1], pp[0], rc); prod(P[3], pp[1], rc); }               // P01, P11
                    if (c == 2) { prod(P[2], pp[0], rc); prod(P[4], pp[1], rc); }               // P02, P12
                    if (c == 3) prod(P[5], pp[1], rc);                                          // P13
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            lstore(WS_STAGE - so);
            __syncthreads();
        }
    }

    // ---- dw = G^T dU G (as conv_wino_wgrad_kernel, one pass: row r is output channel co0 + r)
    float* T = wsm;
    constexpr int TLD = 33;
    const int ci = tid & 31, rw = tid >> 5;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        // dU_(i,0) = P00 - P02, dU_(i,1) = P01 + P02 + P11 + P12, dU_(i,2) = P02 - P01 + P11 - P12, dU_(i,3) = P13 - P11
        const float a0 = P[0][e] - P[2][e], a1 = (P[1][e] + P[2][e]) + (P[3][e] + P[4][e]), a2 = (P[2][e] - P[1][e]) + (P[3][e] - P[4][e]), a3 = P[5][e] - P[3][e];
        const int row = 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        T[((wi * 3 + 0) * 32 + row) * TLD + lr] = a0 + 0.5f * (a1 + a2);
        T[((wi * 3 + 1) * 32 + row) * TLD + lr] = 0.5f * (a1 - a2);
        T[((wi * 3 + 2) * 32 + row) * TLD + lr] = 0.5f * (a1 + a2) + a3;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = rw + 8 * k;
        float* dst = a.dw + ((long)(co0 + row) * 9) * a.Cin + ci0 + ci;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float t0 = T[((0 * 3 + q) * 32 + row) * TLD + ci], t1 = T[((1 * 3 + q) * 32 + row) * TLD + ci], t2 = T[((2 * 3 + q) * 32 + row) * TLD + ci], t3 = T[((3 * 3 + q) * 32 + row) * TLD + ci];
            atomicAdd(dst + (0 * 3 + q) * a.Cin, t0 + 0.5f * (t1 + t2));
            atomicAdd(dst + (1 * 3 + q) * a.Cin, 0.5f * (t1 - t2));
            atomicAdd(dst + (2 * 3 + q) * a.Cin, 0.5f * (t1 + t2) + t3);
        }
    }
    if (do_bias) {          // wave 0 of the block column with input-channel tile 0
        bsum += __shfl_xor(bsum, 32);
        if (lane < 32) atomicAdd(a.db + co0 + lr, bsum);
    }
}

bool conv_wino_wgrad_eligible(const WgradArgs& a) {
    static const bool off = md2_tune_set("MD2_NO_WINO_WGRAD");
    static const long min_m = md2_tune("MD2_WINO_WGRAD_MIN_M", 1);          // (with the split kernel: ahead of the all-taps kernel at every size -- batch 4: 3.79 -> 3.71 ms per step)
    if (off || !g_wino_on || a.bf16) return false;          // (deterministic mode keeps this kernel: one workgroup per weight tile, launch_conv_wino_wgrad)
    if (!(a.KH == 3 && a.KW == 3 && a.stride == 1) || a.Cout % 64 != 0 || a.Cin % 32 != 0) return false;
    if (a.Ho != a.H + 2 * a.pad - 2 || a.Wo != a.W + 2 * a.pad - 2) return false;
    if (a.padmode == PAD_REFLECT && (a.pad != 1 || a.H < 2 || a.W < 2)) return false;
    const long x_bytes = 4 * ((long)a.N * a.x_bs + (long)a.H * a.W * a.x_ps), dy_bytes = 4 * ((long)a.N * a.dy_bs + (long)a.Ho * a.Wo * a.dy_ps);
    if (a.x_ps % 4 != 0 || a.dy_ps % 4 != 0 || x_bytes >= (1L << 32) || dy_bytes >= (1L << 32)) return false;
    return g_wino_on == 2 || g_wino_on == 3 || a.M >= min_m;
}
// opt a kernel into more than 64 KB of dynamic LDS, once per kernel
static void wino_allow_lds(const void* kern, size_t bytes) {
    static std::vector<const void*> done;
    for (const void* k : done) if (k == kern) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    done.push_back(kern);
}
void launch_conv_wino_wgrad(const WgradArgs& a, hipStream_t s) {
    WinoWgGeo g;
    g.TH = (a.Ho + 1) / 2; g.TW = (a.Wo + 1) / 2; g.BPR = (g.TW + WG_TB - 1) / WG_TB; g.nrows = a.N * g.TH;
    g.RB = 1; g.WPS = WG_WP;
    g.xbytes = (unsigned)std::min<long>(4 * ((long)a.N * a.x_bs + (long)a.H * a.W * a.x_ps), (long)W_OOB - 16);          // (conv_wino_wgrad_eligible: both spans are below 2^32)
    g.dybytes = (unsigned)std::min<long>(4 * ((long)a.N * a.dy_bs + (long)a.Ho * a.Wo * a.dy_ps), (long)W_OOB - 16);
    static const bool no_rows = md2_tune_set("MD2_WINO_WGRAD_NO_ROWPACK");
    if (!no_rows && 2 * g.TW <= WG_TB && g.nrows > 1) {          // short tile rows: several per batch (as many as fill the 16 tiles and the 160 staged patch slots)
        g.WPS = 2 * g.TW + 2;
        g.RB = std::min(std::min(WG_TB / g.TW, WG_XSLOTS / (4 * g.WPS)), g.nrows);
        if (g.RB < 2) { g.RB = 1; g.WPS = WG_WP; }
    }
    g.nb = g.RB > 1 ? (g.nrows + g.RB - 1) / g.RB : a.N * g.TH * g.BPR;
    g.nco = a.Cout / 64; g.nci = a.Cin / 32;
    // workgroups a launch is split-K'd up to: the chip's 512 slots when it runs alone, 256 (round 6) on the weight-gradient stream BESIDE the dependency chain -- at batch 4 a
    // launch that takes every CU slows the chain it overlaps by more than it gains itself: 3.19-3.21 -> 3.11-3.14 ms per step, batch 32 and ResNet-50 unchanged
    // (tools/run_wgrad_blocks_ab.sh, profiles/r06_wgrad_blocks.txt; alone -- one stream, the op-level call -- 256 costs 5 % of a batch-32 step)
    static const int want_env = md2_tune("MD2_WINO_WGRAD_BLOCKS", 0);
    const int want = want_env > 0 ? want_env : (a.overlapped ? 256 : 512);
    int ksn = std::max(1, std::min(g.nb, (want + g.nco * g.nci - 1) / (g.nco * g.nci)));
    g.bpb = (g.nb + ksn - 1) / ksn; ksn = (g.nb + g.bpb - 1) / g.bpb;
    // bit-reproducible mode: ONE workgroup per weight tile walks every batch, so each weight (and bias) element receives exactly one atomic add onto zero
    if (g_deterministic) { ksn = 1; g.bpb = g.nb; }
    const int pieces = wgrad_resolve_pieces_wino(a.pieces_mode1 ? a.pieces_mode1 - 1 : g_wgrad_pieces, a.M, a.Cout, a.Cin);
    if (pieces) {          // products on the bf16 matrix pipe (3: exact three-piece split; 2: reduced precision, opt-in); 32 x 32-channel blocks
        g.nco = a.Cout / 32;
        ksn = std::max(1, std::min(g.nb, (want + g.nco * g.nci - 1) / (g.nco * g.nci)));
        if (g_deterministic) ksn = 1;
        g.bpb = (g.nb + ksn - 1) / ksn; ksn = (g.nb + g.bpb - 1) / g.bpb;
        constexpr size_t lds_s = std::max((size_t)2 * WS_STAGE * sizeof(float), (size_t)12 * 32 * 33 * sizeof(float));
        const int grid_s = g.nco * g.nci * ksn;
        auto go = [&](auto kern) {
            wino_allow_lds(reinterpret_cast<const void*>(kern), lds_s);
            hipLaunchKernelGGL(kern, dim3(grid_s), dim3(256), lds_s, s, a, g);
        };
        const int xf = a.in_scale ? 1 : 0, mr = g.RB > 1 ? 1 : 0;
        if (pieces == 3) {
            if (xf && mr) go(conv_wino_wgrad_s_kernel<1, 3, 1>); else if (xf) go(conv_wino_wgrad_s_kernel<1, 3, 0>); else if (mr) go(conv_wino_wgrad_s_kernel<0, 3, 1>); else go(conv_wino_wgrad_s_kernel<0, 3, 0>);
        } else {
            if (xf && mr) go(conv_wino_wgrad_s_kernel<1, 2, 1>); else if (xf) go(conv_wino_wgrad_s_kernel<1, 2, 0>); else if (mr) go(conv_wino_wgrad_s_kernel<0, 2, 1>); else go(conv_wino_wgrad_s_kernel<0, 2, 0>);
        }
        return;
    }
    constexpr size_t lds_pipe = (size_t)2 * WG_STAGE * sizeof(float), lds_epi = (size_t)12 * 32 * 33 * sizeof(float);
    constexpr size_t lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    const int grid = g.nco * g.nci * ksn;
    auto go = [&](auto kern) {
        wino_allow_lds(reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, g);
    };
    const int xf = a.in_scale ? 1 : 0, mr = g.RB > 1 ? 1 : 0;
    if (xf && mr) go(conv_wino_wgrad_kernel<1, 1>); else if (xf) go(conv_wino_wgrad_kernel<1, 0>); else if (mr) go(conv_wino_wgrad_kernel<0, 1>); else go(conv_wino_wgrad_kernel<0, 0>);
}

// load-time defaults (md2_set_winograd / md2_set_conv_products change them): MD2_WINO_MODE=2/3 pins one kernel, 4 / 5 = mode 1 with two / three bf16 pieces per operand;
// MD2_WINO_PIECES / MD2_WGRAD_PIECES = 0 (fp32 MFMA), 3 (exact split), 2 (reduced precision)
static int env_wino_mode() { return getenv("MD2_NO_WINO") ? 0 : (getenv("MD2_WINO_MODE") ? atoi(getenv("MD2_WINO_MODE")) : 1); }
static int env_pieces(const char* name, int dflt) { const char* e = getenv(name); const int v = e ? atoi(e) : dflt; return (v >= 1 && v <= 3) ? v : 0; }
int g_wino_on = env_wino_mode() >= 4 ? 1 : env_wino_mode();
int g_wino_pieces = env_pieces("MD2_WINO_PIECES", env_wino_mode() == 4 ? 2 : (env_wino_mode() == 5 ? 3 : MD2_DEFAULT_WINO_PIECES));
int g_wgrad_pieces = env_pieces("MD2_WGRAD_PIECES", MD2_DEFAULT_WGRAD_PIECES);
// (round 4: 48 instead of 96 64-tile equivalents -- with the split kernel the 256- and 512-channel layers of a batch-4 step are faster as one under-filled round of Winograd
//  workgroups than as direct convolutions with split-K: 3.42 -> 3.26 ms per step at batch 4, unchanged at batch 32; 16 is too low: 3.32)
int g_wino_min_blocks_default = getenv("MD2_WINO_MIN_BLOCKS") ? atoi(getenv("MD2_WINO_MIN_BLOCKS")) : 48;
int g_wino_min_blocks = g_wino_min_blocks_default;

// 3x3, stride 1, dense output lattice, no split-K, fp32; Cin % 16, Cout % 32.  Automatic choice (md2_set_winograd(1, n)): the 64-tile x 32-channel kernel when it
// launches at least n workgroups (default 48 64-tile equivalents; 96 until round 4: in-model per-launch timings at batch 32 and 4, tools/run_layer_compare.sh -- below that the direct implicit GEMM with its split-K
// wins), the direct convolution otherwise; the first (32-tile x 64-channel) kernel only when pinned (mode 2; mode 3 pins conv_wino3_kernel, whose workgroup is 64 or 32 tiles
// by workgroup count, wino3_eligible).
static bool wino3_eligible(const ConvArgs& a, Wino2Geo& g, int& nh);
bool conv_wino_eligible(const ConvArgs& a) {
    if (!g_wino_on || !a.wino_u || a.bf16 == 1) return false;
    if (a.bf16 == 2 && (!g_wino_bf16 || a.wino_pieces != 4 || a.in_scale)) return false;          // bf16 tensors: conv_wino3b_kernel only (U as one bf16 piece, no input transform), and only on request (md2_set_winograd(on + 32, .))
    if (a.bf16 != 2 && a.wino_pieces == 4) return false;
    if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.dil == 1 && a.ostep == 1 && a.nkh == 3 && a.nkw == 3 && a.oy0 == 0 && a.ox0 == 0 && a.Hl == a.Ho && a.Wl == a.Wo)) return false;
    if (a.Cin % 16 != 0 || a.Cout % 32 != 0 || a.acc_lat) return false;
    if (a.Ho != a.H + 2 * a.pad - 2 || a.Wo != a.W + 2 * a.pad - 2) return false;
    if (a.padmode == PAD_REFLECT && (a.pad != 1 || a.H < 2 || a.W < 2)) return false;
    Wino2Geo g; int nh = 2;
    if (g_wino_on == 1 || a.wino_pieces) return wino3_eligible(a, g, nh);          // (U in bf16 pieces: only conv_wino3s_kernel reads that layout)
    const long tiles = (long)a.N * ((a.Ho + 1) / 2) * ((a.Wo + 1) / 2);
    const int cob = a.Cout % 64 == 0 ? 64 : 32;
    return ((tiles + WT - 1) / WT) * (a.Cout / cob) >= g_wino_min_blocks;
}
size_t conv_wino_u_floats(int Cout, int Cin) { return (size_t)24 * Cout * Cin; }          // 16 fp32 per weight pair, or 16 x 3 bf16 pieces in the exact split mode

template <int NS>
static void launch_wino_t(const ConvArgs& a, hipStream_t s) {
    const int TH = (a.Ho + 1) / 2, TW = (a.Wo + 1) / 2, Mt = a.N * TH * TW, ntp = (Mt + WT - 1) / WT;
    const int grid = ntp * (a.Cout / (32 * NS));
    constexpr size_t lds_pipe = 2 * W_STAGE * sizeof(float), lds_epi = (size_t)8 * 32 * (32 * NS + 4) * sizeof(float);
    constexpr size_t lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<NS, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<NS, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<NS, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (a.in_scale) hipLaunchKernelGGL((conv_wino_kernel<NS, 1, 0>), dim3(grid), dim3(256), lds, s, a, TH, TW, Mt, ntp);
    else if (a.bst) hipLaunchKernelGGL((conv_wino_kernel<NS, 0, 1>), dim3(grid), dim3(256), lds, s, a, TH, TW, Mt, ntp);
    else hipLaunchKernelGGL((conv_wino_kernel<NS, 0, 0>), dim3(grid), dim3(256), lds, s, a, TH, TW, Mt, ntp);
}

// the caller (launch_conv_igemm) has checked conv_wino_eligible(a)
void launch_conv_wino(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const long x_bytes = 4 * ((long)a.N * a.x_bs + (long)a.H * a.W * a.x_ps);
    if (a.x_ps % 4 != 0 || a.y_ps % 4 != 0 || (a.dact_src && a.da_ps % 4 != 0) || (a.res && a.res_ps % 4 != 0) || x_bytes >= (1L << 32) || (a.in_scale && a.bst)) {
        fprintf(stderr, "md2: conv_wino: unsupported view (x_ps=%d y_ps=%d)\n", a.x_ps, a.y_ps); abort();
    }
    // statistics / pullback sums are fused unless the run must be bit-reproducible (double atomics): then the separate column passes
    const bool fuse = !g_deterministic;
    const bool stats_ok = fuse && !a.bias && a.act == ACT_NONE && !a.accumulate && !a.dact_src;
    if (a.stats && !stats_ok) a.stats = nullptr;
    if (a.bst && !fuse) { a.bst = nullptr; a.bstB = nullptr; }
    Wino2Geo g2; int nh = 2;
    if (wino3_eligible(a, g2, nh)) launch_wino3(a, g2, nh, s);
    else if (a.bf16 == 2) { fprintf(stderr, "md2: conv_wino: bf16 tensors have the 32-tile split kernel only\n"); abort(); }          // (conv_wino_eligible never lets such a launch through)
    else if (a.Cout % 64 == 0) launch_wino_t<2>(a, s); else launch_wino_t<1>(a, s);
    View yv; yv.p = a.y; yv.N = a.N; yv.H = a.Ho; yv.W = a.Wo; yv.C = a.Cout; yv.bs = a.y_bs; yv.ps = a.y_ps; yv.bf = a.bf16 == 2;
    if (a_in.stats && !a.stats) {
        if (a_in.bias || a_in.act != ACT_NONE) { fprintf(stderr, "md2: conv_wino: BatchNorm statistics need a bias-free, linear output\n"); abort(); }
        launch_colstats(yv, a_in.stats, s);
    }
    if (a_in.bst && !a.bst) {
        View z = yv; z.p = const_cast<float*>(a_in.bz); z.bs = a_in.bz_bs; z.ps = a_in.bz_ps;
        launch_bn_bwd_reduce(yv, nullptr, ACT_NONE, z, a_in.bstat, a_in.bst, s);
        if (a_in.bstB) { View zb = yv; zb.p = const_cast<float*>(a_in.bzB); zb.bs = a_in.bzB_bs; zb.ps = a_in.bzB_ps; launch_bn_bwd_reduce(yv, nullptr, ACT_NONE, zb, a_in.bstatB, a_in.bstB, s); }
    }
}

}  // namespace md2
