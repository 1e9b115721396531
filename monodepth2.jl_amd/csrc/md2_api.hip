// md2_api.hip — handle, network wiring and the C ABI of include/md2hip.h.
//
// Host-side mirror of src/model.jl (Model forward), src/depth_decoder.jl, src/pose_decoder.jl, the ResNet-18 trunk
// (torchvision convention), src/training.jl:21-78 (train_loss) and their hand-written pullbacks
// (the reference gets them from Zygote).  A step is enqueued on the caller's stream plus internal side streams (weight gradients,
// pose branch, RCCL buckets) forked and joined by events; no allocation or host synchronisation happens inside a step.
#include "md2_internal.h"
#include "../../include/md2hip.h"
#include <vector>
#include <string>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>        // types only: the RCCL entry points are resolved with dlsym at md2_comm_init (no link-time dependency)

using namespace md2;

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e_); return 1; } } while (0)
#define KCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e_); return 1; } } while (0)

static std::string g_create_err;

namespace {

struct ConvL {
    int cin = 0, cout = 0, k = 1, stride = 1, pad = 0; bool bias = false;
    int cinP = 0, coutP = 0, taps = 1;   // internal (padded) geometry: weights [coutP][taps][cinP]
    int kind = 0;                        // 0 igemm, 1 stem (direct 7x7 kernels, weights [64][Kpad]: taps=1, cinP=Kpad), 2 head (cout 1, direct kernel)
    long w = 0, b = 0;                   // offsets (floats) into the internal flat buffers
    long fw = 0, fb = 0;                 // offsets into the Flux-layout flat vector
    long uf = -1, ud = -1;               // offsets of the layer's Winograd operands (forward / input gradient) in md2_handle::UF / UD; -1 = the layer never takes the Winograd kernel
    long wn() const { return (long)coutP * taps * cinP; }
};
struct BNL { int c = 0; long g = 0, b = 0, fg = 0, fb = 0; long rs = 0; float* stats = nullptr; double* acc = nullptr; double* acc_b = nullptr; };   // rs: offset of (running mean[c], running var[c]) in md2_handle::RS
// BasicBlock (ResNet-18): c1 3x3(stride) - b1 - relu - c2 3x3 - b2 (+ shortcut) - relu
// Bottleneck (ResNet-50): c1 1x1 - b1 - relu - c2 3x3(stride) - b2 - relu - c3 1x1 x4 - b3 (+ shortcut) - relu     (torchvision convention)
// fuse (BasicBlocks whose kernels support it): a1 = relu(bn1(z1)) is never materialised -- conv2 and its weight gradient apply bn1 + ReLU in their loaders, the
// BatchNorm-pullback sums come out of the input-gradient epilogues, and the shortcut BatchNorm is applied inside the block's final bn + add + ReLU pass
struct Block { ConvL c1, c2, c3, cd; BNL b1, b2, b3, bd; bool down = false, bottleneck = false, stage_end = false, fuse = false; int stride = 1;
               View in, z1, a1, z2, a2, z3, zd, sd, out; View dz1, dz2, dz3, dzd; };
struct ProfEvt { hipEvent_t a, b; int cls; double flops; char desc[144]; };
enum { CLS_IGEMM = 0, CLS_WGRAD = 1, CLS_LOSS_FWD = 2, CLS_LOSS_BWD = 3, CLS_OTHER = 4 };      // CLS_LOSS_BWD: the training step's fused loss value + pullback kernel (bench.py: roofline_loss_train)

}  // namespace

struct md2_handle {
    md2_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    int N = 0, NL = 0, C = 1, H = 0, W = 0;
    int tgt = 1, src[2] = {0, 2};            // 0-based target / source frames of the triplet (md2_config.target_id / source_ids; src/training.jl:25-32)
    int pose_fr[4] = {0, 1, 1, 2};           // temporally ordered (first, second) frames of the two pose pairs (src/model.jl:30-35)
    float K[9], invK[9];
    long n_int = 0, n_flux = 0;
    float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr, *WD = nullptr;
    bool bs = false; uint16_t *Pb = nullptr, *WDb = nullptr;      // precision 2: activations in bf16; bf16 copies of the weights, refreshed from P / WD once per step
    double beta1p = 1.0, beta2p = 1.0;
    ConvL stem; BNL stem_bn; Block blk[16]; int nblk = 8, depth = 18, bf16 = 0; int stages[5] = {64, 64, 128, 256, 512};
    ConvL dc1[5], dc2[5], head[5], squeeze, pose[3];
    // DepthDecoder(; scale_levels) (src/depth_decoder.jl:26-47): ns outputs at levels lv[0] < ... < lv[ns-1] (1..5), branches 1..nbr = lv[ns-1]; out_of[b] = output index of
    // the head on branch b (0-based) or -1; scl[i] = TrainCache.scales[i] (the smoothness weight of output i, src/training.jl:67)
    int ns = 4, lv[5] = {2, 3, 4, 5, 0}, nbr = 5, out_of[5] = {-1, 0, 1, 2, 3}; double scl[5] = {0.125, 0.25, 0.5, 1.0, 0.0};
    std::vector<ConvL*> convs; std::vector<BNL*> bns;
    char* arena = nullptr; size_t arena_off = 0;
    double* acc_all = nullptr; size_t acc_count = 0; double* acc_bwd = nullptr; size_t acc_fwd_count = 0, acc_bwd_count = 0;
    int Kpad = 64;
    View z0, a0, p0; uint8_t* pool_arg = nullptr;
    float* bn_ident = nullptr;       // [mean = 0 | 1 / sigma = 1] x 2048 channels: with these as its statistics, bn_apply(gamma = scale, beta = shift) is the loaders' fma(z, scale, shift), bit for bit
    float* a_fwd = nullptr;          // bf16 tensors: relu(bn(z)) of a fused block, materialised for the FORWARD convolution only (one buffer, the largest block's size); the weight gradient still transforms z in its loader
    View feat[5], y1[5], cat[5], y2[5], dispv[5];
    View sq3, dsq3; View sqcat, py1, py2, py3; float *rvec = nullptr, *tvec = nullptr, *Rt = nullptr;
    double *loss_part = nullptr, *dRt_acc = nullptr; size_t ddisp_lo_count = 0; float *totals = nullptr, *loss_dev = nullptr, *ddisp_full = nullptr, *drvec = nullptr, *dtvec = nullptr;
    const float* x_last = nullptr; const float* auto_last = nullptr; bool have_fwd = false;
    View Gfeat[5], Gblk[16], Gpool; char* gzero_begin = nullptr; size_t gzero_bytes = 0;
    View gA[5], dZ1[5], dupv[5]; float* dxpad = nullptr; float* ddisp_lo[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float *E1 = nullptr, *E2 = nullptr, *E3 = nullptr;
    View dpy3, dpy2, dpy1, dsq;
    long* wdesc_dev = nullptr; int n_wdesc = 0, n_wblocks = 0;
    // Winograd operands U = G g G^T of every 3x3 / stride-1 layer, refreshed once per step from P (forward) and WD (input gradient)
    float *UF = nullptr, *UD = nullptr; long n_uf = 0, n_ud = 0; long *udesc_f = nullptr, *udesc_d = nullptr; int n_udf = 0, n_udd = 0, ublk_f = 0, ublk_d = 0; hipEvent_t ev_uf = nullptr;
    int wg_pieces = MD2_DEFAULT_WGRAD_PIECES;      // g_wgrad_pieces as it stood when this handle's pullback was prepared: every weight-gradient launch of the step takes this mode (WgradArgs::pieces_mode1)
    int u_pieces_f = 0, u_pieces_d = 0;      // layout UF / UD were last written in (g_wino_pieces at that refresh): every launch that reads them is told through ConvArgs::wino_pieces,
                                             // so a md2_set_conv_products between md2_train_loss and md2_backward (or from another handle) cannot make a kernel misread its operand
    // weight gradients run on a second stream, concurrently with the input-gradient / BN-pullback chain of the main stream
    hipStream_t s3 = nullptr; hipStream_t cur = nullptr; hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;   // pose branch stream; `cur` = stream of the chain being enqueued
    hipStream_t s4 = nullptr; int wg_rr = 0; int n_wg_streams = 1; hipEvent_t ev_join4 = nullptr;
    hipStream_t s2 = nullptr; hipEvent_t ev_pool[64]; int ev_next = 0; hipEvent_t ev_join = nullptr, ev_p0 = nullptr, ev_p1 = nullptr; bool prep_done = false; bool two_streams = true; bool forked = false; View dz0;
    int wg_idx = 0, wg_total = 0;       // weight-gradient launches of the running pullback / of the last complete one (wgrad_stream)
    bool acc_zeroed = false;            // md2_train_loss: the forward pass's memset covers the pullback / loss accumulators too
    // launches md2_train_step takes off the caller's stream (round 6, batch 4 is a chain of ~200 dependent launches): the coarse disparity gradients are zeroed on the
    // preparation stream at the head of the step (ddisp_zeroed), the pose stream composes Rt right behind the pose decoder (Rt_ready), and the loss's tail -- the
    // scalar's finalisation and the pose pullback's first kernel, neither read by the depth decoder's pullback -- opens the pose stream's part of model_backward (tail)
    bool ddisp_zeroed = false, Rt_ready = false, defer_tail = false;
    struct LossTail { bool pending = false; LossArgs a; float* loss_out = nullptr; const float *rvec = nullptr, *tvec = nullptr; } tail;
    bool net_ok = true;                 // false: the image is too small for the network (H or W < 64): only the loss-half entry points work
    float* RS = nullptr; long n_rs = 0; bool testmode = false;     // BatchNorm running mean / variance (Flux BatchNorm mu, sigma^2; momentum 0.1) and Flux.testmode!
    double* loss_blk = nullptr;         // deterministic mode: per-block partial sums of the loss kernels
    float* ws_main = nullptr; float* ws_pose = nullptr; long ws_cap = 0;   // split-K workspaces of the implicit GEMM: one per stream that runs convolutions concurrently
    // data parallelism: RCCL communicator + bucketed all-reduce of the flat gradient buffer, overlapped with the rest of the pullback
    void* comm = nullptr; int rank = 0, world = 1; hipStream_t scomm = nullptr; hipEvent_t ev_cm[8], ev_cs[8], ev_cdone = nullptr; bool grads_reduced = false; long bucket_lo[8]; int n_bucket_done = 0;
    bool prof = false; std::vector<ProfEvt> evts; double prof_flops[5] = {0, 0, 0, 0, 0}, prof_bytes[5] = {0, 0, 0, 0, 0}, prof_exec[5] = {0, 0, 0, 0, 0};

    template <class T> T* alloc(size_t count) {
        arena_off = (arena_off + 255) / 256 * 256;
        T* p = reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(arena) + arena_off);
        arena_off += count * sizeof(T);
        return p;
    }
    // activation / gradient tensors: bf16 elements when the handle stores activations in bf16 (md2_config.precision = 2), else fp32
    View alloc_view(int n, int hh, int ww, int c) {
        if (bs) return make_view(reinterpret_cast<float*>(alloc<uint16_t>((size_t)n * hh * ww * c + 8)), n, hh, ww, c, 1);
        return make_view(alloc<float>((size_t)n * hh * ww * c), n, hh, ww, c);
    }
    View alloc_view_f32(int n, int hh, int ww, int c) { return make_view(alloc<float>((size_t)n * hh * ww * c), n, hh, ww, c); }
    View scratch_view(float* p, int n, int hh, int ww, int c) const { return make_view(p, n, hh, ww, c, bs ? 1 : 0); }          // a view of shared fp32-sized scratch in the handle's storage type
    // convolution weights as the kernels of this handle read them: the fp32 master (P / WD) or its per-step bf16 copy (Pb / WDb)
    const float* wfwd(long off) const { return bs ? reinterpret_cast<const float*>(Pb + off) : P + off; }
    const float* wdg(long off) const { return bs ? reinterpret_cast<const float*>(WDb + off) : WD + off; }
    hipStream_t chain() const { return cur ? cur : stream; }
    void pbegin(int cls, double flops = 0, const char* desc = "", hipStream_t st = nullptr, bool own = false) { if (prof) { ProfEvt e; e.cls = cls; e.flops = flops; snprintf(e.desc, sizeof(e.desc), "%s", desc); (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); (void)hipEventRecord(e.a, own ? st : chain()); evts.push_back(e); } }
    void pend(hipStream_t st = nullptr, bool own = false) { if (prof) (void)hipEventRecord(evts.back().b, own ? st : chain()); }
};

static int comm_bucket(md2_handle* h, long lo, long hi);      // RCCL all-reduce of a finished slice of the gradient buffer on the communication stream (below)

// ---------------------------------------------------------------------------------------------------------------------
// layer table
// ---------------------------------------------------------------------------------------------------------------------
static void build_layers(md2_handle* h) {
    long io = 0, fo = 0;
    auto al = [&](long n) { long r = io; io += (n + 15) / 16 * 16; return r; };
    auto mk = [&](ConvL& c, int cin, int cout, int k, int s, int p, bool bias, int kind) {
        c.cin = cin; c.cout = cout; c.k = k; c.stride = s; c.pad = p; c.bias = bias; c.kind = kind;
        if (kind == 1) { c.taps = 1; c.cinP = (49 * cin + 15) / 16 * 16; c.coutP = cout; }
        else if (kind == 2) { c.taps = k * k; c.cinP = cin; c.coutP = 1; }
        else { c.taps = k * k; c.cinP = (cin + 15) / 16 * 16; c.coutP = (cout + 15) / 16 * 16; }
        c.w = al(c.wn()); c.fw = fo; fo += (long)cin * cout * k * k;
        if (bias) { c.b = al(c.coutP); c.fb = fo; fo += cout; }
        h->convs.push_back(&c);
    };
    long ro = 0;
    auto mkbn = [&](BNL& b, int c) { b.c = c; b.g = al(c); b.b = al(c); b.fg = fo; b.fb = fo + c; fo += 2 * c; b.rs = ro; ro += 2 * c; h->bns.push_back(&b); };
    mk(h->stem, h->C, 64, 7, 2, 3, false, 1); mkbn(h->stem_bn, 64);
    h->Kpad = h->stem.cinP;
    const int planes[4] = {64, 128, 256, 512}; int cin = 64;
    const int nb18[4] = {2, 2, 2, 2}, nb50[4] = {3, 4, 6, 3};
    const int* nb = h->depth == 50 ? nb50 : nb18;
    h->nblk = 0; h->stages[0] = 64;
    for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < nb[l]; ++b) {
            Block& B = h->blk[h->nblk++]; const int s = (b == 0 && l > 0) ? 2 : 1;
            B.stride = s; B.bottleneck = h->depth == 50; B.stage_end = (b == nb[l] - 1);
            int co;
            if (!B.bottleneck) { co = planes[l]; mk(B.c1, cin, co, 3, s, 1, false, 0); mkbn(B.b1, co); mk(B.c2, co, co, 3, 1, 1, false, 0); mkbn(B.b2, co); }
            else { co = 4 * planes[l]; mk(B.c1, cin, planes[l], 1, 1, 0, false, 0); mkbn(B.b1, planes[l]); mk(B.c2, planes[l], planes[l], 3, s, 1, false, 0); mkbn(B.b2, planes[l]);
                   mk(B.c3, planes[l], co, 1, 1, 0, false, 0); mkbn(B.b3, co); }
            B.down = (s != 1 || cin != co);
            if (B.down) { mk(B.cd, cin, co, 1, s, 0, false, 0); mkbn(B.bd, co); }
            cin = co;
        }
        h->stages[l + 1] = cin;
    }
    // DepthDecoder channel arithmetic from the encoder stages (src/depth_decoder.jl:31-35)
    const int dec[5] = {256, 128, 64, 32, 16};
    const int inc[5] = {h->stages[4], 256, 128, 64, 32}, skc[5] = {h->stages[3], h->stages[2], h->stages[1], h->stages[0], 0};
    for (int b = 0; b < h->nbr; ++b) { mk(h->dc1[b], inc[b], dec[b], 3, 1, 0, true, 0); mk(h->dc2[b], dec[b] + skc[b], dec[b], 3, 1, 0, true, 0); }
    for (int i = 0; i < h->ns; ++i) mk(h->head[i], dec[h->lv[i] - 1], 1, 3, 1, 0, true, 2);          // Flux.params order: all branches, then the decoders (heads)
    mk(h->squeeze, h->stages[4], 256, 1, 1, 0, true, 0); mk(h->pose[0], 512, 256, 3, 1, 1, true, 0); mk(h->pose[1], 256, 256, 3, 1, 1, true, 0); mk(h->pose[2], 256, 6, 1, 1, 0, true, 0);
    h->n_int = io; h->n_flux = fo; h->n_rs = ro;
}

// Flux (kW,kH,Cin,Cout) true-convolution weights  <->  internal [coutP][taps][cinP] cross-correlation taps
static void convert_params(md2_handle* h, float* internal, float* flux, bool to_internal) {
    if (to_internal) std::fill(internal, internal + h->n_int, 0.f);
    for (ConvL* cp : h->convs) {
        const ConvL& c = *cp; const int k = c.k;
        for (int co = 0; co < c.cout; ++co)
            for (int ci = 0; ci < c.cin; ++ci)
                for (int kh = 0; kh < k; ++kh)
                    for (int kw = 0; kw < k; ++kw) {
                        const long f = c.fw + (((long)co * c.cin + ci) * k + kh) * k + kw;
                        const int tap = (k - 1 - kh) * k + (k - 1 - kw);          // flip: true convolution -> cross-correlation
                        const long i = (c.kind == 1) ? c.w + (long)co * c.cinP + (long)tap * c.cin + ci
                                                     : c.w + ((long)co * c.taps + tap) * c.cinP + ci;
                        if (to_internal) internal[i] = flux[f]; else flux[f] = internal[i];
                    }
        if (c.bias) for (int co = 0; co < c.cout; ++co) { if (to_internal) internal[c.b + co] = flux[c.fb + co]; else flux[c.fb + co] = internal[c.b + co]; }
    }
    for (BNL* bp : h->bns)
        for (int i = 0; i < bp->c; ++i) {
            if (to_internal) { internal[bp->g + i] = flux[bp->fg + i]; internal[bp->b + i] = flux[bp->fb + i]; }
            else { flux[bp->fg + i] = internal[bp->g + i]; flux[bp->fb + i] = internal[bp->b + i]; }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// buffers (called twice: once with arena == nullptr to size the arena, once to carve it)
// ---------------------------------------------------------------------------------------------------------------------
static void plan_buffers(md2_handle* h) {
    const int N = h->N, NL = h->NL, H = h->H, W = h->W;
    h->arena_off = 0;
    h->z0 = h->alloc_view(NL, H / 2, W / 2, 64); h->a0 = h->alloc_view(NL, H / 2, W / 2, 64);
    h->p0 = h->alloc_view(NL, H / 4, W / 4, 64); h->pool_arg = h->alloc<uint8_t>((size_t)NL * (H / 4) * (W / 4) * 64);
    h->feat[0] = h->a0;
    int hh = H / 4, ww = W / 4; View cur = h->p0;
    size_t a_fwd_elems = 0;
    for (int i = 0, stage = 1; i < h->nblk; ++i) {
        Block& B = h->blk[i];
        const int hi = hh, wi = ww;                      // resolution of the block input (the bottleneck's 1x1 conv1 keeps it)
        if (B.stride == 2) { hh /= 2; ww /= 2; }
        B.in = cur;
        if (!B.bottleneck) {
            const int co = B.c1.cout;
            static const bool no_fuse = md2_tune_set("MD2_NO_BN_FUSE");
            B.fuse = !no_fuse && conv_wgrad_supports_in_transform(hh, ww, co, co);        // conv2 and its weight gradient can fold bn1 + ReLU into their loaders at this size
            B.z1 = h->alloc_view(NL, hh, ww, co); B.z2 = h->alloc_view(NL, hh, ww, co);
            B.a1 = B.fuse ? h->scratch_view(nullptr, NL, hh, ww, co) : h->alloc_view(NL, hh, ww, co);      // fused: never materialised (the view only carries the shape)
            if (B.fuse) a_fwd_elems = std::max(a_fwd_elems, (size_t)NL * hh * ww * co);
        } else {
            const int pl = B.c1.cout, co = B.c3.cout;
            static const bool no_fuse = md2_tune_set("MD2_NO_BN_FUSE");
            B.fuse = !no_fuse;                // 1x1 and 3x3 loaders (forward and weight gradient) all take the BatchNorm + ReLU transform at these channel counts (multiples of 64)
            B.z1 = h->alloc_view(NL, hi, wi, pl); B.a1 = B.fuse ? h->scratch_view(nullptr, NL, hi, wi, pl) : h->alloc_view(NL, hi, wi, pl);
            B.z2 = h->alloc_view(NL, hh, ww, pl); B.a2 = B.fuse ? h->scratch_view(nullptr, NL, hh, ww, pl) : h->alloc_view(NL, hh, ww, pl); B.z3 = h->alloc_view(NL, hh, ww, co);
            if (B.fuse) a_fwd_elems = std::max(a_fwd_elems, (size_t)NL * hi * wi * pl);
        }
        const int co = B.bottleneck ? B.c3.cout : B.c1.cout;
        if (B.down) { B.zd = h->alloc_view(NL, hh, ww, co); B.sd = B.fuse ? h->scratch_view(nullptr, NL, hh, ww, co) : h->alloc_view(NL, hh, ww, co); }
        B.out = h->alloc_view(NL, hh, ww, co);
        cur = B.out;
        if (B.stage_end) h->feat[stage++] = B.out;
    }
    h->a_fwd = (h->bs && a_fwd_elems) ? reinterpret_cast<float*>(h->alloc<uint16_t>(a_fwd_elems + 8)) : nullptr;
    h->bn_ident = h->a_fwd ? h->alloc<float>(2 * 2048) : nullptr;
    const int dec[5] = {256, 128, 64, 32, 16}, skc[5] = {h->stages[3], h->stages[2], h->stages[1], h->stages[0], 0}, inc[5] = {h->stages[4], 256, 128, 64, 32};
    int dh = H / 32, dw = W / 32;
    size_t dxpad_max = 0;
    for (int b = 0; b < h->nbr; ++b) {
        h->y1[b] = h->alloc_view(N, dh, dw, dec[b]);
        dxpad_max = std::max(dxpad_max, (size_t)N * (dh + 2) * (dw + 2) * inc[b]);
        dh *= 2; dw *= 2;
        h->cat[b] = h->alloc_view(N, dh, dw, dec[b] + skc[b]);
        h->y2[b] = h->alloc_view(N, dh, dw, dec[b]);
        dxpad_max = std::max(dxpad_max, (size_t)N * (dh + 2) * (dw + 2) * (dec[b] + skc[b]));
        if (h->out_of[b] >= 0) h->dispv[h->out_of[b]] = h->alloc_view_f32(N, dh, dw, 1);          // (the loss reads the disparity maps: fp32 in every mode)
    }
    // a loss-only handle (H, W multiples of 8 but not of 32) has no decoder to derive the disparity sizes from: the loss reads / writes H >> (3 - l) x W >> (3 - l)
    if (!h->net_ok) for (int l = 0; l < h->ns; ++l) h->dispv[l] = h->alloc_view_f32(N, H >> (5 - h->lv[l]), W >> (5 - h->lv[l]), 1);
    const int ph = H / 32, pw = W / 32;
    h->sq3 = h->alloc_view(3 * N, ph, pw, 256); h->dsq3 = h->alloc_view(3 * N, ph, pw, 256);
    h->sqcat = h->alloc_view(2 * N, ph, pw, 512); h->py1 = h->alloc_view(2 * N, ph, pw, 256); h->py2 = h->alloc_view(2 * N, ph, pw, 256); h->py3 = h->alloc_view(2 * N, ph, pw, 16);
    h->rvec = h->alloc<float>(6 * N); h->tvec = h->alloc<float>(6 * N); h->Rt = h->alloc<float>(24 * N);
    h->totals = h->alloc<float>(4 * MD2_MAX_SCALES * N); h->loss_dev = h->alloc<float>(4); h->ddisp_full = h->alloc<float>((size_t)h->ns * N * H * W);
    h->ws_cap = 8L << 20; h->ws_main = h->alloc<float>((size_t)h->ws_cap); h->ws_pose = h->alloc<float>((size_t)h->ws_cap);      // 2 x 32 MB
    h->drvec = h->alloc<float>(6 * N); h->dtvec = h->alloc<float>(6 * N);
    h->loss_blk = h->alloc<double>((size_t)MD2_MAX_SCALES * N * ((W + 15) / 16) * ((H + 7) / 8) * 25);        // deterministic mode: [scale*N][block][24] partials (bound for loss tiles >= 16x8)
    for (BNL* b : h->bns) b->stats = h->alloc<float>(9 * b->c);                 // mean | invstd | scale | shift | pullback m1 | m2 | A | B | C0 (dz = A g + B z + C0)
    size_t nd = 0;
    for (BNL* b : h->bns) nd += (size_t)(4 * BN_STAT_SLOTS) * b->c;
    nd += (size_t)4 * MD2_MAX_SCALES * N + 24 * N;
    h->acc_count = nd; h->acc_all = h->alloc<double>(nd);
    double* ap = h->acc_all;
    for (BNL* b : h->bns) { b->acc = ap; ap += (size_t)2 * BN_STAT_SLOTS * b->c; }      // [slot][sum | sum of squares][c]
    h->acc_fwd_count = ap - h->acc_all;
    h->acc_bwd = ap;
    for (BNL* b : h->bns) { b->acc_b = ap; ap += (size_t)2 * BN_STAT_SLOTS * b->c; }    // [slot][sum g | sum g*xhat][c]: filled by the producing input gradient's epilogue
    h->acc_bwd_count = ap - h->acc_bwd;
    h->loss_part = ap; ap += 4 * MD2_MAX_SCALES * N; h->dRt_acc = ap; ap += 24 * N;
    // ---- backward: gradient accumulators, zeroed together at the start of every backward
    h->arena_off = (h->arena_off + 255) / 256 * 256;
    const size_t gz0 = h->arena_off;
    for (int i = 0; i < 5; ++i) h->Gfeat[i] = h->alloc_view(h->feat[i].N, h->feat[i].H, h->feat[i].W, h->feat[i].C);
    for (int i = 0, stage = 1; i < h->nblk; ++i) {       // the gradient of a stage output IS the stage's feature gradient buffer
        if (h->blk[i].stage_end) h->Gblk[i] = h->Gfeat[stage++];
        else h->Gblk[i] = h->alloc_view(h->blk[i].out.N, h->blk[i].out.H, h->blk[i].out.W, h->blk[i].out.C);
    }
    h->Gpool = h->alloc_view(NL, H / 4, W / 4, 64);
    h->arena_off = (h->arena_off + 255) / 256 * 256;
    h->gzero_begin = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(h->arena) + gz0); h->gzero_bytes = h->arena_off - gz0;
    // ---- backward scratch
    for (int b = 0; b < h->nbr; ++b) {
        h->gA[b] = h->alloc_view(h->y2[b].N, h->y2[b].H, h->y2[b].W, h->y2[b].C);
        h->dZ1[b] = h->alloc_view(h->y1[b].N, h->y1[b].H, h->y1[b].W, h->y1[b].C);
        h->dupv[b] = h->alloc_view(h->y2[b].N, h->y2[b].H, h->y2[b].W, h->y2[b].C);
    }
    h->dxpad = h->alloc<float>(dxpad_max);
    {   // one block, coarse scales first: the fused loss kernel accumulates into levels 0..2 with atomics, so they are zeroed by a single memset
        size_t tot = 0; for (int l = 0; l < h->ns; ++l) tot += (size_t)N * h->dispv[l].H * h->dispv[l].W;
        float* base = h->alloc<float>(tot); h->ddisp_lo_count = tot;
        for (int l = 0; l < h->ns; ++l) { h->ddisp_lo[l] = base; base += (size_t)N * h->dispv[l].H * h->dispv[l].W; }
    }
    const size_t emax = (size_t)NL * (H / 2) * (W / 2) * 64;
    h->E1 = h->alloc<float>(emax); h->E2 = h->alloc<float>(emax); h->E3 = h->alloc<float>(emax);
    // pre-activation gradients of every encoder conv: NOT shared scratch, the weight-gradient stream reads them while the main stream moves on
    h->dz0 = h->alloc_view(NL, H / 2, W / 2, 64);
    for (int i = 0; i < h->nblk; ++i) { Block& B = h->blk[i]; B.dz1 = h->alloc_view(NL, B.z1.H, B.z1.W, B.z1.C); B.dz2 = h->alloc_view(NL, B.z2.H, B.z2.W, B.z2.C); if (B.bottleneck) B.dz3 = h->alloc_view(NL, B.z3.H, B.z3.W, B.z3.C); if (B.down) B.dzd = h->alloc_view(NL, B.zd.H, B.zd.W, B.zd.C); }
    h->dpy3 = h->alloc_view(2 * N, ph, pw, 16); h->dpy2 = h->alloc_view(2 * N, ph, pw, 256); h->dpy1 = h->alloc_view(2 * N, ph, pw, 256); h->dsq = h->alloc_view(2 * N, ph, pw, 512);
}

// ---------------------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------------------
static ConvArgs conv_args(const View& x, const float* w, const View& y, int KH, int stride, int pad, int dil, int padmode) {
    ConvArgs a{};
    a.x = x.p; a.x_bs = x.bs; a.x_ps = x.ps; a.H = x.H; a.W = x.W; a.Cin = x.C;
    a.w = w; a.y = y.p; a.y_bs = y.bs; a.y_ps = y.ps; a.Ho = y.H; a.Wo = y.W; a.Cout = y.C;
    a.N = y.N; a.KH = KH; a.KW = KH; a.stride = stride; a.pad = pad; a.dil = dil; a.padmode = padmode;
    a.bias = nullptr; a.act = ACT_NONE; a.accumulate = 0; a.dact_src = nullptr; a.da_bs = 0; a.da_ps = 0; a.dact = ACT_NONE;
    a.M = y.N * y.H * y.W;
    a.oy0 = 0; a.ox0 = 0; a.ostep = 1; a.Hl = y.H; a.Wl = y.W; a.kh0 = 0; a.khs = 1; a.nkh = KH; a.kw0 = 0; a.kws = 1; a.nkw = KH;
    a.accumulate_c00 = -1; a.acc_frame = 1;
    return a;
}
static void run_igemm(md2_handle* h, const ConvArgs& a_in) {
    ConvArgs a = a_in; a.bf16 = h->bf16;
    a.split3 = h->bf16 ? 0 : igemm_resolve_split(h->u_pieces_f, a.dil > 1 && a.ostep == 1 ? std::max(1, a.KH * a.KW / (a.dil * a.dil)) : a.nkh * a.nkw, a.Cin, a.Cout);          // (the product mode this handle's step was prepared under)
    a.ws = (h->cur && h->cur == h->s3) ? h->ws_pose : h->ws_main; a.ws_cap = h->ws_cap;
    char d[144] = "";
    // (alg: the route launch_conv_igemm takes -- conv_igemm_executed_flops applies the same test -- so that a test can see which kernels a step ran: md2_profile_dump)
    // ep: output-sized tensors the epilogue READS besides writing y (residual, activation-derivative source, BatchNorm inputs of the fused pullback sums, y itself when it
    // accumulates) -- distinct addresses only: part of the launch's algorithmic bytes (bench.py)
    int ep = 0;
    { const void* seen[5]; int ns = 0; const void* cand[5] = {a.res, a.dact_src, a.bz, a.bzB, a.accumulate ? (const void*)a.y : nullptr};
      for (const void* c : cand) { bool dup = c == nullptr; for (int i = 0; i < ns; ++i) dup = dup || seen[i] == c; if (!dup) { seen[ns++] = c; ++ep; } } }
    const bool wino = conv_igemm_executed_flops(a) < conv_igemm_flops(a);
    if (h->prof) snprintf(d, sizeof(d), "igemm M=%d Cout=%d Cin=%d k=%d s=%d dil=%d pm=%d HxW=%dx%d ep=%d alg=%s prod=%s", a.M, a.Cout, a.Cin, a.KH, a.stride, a.dil, a.padmode, a.H, a.W, ep,
                          wino ? (a.wino_pieces ? "winograd_split" : "winograd") : "direct",
                          // prod: how the launch forms its products -- fp32 (v_mfma_f32_*), split3 / split2 (bf16 pieces of fp32 operands on the bf16 pipe), bf16 (the bf16 modes); bench.py prices the matrix-pipe floor from it
                          a.bf16 ? "bf16" : (wino ? (a.wino_pieces == 3 ? "split3" : (a.wino_pieces == 2 ? "split2" : "fp32")) : ((a.split3 && a.Cout % 64 == 0) ? "split3" : "fp32")));
    h->pbegin(CLS_IGEMM, conv_igemm_flops(a), d); launch_conv_igemm(a, h->chain()); h->pend();
    if (h->prof) { h->prof_flops[CLS_IGEMM] += conv_igemm_flops(a); h->prof_exec[CLS_IGEMM] += conv_igemm_executed_flops(a); }
}
// A short independent branch on the pose stream (idle while the encoder runs): side_fork marks the fork point on the chain (everything enqueued so far happens-before the branch), side_enter makes
// it the chain being enqueued; side_end returns the event the caller's stream must wait for before it reads the branch's results.  Used for the 1x1 shortcut of a
// down-sampling block, forward and pullback: small, latency-bound launches that need not stand in the dependency chain (round 6).
static bool side_branch_ok(const md2_handle* h, bool pullback) {
    static const bool off = md2_tune_set("MD2_NO_SIDE_SHORTCUT"), off_f = md2_tune_set("MD2_NO_SIDE_SHORTCUT_FWD"), off_b = md2_tune_set("MD2_NO_SIDE_SHORTCUT_BWD");
    static const long big = md2_tune("MD2_SIDE_SHORTCUT_MAXPIX", 1500000L);
    // measured on one box (tools/env_probe.sh, profiles/r06_side_shortcut_probe.txt): batch 4 2.97 -> 2.92 ms with both halves; at batch 16 / 32 the forward half alone is
    // best or equal (6.14 -> 6.11, 10.69 -> 10.69) and the pullback half costs 0.3 % -- the pullback already shares the chip with the weight-gradient stream there -- so the
    // pullback half is for steps below 1.5 M target pixels
    return !off && !(pullback ? off_b : off_f) && (!pullback || (long)h->N * h->H * h->W < big) && h->two_streams && h->s3 && !h->cur;
}
// (side_fork marks the point on the chain; the branch itself is ENQUEUED after the main path's launches -- side_enter -- so that a host that is only just ahead of the
//  GPU feeds the dependency chain first; on the GPU the branch may start as soon as the fork point is reached)
static hipEvent_t side_fork(md2_handle* h) {
    hipEvent_t ev = h->ev_pool[h->ev_next]; h->ev_next = (h->ev_next + 1) % 64;
    (void)hipEventRecord(ev, h->stream);
    return ev;
}
static void side_enter(md2_handle* h, hipEvent_t fork) { (void)hipStreamWaitEvent(h->s3, fork, 0); h->cur = h->s3; }
static hipEvent_t side_end(md2_handle* h) {
    hipEvent_t ev = h->ev_pool[h->ev_next]; h->ev_next = (h->ev_next + 1) % 64;
    (void)hipEventRecord(ev, h->s3); h->cur = nullptr;
    return ev;
}
// forward conv (kind 0) on views.  x.C must equal c.cinP, y.C == c.coutP
static void conv_fwd(md2_handle* h, const ConvL& c, const View& x, const View& y, int padmode, int act, double* bn_stats = nullptr) {
    ConvArgs a = conv_args(x, h->wfwd(c.w), y, c.k, c.stride, padmode == PAD_REFLECT ? 1 : c.pad, 1, padmode);
    a.bias = c.bias ? h->P + c.b : nullptr; a.act = act; a.stats = h->testmode ? nullptr : bn_stats;
    a.wino_u = c.uf >= 0 ? h->UF + c.uf : nullptr; a.wino_pieces = wino_resolve_pieces(h->u_pieces_f, a.Cin);
    run_igemm(h, a);
}
// input gradient: dy (view of the conv output) -> dx view (dims of the conv input; or of the PADDED input for reflect convs)
static ConvArgs dgrad_args(md2_handle* h, const ConvL& c, const View& dy, const View& dx, int pad_fwd, int accumulate, const View* dact_src, int dact, const View* res = nullptr) {
    ConvArgs a = conv_args(dy, h->wdg(c.w), dx, c.k, 1, c.k - 1 - pad_fwd, c.stride, PAD_ZERO);
    a.accumulate = accumulate; a.acc_frame = h->tgt;
    a.wino_u = c.ud >= 0 ? h->UD + c.ud : nullptr; a.wino_pieces = wino_resolve_pieces(h->u_pieces_d, a.Cin);
    if (dact_src) { a.dact_src = dact_src->p; a.da_bs = dact_src->bs; a.da_ps = dact_src->ps; a.dact = dact; }
    if (res) { a.res = res->p; a.res_bs = res->bs; a.res_ps = res->ps; }
    return a;
}
static void conv_dgrad(md2_handle* h, const ConvL& c, const View& dy, const View& dx, int pad_fwd, int accumulate, const View* dact_src, int dact, const View* res = nullptr) {
    run_igemm(h, dgrad_args(h, c, dy, dx, pad_fwd, accumulate, dact_src, dact, res));
}
// stream a weight gradient is enqueued on: the side stream, forked by event from the chain at this point (its dy is complete)
static hipStream_t wgrad_stream(md2_handle* h) {
    hipStream_t st = h->chain();
    // The pullback's last two weight gradients stay on the chain where the weight-gradient stream ends the step with a backlog -- large batches: its launches are capped at
    // 256 workgroups and run ~40 % longer beside the chain, so at batch 32 the stream drains alone, half-wide, after the chain's last kernel.  Measured on one box
    // (tools/env_probe.sh, profiles/r06_wgrad_tail_probe.txt): batch 32 10.86 -> 10.71 ms with 2 (3: 10.72, 4: 10.76, 8: 10.88); batch 4 2.98 -> 3.01 and batch 16
    // 6.14 -> 6.17 (no backlog there: the launches only lose their overlap); ResNet-50 640x192 and 1024x320 unchanged.  Hence by size: from 1.5 M target pixels per step.
    // (the count of launches is the previous pullback's; MD2_WGRAD_TAIL_CHAIN / MD2_WGRAD_HEAD_CHAIN: tuning builds)
    static const int tail_env = md2_tune("MD2_WGRAD_TAIL_CHAIN", -1), head_on_chain = md2_tune("MD2_WGRAD_HEAD_CHAIN", 0);
    const int tail_on_chain = tail_env >= 0 ? tail_env : ((long)h->N * h->H * h->W >= 1500000L ? 2 : 0);
    const int idx = h->wg_idx++;
    if (tail_on_chain > 0 && h->wg_total > 0 && idx >= h->wg_total - tail_on_chain) return st;
    if (idx < head_on_chain) return st;
    if (h->two_streams && h->s2) {          // fork: everything enqueued on this chain's stream so far happens-before this weight gradient
        hipEvent_t ev = h->ev_pool[h->ev_next]; h->ev_next = (h->ev_next + 1) % 64;
        hipStream_t tgt = (h->n_wg_streams > 1 && (h->wg_rr++ & 1)) ? h->s4 : h->s2;      // round-robin over the weight-gradient streams
        (void)hipEventRecord(ev, h->chain()); (void)hipStreamWaitEvent(tgt, ev, 0);
        st = tgt; h->forked = true;
    }
    return st;
}
static void conv_wgrad(md2_handle* h, const ConvL& c, const View& x, const View& dy, int padmode, const BNL* xbn = nullptr) {
    WgradArgs a{};
    if (xbn) { a.in_scale = xbn->stats + 2 * xbn->c; a.in_shift = xbn->stats + 3 * xbn->c; }      // x is the raw conv output of the layer before: relu(bn(x)) on the fly
    a.x = x.p; a.x_bs = x.bs; a.x_ps = x.ps; a.H = x.H; a.W = x.W; a.Cin = x.C;
    a.dy = dy.p; a.dy_bs = dy.bs; a.dy_ps = dy.ps; a.Ho = dy.H; a.Wo = dy.W; a.Cout = dy.C;
    a.dw = h->G + c.w; a.N = dy.N; a.KH = c.kind == 1 ? 1 : c.k; a.KW = a.KH; a.stride = c.kind == 1 ? 1 : c.stride; a.pad = c.kind == 1 ? 0 : (padmode == PAD_REFLECT ? 1 : c.pad); a.padmode = padmode;
    a.M = dy.N * dy.H * dy.W; a.db = c.bias ? h->G + c.b : nullptr; a.bf16 = h->bf16; a.pieces_mode1 = h->wg_pieces + 1;
    char d[112] = "";
    if (h->prof) snprintf(d, sizeof(d), "wgrad M=%d Cout=%d Cin=%d k=%d s=%d pm=%d alg=%s pieces=%d", a.M, a.Cout, a.Cin, a.KH, a.stride, a.padmode, conv_wino_wgrad_eligible(a) ? "winograd" : "direct",
                          conv_wino_wgrad_eligible(a) ? wgrad_resolve_pieces_wino(h->wg_pieces, a.M, a.Cout, a.Cin) : wgrad_resolve_pieces_tap(h->wg_pieces));
    hipStream_t st = wgrad_stream(h);
    a.overlapped = st != h->chain();
    h->pbegin(CLS_WGRAD, 2.0 * a.M * a.Cout * a.KH * a.KW * a.Cin, d, st, true); launch_conv_wgrad(a, st); h->pend(st, true);
    if (h->prof) {
        const double fl = 2.0 * a.M * a.Cout * a.KH * a.KW * a.Cin;
        h->prof_flops[CLS_WGRAD] += fl;
        h->prof_exec[CLS_WGRAD] += conv_wino_wgrad_eligible(a) ? 2.0 * 16.0 * (double)a.N * ((a.Ho + 1) / 2) * ((a.Wo + 1) / 2) * a.Cout * a.Cin : fl;      // Winograd route: 16 instead of 36 multiplies per tile
    }
}
// the batch statistics were accumulated into b.acc by the convolution that produced z (conv_fwd(..., b.acc))
// (test mode, Flux.testmode!: the tracked running statistics instead; train mode also updates them, momentum 0.1)
// b.stats = mean | invstd | scale | shift: the last two let a consumer apply y = z * scale + shift in its loader instead of reading a materialised y
static void bn_stats(md2_handle* h, const BNL& b, const View& z) {
    if (h->testmode) launch_bn_stats_from_running(h->RS + b.rs, b.stats, b.c, h->chain(), h->P + b.g, h->P + b.b);
    else launch_bn_finalize(b.acc, b.stats, b.c, (double)z.pixels(), BN_STAT_SLOTS, h->RS + b.rs, 0.1f, h->chain(), h->P + b.g, h->P + b.b);
}
// the same for a BatchNorm whose first reader is an elementwise kernel: that kernel finalises the statistics in its prologue (BnFin, md2_internal.h) -- no launch here.
// (test mode has nothing to finalise: the caller runs bn_stats as before when this returns false)
static bool bn_fin(md2_handle* h, const BNL& b, const View& z, BnFin& f) {
    static const bool off = md2_tune_set("MD2_NO_BN_FIN");
    if (h->testmode || off) return false;
    f.acc = b.acc; f.count = (double)z.pixels(); f.slots = BN_STAT_SLOTS; f.stats = b.stats; f.running = h->RS + b.rs; f.momentum = 0.1f;
    return true;
}
static void bn_fwd(md2_handle* h, const BNL& b, const View& z, const View* residual, int act, const View& y) {
    BnFin f; const bool fin = bn_fin(h, b, z, f);
    if (!fin) bn_stats(h, b, z);
    launch_bn_apply(z, b.stats, h->P + b.g, h->P + b.b, residual, act, y, h->chain(), nullptr, nullptr, nullptr, fin ? &f : nullptr);
}
// dy: gradient wrt the BN output (already masked by any following ReLU) -> dz.  have_sums: the producer of dy already accumulated sum dy / sum dy*xhat into b.acc_b
static void bn_bwd(md2_handle* h, const BNL& b, const View& dy, const View& z, const View& dz, bool have_sums = false) {
    if (!have_sums) launch_bn_bwd_reduce(dy, nullptr, ACT_NONE, z, b.stats, b.acc_b, h->chain());
    launch_bn_bwd_apply(dy, nullptr, ACT_NONE, z, b.stats, h->P + b.g, b.acc_b, (double)z.pixels(), dz, h->G + b.g, h->G + b.b, h->chain(), BN_STAT_SLOTS, b.stats + 4 * b.c);
}
// ask the launch that produces gradient `g` to also accumulate the BatchNorm-pullback sums of BatchNorm b (input z) -- and of a second one that consumes the same g
static void want_bn_sums(ConvArgs& a, const BNL& b, const View& z, const BNL* b2 = nullptr, const View* z2 = nullptr) {
    a.bst = b.acc_b; a.bz = z.p; a.bz_bs = z.bs; a.bz_ps = z.ps; a.bstat = b.stats;
    if (b2) { a.bstB = b2->acc_b; a.bzB = z2->p; a.bzB_bs = z2->bs; a.bzB_ps = z2->ps; a.bstatB = b2->stats; }
}

static LossArgs loss_args(md2_handle* h, const float* x, const float* const disp[MD2_MAX_SCALES], const float* Rt, const float* auto_loss) {
    LossArgs a{};
    a.x = x; a.N = h->N; a.C = h->C; a.H = h->H; a.W = h->W; a.nscales = h->ns;
    for (int i = 0; i < h->ns; ++i) { a.disp[i] = disp[i]; a.dh[i] = h->H >> (5 - h->lv[i]); a.dw[i] = h->W >> (5 - h->lv[i]); a.smooth_w[i] = (float)h->cfg.disparity_smoothness * (float)h->scl[i]; }
    a.Rt = Rt; a.auto_loss = h->cfg.automasking ? auto_loss : nullptr;
    std::memcpy(a.K, h->K, sizeof(a.K)); std::memcpy(a.invK, h->invK, sizeof(a.invK));
    a.min_disp = (float)(1.0 / h->cfg.max_depth); a.max_disp = (float)(1.0 / h->cfg.min_depth);
    a.tgt = h->tgt; a.src0 = h->src[0]; a.src1 = h->src[1]; a.normalize = 1;
    return a;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// the coarse levels' disparity gradients are scatter-accumulated (float atomics) unless the reductions are ordered; those levels come first in the block, a
// full-resolution last level is written, not accumulated -- the bytes a scattering loss pass needs zeroed
static bool loss_scatters() { static const bool no_scatter = md2_tune("MD2_LOSS_NOSCATTER", 0) != 0; return !g_deterministic && !no_scatter; }
static size_t ddisp_lo_zero_bytes(const md2_handle* h) {
    const bool last_full = (h->H >> (5 - h->lv[h->ns - 1])) == h->H && (h->W >> (5 - h->lv[h->ns - 1])) == h->W;
    return last_full ? (size_t)((const char*)h->ddisp_lo[h->ns - 1] - (const char*)h->ddisp_lo[0]) : h->ddisp_lo_count * sizeof(float);
}
// Temporarily re-batch every activation view: n_enc encoder images, n_dec decoder samples (<= the sizes the arena was planned for).
static void set_batch(md2_handle* h, int n_enc, int n_dec) {
    h->z0.N = h->a0.N = h->p0.N = n_enc;
    for (int i = 0; i < h->nblk; ++i) { Block& B = h->blk[i]; B.in.N = B.z1.N = B.a1.N = B.z2.N = B.out.N = n_enc; if (B.bottleneck) B.a2.N = B.z3.N = n_enc; if (B.down) B.zd.N = B.sd.N = n_enc; }
    for (int i = 0; i < 5; ++i) { h->feat[i].N = n_enc; h->y1[i].N = h->cat[i].N = h->y2[i].N = n_dec; }
    for (int i = 0; i < h->ns; ++i) h->dispv[i].N = n_dec;
}

// pose decoder, both sources batched (src/model.jl:22-35, src/pose_decoder.jl:23-32); enqueued on h->chain()
static void pose_forward(md2_handle* h, int N) {
    // the squeezer (1x1 conv + ReLU) runs once over all 3N frames -- frame 2's features feed both pairs, the reference squeezes them twice (src/pose_decoder.jl:23-25) --
    // then the two temporally ordered pairs are gathered: (f1, f2) for source 1, (f2, f3) for source 3 (src/model.jl:30-35)
    View sq3 = h->sq3; sq3.N = 3 * N;
    conv_fwd(h, h->squeeze, h->feat[4], sq3, PAD_ZERO, ACT_RELU);
    launch_pose_pairs_fwd(h->sq3.p, h->sqcat.p, N, h->sq3.H * h->sq3.W, h->pose_fr, h->chain(), h->bs);
    conv_fwd(h, h->pose[0], h->sqcat, h->py1, PAD_ZERO, ACT_RELU);
    conv_fwd(h, h->pose[1], h->py1, h->py2, PAD_ZERO, ACT_RELU);
    conv_fwd(h, h->pose[2], h->py2, h->py3, PAD_ZERO, ACT_NONE);
    launch_pose_mean_fwd(h->py3, h->rvec, h->tvec, h->chain());
}

// L frames per sample (3 for training, 1 for eval_disparity); tgt = index of the frame the depth decoder sees
// Winograd operands of the forward convolutions from the current parameters (one launch; ~16/9 of the 3x3 weights are written)
// bf16 tensors (precision 2, round 6): U as ONE bf16 piece of G g G^T of the fp32 master weights (layout code 4, conv_wino3b_kernel); precision 1 keeps the direct kernels
static void refresh_wino_fwd(md2_handle* h, hipStream_t s) {
    h->u_pieces_f = h->bf16 == 2 ? 4 : g_wino_pieces;
    if (h->bf16 == 1 || (h->bf16 == 2 && !g_wino_bf16)) return;
    launch_wino_weights_batched(h->P, h->UF, h->udesc_f, h->n_udf, h->ublk_f, h->u_pieces_f, s);
}
static int model_forward(md2_handle* h, const float* x, int L = 3, int tgt = -1, bool with_pose = true, bool with_depth = true, bool wino_on_side = false) {
    hipStream_t s = h->stream;
    if (tgt < 0) tgt = h->tgt;                  // triplets: the configured target frame
    const int NL = h->z0.N, N = NL / L, H = h->H, W = h->W;
    if (!wino_on_side) refresh_wino_fwd(h, s);          // (md2_train_loss runs it on the side stream, overlapping the stem)
    if (h->bs) launch_cast_bf16(h->P, h->Pb, h->n_int, s);          // this step's bf16 copy of the forward weights (the stem and the heads read the fp32 master)
    // every double accumulator of the step in ONE memset: forward statistics | pullback sums | loss partials | pose-gradient sums are adjacent (plan_buffers).  A training
    // step says so up front (acc_zeroed), and its loss / pullback-preparation phases skip their own memsets.
    HIPCHK(hipMemsetAsync(h->acc_all, 0, (h->acc_zeroed ? h->acc_count : h->acc_fwd_count) * sizeof(double), s));
    // ---- encoder: stem + BN + ReLU, maxpool, 8 basic blocks
    {   // direct 7x7 stride-2 convolution on the boundary tensor (+ its BatchNorm statistics)
        const double fl = 2.0 * NL * (H / 2) * (W / 2) * 64.0 * 49.0 * h->C;
        h->pbegin(CLS_IGEMM, fl, "stem fwd 7x7 s2 (direct)"); launch_stem_fwd(x, NL, h->C, H, W, h->P + h->stem.w, h->Kpad, h->z0, (h->testmode || g_deterministic) ? nullptr : h->stem_bn.acc, s, h->bs); h->pend();          // (bf16 products in the stem: the bf16-tensor mode only -- precision 1 keeps its round-3 definition)
        if (g_deterministic && !h->testmode) launch_colstats(h->z0, h->stem_bn.acc, s);
        if (h->prof) { h->prof_flops[CLS_IGEMM] += fl; h->prof_exec[CLS_IGEMM] += fl; }
    }
    // stem tail in one pass: BatchNorm + ReLU + max-pool; a0 is written only where the decoder's skip connection reads it (the target frames)
    {   BnFin f; const bool fin = bn_fin(h, h->stem_bn, h->z0, f);          // (the statistics are finalised by the pass that applies them)
        if (!fin) bn_stats(h, h->stem_bn, h->z0);
        launch_bn_relu_maxpool(h->z0, h->stem_bn.stats, h->P + h->stem_bn.g, h->P + h->stem_bn.b, h->a0, L, tgt, h->p0, h->pool_arg, s, fin ? &f : nullptr); }
    if (wino_on_side) (void)hipStreamWaitEvent(s, h->ev_uf, 0);          // the first Winograd convolution is block 0's conv1
    for (int i = 0; i < h->nblk; ++i) {
        Block& B = h->blk[i];
        if (B.fuse) {
            // the 1x1 shortcut convolution of a down-sampling block reads only the block input: beside the main path, on the pose stream
            const hipEvent_t ev_fork = (B.down && side_branch_ok(h, false)) ? side_fork(h) : nullptr;
            conv_fwd(h, B.c1, B.in, B.z1, PAD_ZERO, ACT_NONE, B.b1.acc); bn_stats(h, B.b1, B.z1);
            {   // conv2 gathers relu(bn1(z1)) straight from z1 -- except on bf16 tensors, where the plain launch (LDS loader, 128 x 64 tiles) plus one elementwise pass into a
                // shared buffer is ahead of the launch with the transform in its register-staged loader (round 5: 55 + 10...35 against 114 us at configs[4])
                View src = B.z1;
                // (the pass applies the SAME fused scale / shift as the loaders -- the weight gradient still transforms z1 in its loader and must see the values this convolution saw)
                if (h->a_fwd && (B.b1.c > 2048 || B.b1.c % 4 != 0)) { fprintf(stderr, "md2: fused block: the identity-statistics table covers 2048 channels in quads (C=%d)\n", B.b1.c); abort(); }
                if (h->a_fwd) { src = h->scratch_view(h->a_fwd, B.z1.N, B.z1.H, B.z1.W, B.z1.C); launch_bn_apply(B.z1, h->bn_ident + 2048 - B.b1.c, B.b1.stats + 2 * B.b1.c, B.b1.stats + 3 * B.b1.c, nullptr, ACT_RELU, src, s); }
                ConvArgs a = conv_args(src, h->wfwd(B.c2.w), B.z2, B.c2.k, B.c2.stride, B.c2.pad, 1, PAD_ZERO);
                if (!h->a_fwd) { a.in_scale = B.b1.stats + 2 * B.b1.c; a.in_shift = B.b1.stats + 3 * B.b1.c; }
                a.stats = h->testmode ? nullptr : B.b2.acc;
                a.wino_u = B.c2.uf >= 0 ? h->UF + B.c2.uf : nullptr; a.wino_pieces = wino_resolve_pieces(h->u_pieces_f, a.Cin);
                run_igemm(h, a);
            }
            if (B.bottleneck) {           // ... and the 1x1 conv3 gathers relu(bn2(z2)) from z2
                bn_stats(h, B.b2, B.z2);
                ConvArgs a = conv_args(B.z2, h->wfwd(B.c3.w), B.z3, B.c3.k, B.c3.stride, B.c3.pad, 1, PAD_ZERO);
                a.in_scale = B.b2.stats + 2 * B.b2.c; a.in_shift = B.b2.stats + 3 * B.b2.c; a.stats = h->testmode ? nullptr : B.b3.acc;
                run_igemm(h, a);
            }
            // the block's last BatchNorm (and the shortcut's) is read first by the elementwise pass below, which finalises the statistics itself
            const BNL& bo = B.bottleneck ? B.b3 : B.b2; const View& zo = B.bottleneck ? B.z3 : B.z2;
            BnFin fo, fd; const bool fin = bn_fin(h, bo, zo, fo);
            if (!fin) bn_stats(h, bo, zo);
            if (B.down) {                 // out = relu(bn(z) + bnd(zd)): the shortcut's BatchNorm output is never written
                if (ev_fork) { side_enter(h, ev_fork); conv_fwd(h, B.cd, B.in, B.zd, PAD_ZERO, ACT_NONE, B.bd.acc); (void)hipStreamWaitEvent(s, side_end(h), 0); }
                else conv_fwd(h, B.cd, B.in, B.zd, PAD_ZERO, ACT_NONE, B.bd.acc);
                if (!bn_fin(h, B.bd, B.zd, fd)) bn_stats(h, B.bd, B.zd);
                launch_bn_apply(zo, bo.stats, h->P + bo.g, h->P + bo.b, &B.zd, ACT_RELU, B.out, s, B.bd.stats, h->P + B.bd.g, h->P + B.bd.b, fin ? &fo : nullptr, fin ? &fd : nullptr);
            } else launch_bn_apply(zo, bo.stats, h->P + bo.g, h->P + bo.b, &B.in, ACT_RELU, B.out, s, nullptr, nullptr, nullptr, fin ? &fo : nullptr);
            continue;
        }
        conv_fwd(h, B.c1, B.in, B.z1, PAD_ZERO, ACT_NONE, B.b1.acc); bn_fwd(h, B.b1, B.z1, nullptr, ACT_RELU, B.a1);
        conv_fwd(h, B.c2, B.a1, B.z2, PAD_ZERO, ACT_NONE, B.b2.acc);
        if (B.bottleneck) { bn_fwd(h, B.b2, B.z2, nullptr, ACT_RELU, B.a2); conv_fwd(h, B.c3, B.a2, B.z3, PAD_ZERO, ACT_NONE, B.b3.acc); }
        if (B.down) {
            conv_fwd(h, B.cd, B.in, B.zd, PAD_ZERO, ACT_NONE, B.bd.acc);
            bn_fwd(h, B.bd, B.zd, nullptr, ACT_NONE, B.sd);                 // raw zd is kept for the pullback
        }
        if (B.bottleneck) bn_fwd(h, B.b3, B.z3, B.down ? &B.sd : &B.in, ACT_RELU, B.out);
        else bn_fwd(h, B.b2, B.z2, B.down ? &B.sd : &B.in, ACT_RELU, B.out);
    }
    // ---- pose decoder on its own stream, concurrently with the depth decoder (both only read the encoder features)
    const bool pose_async = with_pose && h->two_streams && h->s3;
    // (a training step -- wino_on_side marks md2_train_loss -- also composes Rt there, and the pose stream then takes the heads of every branch but the last: each
    //  waits for its branch only, and none of them stands between two convolutions of the decoder chain)
    static const bool heads_aside_on = !md2_tune_set("MD2_NO_HEADS_ASIDE");
    const bool heads_aside = pose_async && with_depth && heads_aside_on;
    if (with_pose) {
        if (pose_async) { (void)hipEventRecord(h->ev_a, h->stream); (void)hipStreamWaitEvent(h->s3, h->ev_a, 0); h->cur = h->s3; }
        // (the scattering loss pass accumulates the coarse levels' disparity gradients: zeroed here, off the caller's stream; ordered before the loss through ev_b)
        if (pose_async && wino_on_side && with_depth && h->ns > 0 && loss_scatters() && ddisp_lo_zero_bytes(h) > 0) { HIPCHK(hipMemsetAsync(h->ddisp_lo[0], 0, ddisp_lo_zero_bytes(h), h->s3)); h->ddisp_zeroed = true; }
        pose_forward(h, N);
        if (pose_async && wino_on_side) { launch_pose_compose(h->rvec, h->tvec, h->Rt, h->N, h->src[0] < h->tgt, h->src[1] < h->tgt, h->s3); h->Rt_ready = true; }
        if (pose_async) { if (!heads_aside) (void)hipEventRecord(h->ev_b, h->s3); h->cur = nullptr; }
    }
    // ---- depth decoder on the target-frame features (src/model.jl:16-17, src/depth_decoder.jl:49-65)
    View x5 = frame_view(h->feat[4], L, tgt);
    for (int b = 0; b < (with_depth ? h->nbr : 0); ++b) {
        const View& in = (b == 0) ? x5 : h->y2[b - 1];
        conv_fwd(h, h->dc1[b], in, h->y1[b], PAD_REFLECT, ACT_ELU);
        // cat(upsample(y1), skip) in one pass over the concat buffer
        if (b < 4) { View sk = frame_view(h->feat[3 - b], L, tgt); launch_upsample2x_fwd(h->y1[b], h->cat[b], s, &sk); }
        else launch_upsample2x_fwd(h->y1[b], chan_slice(h->cat[b], 0, h->y1[b].C), s);
        conv_fwd(h, h->dc2[b], h->cat[b], h->y2[b], PAD_REFLECT, ACT_ELU);
        if (const int o = h->out_of[b]; o >= 0) {
            hipStream_t hs = s;
            if (heads_aside && b < h->nbr - 1) {          // fork: y2[b] is complete on the chain
                hipEvent_t ev = h->ev_pool[h->ev_next]; h->ev_next = (h->ev_next + 1) % 64;
                (void)hipEventRecord(ev, s); (void)hipStreamWaitEvent(h->s3, ev, 0); hs = h->s3;
            }
            launch_head_fwd(h->y2[b], h->P + h->head[o].w, h->P + h->head[o].b, h->dispv[o], hs);
        }
    }
    if (heads_aside) (void)hipEventRecord(h->ev_b, h->s3);
    if (pose_async) (void)hipStreamWaitEvent(h->stream, h->ev_b, 0);       // join: the loss needs the poses (and the disparities of the heads on the pose stream)
    KCHK();
    return 0;
}

static void simple_mode(md2_handle* h, LossArgs& a) {      // slow_depth objective, src/simple_depth.jl:25-42
    a.nscales = 1; a.dh[0] = h->H; a.dw[0] = h->W; a.smooth_w[0] = 1.f; a.normalize = 0; a.auto_loss = nullptr;
}
static int loss_forward(md2_handle* h, const float* x, const float* const disp[MD2_MAX_SCALES], const float* rvec, const float* tvec, const float* auto_loss, float* loss_out, bool simple = false) {
    hipStream_t s = h->stream;
    HIPCHK(hipMemsetAsync(h->loss_part, 0, (size_t)4 * MD2_MAX_SCALES * h->N * sizeof(double), s));
    launch_pose_compose(rvec, tvec, h->Rt, h->N, h->src[0] < h->tgt, h->src[1] < h->tgt, s);       // inverse_transform = source_ids .< target_id = [true, false] (src/training.jl:29)
    LossArgs a = loss_args(h, x, disp, h->Rt, auto_loss);
    if (simple) simple_mode(h, a);
    h->pbegin(CLS_LOSS_FWD); launch_loss_fwd(a, h->loss_part, g_deterministic ? h->loss_blk : nullptr, s); h->pend();
    if (h->prof) h->prof_bytes[CLS_LOSS_FWD] += 4.0 * h->H * h->W * (1 + 3 * h->C) * a.nscales * h->N;
    launch_loss_finalize(a, h->loss_part, h->totals, loss_out ? loss_out : h->loss_dev, s);          // (straight into the caller's device scalar: no 4-byte copy behind it)
    KCHK();
    return 0;
}
// what follows the fused loss kernel and is not read by the depth decoder's pullback: the scalar loss and the pose pullback's first step
static int loss_tail(md2_handle* h, const LossArgs& a, float* loss_out, const float* rvec, const float* tvec, hipStream_t s) {
    launch_loss_finalize(a, h->loss_part, h->totals, loss_out ? loss_out : h->loss_dev, s);          // (straight into the caller's device scalar: no 4-byte copy behind it)
    launch_pose_compose_bwd(rvec, tvec, h->dRt_acc, h->drvec, h->dtvec, h->N, h->src[0] < h->tgt, h->src[1] < h->tgt, s);
    return 0;
}
// loss value and its pullback in one pass (training): fills the loss, ddisp_lo[4], drvec, dtvec
static int loss_train(md2_handle* h, const float* x, const float* const disp[MD2_MAX_SCALES], const float* rvec, const float* tvec, const float* auto_loss, float* loss_out, bool simple = false) {
    hipStream_t s = h->stream;
    if (!h->acc_zeroed) HIPCHK(hipMemsetAsync(h->loss_part, 0, (size_t)(4 * MD2_MAX_SCALES + 24) * h->N * sizeof(double), s));           // loss_part [20N] and dRt_acc [24N] are adjacent
    h->acc_zeroed = false;
    if (!(h->Rt_ready && rvec == h->rvec && tvec == h->tvec)) launch_pose_compose(rvec, tvec, h->Rt, h->N, h->src[0] < h->tgt, h->src[1] < h->tgt, s);
    h->Rt_ready = false;
    LossArgs a = loss_args(h, x, disp, h->Rt, auto_loss);
    if (simple) simple_mode(h, a);
    const size_t P = (size_t)h->H * h->W;
    // gradient wrt the disparity maps: the full-resolution scale is written directly; the coarse scales receive the adjoint of the bilinear resize from inside the
    // kernel (per-tile partial sums, float atomics) -- or, in deterministic mode, full-resolution maps that the ordered adjoint kernel reduces afterwards
    float *dd[MD2_MAX_SCALES] = {}, *dlo[MD2_MAX_SCALES] = {};
    const bool scatter = !simple && loss_scatters();
    bool any_lo = false;
    for (int i = 0; i < a.nscales; ++i) {
        const bool full = !simple && a.dh[i] == h->H && a.dw[i] == h->W;
        dd[i] = full ? h->ddisp_lo[i] : h->ddisp_full + (size_t)i * h->N * P;
        dlo[i] = (scatter && !full) ? h->ddisp_lo[i] : nullptr; any_lo |= dlo[i] != nullptr;
    }
    // (the coarse levels come first in the block; a full-resolution last level is written, not accumulated)
    if (any_lo && !h->ddisp_zeroed) HIPCHK(hipMemsetAsync(h->ddisp_lo[0], 0, ddisp_lo_zero_bytes(h), s));
    h->ddisp_zeroed = false;
    launch_loss_smooth_sums(a, h->loss_part, g_deterministic ? h->loss_blk : nullptr, s);
    h->pbegin(CLS_LOSS_BWD); launch_loss_train(a, h->loss_part, dd, dlo, h->dRt_acc, g_deterministic ? h->loss_blk : nullptr, s); h->pend();
    if (h->prof) h->prof_bytes[CLS_LOSS_BWD] += 4.0 * h->H * h->W * (2 + 3 * h->C) * a.nscales * h->N;
    if (!simple) for (int i = 0; i < a.nscales; ++i) if (!dlo[i] && dd[i] != h->ddisp_lo[i]) launch_upsample_planar_bwd(dd[i], h->ddisp_lo[i], h->N, a.dh[i], a.dw[i], h->H, h->W, s);
    if (h->defer_tail && !simple) { h->tail.pending = true; h->tail.a = a; h->tail.loss_out = loss_out; h->tail.rvec = rvec; h->tail.tvec = tvec; }      // model_backward runs it on the pose stream
    else { h->tail.pending = false; if (int r = loss_tail(h, a, loss_out, rvec, tvec, s)) return r; }
    h->defer_tail = false;
    KCHK();
    return 0;
}

// pose decoder pullback on h->chain(); part 0: mean/pose3/pose2/pose1 (independent of the depth decoder), part 1: the squeezer
// (writes Gfeat[4]; frame 1 accumulates onto the depth decoder's contribution)
static void pose_backward(md2_handle* h, int N, int part) {
    if (part == 0) {
        launch_pose_mean_bwd(h->drvec, h->dtvec, h->dpy3, h->chain());
        conv_wgrad(h, h->pose[2], h->py2, h->dpy3, PAD_ZERO);
        conv_dgrad(h, h->pose[2], h->dpy3, h->dpy2, 0, 0, &h->py2, ACT_RELU);
        conv_wgrad(h, h->pose[1], h->py1, h->dpy2, PAD_ZERO);
        conv_dgrad(h, h->pose[1], h->dpy2, h->dpy1, 1, 0, &h->py1, ACT_RELU);
        conv_wgrad(h, h->pose[0], h->sqcat, h->dpy1, PAD_ZERO);
        conv_dgrad(h, h->pose[0], h->dpy1, h->dsq, 1, 0, &h->sqcat, ACT_RELU);
        return;
    }
    // squeezer: scatter-add the pair gradients back onto the 3N frames (frame 2 collects both of its uses), then ONE weight gradient and ONE input gradient.
    // Gfeat[4]: frames 1 and 3 get exactly this contribution (overwrite), frame 2 adds to the depth decoder's (accumulate mode 2)
    launch_pose_pairs_bwd(h->dsq.p, h->dsq3.p, N, h->dsq3.H * h->dsq3.W, h->pose_fr, h->chain(), h->bs);
    conv_wgrad(h, h->squeeze, h->feat[4], h->dsq3, PAD_ZERO);
    conv_dgrad(h, h->squeeze, h->dsq3, h->Gfeat[4], 0, 2, &h->feat[4], ACT_RELU);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward of the model (Zygote pullback of src/model.jl:8-20), accumulating into h->G
// ---------------------------------------------------------------------------------------------------------------------
// What the pullback needs besides the activations: zeroed parameter-gradient / statistics accumulators and the transposed, flipped
// weight copies for the input-gradient GEMMs (one launch for all layers).  None of it depends on the forward pass, so md2_train_loss
// runs it on the (otherwise idle) second stream while the forward pass computes; without that stream it runs at the head of backward.
static int backward_prepare(md2_handle* h, hipStream_t s) {
    launch_weight_prepare(h->P, h->WD, h->wdesc_dev, h->n_wdesc, h->n_wblocks, h->G, h->n_int, s);          // transposed / flipped weight copies + the zeroed gradient buffer, one launch
    if (h->bs) launch_cast_bf16(h->WD, h->WDb, h->n_int, s);
    if (!h->acc_zeroed) HIPCHK(hipMemsetAsync(h->acc_bwd, 0, h->acc_bwd_count * sizeof(double), s));
    h->u_pieces_d = h->bf16 == 2 ? 4 : g_wino_pieces; h->wg_pieces = g_wgrad_pieces;
    if (h->bf16 == 0 || (h->bf16 == 2 && g_wino_bf16)) launch_wino_weights_batched(h->WD, h->UD, h->udesc_d, h->n_udd, h->ublk_d, h->u_pieces_d, s);          // Winograd operands of the input-gradient convolutions (bf16 tensors: one bf16 piece, from the fp32 transposed copy)
    KCHK(); return 0;
}
// The flat gradient buffer is produced back to front; its four all-reduce buckets in the order they complete: [0] depth decoder + heads + pose decoder, [1] encoder layer 4,
// [2] encoder layers 2-3, [3] stem + encoder layer 1.  They must tile [0, n_int) exactly once for every encoder / scale_levels choice (tests/test_host_and_abi.py checks
// md2_comm_bucket_plan on the host for both trunks and every level subset); an empty bucket (ResNet layouts where two boundaries coincide) is skipped by comm_bucket.
struct BucketPlan { long lo[4], hi[4]; int first_l4, first_l2; };
static BucketPlan bucket_plan(const md2_handle* h) {
    int stage_first[4] = {0, 0, 0, 0};                                 // first block of encoder layers 1..4
    for (int i = 0, l = 0; i < h->nblk && l < 4; ++i) if (i == 0 || h->blk[i - 1].stage_end) stage_first[l++] = i;
    BucketPlan p;
    p.first_l4 = stage_first[3]; p.first_l2 = stage_first[1];
    p.lo[0] = h->dc1[0].w;                    p.hi[0] = h->n_int;
    p.lo[1] = h->blk[stage_first[3]].c1.w;    p.hi[1] = h->dc1[0].w;
    p.lo[2] = h->blk[stage_first[1]].c1.w;    p.hi[2] = h->blk[stage_first[3]].c1.w;
    p.lo[3] = 0;                              p.hi[3] = h->blk[stage_first[1]].c1.w;
    return p;
}
static int model_backward(md2_handle* h) {
    hipStream_t s = h->stream;
    const int N = h->N, NL = h->NL, H = h->H, W = h->W;
    if (h->prep_done) { (void)hipStreamWaitEvent(s, h->ev_p1, 0); h->prep_done = false; }     // zeroed gradients + transposed weights were prepared during the forward pass
    else if (int r = backward_prepare(h, s)) return r;
    // data parallel: the gradient buffer is all-reduced in up to 4 buckets as they complete (decoders | stage 5 | stages 3-4 | stage 2 + stem), MD2_COMM_BUCKETS=0: one
    // all-reduce by md2_allreduce_grads after the pullback
    static const bool buckets_on = md2_tune("MD2_COMM_BUCKETS", 1) != 0;
    h->grads_reduced = false; h->n_bucket_done = (h->comm && buckets_on) ? 0 : -1; h->wg_idx = 0;
    const BucketPlan bp = bucket_plan(h);
    // The gradient accumulators (Gfeat / Gblk / Gpool, 715 MB at B=32) are NOT zeroed: every buffer's first writer overwrites
    // (accumulate 0) and later writers add.  Stage buffers receive decoder / pose contributions in the target frame before the
    // encoder pullback reaches them, so the encoder's first writer uses accumulate mode 2 (add only in target frames).

    // ---- pose decoder pullback on its own stream, overlapping the depth decoder pullback.  Its last four launches accumulate into
    //      Gfeat[4] (frame 1 on top of the depth decoder's first write): they follow the depth decoder on the caller's stream — see pose_backward.
    const bool pose_async = h->two_streams && h->s3;
    const int nbr = h->nbr;
    // The heads' pullbacks are independent of each other and latency-bound (their epilogues are weight-gradient atomics).  The decoder chain below starts from the LAST
    // branch and meets the other heads' gradients only when it folds into gA[b - 1]: with the pose stream available the other heads run there, beside the chain's first
    // convolutions (ev_heads orders the first fold after them); alone on the stream they run back to back as before.
    static const bool heads_aside_on = !md2_tune_set("MD2_NO_HEADS_ASIDE");
    const bool heads_aside = pose_async && heads_aside_on && h->ns > 1;
    hipEvent_t ev_heads = nullptr;
    auto head_bwd = [&](int l, hipStream_t st) {
        const int b = h->lv[l] - 1;
        View dyv = make_view(h->ddisp_lo[l], N, h->dispv[l].H, h->dispv[l].W, 1);
        // last branch: only its head consumes it, so the ELU derivative of its y2 is applied right there
        launch_head_bwd(h->y2[b], h->P + h->head[l].w, h->dispv[l], dyv, h->gA[b], 0, b == nbr - 1 ? ACT_ELU : ACT_NONE, h->G + h->head[l].w, h->G + h->head[l].b, st);
    };
    if (pose_async) {
        (void)hipEventRecord(h->ev_a, h->stream);
        if (heads_aside) head_bwd(h->ns - 1, s);          // (host order: the chain's own head before the pose stream's ~30 enqueues; the pose stream does not wait for it)
        (void)hipStreamWaitEvent(h->s3, h->ev_a, 0);
        if (h->tail.pending) { h->tail.pending = false; if (int r = loss_tail(h, h->tail.a, h->tail.loss_out, h->tail.rvec, h->tail.tvec, h->s3)) return r; }
        if (heads_aside) {
            for (int l = h->ns - 2; l >= 0; --l) head_bwd(l, h->s3);          // (finest first: the order the chain needs them in)
            ev_heads = h->ev_pool[h->ev_next]; h->ev_next = (h->ev_next + 1) % 64; (void)hipEventRecord(ev_heads, h->s3);
        }
        h->cur = h->s3; pose_backward(h, N, 0); h->cur = nullptr;
    }
    // ---- depth decoder
    if (!heads_aside) for (int l = 0; l < h->ns; ++l) head_bwd(l, s);
    for (int b = nbr - 1; b >= 0; --b) {
        // gA[b] now holds d loss / d (pre-ELU output of c2[b])
        const View& dZ2 = h->gA[b];
        conv_wgrad(h, h->dc2[b], h->cat[b], dZ2, PAD_REFLECT);
        View dcp = h->scratch_view(h->dxpad, N, h->cat[b].H + 2, h->cat[b].W + 2, h->cat[b].C);
        conv_dgrad(h, h->dc2[b], dZ2, dcp, 0, 0, nullptr, ACT_NONE);
        const int dc = h->y1[b].C;
        const bool gather_up = h->cat[b].H >= 4 && h->cat[b].W >= 4;
        static const bool fold_pair = !md2_tune_set("MD2_NO_FOLD_PAIR");
        if (b < 4) {
            View fv = frame_view(h->feat[3 - b], 3, h->tgt), gv = frame_view(h->Gfeat[3 - b], 3, h->tgt);
            // first writer of this stage's target frames -- in one launch with the upsampled half's adjoint where that is the gather pass
            if (gather_up && fold_pair) launch_fold_cat_bwd(chan_slice(dcp, 0, dc), h->dZ1[b], &h->y1[b], ACT_ELU, chan_slice(dcp, dc, fv.C), gv, 0, &fv, ACT_RELU, s);
            else launch_fold_reflect(chan_slice(dcp, dc, fv.C), gv, 0, &fv, ACT_RELU, s);
        }
        // the upsampled half of the concat: reflect-pad adjoint and the x2 adjoint in one gather pass (deterministic either way: no atomics); maps narrower than 4 keep the two passes
        if (gather_up) { if (!(b < 4 && fold_pair)) launch_fold_up2_bwd(chan_slice(dcp, 0, dc), h->dZ1[b], &h->y1[b], ACT_ELU, s); }
        else { launch_fold_reflect(chan_slice(dcp, 0, dc), h->dupv[b], 0, nullptr, ACT_NONE, s); launch_upsample2x_bwd(h->dupv[b], h->dZ1[b], &h->y1[b], ACT_ELU, s); }
        const View& in = (b == 0) ? frame_view(h->feat[4], 3, h->tgt) : h->y2[b - 1];
        conv_wgrad(h, h->dc1[b], in, h->dZ1[b], PAD_REFLECT);
        View dip = h->scratch_view(h->dxpad, N, in.H + 2, in.W + 2, in.C);
        conv_dgrad(h, h->dc1[b], h->dZ1[b], dip, 0, 0, nullptr, ACT_NONE);
        if (b == 0) { View fv = frame_view(h->feat[4], 3, h->tgt), gv = frame_view(h->Gfeat[4], 3, h->tgt); launch_fold_reflect(dip, gv, 0, &fv, ACT_RELU, s); }   // first writer of Gfeat[4] frame 1
        else {
            if (ev_heads && h->out_of[b - 1] >= 0) { (void)hipStreamWaitEvent(s, ev_heads, 0); ev_heads = nullptr; }    // the heads on the pose stream have written their gA
            launch_fold_reflect(dip, h->gA[b - 1], h->out_of[b - 1] >= 0 ? 1 : 0, &h->y2[b - 1], ACT_ELU, s);        // (adds to the head's contribution where branch b-1 has one)
        }
    }
    // ---- pose branch, second half: accumulate into Gfeat[4] after the depth decoder's write; then join (the encoder pullback reads Gfeat[4])
    // (on the caller's stream: the encoder pullback needs it at once, and a hop to the pose stream and back only adds two event latencies to the chain -- batch 4:
    //  48 us between the decoder's last fold and the encoder's first kernel, profiles/r06_b4_two_stream_trace.txt; the pose stream's first half finished long before)
    if (pose_async) {
        (void)hipEventRecord(h->ev_c, h->s3); (void)hipStreamWaitEvent(h->stream, h->ev_c, 0);
        pose_backward(h, N, 1);
    } else { pose_backward(h, N, 0); pose_backward(h, N, 1); }
    if (int r = comm_bucket(h, bp.lo[0], bp.hi[0])) return r;           // depth decoder, heads, pose decoder
    // ---- encoder
    // BatchNorm pullbacks need sum g and sum g*xhat over the whole batch before the first dz can be formed.  Where the kernels allow it these sums are accumulated by
    // the epilogue of the input-gradient launch that PRODUCES g (its last writer), so the separate reduction pass over (g, z) disappears; `sums_ready[i]` says whether
    // block i's output-side BatchNorms (bn2 and the shortcut's) got theirs that way.
    bool sums_ready[16]; for (int i = 0; i < 16; ++i) sums_ready[i] = false;
    // stage_below[i]: index of the stage feature that is block i's input when block i-1 ends a stage; stage_used[j]: the decoder wrote Gfeat[j] (skip of branch 3-j, head feature 4)
    int stage_below[16]; { int st = 1; for (int i = 0; i < h->nblk; ++i) { stage_below[i] = st - 1; if (h->blk[i].stage_end) ++st; } }
    bool stage_used[5]; for (int j = 0; j < 5; ++j) stage_used[j] = (j == 4) || (3 - j) < nbr;
    for (int i = h->nblk - 1; i >= 0; --i) {
        Block& B = h->blk[i];
        const View& Gout = h->Gblk[i];                                  // d / d(pre-ReLU block output), already masked
        const View* Gin = (i == 0) ? &h->Gpool : &h->Gblk[i - 1];
        const View* in_act = (i == 0) ? nullptr : &B.in;                 // ReLU mask of the block input (stage 0: applied in the maxpool pullback)
        const View& dz1 = B.dz1; View da1 = h->scratch_view(h->E2, NL, B.a1.H, B.a1.W, B.a1.C);
        // first encoder writer of Gin: own buffers (and the pool gradient) are overwritten; stage buffers already hold the decoder's contribution in the target frames
        // (a stage whose skip connection the decoder does not reach -- scale_levels ending below 5 -- has no such contribution: plain overwrite)
        const int first_mode = (i == 0 || !h->blk[i - 1].stage_end || !stage_used[stage_below[i]]) ? 0 : 2;
        // the launch that completes Gin also reduces the BatchNorm-pullback sums of the block below (bn2 / bn3 and, for a down-sampling block, the shortcut's BatchNorm)
        Block* Bp = (i > 0 && h->blk[i - 1].fuse) ? &h->blk[i - 1] : nullptr;
        if (B.fuse) {
            // the shortcut's pullback (BatchNorm, weight gradient, input gradient into its quarter of Gin) depends on Gout alone: beside the main path, on the pose
            // stream; the main path's last launch -- conv1's input gradient, which completes Gin on top of it -- waits for it
            const hipEvent_t ev_fork = (B.down && side_branch_ok(h, true)) ? side_fork(h) : nullptr;
            auto shortcut_bwd = [&]() {
                bn_bwd(h, B.bd, Gout, B.zd, B.dzd, sums_ready[i]);
                conv_wgrad(h, B.cd, B.in, B.dzd, PAD_ZERO);
                ConvArgs ad = dgrad_args(h, B.cd, B.dzd, *Gin, 0, first_mode, in_act, ACT_RELU); ad.skip_empty = 1;
                run_igemm(h, ad);
            };
            if (B.bottleneck) {
                View da2 = h->scratch_view(h->E3, NL, B.a2.H, B.a2.W, B.a2.C);
                bn_bwd(h, B.b3, Gout, B.z3, B.dz3, sums_ready[i]);
                conv_wgrad(h, B.c3, B.z2, B.dz3, PAD_ZERO, &B.b2);           // x = relu(bn2(z2)) formed in the loader
                ConvArgs a = dgrad_args(h, B.c3, B.dz3, da2, 0, 0, &B.z2, ACT_RELU);       // da2 = (W3^T dz3) * relu'(bn2(z2)) + the sums of bn2's pullback
                a.dact_scale = B.b2.stats + 2 * B.b2.c; a.dact_shift = B.b2.stats + 3 * B.b2.c;
                want_bn_sums(a, B.b2, B.z2);
                run_igemm(h, a);
                bn_bwd(h, B.b2, da2, B.z2, B.dz2, true);
            } else
            bn_bwd(h, B.b2, Gout, B.z2, B.dz2, sums_ready[i]);
            conv_wgrad(h, B.c2, B.z1, B.dz2, PAD_ZERO, &B.b1);               // x = relu(bn1(z1)) formed in the loader
            {   // da1 = (W2^T dz2) * relu'(bn1(z1)), and the sums of bn1's pullback from the same epilogue
                ConvArgs a = dgrad_args(h, B.c2, B.dz2, da1, 1, 0, &B.z1, ACT_RELU);
                a.dact_scale = B.b1.stats + 2 * B.b1.c; a.dact_shift = B.b1.stats + 3 * B.b1.c;
                want_bn_sums(a, B.b1, B.z1);
                run_igemm(h, a);
            }
            bn_bwd(h, B.b1, da1, B.z1, dz1, true);
            conv_wgrad(h, B.c1, B.in, dz1, PAD_ZERO);
            if (B.down) {           // the 1x1 / stride-2 shortcut writes its quarter of the pixels FIRST; the 3x3's residue classes then complete every pixel exactly once
                if (ev_fork) { side_enter(h, ev_fork); shortcut_bwd(); (void)hipStreamWaitEvent(s, side_end(h), 0); } else shortcut_bwd();
            }
            ConvArgs a = dgrad_args(h, B.c1, dz1, *Gin, B.c1.pad, first_mode, in_act, ACT_RELU, B.down ? nullptr : &Gout);
            if (B.down) { if (B.bottleneck) a.acc_lat = B.cd.stride; else a.accumulate_c00 = 1; }      // (the bottleneck's conv1 is a dense 1x1: it adds on the shortcut's pixel lattice)
            if (Bp) { want_bn_sums(a, Bp->bottleneck ? Bp->b3 : Bp->b2, Bp->bottleneck ? Bp->z3 : Bp->z2, Bp->down ? &Bp->bd : nullptr, Bp->down ? &Bp->zd : nullptr); sums_ready[i - 1] = true; }
            run_igemm(h, a);
        } else {
        if (!B.bottleneck) {
            bn_bwd(h, B.b2, Gout, B.z2, B.dz2);
            conv_wgrad(h, B.c2, B.a1, B.dz2, PAD_ZERO);
            conv_dgrad(h, B.c2, B.dz2, da1, 1, 0, &B.a1, ACT_RELU);
        } else {
            View da2 = h->scratch_view(h->E3, NL, B.a2.H, B.a2.W, B.a2.C);
            bn_bwd(h, B.b3, Gout, B.z3, B.dz3);
            conv_wgrad(h, B.c3, B.a2, B.dz3, PAD_ZERO);
            conv_dgrad(h, B.c3, B.dz3, da2, 0, 0, &B.a2, ACT_RELU);
            bn_bwd(h, B.b2, da2, B.z2, B.dz2);
            conv_wgrad(h, B.c2, B.a1, B.dz2, PAD_ZERO);
            conv_dgrad(h, B.c2, B.dz2, da1, 1, 0, &B.a1, ACT_RELU);          // carries the block's stride
        }
        bn_bwd(h, B.b1, da1, B.z1, dz1);
        conv_wgrad(h, B.c1, B.in, dz1, PAD_ZERO);
        // identity shortcut: d(block input) = relu'(in) * (conv1 pullback + Gout) -- Gout is added in the same epilogue (was an axpy pass)
        if (B.down) {           // (the last writer of Gin must be a launch that covers every pixel: the shortcut's input gradient goes first when the block below wants its sums)
            const View& dzd = B.dzd;
            bn_bwd(h, B.bd, Gout, B.zd, dzd);
            conv_wgrad(h, B.cd, B.in, dzd, PAD_ZERO);
            if (Bp) { ConvArgs ad = dgrad_args(h, B.cd, dzd, *Gin, 0, first_mode, in_act, ACT_RELU); ad.skip_empty = 1; run_igemm(h, ad); }
        }
        {
            ConvArgs a = dgrad_args(h, B.c1, dz1, *Gin, B.c1.pad, first_mode, in_act, ACT_RELU, B.down ? nullptr : &Gout);
            if (B.down && Bp) a.accumulate_c00 = 1;
            if (Bp) { want_bn_sums(a, Bp->bottleneck ? Bp->b3 : Bp->b2, Bp->bottleneck ? Bp->z3 : Bp->z2, Bp->down ? &Bp->bd : nullptr, Bp->down ? &Bp->zd : nullptr); sums_ready[i - 1] = true; }
            run_igemm(h, a);
        }
        if (B.down && !Bp) conv_dgrad(h, B.cd, B.dzd, *Gin, 0, 1, in_act, ACT_RELU);
        }
        if (i == bp.first_l4) { if (int r = comm_bucket(h, bp.lo[1], bp.hi[1])) return r; }                                  // encoder layer 4
        else if (i == bp.first_l2 && bp.first_l2 != bp.first_l4) { if (int r = comm_bucket(h, bp.lo[2], bp.hi[2])) return r; }   // layers 2-3
    }
    {   // stem tail pullback: the max-pool pullback also reduces the stem BatchNorm's sums, and the stem's weight gradient forms dz = A g + B z + C0 in its loader:
        // no reduction pass, no dz tensor (Gfeat[0]: decoder skip gradient in the target frames only; ReLU mask from the pooled output)
        const BNL& b = h->stem_bn;
        const bool fuse = !g_deterministic && !md2_tune_set("MD2_NO_BN_FUSE");
        const int acc_frame = stage_used[0] ? h->tgt : -1;           // no decoder skip gradient in Gfeat[0] when the fifth branch is absent
        if (fuse) launch_maxpool_bwd(h->Gpool, h->pool_arg, h->Gfeat[0], 2, &h->p0, s, &h->z0, b.stats, b.acc_b, acc_frame);
        else { launch_maxpool_bwd(h->Gpool, h->pool_arg, h->Gfeat[0], 2, &h->p0, s, nullptr, nullptr, nullptr, acc_frame); launch_bn_bwd_reduce(h->Gfeat[0], nullptr, ACT_NONE, h->z0, b.stats, b.acc_b, s); }
        launch_bn_bwd_finalize(b.acc_b, BN_STAT_SLOTS, b.c, (double)h->z0.pixels(), b.stats + 4 * b.c, h->G + b.g, h->G + b.b, b.stats, h->P + b.g, s);
        const double fl = 2.0 * NL * (H / 2) * (W / 2) * 64.0 * 49.0 * h->C;
        // the last launch of the pullback stays on the caller's stream, which has nothing else left: beside the weight-gradient stream's tail instead of behind it, and
        // without the fork / join hop (batch 4: 56 us between bn_bwd_finalize and ADAM for this 35 us kernel)
        static const bool stem_on_chain = !md2_tune_set("MD2_STEM_WGRAD_FORK");
        hipStream_t st = stem_on_chain ? s : wgrad_stream(h);
        h->pbegin(CLS_WGRAD, fl, "stem wgrad 7x7 s2 (direct, BatchNorm pullback in the loader)", st, true);
        launch_stem_wgrad(h->x_last, NL, h->C, H, W, h->Gfeat[0], h->G + h->stem.w, h->Kpad, st, &h->z0, b.stats + 6 * b.c, h->bs);
        h->pend(st, true);
        if (h->prof) { h->prof_flops[CLS_WGRAD] += fl; h->prof_exec[CLS_WGRAD] += fl; }
    }
    if (h->n_bucket_done >= 0) {            // last bucket (layer 1 + stem), then the caller's stream waits for the communication stream
        if (int r = comm_bucket(h, bp.lo[3], bp.hi[3])) return r;
        (void)hipEventRecord(h->ev_cdone, h->scomm); (void)hipStreamWaitEvent(h->stream, h->ev_cdone, 0);
        h->grads_reduced = true;
    }
    h->wg_total = h->wg_idx;
    if (h->forked) {                        // join: the gradient buffer is complete only when the weight-gradient stream has drained
        (void)hipEventRecord(h->ev_join, h->s2); (void)hipStreamWaitEvent(h->stream, h->ev_join, 0); h->forked = false;
        if (h->n_wg_streams > 1) { (void)hipEventRecord(h->ev_join4, h->s4); (void)hipStreamWaitEvent(h->stream, h->ev_join4, 0); }
    }
    KCHK();
    return 0;
}

// =====================================================================================================================
// RCCL (data parallelism): one process per GPU, all-reduce(sum) of the flat gradient buffer over xGMI.
// The library calls RCCL itself (no torch, no host framework in between) so that a Julia host gets the same multi-GPU step; the entry points are
// resolved with dlsym at md2_comm_init -- librccl.so.1 already mapped into the process (e.g. by a host framework) is reused, else it is loaded.
// =====================================================================================================================
namespace {
struct RcclApi {
    void* lib = nullptr; bool tried = false; std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr; decltype(&ncclBroadcast) Broadcast = nullptr; decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr;
};
RcclApi g_rccl;
bool rccl_load() {
    RcclApi& r = g_rccl;
    if (r.tried) return r.lib != nullptr;
    r.tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // already in the process: share it
    if (!r.lib) for (const char* n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.lib) { r.err = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "?"); return false; }
#define MD2_SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f)); if (!r.f) { r.err = "librccl lacks nccl" #f; r.lib = nullptr; return false; }
    MD2_SYM(GetUniqueId) MD2_SYM(CommInitRank) MD2_SYM(CommDestroy) MD2_SYM(AllReduce) MD2_SYM(Broadcast) MD2_SYM(GetErrorString) MD2_SYM(GroupStart) MD2_SYM(GroupEnd)
#undef MD2_SYM
    return true;
}
}  // namespace
#define NCCLCHK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { h->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r_); return 1; } } while (0)

// Gradient bucket [lo, hi) of the flat buffer is complete on the streams that write it: order the communication stream after them and enqueue its all-reduce there,
// so that it runs while the rest of the pullback computes.  (The buckets are walked from the end of the buffer: decoders, then encoder stages 4 .. 1.)
static int comm_bucket(md2_handle* h, long lo, long hi) {
    if (!h->comm || h->n_bucket_done < 0 || hi <= lo) return 0;
    const int i = h->n_bucket_done++;
    if (i >= 8) { h->err = "comm_bucket: too many buckets"; return 1; }
    (void)hipEventRecord(h->ev_cm[i], h->stream); (void)hipStreamWaitEvent(h->scomm, h->ev_cm[i], 0);
    if (h->forked) { (void)hipEventRecord(h->ev_cs[i], h->s2); (void)hipStreamWaitEvent(h->scomm, h->ev_cs[i], 0); }
    if (h->forked && h->n_wg_streams > 1) { (void)hipEventRecord(h->ev_join4, h->s4); (void)hipStreamWaitEvent(h->scomm, h->ev_join4, 0); }
    NCCLCHK(g_rccl.AllReduce(h->G + lo, h->G + lo, (size_t)(hi - lo), ncclFloat, ncclSum, (ncclComm_t)h->comm, h->scomm));
    return 0;
}
static void comm_destroy(md2_handle* h) {
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr; h->world = 1; h->rank = 0;
}

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

const char* md2_last_error(md2_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static void destroy_handle(md2_handle* h);

// argument checks of md2_create: pure host work, before anything touches the device (also the front end of md2_comm_bucket_plan)
static int validate_cfg(const md2_config* cfg) {
    if (!cfg) { g_create_err = "null argument"; return 2; }
    // The network needs height/width multiples of 32 and >= 64 (five stride-2 stages, then reflect padding at 1/32 scale).  Smaller images
    // (multiples of 8, >= 16) give a LOSS-ONLY handle: md2_loss_fwd_bwd / md2_simple_depth_loss / md2_automasking_loss work, the model entry points fail.
    if (cfg->batch < 1 || (cfg->channels != 1 && cfg->channels != 3) || cfg->height % 8 || cfg->width % 8 || cfg->height < 16 || cfg->width < 16) {
        g_create_err = "md2_create: need batch >= 1, channels in {1,3}, height/width multiples of 8 and >= 16 (multiples of 32 and >= 64 for the network)"; return 2;
    }
    if (cfg->encoder_depth != 0 && cfg->encoder_depth != 18 && cfg->encoder_depth != 50) { g_create_err = "md2_create: encoder_depth must be 18 (or 0) or 50"; return 2; }
    if (cfg->precision < 0 || cfg->precision > 2) { g_create_err = "md2_create: precision must be 0 (fp32), 1 (bf16 MFMA operands, fp32 tensors) or 2 (bf16 activation tensors)"; return 2; }
    {   // shape limits of the kernels (32-bit byte offsets inside one activation tensor, 24-bit pixel counts in the weight gradients): refuse here, with a
        // status code, instead of aborting inside a launcher in the middle of a step
        const double px_half = 3.0 * cfg->batch * (cfg->height / 2.0) * (cfg->width / 2.0);          // pixels of the largest activation (stem output, 64 channels)
        const double widest = px_half * 64.0 * 4.0, dec_full = (double)cfg->batch * (cfg->height + 2.0) * (cfg->width + 2.0) * 32.0 * 4.0;
        if (widest >= 4294967296.0 || dec_full >= 4294967296.0 || px_half >= 16777216.0) {
            g_create_err = "md2_create: batch x resolution too large for one handle (an activation tensor must stay below 4 GiB and 2^24 pixels); use a smaller per-GPU batch"; return 2;
        }
    }
    {   // TrainCache.target_id / source_ids and DepthDecoder scale_levels: pure argument checks, before anything touches the device
        const int t = cfg->target_id ? cfg->target_id : 2, s0 = cfg->source_ids[0] ? cfg->source_ids[0] : (cfg->source_ids[1] ? 0 : 1), s1 = cfg->source_ids[1] ? cfg->source_ids[1] : (cfg->source_ids[0] ? 0 : 3);
        if (t < 1 || t > 3 || s0 < 1 || s0 > 3 || s1 < 1 || s1 > 3 || s0 == s1 || s0 == t || s1 == t) { g_create_err = "md2_create: target_id / source_ids must be three distinct frames of the triplet (1..3), two of them sources"; return 2; }
        bool ended = false;
        for (int i = 0; i < 5; ++i) {
            const int l = cfg->scale_levels[i];
            if (l == 0) { ended = true; continue; }
            if (ended || l < 1 || l > 5 || (i && l <= cfg->scale_levels[i - 1])) { g_create_err = "md2_create: `scale_levels` should be at most of length 5, ascending, with values in [1, 5] (0-terminated)"; return 2; }
        }
        // scales: all five 0.0 (a zero-initialised config) = the reference's defaults; otherwise every listed level's value is taken as given -- an explicit 0 switches that
        // level's smoothness term off like in the reference (src/training.jl:67) -- and a negative entry asks for that level's default
        for (int i = 0; i < 5; ++i) if (cfg->scales[i] != cfg->scales[i]) { g_create_err = "md2_create: `scales` must not be NaN (negative = the default 1 / 2^(5 - level))"; return 2; }
    }
    return 0;
}
static void apply_cfg(md2_handle* h, const md2_config* cfg);
int32_t md2_create(const md2_config* cfg, md2_handle** out) {
    if (!out) { g_create_err = "null argument"; return 2; }
    if (int r = validate_cfg(cfg)) return r;
    const bool net_ok = cfg->height % 32 == 0 && cfg->width % 32 == 0 && cfg->height >= 64 && cfg->width >= 64;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "md2_create: no HIP device (libmd2hip needs an MI355X; there is no CPU fallback)"; return 3; }
    if (hipSetDevice(cfg->device) != hipSuccess) { g_create_err = "md2_create: hipSetDevice failed"; return 3; }
    if (const char* e = getenv("MD2_DETERMINISTIC")) g_deterministic = atoi(e) != 0;
    md2_handle* h = new md2_handle();
    apply_cfg(h, cfg); h->net_ok = net_ok;
    for (int i = 0; i < 64; ++i) h->ev_pool[i] = nullptr;
    for (int i = 0; i < 8; ++i) { h->ev_cm[i] = nullptr; h->ev_cs[i] = nullptr; }
    build_layers(h);
    plan_buffers(h);                        // sizing pass
    const size_t need = h->arena_off + 4096;
    auto fail = [&](const char* what) { g_create_err = std::string("md2_create: ") + what; destroy_handle(h); return 4; };
    if (hipMalloc((void**)&h->arena, need) != hipSuccess) return fail("hipMalloc(arena) failed");
    plan_buffers(h);
    const size_t pb = (size_t)h->n_int * sizeof(float);
    if (hipMalloc((void**)&h->P, pb) != hipSuccess || hipMalloc((void**)&h->G, pb) != hipSuccess || hipMalloc((void**)&h->M, pb) != hipSuccess ||
        hipMalloc((void**)&h->V, pb) != hipSuccess || hipMalloc((void**)&h->WD, pb) != hipSuccess) return fail("hipMalloc(params) failed");
    if (h->bs) {          // bf16 copies of the forward / input-gradient weights (+ 64 elements: the paired-tap loader of the 16-channel layers reads one tap past a weight row)
        const size_t hb = ((size_t)h->n_int + 64) * sizeof(uint16_t);
        if (hipMalloc((void**)&h->Pb, hb) != hipSuccess || hipMalloc((void**)&h->WDb, hb) != hipSuccess) return fail("hipMalloc(bf16 weights) failed");
        (void)hipMemset(h->Pb, 0, hb); (void)hipMemset(h->WDb, 0, hb);
    }
    (void)hipMemset(h->P, 0, pb); (void)hipMemset(h->G, 0, pb); (void)hipMemset(h->M, 0, pb); (void)hipMemset(h->V, 0, pb); (void)hipMemset(h->WD, 0, pb);
    (void)hipMemset(h->arena, 0, need);
    if (h->bn_ident) { std::vector<float> id(2 * 2048, 0.f); for (int i = 2048; i < 4096; ++i) id[i] = 1.f; (void)hipMemcpy(h->bn_ident, id.data(), id.size() * sizeof(float), hipMemcpyHostToDevice); }
    {   // BatchNorm running statistics: Flux initialises mu = 0, sigma^2 = 1
        std::vector<float> rs((size_t)h->n_rs, 0.f);
        for (BNL* b : h->bns) for (int i = 0; i < b->c; ++i) rs[b->rs + b->c + i] = 1.f;
        if (hipMalloc((void**)&h->RS, rs.size() * sizeof(float)) != hipSuccess) return fail("hipMalloc(bn state) failed");
        (void)hipMemcpy(h->RS, rs.data(), rs.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    {   // layer table of the batched weight transform
        std::vector<long> d;
        long first = 0;
        for (ConvL* c : h->convs) if (c->kind == 0) { d.push_back(c->w); d.push_back(c->coutP); d.push_back(c->cinP); d.push_back(c->taps); d.push_back(first); first += weight_prepare_blocks(c->coutP, c->cinP, c->taps); }
        h->n_wdesc = (int)d.size() / 5; h->n_wblocks = (int)first;
        if (hipMalloc((void**)&h->wdesc_dev, d.size() * sizeof(long)) != hipSuccess) return fail("hipMalloc(wdesc) failed");
        (void)hipMemcpy(h->wdesc_dev, d.data(), d.size() * sizeof(long), hipMemcpyHostToDevice);
    }
    {   // Winograd operand tables: forward U from P [coutP][9][cinP], input-gradient U from WD [cinP][9][coutP] (the same transform with the channel roles swapped)
        std::vector<long> df, dd; long of = 0, od = 0;
        for (ConvL* c : h->convs) if (c->kind == 0 && c->k == 3 && c->stride == 1) {
            if (c->coutP % 32 == 0 && c->cinP % 16 == 0) { c->uf = of; const long nb = wino_weight_blocks(c->coutP, c->cinP); df.insert(df.end(), {c->w, of, (long)c->coutP, (long)c->cinP, nb}); h->ublk_f += (int)nb; of += (long)conv_wino_u_floats(c->coutP, c->cinP); }
            if (c->cinP % 32 == 0 && c->coutP % 16 == 0) { c->ud = od; const long nb = wino_weight_blocks(c->cinP, c->coutP); dd.insert(dd.end(), {c->w, od, (long)c->cinP, (long)c->coutP, nb}); h->ublk_d += (int)nb; od += (long)conv_wino_u_floats(c->cinP, c->coutP); }
        }
        h->n_uf = of; h->n_ud = od; h->n_udf = (int)df.size() / 5; h->n_udd = (int)dd.size() / 5;
        if (hipMalloc((void**)&h->UF, std::max<long>(of, 4) * sizeof(float)) != hipSuccess || hipMalloc((void**)&h->UD, std::max<long>(od, 4) * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&h->udesc_f, std::max<size_t>(df.size(), 5) * sizeof(long)) != hipSuccess || hipMalloc((void**)&h->udesc_d, std::max<size_t>(dd.size(), 5) * sizeof(long)) != hipSuccess) return fail("hipMalloc(winograd operands) failed");
        if (!df.empty()) (void)hipMemcpy(h->udesc_f, df.data(), df.size() * sizeof(long), hipMemcpyHostToDevice);
        if (!dd.empty()) (void)hipMemcpy(h->udesc_d, dd.data(), dd.size() * sizeof(long), hipMemcpyHostToDevice);
        if (hipEventCreateWithFlags(&h->ev_uf, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
    }
    // default intrinsics: Depth10k (src/dtk.jl:16-22) scaled to the image size
    const double f = 2648.0 / 4.63461538462 * (h->W / 416.0), cx = h->W / 2.0, cy = h->H / 2.0;
    const float Kc[9] = {(float)f, 0, 0, 0, (float)f, 0, (float)cx, (float)cy, 1};
    const float iKc[9] = {(float)(1 / f), 0, 0, 0, (float)(1 / f), 0, (float)(-cx / f), (float)(-cy / f), 1};
    std::memcpy(h->K, Kc, sizeof(Kc)); std::memcpy(h->invK, iKc, sizeof(iKc));
    {   // the weight-gradient stream only fills in around the dependency chain on the caller's stream: lowest priority (MD2_S2_PRIO: 0 default queue priority)
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int want = md2_tune("MD2_S2_PRIO", 1);
        const int prio = want > 0 ? lo : (want < 0 ? hi : 0);
        if (hipStreamCreateWithPriority(&h->s2, hipStreamNonBlocking, prio) != hipSuccess || hipStreamCreateWithFlags(&h->s3, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    }
    if (hipEventCreateWithFlags(&h->ev_a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_b, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_c, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
    for (int i = 0; i < 64; ++i) if (hipEventCreateWithFlags(&h->ev_pool[i], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
    if (hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_p0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_p1, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
    h->two_streams = getenv("MD2_ONE_STREAM") == nullptr;
    if (hipStreamCreateWithFlags(&h->s4, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_join4, hipEventDisableTiming) != hipSuccess) return fail("hipStreamCreate failed");
    h->n_wg_streams = md2_tune("MD2_WG_STREAMS", 1);
    *out = h;
    return 0;
}
// the handle's host-side description of a configuration (no device work): sizes, frame roles, scale levels -- what build_layers needs
static void apply_cfg(md2_handle* h, const md2_config* cfg) {
    h->cfg = *cfg; h->N = cfg->batch; h->NL = 3 * cfg->batch; h->C = cfg->channels; h->H = cfg->height; h->W = cfg->width;
    h->depth = cfg->encoder_depth == 50 ? 50 : 18; h->bf16 = cfg->precision; h->bs = cfg->precision == 2;
    {   // TrainCache.target_id / source_ids (1-based at the boundary; validated above)
        const int t = cfg->target_id ? cfg->target_id : 2, s0 = cfg->source_ids[0] ? cfg->source_ids[0] : (cfg->source_ids[1] ? 0 : 1), s1 = cfg->source_ids[1] ? cfg->source_ids[1] : (cfg->source_ids[0] ? 0 : 3);
        h->tgt = t - 1; h->src[0] = s0 - 1; h->src[1] = s1 - 1;
        // DepthDecoder scale_levels / TrainCache.scales (all zero: 2:5 and 1 / 2^(5 - level), src/Monodepth.jl:106-107)
        if (cfg->scale_levels[0]) {
            h->ns = 0; for (int b = 0; b < 5; ++b) h->out_of[b] = -1;
            for (int i = 0; i < 5 && cfg->scale_levels[i]; ++i) {
                const int l = cfg->scale_levels[i];
                h->lv[i] = l; h->out_of[l - 1] = i; h->ns = i + 1;
            }
            h->nbr = h->lv[h->ns - 1];
        }
        bool scales_given = false; for (int i = 0; i < 5; ++i) scales_given |= cfg->scales[i] != 0.0;
        for (int i = 0; i < h->ns; ++i) h->scl[i] = (scales_given && cfg->scales[i] >= 0.0) ? cfg->scales[i] : 1.0 / (double)(1 << (5 - h->lv[i]));
        for (int k = 0; k < 2; ++k) { const bool before = h->src[k] < h->tgt; h->pose_fr[2 * k] = before ? h->src[k] : h->tgt; h->pose_fr[2 * k + 1] = before ? h->tgt : h->src[k]; }
    }
}
static void comm_destroy(md2_handle* h);
static void destroy_handle(md2_handle* h) {          // also the failure path of md2_create: every member is null until it was created
    (void)hipDeviceSynchronize();
    comm_destroy(h);
    for (auto& e : h->evts) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int i = 0; i < 64; ++i) if (h->ev_pool[i]) (void)hipEventDestroy(h->ev_pool[i]);
    for (int i = 0; i < 8; ++i) { if (h->ev_cm[i]) (void)hipEventDestroy(h->ev_cm[i]); if (h->ev_cs[i]) (void)hipEventDestroy(h->ev_cs[i]); }
    hipEvent_t evs[] = {h->ev_join, h->ev_p0, h->ev_p1, h->ev_a, h->ev_b, h->ev_c, h->ev_join4, h->ev_cdone, h->ev_uf};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    hipStream_t sts[] = {h->s2, h->s3, h->s4, h->scomm};
    for (hipStream_t st : sts) if (st) (void)hipStreamDestroy(st);
    (void)hipFree(h->wdesc_dev); (void)hipFree(h->arena); (void)hipFree(h->P); (void)hipFree(h->G); (void)hipFree(h->M); (void)hipFree(h->V); (void)hipFree(h->WD); (void)hipFree(h->Pb); (void)hipFree(h->WDb); (void)hipFree(h->RS);
    (void)hipFree(h->UF); (void)hipFree(h->UD); (void)hipFree(h->udesc_f); (void)hipFree(h->udesc_d);
    delete h;
}
int32_t md2_destroy(md2_handle* h) {
    if (!h) return 0;
    destroy_handle(h); return 0;
}
int32_t md2_set_stream(md2_handle* h, void* st) { h->stream = (hipStream_t)st; return 0; }
int32_t md2_synchronize(md2_handle* h) { HIPCHK(hipStreamSynchronize(h->stream)); return 0; }
int32_t md2_set_intrinsics(md2_handle* h, const float* K, const float* invK) { std::memcpy(h->K, K, 36); std::memcpy(h->invK, invK, 36); return 0; }

int64_t md2_params_count(md2_handle* h) { return h->n_flux; }
int32_t md2_params_set(md2_handle* h, const float* host, int64_t n) {
    if (n != h->n_flux) { h->err = "md2_params_set: wrong parameter count"; return 2; }
    h->have_fwd = false; h->prep_done = false;          // a pending md2_backward would mix old activations / weight copies with the new parameters
    std::vector<float> in((size_t)h->n_int);
    convert_params(h, in.data(), const_cast<float*>(host), true);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->P, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(h->M, 0, in.size() * sizeof(float))); HIPCHK(hipMemset(h->V, 0, in.size() * sizeof(float)));
    h->beta1p = h->beta2p = 1.0;
    return 0;
}
static int get_flat(md2_handle* h, const float* dev, float* host, int64_t n) {
    if (n != h->n_flux) { h->err = "wrong parameter count"; return 2; }
    std::vector<float> in((size_t)h->n_int);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(in.data(), dev, in.size() * sizeof(float), hipMemcpyDeviceToHost));
    convert_params(h, in.data(), host, false);
    return 0;
}
int32_t md2_params_get(md2_handle* h, float* host, int64_t n) { return get_flat(h, h->P, host, n); }
int32_t md2_grads_get(md2_handle* h, float* host, int64_t n) { return get_flat(h, h->G, host, n); }
int32_t md2_grad_buffer(md2_handle* h, void** p, int64_t* count) { *p = h->G; *count = h->n_int; return 0; }
int32_t md2_param_buffer(md2_handle* h, void** p, int64_t* count) { *p = h->P; *count = h->n_int; return 0; }

static int copy_outputs(md2_handle* h, float* d0, float* d1, float* d2, float* d3, float* d4, float* rvec, float* tvec) {
    float* ds[MD2_MAX_SCALES] = {d0, d1, d2, d3, d4};
    for (int i = 0; i < h->ns; ++i)
        if (ds[i]) HIPCHK(hipMemcpyAsync(ds[i], h->dispv[i].p, (size_t)h->N * h->dispv[i].H * h->dispv[i].W * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (rvec) HIPCHK(hipMemcpyAsync(rvec, h->rvec, 6 * h->N * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (tvec) HIPCHK(hipMemcpyAsync(tvec, h->tvec, 6 * h->N * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}
int32_t md2_model_forward(md2_handle* h, const float* x, float* d0, float* d1, float* d2, float* d3, float* d4, float* rvec, float* tvec) {
    if (!h->net_ok) { h->err = "this handle is loss-only (height/width below 64 or not multiples of 32): the model entry points are unavailable"; return 2; }
    h->have_fwd = false;                    // overwrites the activations a pending md2_backward would need
    if (int r = model_forward(h, x)) return r;
    return copy_outputs(h, d0, d1, d2, d3, d4, rvec, tvec);
}
int32_t md2_eval_poses(md2_handle* h, const float* x, float* rvec, float* tvec) {
    if (!h->net_ok) { h->err = "this handle is loss-only (height/width below 64 or not multiples of 32): the model entry points are unavailable"; return 2; }
    h->have_fwd = false;
    if (int r = model_forward(h, x, 3, -1, true, false)) return r;          // encoder + pose decoder; the depth decoder is skipped
    return copy_outputs(h, nullptr, nullptr, nullptr, nullptr, nullptr, rvec, tvec);
}
int32_t md2_train_loss(md2_handle* h, const float* x, const float* auto_loss, float* loss) {
    if (!h->net_ok) { h->err = "this handle is loss-only (height/width below 64 or not multiples of 32): the model entry points are unavailable"; return 2; }
    if (h->cfg.automasking && !auto_loss) { h->err = "md2_train_loss: automasking is on but auto_loss is NULL"; return 2; }
    if (h->testmode) { h->err = "md2_train_loss: the handle is in test mode (md2_set_testmode(h, 0) = Flux.trainmode! first)"; return 2; }
    h->have_fwd = false; h->prep_done = false; h->ddisp_zeroed = false; h->Rt_ready = false;
    h->acc_zeroed = true;                   // model_forward zeroes every accumulator of this step at once (consumed by loss_train below)
    if (h->two_streams && h->s2) {          // ordered after everything already on the caller's stream (the previous ADAM step reads G and writes P)
        (void)hipEventRecord(h->ev_p0, h->stream); (void)hipStreamWaitEvent(h->s2, h->ev_p0, 0);
        refresh_wino_fwd(h, h->s2); (void)hipEventRecord(h->ev_uf, h->s2);
        if (int r = backward_prepare(h, h->s2)) { h->acc_zeroed = false; return r; }
        (void)hipEventRecord(h->ev_p1, h->s2); h->prep_done = true;
    }
    if (int r = model_forward(h, x, 3, -1, true, true, h->prep_done)) { h->acc_zeroed = false; h->ddisp_zeroed = false; h->Rt_ready = false; return r; }
    const float* disp[MD2_MAX_SCALES] = {h->dispv[0].p, h->dispv[1].p, h->dispv[2].p, h->dispv[3].p, h->dispv[4].p};
    if (int r = loss_train(h, x, disp, h->rvec, h->tvec, auto_loss, loss)) return r;          // the loss pullback comes out of the same pass: md2_backward starts at the network
    h->x_last = x; h->auto_last = auto_loss; h->have_fwd = true;
    return 0;
}
int32_t md2_backward(md2_handle* h) {
    if (!h->have_fwd) { h->err = "md2_backward: call md2_train_loss first"; return 2; }
    return model_backward(h);                   // ddisp_lo / drvec / dtvec were filled by md2_train_loss
}
int32_t md2_adam_step(md2_handle* h, float grad_scale) {
    h->beta1p *= 0.9; h->beta2p *= 0.999;
    launch_adam(h->P, h->G, h->M, h->V, h->n_int, (float)(1.0 - h->beta1p), (float)(1.0 - h->beta2p), (float)h->cfg.lr, grad_scale, h->stream);
    KCHK(); return 0;
}
int32_t md2_train_step(md2_handle* h, const float* x, const float* auto_loss, float* loss, int32_t do_update) {
    static const bool no_defer = md2_tune_set("MD2_NO_DEFER_TAIL");
    h->defer_tail = h->two_streams && h->s3 && !no_defer;            // the pullback follows at once: the loss's tail opens its pose stream (loss_train / model_backward)
    if (int r = md2_train_loss(h, x, auto_loss, loss)) { h->defer_tail = false; h->tail.pending = false; return r; }
    if (int r = md2_backward(h)) return r;
    if (h->comm) { if (int r = md2_allreduce_grads(h)) return r; }                  // data parallel: gradient of the global batch = mean over the ranks
    if (do_update) return md2_adam_step(h, h->comm ? 1.f / (float)h->world : 1.f);
    return 0;
}
int32_t md2_automasking_loss(md2_handle* h, const float* x, float* out) {
    // photometric of the UNWARPED sources vs the target, min over sources (src/training.jl:9-11); the frames are read in place
    if (h->H < 2 || h->W < 2) { h->err = "md2_automasking_loss: image too small"; return 2; }
    launch_automask(x, out, h->N, h->C, h->H, h->W, h->src[0], h->src[1], h->tgt, h->stream);
    KCHK(); return 0;
}
int32_t md2_loss_fwd_bwd(md2_handle* h, const float* x, const float* d0, const float* d1, const float* d2, const float* d3, const float* d4, const float* rvec, const float* tvec,
                         const float* auto_loss, float* loss, float* dd0, float* dd1, float* dd2, float* dd3, float* dd4, float* drvec, float* dtvec) {
    const float* disp[MD2_MAX_SCALES] = {d0, d1, d2, d3, d4};
    for (int i = 0; i < h->ns; ++i) if (!disp[i]) { h->err = "md2_loss_fwd_bwd: a disparity of a configured scale level is NULL"; return 2; }
    h->have_fwd = false;                    // overwrites Rt / totals of a pending md2_train_loss
    if (!(dd0 || dd1 || dd2 || dd3 || dd4 || drvec || dtvec)) return loss_forward(h, x, disp, rvec, tvec, auto_loss, loss);
    if (int r = loss_train(h, x, disp, rvec, tvec, auto_loss, loss)) return r;
    float* dds[MD2_MAX_SCALES] = {dd0, dd1, dd2, dd3, dd4};
    for (int i = 0; i < h->ns; ++i)
        if (dds[i]) HIPCHK(hipMemcpyAsync(dds[i], h->ddisp_lo[i], (size_t)h->N * h->dispv[i].H * h->dispv[i].W * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (drvec) HIPCHK(hipMemcpyAsync(drvec, h->drvec, 6 * h->N * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (dtvec) HIPCHK(hipMemcpyAsync(dtvec, h->dtvec, 6 * h->N * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int32_t md2_simple_depth_loss(md2_handle* h, const float* x, const float* disp, const float* rvec, const float* tvec, float* loss, float* ddisp, float* drvec, float* dtvec) {
    const float* d4[MD2_MAX_SCALES] = {disp, disp, disp, disp, disp};
    h->have_fwd = false;
    if (!(ddisp || drvec || dtvec)) return loss_forward(h, x, d4, rvec, tvec, nullptr, loss, true);
    if (int r = loss_train(h, x, d4, rvec, tvec, nullptr, loss, true)) return r;
    if (ddisp) HIPCHK(hipMemcpyAsync(ddisp, h->ddisp_full, (size_t)h->N * h->H * h->W * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (drvec) HIPCHK(hipMemcpyAsync(drvec, h->drvec, 6 * h->N * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (dtvec) HIPCHK(hipMemcpyAsync(dtvec, h->dtvec, 6 * h->N * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

// eval_disparity(m, x), src/model.jl:28: encoder + depth decoder on n single frames x [n][C][H][W], n <= batch.
// BatchNorm follows the handle's mode like Flux's: batch statistics under trainmode! (the default, src/Monodepth.jl:126), the tracked running
// statistics after md2_set_testmode(h, 1) -- what the reference's eval_image / eval_video do (testmode!, src/Monodepth.jl:209,237).
int32_t md2_eval_disparity(md2_handle* h, const float* x, int32_t n, float* d0, float* d1, float* d2, float* d3, float* d4) {
    if (n < 1 || n > h->N) { h->err = "md2_eval_disparity: need 1 <= n <= batch"; return 2; }
    if (!h->net_ok) { h->err = "this handle is loss-only (height/width below 64 or not multiples of 32): the model entry points are unavailable"; return 2; }
    set_batch(h, n, n);
    int r = model_forward(h, x, 1, 0, false);
    if (!r) {
        float* ds[MD2_MAX_SCALES] = {d0, d1, d2, d3, d4};
        for (int i = 0; i < h->ns && !r; ++i)
            if (ds[i] && hipMemcpyAsync(ds[i], h->dispv[i].p, (size_t)n * h->dispv[i].H * h->dispv[i].W * sizeof(float), hipMemcpyDeviceToDevice, h->stream) != hipSuccess) { h->err = "md2_eval_disparity: copy failed"; r = 1; }
    }
    set_batch(h, h->NL, h->N);
    h->have_fwd = false;                 // the training activations were overwritten
    return r;
}
// ADAM state (moments in Flux parameter layout + running beta powers) for true resume; the reference saves weights only
// (src/Monodepth.jl:191-194) — this is the added optimiser state.
int32_t md2_opt_state_get(md2_handle* h, float* host_m, float* host_v, int64_t n, double* beta1p, double* beta2p) {
    if (int r = get_flat(h, h->M, host_m, n)) return r;
    if (int r = get_flat(h, h->V, host_v, n)) return r;
    *beta1p = h->beta1p; *beta2p = h->beta2p; return 0;
}
int32_t md2_opt_state_set(md2_handle* h, const float* host_m, const float* host_v, int64_t n, double beta1p, double beta2p) {
    if (n != h->n_flux) { h->err = "md2_opt_state_set: wrong parameter count"; return 2; }
    std::vector<float> in((size_t)h->n_int);
    HIPCHK(hipStreamSynchronize(h->stream));
    convert_params(h, in.data(), const_cast<float*>(host_m), true); HIPCHK(hipMemcpy(h->M, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice));
    convert_params(h, in.data(), const_cast<float*>(host_v), true); HIPCHK(hipMemcpy(h->V, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice));
    h->beta1p = beta1p; h->beta2p = beta2p; return 0;
}

// weight gradients on the second stream (default on).  Off = every kernel runs alone on the main stream: used while timing single
// kernels for the roofline, because overlapped launches inflate each other's duration.
int32_t md2_set_two_streams(md2_handle* h, int32_t on) { HIPCHK(hipStreamSynchronize(h->stream)); h->two_streams = on != 0; return 0; }

int32_t md2_profile_enable(md2_handle* h, int32_t on) { h->prof = on != 0; return 0; }
int32_t md2_profile_reset(md2_handle* h) {
    (void)hipStreamSynchronize(h->stream);
    for (auto& e : h->evts) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    h->evts.clear();
    for (int i = 0; i < 5; ++i) { h->prof_flops[i] = 0; h->prof_bytes[i] = 0; h->prof_exec[i] = 0; }
    return 0;
}
int32_t md2_profile_get(md2_handle* h, int32_t cls, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    HIPCHK(hipStreamSynchronize(h->stream));
    double t = 0; int64_t n = 0;
    for (auto& e : h->evts) if (e.cls == cls) { float ms = 0; if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { t += ms; ++n; } }
    if (total_ms) *total_ms = t; if (launches) *launches = n;
    if (flops) *flops = (cls >= 0 && cls < 5) ? h->prof_flops[cls] : 0; if (bytes) *bytes = (cls >= 0 && cls < 5) ? h->prof_bytes[cls] : 0;
    return 0;
}

int32_t md2_profile_get_executed(md2_handle* h, int32_t cls, double* flops_executed) {
    if (flops_executed) *flops_executed = (cls == CLS_IGEMM || cls == CLS_WGRAD) ? h->prof_exec[cls] : ((cls >= 0 && cls < 5) ? h->prof_flops[cls] : 0);
    return 0;
}
int32_t md2_profile_dump(md2_handle* h, const char* path) {
    HIPCHK(hipStreamSynchronize(h->stream));
    FILE* f = fopen(path, "w"); if (!f) { h->err = "md2_profile_dump: cannot open file"; return 2; }
    fprintf(f, "cls,ms,gflop,tflops,desc\n");
    for (auto& e : h->evts) { float ms = 0; (void)hipEventElapsedTime(&ms, e.a, e.b); fprintf(f, "%d,%.5f,%.4f,%.2f,%s\n", e.cls, ms, e.flops * 1e-9, ms > 0 ? e.flops / (ms * 1e-3) * 1e-12 : 0.0, e.desc); }
    fclose(f); return 0;
}

// ---- op-level layer entry points (NCHW in / out; converted to NHWC views in temporaries) ----------------------------------
static float* tmp_alloc(size_t n, hipStream_t s) { float* p = nullptr; if (hipMallocAsync((void**)&p, n * sizeof(float), s) != hipSuccess) return nullptr; (void)hipMemsetAsync(p, 0, n * sizeof(float), s); return p; }
static int pad16(int c) { return (c + 15) / 16 * 16; }
// md2_set_op_storage(): the op-level convolution entry points (md2_conv2d_fwd / md2_conv2d_bn_fwd / md2_conv2d_bwd) run on bf16 TENSORS with bf16 weight copies -- the kernels
// of md2_config.precision = 2 -- so that tests can hold each of them against a plain fp32 reference of the same op.  Process-wide, 0 = fp32 tensors (default).
static int g_op_bs = 0;
int32_t md2_set_op_storage(int32_t bf16) { if (bf16 != 0 && bf16 != 1) return 2; g_op_bs = bf16; return 0; }
static int padc(int c) { return g_op_bs ? (c <= 16 ? 16 : (c + 31) / 32 * 32) : pad16(c); }          // bf16 tensors: 16 channels (paired taps) or a multiple of 32 (64-byte K-steps)
static const float* op_weights(const float* w_f32, size_t n, hipStream_t s) {          // bf16 tensors: a bf16 copy of the (internal-layout) weights, + 64 elements the paired-tap loader may read past a row
    if (!g_op_bs) return w_f32;
    uint16_t* wb = nullptr; const size_t n4 = (n + 3) / 4 * 4;
    if (hipMallocAsync((void**)&wb, (n4 + 64) * sizeof(uint16_t), s) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(wb, 0, (n4 + 64) * sizeof(uint16_t), s);
    launch_cast_bf16(w_f32, wb, (long)(n / 4 * 4), s);
    return reinterpret_cast<const float*>(wb);
}
// weights Flux [Cout][Cin][k][k] (device) -> internal [CoutP][taps][CinP] (device) via host
static int upload_weights(const float* w_dev, float* w_int, int cin, int cout, int k, int cinP, int coutP, hipStream_t s) {
    std::vector<float> f((size_t)cin * cout * k * k), in((size_t)coutP * k * k * cinP, 0.f);
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpy(f.data(), w_dev, f.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int co = 0; co < cout; ++co) for (int ci = 0; ci < cin; ++ci) for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw)
        in[((size_t)co * k * k + (k - 1 - kh) * k + (k - 1 - kw)) * cinP + ci] = f[(((size_t)co * cin + ci) * k + kh) * k + kw];
    return hipMemcpy(w_int, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess;
}
static int conv2d_fwd_impl(const float* x, const float* w, const float* b, float* y, float* bn_mean, float* bn_invstd, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                           int32_t pad_mode, int32_t act, hipStream_t s) {
    const int cinP = padc(Cin), coutP = padc(Cout), bs = g_op_bs;
    const int p_eff = pad_mode == PAD_REFLECT ? 1 : pad;
    const int Ho = (H + 2 * p_eff - k) / stride + 1, Wo = (W + 2 * p_eff - k) / stride + 1;
    float *xi = tmp_alloc((size_t)N * H * W * cinP, s), *wi = tmp_alloc((size_t)coutP * k * k * cinP, s), *yi = tmp_alloc((size_t)N * Ho * Wo * coutP, s), *bi = tmp_alloc(coutP, s);
    float* st = bn_mean ? tmp_alloc(2 * coutP, s) : nullptr; double* acc = bn_mean ? (double*)tmp_alloc((size_t)4 * BN_STAT_SLOTS * coutP, s) : nullptr;
    if (!xi || !wi || !yi || !bi || (bn_mean && (!st || !acc))) return 3;
    View xv = make_view(xi, N, H, W, cinP, bs), yv = make_view(yi, N, Ho, Wo, coutP, bs);
    View xs = xv; xs.C = Cin; launch_nchw_to_view(x, xs, s);
    if (upload_weights(w, wi, Cin, Cout, k, cinP, coutP, s)) return 1;
    if (b) (void)hipMemcpyAsync(bi, b, Cout * sizeof(float), hipMemcpyDeviceToDevice, s);
    const float* wk = op_weights(wi, (size_t)coutP * k * k * cinP, s); if (!wk) return 3;
    ConvArgs a = conv_args(xv, wk, yv, k, stride, p_eff, 1, pad_mode); a.bf16 = bs ? 2 : 0;
    a.bias = b ? bi : nullptr; a.act = act; a.stats = acc;          // stats: fused into the epilogue of the 32-wide tiles, a column-statistics pass otherwise
    float* uw = nullptr;                                            // 3x3 / stride 1: the Winograd operand, as a handle keeps it per layer
    if (k == 3 && stride == 1 && coutP % 32 == 0) { const int up = bs ? 4 : g_wino_pieces; uw = tmp_alloc(conv_wino_u_floats(coutP, cinP), s); if (!uw) return 3; launch_wino_weights(wi, uw, coutP, cinP, up, s); a.wino_u = uw; a.wino_pieces = wino_resolve_pieces(up, a.Cin); }          // (bf16 tensors: one bf16 piece, as a precision-2 handle prepares it)
    // a split-K workspace like a handle's, so that the op-level tests drive the same reduce pass (bias / activation / statistics) the training step uses on small maps
    const long ws_cap = std::min<long>(8L << 20, 8L * a.M * a.Cout);
    float* ws = tmp_alloc((size_t)ws_cap, s); if (!ws) return 3;
    a.ws = ws; a.ws_cap = ws_cap;
    a.split3 = bs ? 0 : igemm_resolve_split(g_wino_pieces, a.nkh * a.nkw, a.Cin, a.Cout);
    launch_conv_igemm(a, s);
    (void)hipFreeAsync(ws, s); if (uw) (void)hipFreeAsync(uw, s);
    if (bn_mean) {
        launch_bn_finalize(acc, st, coutP, (double)yv.pixels(), BN_STAT_SLOTS, nullptr, 0.f, s);
        (void)hipMemcpyAsync(bn_mean, st, Cout * sizeof(float), hipMemcpyDeviceToDevice, s); (void)hipMemcpyAsync(bn_invstd, st + coutP, Cout * sizeof(float), hipMemcpyDeviceToDevice, s);
        (void)hipFreeAsync(st, s); (void)hipFreeAsync(acc, s);
    }
    View ys = yv; ys.C = Cout; launch_nhwc_to_nchw(nullptr, ys, y, s);
    if (wk != wi) (void)hipFreeAsync(const_cast<float*>(wk), s);
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(wi, s); (void)hipFreeAsync(yi, s); (void)hipFreeAsync(bi, s);
    return hipGetLastError() != hipSuccess;
}
int32_t md2_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                       int32_t pad_mode, int32_t act, void* stream) {
    return conv2d_fwd_impl(x, w, b, y, nullptr, nullptr, N, Cin, H, W, Cout, k, stride, pad, pad_mode, act, (hipStream_t)stream);
}
// The encoder's conv -> BatchNorm pair: bias-free convolution whose epilogue also accumulates the training-BatchNorm batch statistics (mean, 1/sqrt(var + eps))
int32_t md2_conv2d_bnstats_fwd(const float* x, const float* w, float* y, float* bn_mean, float* bn_invstd, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t k,
                               int32_t stride, int32_t pad, void* stream) {
    if (!bn_mean || !bn_invstd) return 2;
    return conv2d_fwd_impl(x, w, nullptr, y, bn_mean, bn_invstd, N, Cin, H, W, Cout, k, stride, pad, 0, ACT_NONE, (hipStream_t)stream);
}
int32_t md2_conv2d_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride,
                       int32_t pad, int32_t pad_mode, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int cinP = padc(Cin), coutP = padc(Cout), bs = g_op_bs;
    const int p_eff = pad_mode == PAD_REFLECT ? 1 : pad;
    const int Ho = (H + 2 * p_eff - k) / stride + 1, Wo = (W + 2 * p_eff - k) / stride + 1;
    const size_t wn = (size_t)coutP * k * k * cinP;
    float *xi = tmp_alloc((size_t)N * H * W * cinP, s), *wi = tmp_alloc(wn, s), *wd = tmp_alloc(wn, s), *gi = tmp_alloc(wn, s), *dyi = tmp_alloc((size_t)N * Ho * Wo * coutP, s), *gb = tmp_alloc(coutP, s);
    if (!xi || !wi || !wd || !gi || !dyi || !gb) return 3;
    View xv = make_view(xi, N, H, W, cinP, bs), dyv = make_view(dyi, N, Ho, Wo, coutP, bs);
    View xs = xv; xs.C = Cin; launch_nchw_to_view(x, xs, s);
    View ds = dyv; ds.C = Cout; launch_nchw_to_view(dy, ds, s);
    if (upload_weights(w, wi, Cin, Cout, k, cinP, coutP, s)) return 1;
    if (dx) {
        launch_weight_to_dgrad(wi, wd, coutP, cinP, k * k, s);
        const int Hp = pad_mode == PAD_REFLECT ? H + 2 : H, Wp = pad_mode == PAD_REFLECT ? W + 2 : W;
        float* dxi = tmp_alloc((size_t)N * Hp * Wp * cinP, s); if (!dxi) return 3;
        View dxp = make_view(dxi, N, Hp, Wp, cinP, bs);
        const float* wk = op_weights(wd, wn, s); if (!wk) return 3;
        ConvArgs a = conv_args(dyv, wk, dxp, k, 1, k - 1 - (pad_mode == PAD_REFLECT ? 0 : pad), stride, PAD_ZERO); a.bf16 = bs ? 2 : 0;
        a.accumulate = 1;                                   // dxi is zero-initialised; strided pullbacks may have tap-free residue classes
        const long ws_cap = std::min<long>(8L << 20, 8L * a.M * a.Cout);
        float* ws = tmp_alloc((size_t)ws_cap, s); if (!ws) return 3;
        a.ws = ws; a.ws_cap = ws_cap;
        float* uw = nullptr;
        if (k == 3 && stride == 1 && cinP % 32 == 0) { const int up = bs ? 4 : g_wino_pieces; uw = tmp_alloc(conv_wino_u_floats(cinP, coutP), s); if (!uw) return 3; launch_wino_weights(wd, uw, cinP, coutP, up, s); a.wino_u = uw; a.wino_pieces = wino_resolve_pieces(up, a.Cin); }
        a.split3 = bs ? 0 : igemm_resolve_split(g_wino_pieces, a.dil > 1 ? std::max(1, k * k / (a.dil * a.dil)) : a.nkh * a.nkw, a.Cin, a.Cout);
        launch_conv_igemm(a, s);
        (void)hipFreeAsync(ws, s); if (uw) (void)hipFreeAsync(uw, s); if (wk != wd) (void)hipFreeAsync(const_cast<float*>(wk), s);
        if (pad_mode == PAD_REFLECT) {
            float* dxf = tmp_alloc((size_t)N * H * W * cinP, s); if (!dxf) return 3;
            View dv = make_view(dxf, N, H, W, cinP, bs);
            launch_fold_reflect(dxp, dv, 0, nullptr, ACT_NONE, s);
            View o = dv; o.C = Cin; launch_nhwc_to_nchw(nullptr, o, dx, s);
            (void)hipFreeAsync(dxf, s);
        } else { View o = dxp; o.C = Cin; launch_nhwc_to_nchw(nullptr, o, dx, s); }
        (void)hipFreeAsync(dxi, s);
    }
    if (dw || db) {
        WgradArgs a{};
        a.x = xv.p; a.x_bs = xv.bs; a.x_ps = xv.ps; a.H = H; a.W = W; a.Cin = cinP;
        a.dy = dyv.p; a.dy_bs = dyv.bs; a.dy_ps = dyv.ps; a.Ho = Ho; a.Wo = Wo; a.Cout = coutP;
        a.dw = gi; a.N = N; a.KH = k; a.KW = k; a.stride = stride; a.pad = p_eff; a.padmode = pad_mode; a.M = N * Ho * Wo; a.db = gb; a.bf16 = bs ? 2 : 0;
        launch_conv_wgrad(a, s);
        if (hipStreamSynchronize(s) != hipSuccess) return 1;
        std::vector<float> in(wn), f((size_t)Cin * Cout * k * k), bb(coutP);
        (void)hipMemcpy(in.data(), gi, wn * sizeof(float), hipMemcpyDeviceToHost); (void)hipMemcpy(bb.data(), gb, coutP * sizeof(float), hipMemcpyDeviceToHost);
        for (int co = 0; co < Cout; ++co) for (int ci = 0; ci < Cin; ++ci) for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw)
            f[(((size_t)co * Cin + ci) * k + kh) * k + kw] = in[((size_t)co * k * k + (k - 1 - kh) * k + (k - 1 - kw)) * cinP + ci];
        if (dw) (void)hipMemcpy(dw, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice);
        if (db) (void)hipMemcpy(db, bb.data(), Cout * sizeof(float), hipMemcpyHostToDevice);
    }
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(wi, s); (void)hipFreeAsync(wd, s); (void)hipFreeAsync(gi, s); (void)hipFreeAsync(dyi, s); (void)hipFreeAsync(gb, s);
    return hipGetLastError() != hipSuccess;
}
int32_t md2_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t relu, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 || 256 % (C / 4)) return 2;
    float *xi = tmp_alloc((size_t)N * H * W * C, s), *st = tmp_alloc(2 * C, s); double* acc = (double*)tmp_alloc(4 * C, s);
    if (!xi || !st || !acc) return 3;
    View v = make_view(xi, N, H, W, C); launch_nchw_to_view(x, v, s);
    launch_colstats(v, acc, s); launch_bn_finalize(acc, st, C, (double)v.pixels(), 1, nullptr, 0.f, s);
    launch_bn_apply(v, st, gamma, beta, nullptr, relu ? ACT_RELU : ACT_NONE, v, s);
    launch_nhwc_to_nchw(nullptr, v, y, s);
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(st, s); (void)hipFreeAsync(acc, s);
    return hipGetLastError() != hipSuccess;
}
int32_t md2_bn_train_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 || 256 % (C / 4)) return 2;
    const size_t n = (size_t)N * H * W * C;
    float *xi = tmp_alloc(n, s), *di = tmp_alloc(n, s), *oi = tmp_alloc(n, s), *st = tmp_alloc(4 * C, s); double* acc = (double*)tmp_alloc(8 * C, s);
    if (!xi || !di || !oi || !st || !acc) return 3;
    View v = make_view(xi, N, H, W, C), d = make_view(di, N, H, W, C), o = make_view(oi, N, H, W, C);
    launch_nchw_to_view(x, v, s); launch_nchw_to_view(dy, d, s);
    launch_colstats(v, acc, s); launch_bn_finalize(acc, st, C, (double)v.pixels(), 1, nullptr, 0.f, s);
    (void)hipMemsetAsync(dgamma, 0, C * sizeof(float), s); (void)hipMemsetAsync(dbeta, 0, C * sizeof(float), s);
    launch_bn_bwd_reduce(d, nullptr, ACT_NONE, v, st, acc + 2 * C, s);
    launch_bn_bwd_apply(d, nullptr, ACT_NONE, v, st, gamma, acc + 2 * C, (double)v.pixels(), o, dgamma, dbeta, s, 1, st + 2 * C);
    launch_nhwc_to_nchw(nullptr, o, dx, s);
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(di, s); (void)hipFreeAsync(oi, s); (void)hipFreeAsync(st, s); (void)hipFreeAsync(acc, s);
    return hipGetLastError() != hipSuccess;
}
int32_t md2_maxpool3s2_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream; if (C % 4) return 2;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    float *xi = tmp_alloc((size_t)N * H * W * C, s), *yi = tmp_alloc((size_t)N * Ho * Wo * C, s); uint8_t* arg = (uint8_t*)tmp_alloc((size_t)N * Ho * Wo * C / 4 + 4, s);
    if (!xi || !yi || !arg) return 3;
    View v = make_view(xi, N, H, W, C), o = make_view(yi, N, Ho, Wo, C);
    launch_nchw_to_view(x, v, s); launch_maxpool_fwd(v, o, arg, s); launch_nhwc_to_nchw(nullptr, o, y, s);
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(yi, s); (void)hipFreeAsync(arg, s);
    return hipGetLastError() != hipSuccess;
}
int32_t md2_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream; if (C % 4) return 2;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    float *xi = tmp_alloc((size_t)N * H * W * C, s), *yi = tmp_alloc((size_t)N * Ho * Wo * C, s), *di = tmp_alloc((size_t)N * Ho * Wo * C, s), *oi = tmp_alloc((size_t)N * H * W * C, s);
    uint8_t* arg = (uint8_t*)tmp_alloc((size_t)N * Ho * Wo * C / 4 + 4, s);
    if (!xi || !yi || !di || !oi || !arg) return 3;
    View v = make_view(xi, N, H, W, C), o = make_view(yi, N, Ho, Wo, C), d = make_view(di, N, Ho, Wo, C), g = make_view(oi, N, H, W, C);
    launch_nchw_to_view(x, v, s); launch_nchw_to_view(dy, d, s);
    launch_maxpool_fwd(v, o, arg, s); launch_maxpool_bwd(d, arg, g, 0, nullptr, s); launch_nhwc_to_nchw(nullptr, g, dx, s);
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(yi, s); (void)hipFreeAsync(di, s); (void)hipFreeAsync(oi, s); (void)hipFreeAsync(arg, s);
    return hipGetLastError() != hipSuccess;
}
int32_t md2_adam_raw(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, float grad_scale, void* stream) {
    if (n <= 0 || step < 1) return 2;
    launch_adam(p, g, m, v, (long)n, (float)(1.0 - std::pow(0.9, step)), (float)(1.0 - std::pow(0.999, step)), (float)lr, grad_scale, (hipStream_t)stream);
    return hipGetLastError() != hipSuccess;
}


// ---- Flux.testmode! / trainmode!, the BatchNorm running statistics, deterministic mode ------------------------------------------------
int32_t md2_set_testmode(md2_handle* h, int32_t on) { h->testmode = on != 0; h->have_fwd = false; h->prep_done = false; return 0; }
int64_t md2_bn_state_count(md2_handle* h) { return h->n_rs; }
int32_t md2_bn_state_get(md2_handle* h, float* host, int64_t n) {
    if (n != h->n_rs) { h->err = "md2_bn_state_get: wrong element count"; return 2; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host, h->RS, (size_t)n * sizeof(float), hipMemcpyDeviceToHost)); return 0;
}
int32_t md2_bn_state_set(md2_handle* h, const float* host, int64_t n) {
    if (n != h->n_rs) { h->err = "md2_bn_state_set: wrong element count"; return 2; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->RS, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice)); return 0;
}
int32_t md2_set_deterministic(int32_t on) { g_deterministic = on != 0; return 0; }
int32_t md2_set_winograd(int32_t on, int32_t min_blocks) {
    g_wino_bf16 = (on >= 32 && on <= 35) ? 1 : 0; if (on >= 32 && on <= 35) on -= 32;          // + 32: Winograd for bf16 tensors (conv_wino3b_kernel; measured, reduced gradient fidelity, not the default)
    g_wino_persist = (on >= 16 && on <= 19) ? 1 : 0; if (on >= 16 && on <= 19) on -= 16;          // + 16: the split Winograd kernel in its persistent, tile-looping form (conv_wino3p_kernel; measured, not the default)
    g_wino_wide = (on >= 8 && on <= 11) ? 1 : 0; if (on >= 8 && on <= 11) on -= 8;          // + 8: the split kernel's 64-channel / 512-thread workgroups where Cout % 64 == 0 (conv_wino3w_kernel; measured, not the default)
    if (on == 4 || on == 5) { g_wino_on = 1; g_wino_pieces = on == 4 ? 2 : 3; }          // legacy aliases of md2_set_winograd(1, .) + md2_set_conv_products(2 / 3, -1)
    else g_wino_on = on < 0 || on > 3 ? 1 : on;
    if (min_blocks >= 0) g_wino_min_blocks = min_blocks; else if (min_blocks == -2) g_wino_min_blocks = g_wino_min_blocks_default;
    return 0;
}
int32_t md2_set_conv_products(int32_t fwd_dgrad, int32_t wgrad) {
    if (fwd_dgrad < -1 || fwd_dgrad > 3 || wgrad < -1 || wgrad > 3) return 2;
    if (fwd_dgrad >= 0) g_wino_pieces = fwd_dgrad;
    if (wgrad >= 0) g_wgrad_pieces = wgrad;
    return 0;
}
int32_t md2_get_conv_products(int32_t* fwd_dgrad, int32_t* wgrad) { if (fwd_dgrad) *fwd_dgrad = g_wino_pieces; if (wgrad) *wgrad = g_wgrad_pieces; return 0; }
int32_t md2_set_lr(md2_handle* h, double lr) { if (!(lr > 0)) { h->err = "md2_set_lr: lr must be positive"; return 2; } h->cfg.lr = lr; return 0; }

// ---- op-level hooks for the direct kernels the network uses besides the implicit GEMM (parity tests) ------------------------------------
// ResNet stem (7x7, stride 2, pad 3, no bias): x [N][C][H][W], w Flux [64][C][7][7] -> z [N][64][H/2][W/2]; bn_mean / bn_invstd [64] (optional) are the
// training-BatchNorm statistics the kernel's epilogue accumulates
int32_t md2_stem_conv_fwd(const float* x, const float* w, float* z, float* bn_mean, float* bn_invstd, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if ((C != 1 && C != 3) || (H & 1) || (W & 1) || H < 8 || W < 8) return 2;
    const int Kpad = (49 * C + 15) / 16 * 16, Ho = H / 2, Wo = W / 2;
    std::vector<float> f((size_t)64 * C * 49), in((size_t)64 * Kpad, 0.f);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(f.data(), w, f.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int co = 0; co < 64; ++co) for (int ci = 0; ci < C; ++ci) for (int kh = 0; kh < 7; ++kh) for (int kw = 0; kw < 7; ++kw)
        in[(size_t)co * Kpad + ((6 - kh) * 7 + (6 - kw)) * C + ci] = f[(((size_t)co * C + ci) * 7 + kh) * 7 + kw];
    float *wi = tmp_alloc(in.size(), s), *zi = tmp_alloc((size_t)N * Ho * Wo * 64, s), *st = tmp_alloc(128, s); double* acc = (double*)tmp_alloc((size_t)2 * BN_STAT_SLOTS * 128, s);
    if (!wi || !zi || !st || !acc) return 3;
    if (hipMemcpy(wi, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
    View zv = make_view(zi, N, Ho, Wo, 64, g_op_bs);          // (md2_set_op_storage(1): the bf16 modes' kernel -- bf16 products, z a bf16 tensor)
    const bool fused = !g_deterministic;
    launch_stem_fwd(x, N, C, H, W, wi, Kpad, zv, (bn_mean && fused) ? acc : nullptr, s, g_op_bs);
    if (bn_mean && !fused) launch_colstats(zv, acc, s);
    if (bn_mean) {
        launch_bn_finalize(acc, st, 64, (double)zv.pixels(), BN_STAT_SLOTS, nullptr, 0.f, s);
        (void)hipMemcpyAsync(bn_mean, st, 64 * sizeof(float), hipMemcpyDeviceToDevice, s); (void)hipMemcpyAsync(bn_invstd, st + 64, 64 * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
    launch_nhwc_to_nchw(nullptr, zv, z, s);
    (void)hipFreeAsync(wi, s); (void)hipFreeAsync(zi, s); (void)hipFreeAsync(st, s); (void)hipFreeAsync(acc, s);
    return hipGetLastError() != hipSuccess;
}
// dz [N][64][H/2][W/2] -> dw Flux [64][C][7][7]
int32_t md2_stem_conv_wgrad(const float* x, const float* dz, float* dw, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if ((C != 1 && C != 3) || (H & 1) || (W & 1) || H < 8 || W < 8) return 2;
    const int Kpad = (49 * C + 15) / 16 * 16, Ho = H / 2, Wo = W / 2;
    float *gi = tmp_alloc((size_t)64 * Kpad, s), *di = tmp_alloc((size_t)N * Ho * Wo * 64, s);
    if (!gi || !di) return 3;
    View dv = make_view(di, N, Ho, Wo, 64, g_op_bs); launch_nchw_to_view(dz, dv, s);
    launch_stem_wgrad(x, N, C, H, W, dv, gi, Kpad, s, nullptr, nullptr, g_op_bs);
    std::vector<float> in((size_t)64 * Kpad), f((size_t)64 * C * 49);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(in.data(), gi, in.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int co = 0; co < 64; ++co) for (int ci = 0; ci < C; ++ci) for (int kh = 0; kh < 7; ++kh) for (int kw = 0; kw < 7; ++kw)
        f[(((size_t)co * C + ci) * 7 + kh) * 7 + kw] = in[(size_t)co * Kpad + ((6 - kh) * 7 + (6 - kw)) * C + ci];
    if (hipMemcpy(dw, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
    (void)hipFreeAsync(gi, s); (void)hipFreeAsync(di, s);
    return hipGetLastError() != hipSuccess;
}
// disparity head (src/depth_decoder.jl:43): reflect-pad + 3x3 conv to ONE channel + sigmoid.  x [N][Cin][H][W], w Flux [1][Cin][3][3], b [1] -> y [N][1][H][W]
static int head_weights(const float* w_dev, float* w_int, int Cin, hipStream_t s) {
    std::vector<float> f((size_t)Cin * 9), in((size_t)9 * Cin);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(f.data(), w_dev, f.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int ci = 0; ci < Cin; ++ci) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) in[(size_t)((2 - kh) * 3 + (2 - kw)) * Cin + ci] = f[((size_t)ci * 3 + kh) * 3 + kw];
    return hipMemcpy(w_int, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess;
}
int32_t md2_head_conv_fwd(const float* x, const float* w, const float* b, float* y, int32_t N, int32_t Cin, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if ((Cin != 16 && Cin != 32 && Cin != 64 && Cin != 128 && Cin != 256) || H < 3 || W < 3) return 2;
    float *xi = tmp_alloc((size_t)N * H * W * Cin, s), *wi = tmp_alloc((size_t)9 * Cin, s);
    if (!xi || !wi) return 3;
    View xv = make_view(xi, N, H, W, Cin); launch_nchw_to_view(x, xv, s);
    if (head_weights(w, wi, Cin, s)) return 1;
    launch_head_fwd(xv, wi, b, make_view(y, N, H, W, 1), s);
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(wi, s);
    return hipGetLastError() != hipSuccess;
}
// pullback: dy [N][1][H][W] -> dx [N][Cin][H][W], dw Flux [1][Cin][3][3], db [1].  x_act: 0, or 2 (ELU) when x is the ELU OUTPUT of its producer and
// dx is wanted with respect to the pre-activation (how the last decoder branch uses it)
int32_t md2_head_conv_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t x_act, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if ((Cin != 16 && Cin != 32 && Cin != 64 && Cin != 128 && Cin != 256) || H < 3 || W < 3) return 2;
    const size_t n = (size_t)N * H * W * Cin;
    float *xi = tmp_alloc(n, s), *oi = tmp_alloc(n, s), *wi = tmp_alloc((size_t)9 * Cin, s), *gi = tmp_alloc((size_t)9 * Cin, s), *gb = tmp_alloc(4, s);
    if (!xi || !oi || !wi || !gi || !gb) return 3;
    View xv = make_view(xi, N, H, W, Cin), ov = make_view(oi, N, H, W, Cin); launch_nchw_to_view(x, xv, s);
    if (head_weights(w, wi, Cin, s)) return 1;
    launch_head_bwd(xv, wi, make_view(const_cast<float*>(y), N, H, W, 1), make_view(const_cast<float*>(dy), N, H, W, 1), ov, 0, x_act, gi, gb, s);
    launch_nhwc_to_nchw(nullptr, ov, dx, s);
    std::vector<float> in((size_t)9 * Cin), f((size_t)Cin * 9);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(in.data(), gi, in.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int ci = 0; ci < Cin; ++ci) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) f[((size_t)ci * 3 + kh) * 3 + kw] = in[(size_t)((2 - kh) * 3 + (2 - kw)) * Cin + ci];
    if (hipMemcpy(dw, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
    (void)hipMemcpyAsync(db, gb, sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipFreeAsync(xi, s); (void)hipFreeAsync(oi, s); (void)hipFreeAsync(wi, s); (void)hipFreeAsync(gi, s); (void)hipFreeAsync(gb, s);
    return hipGetLastError() != hipSuccess;
}


// ---- data parallelism over RCCL (one process per GPU; the insertion point in the reference's loop is the update!, src/Monodepth.jl:170) -------------
// rank 0 creates the 128-byte id and hands it to the other ranks by whatever channel the host has (MPI, a file, a TCP store, torch.distributed ...)
int32_t md2_comm_unique_id(void* out128) {
    if (!out128) return 2;
    if (!rccl_load()) { g_create_err = g_rccl.err; return 3; }
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) { g_create_err = "ncclGetUniqueId failed"; return 1; }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, sizeof(id)); return 0;
}
int32_t md2_comm_init(md2_handle* h, int32_t rank, int32_t world, const void* unique_id128) {
    if (world < 1 || rank < 0 || rank >= world || !unique_id128) { h->err = "md2_comm_init: bad rank / world / id"; return 2; }
    if (!rccl_load()) { h->err = g_rccl.err; return 3; }
    if (h->comm) comm_destroy(h);
    HIPCHK(hipSetDevice(h->cfg.device));
    if (!h->scomm) {
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPCHK(hipStreamCreateWithPriority(&h->scomm, hipStreamNonBlocking, hi));              // the collective must not queue behind the compute it overlaps
        for (int i = 0; i < 8; ++i) { HIPCHK(hipEventCreateWithFlags(&h->ev_cm[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_cs[i], hipEventDisableTiming)); }
        HIPCHK(hipEventCreateWithFlags(&h->ev_cdone, hipEventDisableTiming));
    }
    ncclUniqueId id; std::memcpy(&id, unique_id128, sizeof(id));
    ncclComm_t c = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&c, world, id, rank));
    h->comm = c; h->rank = rank; h->world = world; h->grads_reduced = false;
    return 0;
}
int32_t md2_comm_destroy(md2_handle* h) { HIPCHK(hipStreamSynchronize(h->stream)); if (h->scomm) HIPCHK(hipStreamSynchronize(h->scomm)); comm_destroy(h); return 0; }
// the four [lo, hi) slices of the flat gradient buffer md2_backward all-reduces, in completion order, for a configuration -- host-only (no device, no handle), so that the
// tiling of [0, count) can be checked anywhere
int32_t md2_comm_bucket_plan(const md2_config* cfg, int64_t* lo4, int64_t* hi4, int64_t* count) {
    if (!lo4 || !hi4 || !count) { g_create_err = "null argument"; return 2; }
    if (int r = validate_cfg(cfg)) return r;
    md2_handle* h = new md2_handle();
    apply_cfg(h, cfg);
    build_layers(h);
    const BucketPlan p = bucket_plan(h);
    for (int i = 0; i < 4; ++i) { lo4[i] = p.lo[i]; hi4[i] = p.hi[i]; }
    *count = h->n_int;
    delete h;
    return 0;
}
int32_t md2_comm_world(md2_handle* h, int32_t* rank, int32_t* world) { if (rank) *rank = h->rank; if (world) *world = h->world; return 0; }
// gradient(theta) summed over the ranks.  md2_backward already did it bucket by bucket (overlapped with the pullback) when a communicator is attached; this is then a no-op.
int32_t md2_allreduce_grads(md2_handle* h) {
    if (!h->comm) { h->err = "md2_allreduce_grads: call md2_comm_init first"; return 2; }
    if (h->grads_reduced) return 0;
    NCCLCHK(g_rccl.AllReduce(h->G, h->G, (size_t)h->n_int, ncclFloat, ncclSum, (ncclComm_t)h->comm, h->stream));
    h->grads_reduced = true; return 0;
}
// in-place sum over the ranks of any device fp32 buffer on the handle's stream (loss logging, self-checks)
int32_t md2_comm_allreduce(md2_handle* h, float* dev_buf, int64_t n) {
    if (!h->comm) { h->err = "md2_comm_allreduce: call md2_comm_init first"; return 2; }
    NCCLCHK(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)n, ncclFloat, ncclSum, (ncclComm_t)h->comm, h->stream)); return 0;
}
// replicas must start identical: parameters, ADAM moments, BatchNorm running statistics and the beta powers of `root` go to every rank
int32_t md2_broadcast_state(md2_handle* h, int32_t root) {
    if (!h->comm) { h->err = "md2_broadcast_state: call md2_comm_init first"; return 2; }
    h->have_fwd = false; h->prep_done = false;
    double bp[2] = {h->beta1p, h->beta2p}; double* dbp = nullptr;
    HIPCHK(hipMalloc((void**)&dbp, sizeof(bp))); HIPCHK(hipMemcpy(dbp, bp, sizeof(bp), hipMemcpyHostToDevice));
    ncclComm_t c = (ncclComm_t)h->comm;
    NCCLCHK(g_rccl.GroupStart());
    NCCLCHK(g_rccl.Broadcast(h->P, h->P, (size_t)h->n_int, ncclFloat, root, c, h->stream));
    NCCLCHK(g_rccl.Broadcast(h->M, h->M, (size_t)h->n_int, ncclFloat, root, c, h->stream));
    NCCLCHK(g_rccl.Broadcast(h->V, h->V, (size_t)h->n_int, ncclFloat, root, c, h->stream));
    NCCLCHK(g_rccl.Broadcast(h->RS, h->RS, (size_t)h->n_rs, ncclFloat, root, c, h->stream));
    NCCLCHK(g_rccl.Broadcast(dbp, dbp, 2, ncclDouble, root, c, h->stream));
    NCCLCHK(g_rccl.GroupEnd());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(bp, dbp, sizeof(bp), hipMemcpyDeviceToHost)); (void)hipFree(dbp);
    h->beta1p = bp[0]; h->beta2p = bp[1];
    return 0;
}

}  // extern "C"
