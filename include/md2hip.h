/* md2hip.h — C ABI of libmd2hip.so: the MI355X (gfx950) implementation of Monodepth2.jl's
 * self-supervised training step (encoder + depth/pose decoders + warp/SSIM/L1 loss, fwd + bwd + ADAM).
 *
 * The reference has no FFI of its own (pure Julia; kernels come from CUDA.jl/NNlib by array-type
 * dispatch, src/Monodepth.jl:71-73, src/utils.jl:1-11).  Each entry point below names the reference
 * function(s) whose device execution it replaces; the Julia `ccall` shim that binds them is in
 * monodepth2.jl_amd/julia/Monodepth2HIP.jl and INTEGRATION.md.
 *
 * Conventions
 *  - Every function returns 0 on success, non-zero on error; md2_last_error() gives the message
 *    (no exceptions cross the ABI).  A handle is not thread-safe: one handle <-> one GPU <-> one stream.
 *  - All array arguments are DEVICE pointers to fp32 unless named `host_*`.  The caller owns every
 *    buffer it passes; the library owns the handle, weights, ADAM moments and workspaces.
 *  - Layout at the boundary is the reference's: a Julia column-major (W,H,C,N) array is passed as-is,
 *    i.e. it is row-major [N][C][H][W]; (W,H,C,L,N) == [N][L][C][H][W]; grid (2,W,H,N) == [N][H][W][2];
 *    rvec (3,N) == [N][3]; 3x3 matrices are column-major (R[n][j*3+i] = R_ij) like Julia's (3,3,N).
 *  - Calls are asynchronous on the handle's stream; only the host_* getters synchronise.
 */
#ifndef MD2HIP_H
#define MD2HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct md2_handle md2_handle;

/* Params (src/Monodepth.jl:32-42) + the shapes TrainCache (src/Monodepth.jl:44-55) is built for. */
typedef struct md2_config {
    int32_t batch;        /* N: triplets per step on this GPU (Params.batch_size)            */
    int32_t channels;     /* C: 1 (grayscale, train()) or 3                                  */
    int32_t height;       /* H (Params.target_size[2]), multiple of 32                       */
    int32_t width;        /* W (Params.target_size[1]), multiple of 32                       */
    int32_t automasking;  /* Params.automasking                                              */
    int32_t device;       /* HIP device ordinal                                              */
    double min_depth, max_depth, disparity_smoothness;   /* Params                            */
    double lr;            /* ADAM eta (1e-4, src/Monodepth.jl:125)                           */
    int32_t encoder_depth;/* ResidualNetwork(n): 18 (default when 0; src/Monodepth.jl:117) or 50    */
    int32_t precision;    /* 0: fp32 (parity mode, the reference's precision = f32, src/Monodepth.jl:72): fp32 storage, fp32 accumulation, every product at full fp32 operand
                             accuracy (fp32 MFMA, or its exact three-bf16-piece decomposition -- md2_set_conv_products below; nothing is rounded to fewer than 24 mantissa bits);
                             1: bf16 mixed precision -- the MFMA operands of the 32-wide conv / weight-gradient tiles are rounded to
                                bf16, accumulation, activations, master weights, BatchNorm, loss and ADAM stay fp32;
                             2: bf16 activation storage (BASELINE config "1024x320 bf16 mixed precision") -- every activation / skip tensor and its gradient is a bf16 tensor in HBM
                                (round to nearest even on store), the convolutions read per-step bf16 copies of the weights and run v_mfma_f32_32x32x16_bf16 /
                                16x16x32 with fp32 accumulation; fp32 stay: the master weights and ADAM, BatchNorm statistics and their pullback sums, biases, the
                                disparity maps, poses, loss and every parameter gradient.  Not a parity mode (the reference is Float32): tests bound it against the fp32 step */
    int32_t target_id;    /* TrainCache.target_id, 1-based frame of the triplet the depth decoder sees (src/model.jl:16, src/training.jl:25); 0 = the default 2 */
    int32_t source_ids[2];/* TrainCache.source_ids, 1-based, two distinct frames other than target_id in the caller's order (src/model.jl:22-35: a source before the
                             target is paired (source, target) and its pose inverted, src/training.jl:29); {0, 0} = the default {1, 3} (src/dtk.jl, src/kitty.jl)     */
    int32_t scale_levels[5]; /* DepthDecoder(; scale_levels) (src/depth_decoder.jl:26-47): ascending levels in 1..5, 0-terminated when fewer than five; output i has the resolution
                             H >> (5 - level_i); branches 1..max level are built, a head after each listed level.  All zero = the default 2:5 (src/Monodepth.jl:106)            */
    double scales[5];     /* TrainCache.scales (src/Monodepth.jl:54,107): the smoothness weight of output i is disparity_smoothness * scales[i] (src/training.jl:67).
                             All five 0 (a zero-initialised struct) = the reference's 1 / 2^(5 - level_i) for every output; otherwise each entry is taken as given -- an explicit
                             0 switches that output's smoothness term off, as in the reference -- and a negative entry asks for that output's default                         */
} md2_config;

/* ---- handle ------------------------------------------------------------------------------- */
/* height / width: multiples of 32 and >= 64 for the network; multiples of 8 and >= 16 give a LOSS-ONLY handle (md2_loss_fwd_bwd, md2_simple_depth_loss,
 * md2_automasking_loss work, the model entry points return an error).  Shape limits of the kernels are checked here, never by aborting in a step. */
int32_t md2_create(const md2_config* cfg, md2_handle** out);
int32_t md2_destroy(md2_handle* h);
const char* md2_last_error(md2_handle* h);          /* h may be NULL: error of the last failed md2_create */
int32_t md2_set_stream(md2_handle* h, void* hip_stream);   /* default: the null stream */
int32_t md2_synchronize(md2_handle* h);
/* TrainCache.K / .invK (src/Monodepth.jl:113): 3x3 column-major, host pointers */
int32_t md2_set_intrinsics(md2_handle* h, const float* host_K, const float* host_invK);

/* ---- parameters: Flux.params(model) (src/Monodepth.jl:124), flat, order in DESIGN.md;
 *      conv weights in Flux layout (kW,kH,Cin,Cout) true-convolution convention ------------------ */
int64_t md2_params_count(md2_handle* h);
int32_t md2_params_set(md2_handle* h, const float* host_params, int64_t n);
int32_t md2_params_get(md2_handle* h, float* host_params, int64_t n);
int32_t md2_grads_get(md2_handle* h, float* host_grads, int64_t n);     /* gradient(theta) do ... end */
/* The gradients are those of the last md2_backward; the NEXT md2_train_loss starts zeroing them (it prepares the pullback's
 * accumulators on the second stream while the forward pass runs), so read / all-reduce them before calling it. */
/* internal flat gradient buffer (device) for the data-parallel all-reduce (RCCL via the host) */
int32_t md2_grad_buffer(md2_handle* h, void** dev_ptr, int64_t* count);
int32_t md2_param_buffer(md2_handle* h, void** dev_ptr, int64_t* count);

/* ---- model: (m::Model)(x, source_ids, target_id), src/model.jl:8-20 (source_ids=[1,3], target_id=2)
 *      x [N][3][C][H][W]; disp_i [N][1][H>>(5-level_i)][W>>(5-level_i)] for the configured scale levels (2:5 by default: H>>(3-i)), the pointers past the last
 *      level are ignored (NULL); rvec,tvec [2][N][3] (source-major) --- */
int32_t md2_model_forward(md2_handle* h, const float* x, float* disp0, float* disp1, float* disp2, float* disp3, float* disp4,
                          float* rvec, float* tvec);
/* eval_poses(m, features, source_ids, target_id), src/model.jl:22-35, on the frames themselves (the stage features live in the handle):
 * encoder + pose decoder only, for both (source, target) pairs -> rvec, tvec [2][N][3] */
int32_t md2_eval_poses(md2_handle* h, const float* x, float* rvec, float* tvec);
/* ---- train_loss forward (src/training.jl:21-78): model + loss.  auto_loss may be NULL.
 *      loss: device scalar.  Keeps activations for md2_backward.  The loss kernel produces the loss value and its pullback
 *      (d loss / d disparities, d loss / d poses) in the same pass over the warp, so md2_backward starts at the network. --- */
int32_t md2_train_loss(md2_handle* h, const float* x, const float* auto_loss, float* loss);
/* ---- Zygote pullback of the last md2_train_loss into the gradient buffer --------------------- */
int32_t md2_backward(md2_handle* h);
/* ---- Flux.Optimise.update!(ADAM, theta, grads), src/Monodepth.jl:170; grads scaled by grad_scale first */
int32_t md2_adam_step(md2_handle* h, float grad_scale);
/* ADAM's eta (ADAM(1e-4), src/Monodepth.jl:125) can be changed between steps */
int32_t md2_set_lr(md2_handle* h, double lr);
/* ---- one whole step of the loop body src/Monodepth.jl:156-176: fwd + bwd (+ ADAM if do_update) -- */
int32_t md2_train_step(md2_handle* h, const float* x, const float* auto_loss, float* loss, int32_t do_update);
/* eval_disparity(m, x), src/model.jl:28: n single frames x [n][C][H][W] (n <= batch) -> the 4 disparities [n][1][h][w].
 * BatchNorm follows the handle's mode (md2_set_testmode): batch statistics under trainmode! (default), tracked statistics under testmode!. */
int32_t md2_eval_disparity(md2_handle* h, const float* x, int32_t n, float* disp0, float* disp1, float* disp2, float* disp3, float* disp4);
/* Flux.testmode!(model) / trainmode!(model) (src/Monodepth.jl:126,209,237): on = 1 makes every BatchNorm normalise with its tracked running mean /
 * variance (momentum 0.1, updated by every train-mode forward pass like Flux's `_track_stats!`); md2_train_loss needs train mode. */
int32_t md2_set_testmode(md2_handle* h, int32_t on);
/* The BatchNorm running statistics (Flux BatchNorm.mu / .sigma2, part of what `@save model_host` keeps, src/Monodepth.jl:191-194): for every BatchNorm in
 * parameter order, mean[c] then variance[c]. */
int64_t md2_bn_state_count(md2_handle* h);
int32_t md2_bn_state_get(md2_handle* h, float* host_state, int64_t n);
int32_t md2_bn_state_set(md2_handle* h, const float* host_state, int64_t n);
/* Bit-reproducible mode (also MD2_DETERMINISTIC=1 in the environment at md2_create): no split-K, one workgroup per weight-gradient tile, ordered loss /
 * BatchNorm / bias sums.  Process-wide, much slower; for bisecting parity failures. */
int32_t md2_set_deterministic(int32_t on);
/* Algorithm choice for the 3x3 / stride-1 convolutions (the reference leaves it to cuDNN behind Flux `Conv`, src/depth_decoder.jl:13-14): on = 1 (default) routes launches
 * with at least min_blocks workgroups (default 48 64-tile equivalents; -1 keeps the current value, -2 restores the default) to the fused Winograd F(2x2,3x3) kernels, on = 0 forces the direct
 * implicit GEMM; on = 2 / 3 pin one of the two Winograd kernels wherever it applies (tests, tuning).  on = 4 / 5 are legacy aliases of
 * md2_set_winograd(1, .) + md2_set_conv_products(2 / 3, -1).  on + 8 (8 .. 11): as `on`, with the split kernels' 64-channel / 512-thread
 * workgroups wherever Cout % 64 == 0 (round 5: faster launch by launch, slower inside the step, so not the default -- DESIGN.md section 4; tests keep it exercised).
 * on + 32 (32 .. 35): as `on`, and handles with md2_config.precision = 2 (bf16 tensors) take a Winograd kernel on their 3x3 / stride-1 layers as well (round 6,
 * conv_wino3b_kernel: fp32 transforms, V and U rounded to bf16 once).  1.7 - 4 % of a step; the two extra roundings lower the gradient's agreement with the fp32 step, so it
 * is not the default.  The +8 / +16 / +32 forms are exclusive.
 * on + 16 (16 .. 19): as `on`, with the split Winograd kernel in its persistent, tile-looping form (round 6, conv_wino3p_kernel: the resident set of workgroups walks the launch's
 * items and the head of the next item rides in the last chunks of the current one; bit-identical results, measured equal launch by launch and 1 % behind in the step, so
 * not the default -- DESIGN.md section 4; a test keeps it exercised).  Process-wide; also MD2_NO_WINO=1 / MD2_WINO_MODE / MD2_WINO_MIN_BLOCKS in the environment at load time. */
int32_t md2_set_winograd(int32_t on, int32_t min_blocks);
/* How the fp32 products of the convolutions are formed (the role cuDNN's math mode plays behind Flux; the reference computes in fp32, src/Monodepth.jl:72).
 *   fwd_dgrad: the forward / input-gradient kernels -- the fused Winograd kernels and, since round 5, the direct implicit GEMM (1x1, stride 2 and its residue-class input
 *   gradients, 3x3 launches below the Winograd threshold; 64-channel-wide tiles);  wgrad: the weight-gradient kernels (Winograd-domain and per-tap).  -1 keeps the current value.
 *   0 = v_mfma_f32_32x32x2_f32, the fp32 MFMA.
 *   1 = (default) per launch whichever of the two full-accuracy forms 0 and 3 is faster at the launch's shape (the rule is in csrc/md2_internal.h, from in-model timings).
 *   3 = exact split: every fp32 operand is the exact sum of three bf16 pieces (hi + mid + lo = all 24 mantissa bits); a product group takes six
 *       v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped cross terms are below 2^-24 of a product, the size of fp32's own product rounding).  Same accuracy as
 *       the fp32 MFMA -- measured: disparities within 4.6e-6 of the oracle against 4.5e-6 -- on the bf16 matrix pipe, which co-executes with vector instructions.
 *   2 = REDUCED precision, opt-in, never a default: two pieces (hi + lo, ~16-bit operand mantissas), three MFMAs per product group.
 * Defaults: 1 / 1 (MD2_DEFAULT_WINO_PIECES / MD2_DEFAULT_WGRAD_PIECES in csrc/md2_internal.h; MD2_WINO_PIECES / MD2_WGRAD_PIECES in the environment at load time), i.e. full
 * fp32 operand accuracy in every product of md2_config.precision = 0.  Process-wide; the layout a handle's Winograd operands were written in is recorded per handle when they
 * are refreshed (md2_train_loss / md2_model_forward), so changing the mode at any time -- also between md2_train_loss and md2_backward, or from another handle or
 * thread -- cannot make a kernel misread them: the change applies from the next refresh. */
int32_t md2_set_conv_products(int32_t fwd_dgrad, int32_t wgrad);
int32_t md2_get_conv_products(int32_t* fwd_dgrad, int32_t* wgrad);
/* ADAM moments (Flux parameter layout, host) + running beta powers: the optimiser state the reference's BSON checkpoints
 * (src/Monodepth.jl:191-194) omit; with md2_params_get/set this gives true resume. */
int32_t md2_opt_state_get(md2_handle* h, float* host_m, float* host_v, int64_t n, double* beta1p, double* beta2p);
int32_t md2_opt_state_set(md2_handle* h, const float* host_m, const float* host_v, int64_t n, double beta1p, double beta2p);
/* automasking_loss(ssim, x, x[:,:,:,target,:]; source_ids), src/training.jl:9-11 -> [N][1][H][W] */
int32_t md2_automasking_loss(md2_handle* h, const float* x, float* out);

/* ---- data parallelism: one process per GPU, RCCL over xGMI, called by the library itself (the reference is single-GPU, the
 *      insertion point is its update!, src/Monodepth.jl:170).  Rank 0 draws a 128-byte id and passes it to the other ranks through the host's own channel;
 *      after md2_comm_init, md2_backward all-reduces the flat gradient buffer in up to 4 buckets on a communication stream while the rest of the pullback
 *      computes, and md2_train_step applies ADAM to the mean (grad_scale = 1/world).  BatchNorm statistics stay per replica (= the reference at batch B). */
int32_t md2_comm_unique_id(void* out_128_bytes);
int32_t md2_comm_init(md2_handle* h, int32_t rank, int32_t world, const void* unique_id_128_bytes);
int32_t md2_comm_destroy(md2_handle* h);
int32_t md2_comm_world(md2_handle* h, int32_t* rank, int32_t* world);
/* The up-to-four [lo, hi) slices (float offsets into the flat gradient buffer of md2_grad_buffer, in the order md2_backward completes and all-reduces them: decoders |
 * encoder layer 4 | layers 2-3 | stem + layer 1) for a configuration.  Host-only: needs neither a device nor a handle, so the exact tiling of [0, *count) can be
 * asserted on any machine (tests/test_host_and_abi.py does, for both encoders and every scale_levels subset). */
int32_t md2_comm_bucket_plan(const md2_config* cfg, int64_t* lo4, int64_t* hi4, int64_t* count);
int32_t md2_allreduce_grads(md2_handle* h);                 /* sum over ranks of the gradient buffer; no-op when md2_backward already reduced it */
int32_t md2_comm_allreduce(md2_handle* h, float* dev_buf, int64_t n);   /* in-place sum of any device buffer (loss logging, self-checks) */
int32_t md2_broadcast_state(md2_handle* h, int32_t root);   /* parameters, ADAM moments + beta powers, BatchNorm running statistics of `root` to every rank */

/* The loss half alone (src/training.jl:28-77) given disparities and poses; pullbacks may be NULL.
 * ddisp_i like disp_i; drvec/dtvec [2][N][3]. */
int32_t md2_loss_fwd_bwd(md2_handle* h, const float* x, const float* disp0, const float* disp1, const float* disp2,
                         const float* disp3, const float* disp4, const float* rvec, const float* tvec, const float* auto_loss, float* loss,
                         float* ddisp0, float* ddisp1, float* ddisp2, float* ddisp3, float* ddisp4, float* drvec, float* dtvec);

/* slow_depth's objective (src/simple_depth.jl:25-42, BASELINE config #1): mean(prediction_loss(warp(disp, x, Ps))) +
 * smooth_loss(disp) on ONE full-resolution disparity disp [N][1][H][W] (no sigmoid, no mean-normalisation, weight 1) and its
 * pullbacks (any may be NULL): ddisp [N][1][H][W], drvec/dtvec [2][N][3]. */
int32_t md2_simple_depth_loss(md2_handle* h, const float* x, const float* disp, const float* rvec, const float* tvec, float* loss,
                              float* ddisp, float* drvec, float* dtvec);

/* Use the library's internal side streams (default 1; MD2_ONE_STREAM=1 in the environment sets the default to 0): the weight-gradient
 * kernels beside the input-gradient / BatchNorm-pullback chain, and the pose decoder, the disparity heads, the loss's tail and the 1x1
 * shortcut branches beside the depth decoder / encoder.  Every side stream is forked from and joined back to the handle's stream by
 * event, so stream order on the handle's stream is all a caller sees.  0: everything on the handle's stream.  Results are identical up
 * to fp32-atomic ordering, and bit-identical in the bit-reproducible mode (md2_set_deterministic). */
int32_t md2_set_two_streams(md2_handle* h, int32_t on);

/* per-kernel-class device timing (HIP events on the handle's stream) for bench.py's roofline */
int32_t md2_profile_enable(md2_handle* h, int32_t on);
int32_t md2_profile_reset(md2_handle* h);
/* classes: 0 conv_igemm (fwd+dgrad), 1 conv_wgrad, 2 loss_fwd, 3 loss_bwd, 4 other */
int32_t md2_profile_get(md2_handle* h, int32_t cls, double* total_ms, int64_t* launches, double* flops, double* bytes);
/* FLOPs the matrix cores executed in class `cls` since md2_profile_reset: equals the algorithmic count of md2_profile_get except for the convolutions that took the
 * Winograd F(2x2,3x3) route (16 instead of 36 multiplies per 2x2 output tile) -- bench.py reports both fractions of the MFMA peak. */
int32_t md2_profile_get_executed(md2_handle* h, int32_t cls, double* flops_executed);

/* per-launch table of the profiled kernels (csv: class, ms, GFLOP, TFLOP/s, shape) */
int32_t md2_profile_dump(md2_handle* h, const char* path);

/* ---- op-level entry points (stateless; `stream` = hipStream_t or NULL), reference layout ------ */
/* SSIM()(x, y), src/utils.jl:25-39.  [planes][H][W] */
int32_t md2_ssim_fwd(const float* x, const float* y, float* out, int32_t planes, int32_t H, int32_t W, void* stream);
/* ... and its pullback: dout [planes][H][W] -> dx, dy (either may be NULL).  With the other *_bwd entry points below this is what a ChainRulesCore.rrule of the
 * L2 functions calls when Zygote stays in charge of the step (src/Monodepth.jl:170) instead of md2_train_step (monodepth2.jl_amd/julia/Monodepth2HIP.jl). */
int32_t md2_ssim_bwd(const float* x, const float* y, const float* dout, float* dx, float* dy, int32_t planes, int32_t H, int32_t W, void* stream);
/* photometric_loss(ssim, predicted, target), src/training.jl:1-7 -> [N][1][H][W] */
int32_t md2_photometric_loss_fwd(const float* pred, const float* target, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* pullback wrt `predicted` (the target is data): dout [N][1][H][W] -> dpred [N][C][H][W] */
int32_t md2_photometric_loss_bwd(const float* pred, const float* target, const float* dout, float* dpred, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* prediction_loss(ssim, predictions, target), src/training.jl:13-15; preds [k][N][C][H][W] */
int32_t md2_prediction_loss_fwd(const float* preds, int32_t k, const float* target, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* pullback wrt the k predictions: the gradient of `minimum` goes to the first arg-min; dpreds [k][N][C][H][W] */
int32_t md2_prediction_loss_bwd(const float* preds, int32_t k, const float* target, const float* dout, float* dpreds, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* _apply_mask(mask, warp_loss), src/training.jl:17-19: minimum(cat(mask, warp_loss; dims=3); dims=3) of two [N][1][H][W] maps = their elementwise minimum (n elements) */
int32_t md2_apply_mask(const float* mask, const float* warp_loss, float* out, int64_t n, void* stream);
/* disparity_to_depth, src/utils.jl:175-179 */
int32_t md2_disparity_to_depth(const float* d, float* out, int64_t n, double min_depth, double max_depth, void* stream);
/* Backproject(depth, invK), src/utils.jl:63-65: depth [N][P] -> [N][P][3] */
int32_t md2_backproject(const float* depth, const float* invK_dev, float* out, int32_t N, int32_t H, int32_t W, void* stream);
/* Project(points, K, R, t), src/utils.jl:95-99: -> uv [N][P][2] */
int32_t md2_project(const float* pts, const float* K_dev, const float* R, const float* t, float* uv, int32_t N, int32_t H, int32_t W, void* stream);
/* NNlib.grid_sample(input, grid; padding_mode), call site src/training.jl:56; border: 1 = :border, 0 = :zeros */
int32_t md2_grid_sample_fwd(const float* in, const float* grid, float* out, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t border, void* stream);
int32_t md2_grid_sample_bwd_grid(const float* in, const float* grid, const float* dout, float* dgrid, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t border, void* stream);
/* fused disparity -> depth -> Backproject -> Project -> grid_sample(:border)  (src/training.jl:48-57) */
int32_t md2_warp_fwd(const float* disp, const float* src, const float* K_dev, const float* invK_dev, const float* R, const float* t,
                     float* out, int32_t N, int32_t C, int32_t H, int32_t W, double min_depth, double max_depth, void* stream);
/* pullback of md2_warp_fwd wrt the disparity and the pose (the source image is data): dout [N][C][H][W] -> ddisp [N][H][W], dR [N][9] (column-major), dt [N][3] */
int32_t md2_warp_bwd(const float* disp, const float* src, const float* K_dev, const float* invK_dev, const float* R, const float* t, const float* dout,
                     float* ddisp, float* dR, float* dt, int32_t N, int32_t C, int32_t H, int32_t W, double min_depth, double max_depth, void* stream);
/* NNlib.upsample_bilinear(x; size), src/training.jl:45 */
int32_t md2_upsample_bilinear_fwd(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, void* stream);
int32_t md2_upsample_bilinear_bwd(const float* dout, float* dx, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, void* stream);
/* hat / so3_exp_map / composeT, src/utils.jl:101-141,181-188 */
int32_t md2_hat(const float* rvec, float* S, int32_t N, void* stream);
int32_t md2_so3_exp_map(const float* rvec, float* R, int32_t N, void* stream);
int32_t md2_so3_exp_map_bwd(const float* rvec, const float* dR, float* drvec, int32_t N, void* stream);
int32_t md2_composeT(const float* rvec, const float* t, int32_t invert, float* R, float* tout, int32_t N, void* stream);
/* smooth_loss(disparity, image), src/utils.jl:159-173 -> device scalar */
int32_t md2_smooth_loss(const float* disp, const float* image, float* out, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* pullback wrt the disparity (the image is data): dloss = upstream scalar on the device -> ddisp [N][H][W] */
int32_t md2_smooth_loss_bwd(const float* disp, const float* image, const float* dloss, float* ddisp, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* device memory for callers without an array package of their own ("device pointers it obtained from md2_alloc / md2_free") */
int32_t md2_alloc(void** ptr, int64_t bytes);
int32_t md2_free(void* ptr);
/* Flux Conv (true convolution, weights (kW,kH,Cin,Cout)) on NCHW tensors through the MFMA implicit-GEMM kernels.
 * pad_mode 0 = zeros, 1 = reflect-pad(1) then valid conv (DecoderBlock, src/depth_decoder.jl:1-5); act: 0 none 1 relu 2 elu 3 sigmoid */
int32_t md2_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int32_t N, int32_t Cin, int32_t H, int32_t W,
                       int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t pad_mode, int32_t act, void* stream);
/* pullbacks wrt input / weights / bias of the conv above (act = none); any of dx,dw,db may be NULL */
int32_t md2_conv2d_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int32_t N, int32_t Cin,
                       int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t pad_mode, void* stream);
/* Test switch (process-wide): 1 = md2_conv2d_fwd / md2_conv2d_bn_fwd / md2_conv2d_bwd run the kernels of md2_config.precision = 2 -- x, y, dy, dx as bf16 tensors, bf16 weight
 * copies, fp32 accumulation -- so that each of them can be held against a plain fp32 reference of the op on bf16-rounded inputs; 0 (default) = fp32 tensors.  Returns 2 on any other value.
 * Like md2_set_deterministic / md2_set_winograd / md2_set_conv_products this is an unsynchronised process-wide switch: set it from one thread while no call of the library is in flight
 * (a handle snapshots the product modes when it prepares a step, so a step never mixes them; the op-level entry points read the switches at call time). */
int32_t md2_set_op_storage(int32_t bf16);
/* The encoder's conv -> BatchNorm pair (ResNet.jl via src/model.jl:14): bias-free convolution whose epilogue also accumulates the training-BatchNorm batch
 * statistics; bn_mean / bn_invstd [Cout] = mean and 1/sqrt(biased var + 1e-5) over (N,H,W) of y */
int32_t md2_conv2d_bnstats_fwd(const float* x, const float* w, float* y, float* bn_mean, float* bn_invstd, int32_t N, int32_t Cin, int32_t H, int32_t W,
                               int32_t Cout, int32_t k, int32_t stride, int32_t pad, void* stream);
/* The direct kernels of the ResNet stem (7x7, stride 2, pad 3, no bias; ResidualNetwork(...), src/Monodepth.jl:117): x [N][C][H][W], w Flux [64][C][7][7]
 * -> z [N][64][H/2][W/2] (+ the BatchNorm statistics of z, optional); and its weight gradient from dz */
int32_t md2_stem_conv_fwd(const float* x, const float* w, float* z, float* bn_mean, float* bn_invstd, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int32_t md2_stem_conv_wgrad(const float* x, const float* dz, float* dw, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* The disparity heads (src/depth_decoder.jl:43): reflect-pad + 3x3 conv to ONE channel + sigmoid, Cin in {16,32,64,128,256}; x [N][Cin][H][W], w Flux [1][Cin][3][3],
 * b [1] -> y [N][1][H][W]; pullback of dy -> dx, dw, db (x_act = 2: x is an ELU output and dx is taken w.r.t. its pre-activation, as the last branch does) */
int32_t md2_head_conv_fwd(const float* x, const float* w, const float* b, float* y, int32_t N, int32_t Cin, int32_t H, int32_t W, void* stream);
int32_t md2_head_conv_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db, int32_t N, int32_t Cin, int32_t H,
                          int32_t W, int32_t x_act, void* stream);
/* Flux BatchNorm train mode (+ optional relu), NCHW */
int32_t md2_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t relu, void* stream);
int32_t md2_bn_train_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* MaxPool((3,3); pad=1, stride=2), NCHW */
int32_t md2_maxpool3s2_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int32_t md2_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* ADAM on raw buffers: step = 1-based iteration count */
int32_t md2_adam_raw(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, double lr, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
