"""Host-side mirror of the reference's call surface for the training-step hot path, on top of the C ABI.

Julia is not available in the build image, so this Python layer stands where the reference's Julia
module stands (the Julia `ccall` shim a maintainer would use is julia/Monodepth2HIP.jl).  Names,
argument order and meaning follow the reference (src/Monodepth.jl, model.jl, depth_decoder.jl,
pose_decoder.jl, training.jl, utils.jl); ids are 1-based like the reference.  Arrays are torch CUDA
float32 tensors whose shape is the REVERSE of the Julia shape (same bytes): Julia (W,H,C,N) <-> (N,C,H,W),
(W,H,C,L,N) <-> (N,L,C,H,W), rvec (3,N) <-> (N,3), R (3,3,N) <-> (N,3,3) stored column-major.

torch is used only for device memory, streams and torch.distributed; every computation is a HIP kernel
of libmd2hip.so.  There is no CPU path: without the library / a GPU every call raises.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import synthetic


def _check(rc, handle=None, what=""):
    if rc != 0:
        msg = _lib.lib().md2_last_error(handle)
        raise RuntimeError("libmd2hip %s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("expected a contiguous float32 CUDA tensor (the product path has no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _f32(t):
    return t.contiguous().float()


# --------------------------------------------------------------------------------------------
# Params / TrainCache  (src/Monodepth.jl:32-55)
# --------------------------------------------------------------------------------------------
@dataclass
class Params:
    target_size: Tuple[int, int]          # (width, height)
    batch_size: int
    min_depth: float = 0.1
    max_depth: float = 100.0
    disparity_smoothness: float = 1e-3
    frame_ids: List[int] = field(default_factory=lambda: [1, 2, 3])
    automasking: bool = True


@dataclass
class TrainCache:
    ssim: "SSIM"
    backprojections: "Backproject"
    projections: "Project"
    K: np.ndarray                         # 3x3
    invK: np.ndarray
    target_id: int
    source_ids: List[int]
    scales: List[float]


def _colmajor(M):
    return np.ascontiguousarray(np.asarray(M, np.float32).T).reshape(-1)


def _dev_mat(M):
    return torch.from_numpy(_colmajor(M)).cuda()


# --------------------------------------------------------------------------------------------
# L2 ops (src/utils.jl)
# --------------------------------------------------------------------------------------------
class SSIM:
    """SSIM()(x, y): (N,C,H,W) -> (N,C,H,W); lower = more similar (src/utils.jl:13-39)."""

    def __call__(self, x, y):
        x, y = _f32(x), _f32(y); N, Cc, H, W = x.shape; out = torch.empty_like(x)
        _check(_lib.lib().md2_ssim_fwd(_dev(x), _dev(y), _dev(out), N * Cc, H, W, _stream()), None, "ssim_fwd"); return out


class Backproject:
    """Backproject(; width, height)(depth (N,P), invK) -> (N,P,3)  (src/utils.jl:41-65)."""

    def __init__(self, width, height):
        self.width, self.height = width, height

    def __call__(self, depth, invK):
        depth = _f32(depth); N = depth.shape[0]; out = torch.empty(N, self.width * self.height, 3, device=depth.device)
        iK = invK if torch.is_tensor(invK) else _dev_mat(invK)
        _check(_lib.lib().md2_backproject(_dev(depth), _dev(iK), _dev(out), N, self.height, self.width, _stream()), None, "backproject"); return out


class Project:
    """Project(; width, height)(points (N,P,3), K, R (N,9 col-major), t (N,3)) -> uv (N,P,2) in [-1,1]  (src/utils.jl:67-99)."""

    def __init__(self, width, height):
        self.width, self.height = width, height

    def __call__(self, points, K, R, t):
        points = _f32(points); N = points.shape[0]; uv = torch.empty(N, self.width * self.height, 2, device=points.device)
        Kd = K if torch.is_tensor(K) else _dev_mat(K)
        _check(_lib.lib().md2_project(_dev(points), _dev(Kd), _dev(_f32(R)), _dev(_f32(t).reshape(N, 3)), _dev(uv), N, self.height, self.width, _stream()), None, "project")
        return uv


def hat(rvec):
    rvec = _f32(rvec); N = rvec.shape[0]; S = torch.empty(N, 9, device=rvec.device)
    _check(_lib.lib().md2_hat(_dev(rvec), _dev(S), N, _stream())); return S


def so3_exp_map(rvec):
    """rvec (N,3) -> R (N,9) column-major  (src/utils.jl:101-117)."""
    rvec = _f32(rvec); N = rvec.shape[0]; R = torch.empty(N, 9, device=rvec.device)
    _check(_lib.lib().md2_so3_exp_map(_dev(rvec), _dev(R), N, _stream())); return R


def so3_exp_map_pullback(rvec, dR):
    rvec = _f32(rvec); N = rvec.shape[0]; dr = torch.empty(N, 3, device=rvec.device)
    _check(_lib.lib().md2_so3_exp_map_bwd(_dev(rvec), _dev(_f32(dR)), _dev(dr), N, _stream())); return dr


def composeT(rvec, t, invert):
    """-> (R (N,9), t (N,3))  (src/utils.jl:181-188)."""
    rvec = _f32(rvec); N = rvec.shape[0]; R = torch.empty(N, 9, device=rvec.device); to = torch.empty(N, 3, device=rvec.device)
    _check(_lib.lib().md2_composeT(_dev(rvec), _dev(_f32(t).reshape(N, 3)), int(bool(invert)), _dev(R), _dev(to), N, _stream())); return R, to


def smooth_loss(disparity, image):
    """disparity (N,H,W), image (N,C,H,W) -> 0-dim tensor  (src/utils.jl:159-173)."""
    d, im = _f32(disparity), _f32(image); N, H, W = d.shape; out = torch.empty((), device=d.device)
    _check(_lib.lib().md2_smooth_loss(_dev(d), _dev(im), _dev(out), N, im.shape[1], H, W, _stream())); return out


def disparity_to_depth(disparity, min_depth, max_depth):
    d = _f32(disparity); out = torch.empty_like(d)
    _check(_lib.lib().md2_disparity_to_depth(_dev(d), _dev(out), d.numel(), float(min_depth), float(max_depth), _stream())); return out


def grid_sample(input, grid, padding_mode="zeros"):
    """NNlib.grid_sample: input (N,C,H,W), grid (N,Ho,Wo,2)."""
    x, g = _f32(input), _f32(grid); N, Cc, H, W = x.shape; Ho, Wo = g.shape[1:3]; out = torch.empty(N, Cc, Ho, Wo, device=x.device)
    _check(_lib.lib().md2_grid_sample_fwd(_dev(x), _dev(g), _dev(out), N, Cc, H, W, Ho, Wo, int(padding_mode == "border"), _stream())); return out


def grid_sample_pullback_grid(input, grid, dout, padding_mode="zeros"):
    x, g, d = _f32(input), _f32(grid), _f32(dout); N, Cc, H, W = x.shape; Ho, Wo = g.shape[1:3]; dg = torch.empty_like(g)
    _check(_lib.lib().md2_grid_sample_bwd_grid(_dev(x), _dev(g), _dev(d), _dev(dg), N, Cc, H, W, Ho, Wo, int(padding_mode == "border"), _stream())); return dg


def upsample_bilinear(x, size):
    """NNlib.upsample_bilinear(x; size=(W,H)) — size given as (width, height) like the reference."""
    x = _f32(x); N, Cc, H, W = x.shape; Wo, Ho = size; out = torch.empty(N, Cc, Ho, Wo, device=x.device)
    _check(_lib.lib().md2_upsample_bilinear_fwd(_dev(x), _dev(out), N * Cc, H, W, Ho, Wo, _stream())); return out


def upsample_bilinear_pullback(dout, size_in):
    d = _f32(dout); N, Cc, Ho, Wo = d.shape; W, H = size_in; dx = torch.empty(N, Cc, H, W, device=d.device)
    _check(_lib.lib().md2_upsample_bilinear_bwd(_dev(d), _dev(dx), N * Cc, H, W, Ho, Wo, _stream())); return dx


def warp(disparity, source, K, invK, R, t, min_depth, max_depth):
    """The sequence `slow_depth` calls `warp` (undefined in the reference, SURVEY.md §2 note):
    disparity_to_depth -> Backproject -> Project -> grid_sample(:border), src/training.jl:48-57."""
    d, s = _f32(disparity), _f32(source); N, Cc, H, W = s.shape; out = torch.empty_like(s)
    Kd = K if torch.is_tensor(K) else _dev_mat(K); iK = invK if torch.is_tensor(invK) else _dev_mat(invK)
    _check(_lib.lib().md2_warp_fwd(_dev(d), _dev(s), _dev(Kd), _dev(iK), _dev(_f32(R)), _dev(_f32(t).reshape(N, 3)), _dev(out), N, Cc, H, W, float(min_depth), float(max_depth), _stream()))
    return out


# --------------------------------------------------------------------------------------------
# losses (src/training.jl:1-19)
# --------------------------------------------------------------------------------------------
def ssim_pullback(x, y, dout):
    """Pullback of SSIM()(x, y): dout (N,C,H,W) -> (dx, dy)."""
    x, y, g = _f32(x), _f32(y), _f32(dout); N, Cc, H, W = x.shape; dx, dy = torch.empty_like(x), torch.empty_like(x)
    _check(_lib.lib().md2_ssim_bwd(_dev(x), _dev(y), _dev(g), _dev(dx), _dev(dy), N * Cc, H, W, _stream())); return dx, dy


def photometric_loss_pullback(predicted, target, dout):
    """Pullback of photometric_loss wrt `predicted` (the target is data): dout (N,1,H,W) -> (N,C,H,W)."""
    p, t, g = _f32(predicted), _f32(target), _f32(dout); N, Cc, H, W = p.shape; dp = torch.empty_like(p)
    _check(_lib.lib().md2_photometric_loss_bwd(_dev(p), _dev(t), _dev(g), _dev(dp), N, Cc, H, W, _stream())); return dp


def prediction_loss_pullback(predictions, target, dout):
    """Pullback of prediction_loss wrt its predictions: the gradient of the minimum goes to the first arg-min."""
    preds = _f32(torch.stack([_f32(p) for p in predictions], 0)); t, g = _f32(target), _f32(dout); k, N, Cc, H, W = preds.shape; dp = torch.empty_like(preds)
    _check(_lib.lib().md2_prediction_loss_bwd(_dev(preds), k, _dev(t), _dev(g), _dev(dp), N, Cc, H, W, _stream())); return [dp[j] for j in range(k)]


def smooth_loss_pullback(disparity, image, dloss=1.0):
    """Pullback of smooth_loss wrt the disparity (the image is data)."""
    d, im = _f32(disparity), _f32(image); N, H, W = d.shape; dd = torch.empty_like(d)
    g = _f32(dloss) if torch.is_tensor(dloss) else torch.full((1,), float(dloss), device=d.device)
    _check(_lib.lib().md2_smooth_loss_bwd(_dev(d), _dev(im), _dev(g), _dev(dd), N, im.shape[1], H, W, _stream())); return dd


def warp_pullback(disparity, source, K, invK, R, t, min_depth, max_depth, dout):
    """Pullback of `warp` wrt the disparity and the pose: -> (ddisparity like disparity, dR (N,9) column-major, dt (N,3))."""
    d, s, g = _f32(disparity), _f32(source), _f32(dout); N, Cc, H, W = s.shape
    Kd = K if torch.is_tensor(K) else _dev_mat(K); iK = invK if torch.is_tensor(invK) else _dev_mat(invK)
    dd = torch.empty_like(d); dR = torch.empty(N, 9, device=d.device); dt = torch.empty(N, 3, device=d.device)
    _check(_lib.lib().md2_warp_bwd(_dev(d), _dev(s), _dev(Kd), _dev(iK), _dev(_f32(R)), _dev(_f32(t).reshape(N, 3)), _dev(g), _dev(dd), _dev(dR), _dev(dt), N, Cc, H, W,
                                   float(min_depth), float(max_depth), _stream()))
    return dd, dR, dt


def photometric_loss(ssim, predicted, target, alpha=0.85):
    if abs(alpha - 0.85) > 1e-12:
        raise ValueError("the HIP kernel implements the reference's alpha = 0.85")
    p, t = _f32(predicted), _f32(target); N, Cc, H, W = p.shape; out = torch.empty(N, 1, H, W, device=p.device)
    _check(_lib.lib().md2_photometric_loss_fwd(_dev(p), _dev(t), _dev(out), N, Cc, H, W, _stream())); return out


def prediction_loss(ssim, predictions, target):
    preds = _f32(torch.stack([_f32(p) for p in predictions], 0)); t = _f32(target); k, N, Cc, H, W = preds.shape
    out = torch.empty(N, 1, H, W, device=t.device)
    _check(_lib.lib().md2_prediction_loss_fwd(_dev(preds), k, _dev(t), _dev(out), N, Cc, H, W, _stream())); return out


def automasking_loss(ssim, inputs, target, source_ids):
    """inputs (N,L,C,H,W); photometric of the unwarped sources, min over sources (src/training.jl:9-11)."""
    return prediction_loss(ssim, [inputs[:, i - 1].contiguous() for i in source_ids], target)


def _apply_mask(mask, warp_loss):
    """`_apply_mask` (src/training.jl:17-19): minimum over the channel-concatenation of the automask and the warp loss = their elementwise minimum (md2_apply_mask).
    Op-level call surface (the visualization map); the training step applies the mask inside the fused loss kernel (md2_train_loss, ties -> mask)."""
    m, w = _f32(mask), _f32(warp_loss)
    if m.shape != w.shape:
        raise ValueError("_apply_mask: mask %s and warp_loss %s differ in shape" % (tuple(m.shape), tuple(w.shape)))
    out = torch.empty_like(w)
    _check(_lib.lib().md2_apply_mask(_dev(m), _dev(w), _dev(out), m.numel(), _stream())); return out


# --------------------------------------------------------------------------------------------
# model (src/model.jl, src/depth_decoder.jl, src/pose_decoder.jl, ResNet.jl call sites)
# --------------------------------------------------------------------------------------------
@dataclass
class Pose:
    rvec: torch.Tensor   # (N,3)
    tvec: torch.Tensor   # (N,3)  == Julia (3,1,N)


class ResidualNetwork:
    """ResidualNetwork(n; in_channels, classes=nothing): the ResNet-18 (BasicBlock) and ResNet-50 (Bottleneck) trunks."""

    def __init__(self, depth=18, in_channels=3, classes=None):
        if depth not in (18, 50) or classes is not None:
            raise NotImplementedError("hot path implements the ResNet-18 / ResNet-50 trunks without the classifier head")
        self.depth, self.in_channels = depth, in_channels
        self.stages = synthetic.stages(depth)


class DepthDecoder:
    def __init__(self, encoder_channels, scale_levels):
        scale_levels = list(scale_levels)
        if len(scale_levels) > 5 or min(scale_levels) < 1 or max(scale_levels) > 5:
            raise ValueError("`scale_levels` should be at most of length 5 and have values in [1, 5] range.")   # src/depth_decoder.jl:27-29
        if any(b <= a for a, b in zip(scale_levels, scale_levels[1:])):
            raise ValueError("`scale_levels` must be ascending: branch i of the reference's constructor starts where level i-1 ended (src/depth_decoder.jl:38-46)")
        if tuple(encoder_channels) not in (synthetic.stages(18), synthetic.stages(50)):
            raise NotImplementedError("hot path implements the ResNet-18 / ResNet-50 stage channels (src/Monodepth.jl:117-118)")
        self.encoder_channels, self.scale_levels = list(encoder_channels), scale_levels


class PoseDecoder:
    def __init__(self, encoder_out_channels):
        if encoder_out_channels not in (512, 2048):
            raise NotImplementedError("hot path implements PoseDecoder(512) and PoseDecoder(2048)")
        self.encoder_out_channels = encoder_out_channels


PRECISIONS = {"f32": 0, "bf16": 1, "bf16_storage": 2}          # md2_config.precision


class Model:
    """Model(encoder, depth_decoder, pose_decoder); `model(x, source_ids, target_id)` -> (disparities, poses).

    Owns one libmd2hip handle (weights, ADAM moments, workspaces) per (batch, H, W) shape."""

    def __init__(self, encoder, depth_decoder, pose_decoder, lr=1e-4, precision="f32"):
        self.encoder, self.depth_decoder, self.pose_decoder = encoder, depth_decoder, pose_decoder
        if tuple(depth_decoder.encoder_channels) != tuple(encoder.stages) or pose_decoder.encoder_out_channels != encoder.stages[-1]:
            raise ValueError("decoder channels do not match the encoder's stages")
        if precision not in PRECISIONS:
            raise ValueError("precision must be 'f32' (the reference's), 'bf16' (mixed: bf16 MFMA operands, fp32 everything else) or 'bf16_storage' (activations and their gradients "
                             "stored in bf16, bf16 weight copies for the convolutions; fp32 master weights, accumulation, BatchNorm statistics, loss and ADAM)")
        self._h = None
        self.lr, self.precision = lr, precision
        self._shape = None
        self._params_host = None
        self._cfg_extra = dict(min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3, automasking=False, target_id=2, source_ids=(1, 3), scales=None)
        self._K = None
        self._testmode = False
        self._comm_spec = None

    # ADAM's eta lives in the handle (md2_config.lr, read by md2_adam_step / md2_train_step): assigning model.lr must reach it, also on a handle that already exists
    # (train(model=resumed, lr=new) used to keep stepping at the handle's creation-time rate)
    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        if getattr(self, "_h", None) is not None:
            _check(_lib.lib().md2_set_lr(self._h, self._lr), self._h, "set_lr")

    # -- handle management --------------------------------------------------------------------
    def _ensure(self, x, parameters: Optional[Params] = None, cache: Optional[TrainCache] = None, frames=None):
        N, L, Cc, H, W = x.shape
        if L != 3 or Cc != self.encoder.in_channels:
            raise ValueError("x must be (N,3,C,H,W) with C == encoder.in_channels")
        extra = dict(self._cfg_extra)
        if parameters is not None:
            extra.update(min_depth=parameters.min_depth, max_depth=parameters.max_depth,
                         disparity_smoothness=parameters.disparity_smoothness, automasking=bool(parameters.automasking))
        if cache is not None:                      # TrainCache.target_id / source_ids / scales (src/Monodepth.jl:44-55)
            frames = (cache.source_ids, cache.target_id)
            if len(cache.scales) != len(self.depth_decoder.scale_levels):
                raise ValueError("TrainCache.scales needs one entry per DepthDecoder scale level (zip(disparities, cache.scales), src/training.jl:42)")
            extra.update(scales=tuple(float(v) for v in cache.scales))
            # md2_config.scales: an all-zero array means "not given" (zero-initialised struct = the defaults), a 0 among other entries is taken as given.  A caller
            # whose scales are ALL 0 wants every smoothness weight disparity_smoothness * scale_i (src/training.jl:67) to be 0: say that through the other factor.
            if not any(extra["scales"]):
                extra.update(scales=None, disparity_smoothness=0.0)
        if frames is not None:
            sids, tid = [int(v) for v in frames[0]], int(frames[1])
            if len(sids) != 2 or sorted(sids + [tid]) != [1, 2, 3]:
                raise NotImplementedError("the hot path takes triplets: target_id and two source_ids, three distinct frames in 1..3 (any order; src/model.jl:22-35)")
            extra.update(target_id=tid, source_ids=tuple(sids))
        key = (N, Cc, H, W, tuple(sorted(extra.items())), x.device.index)
        if self._h is not None and key == self._shape:
            if cache is not None:                 # the reference reads cache.K / cache.invK on every call (src/training.jl:48-57)
                self.set_intrinsics(cache.K, cache.invK)
            return
        # a new shape / Params needs a new handle: carry the weights AND the optimiser / BatchNorm state across, or training would silently restart ADAM
        if self._h is not None and self._comm_spec is not None:
            # refuse BEFORE tearing anything down: destroying this rank's communicator alone would leave the other ranks hanging in their next collective
            raise RuntimeError("a new batch / resolution / Params needs a new handle, but an RCCL communicator is attached: comm_destroy() on every rank first, then comm_init again")
        saved = self.params() if self._h is not None else self._params_host
        state = self._opt_and_bn_state() if self._h is not None else None
        self.close()
        cfg = _lib.Md2Config(N, Cc, H, W, int(extra["automasking"]), x.device.index or 0, extra["min_depth"], extra["max_depth"], extra["disparity_smoothness"], self.lr,
                             self.encoder.depth, PRECISIONS[self.precision], int(extra["target_id"]), (C.c_int32 * 2)(*extra["source_ids"]),
                             (C.c_int32 * 5)(*self.depth_decoder.scale_levels), (C.c_double * 5)(*(extra["scales"] or ())))
        h = C.c_void_p()
        _check(_lib.lib().md2_create(C.byref(cfg), C.byref(h)), None, "md2_create")
        self._h, self._shape, self._cfg_extra = h, key, extra
        if saved is not None:
            self.load_params(saved)
        if state is not None:
            self._restore_opt_and_bn_state(state)
        if self._testmode:
            _check(_lib.lib().md2_set_testmode(self._h, 1), self._h)
        if cache is not None:
            self.set_intrinsics(cache.K, cache.invK)
        elif self._K is not None:
            self.set_intrinsics(*self._K)

    def _opt_and_bn_state(self):
        L = _lib.lib(); n = L.md2_params_count(self._h)
        mm = np.empty(n, np.float32); vv = np.empty(n, np.float32); b1 = C.c_double(); b2 = C.c_double()
        _check(L.md2_opt_state_get(self._h, mm.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), n, C.byref(b1), C.byref(b2)), self._h, "opt_state_get")
        return mm, vv, b1.value, b2.value, self.bn_state()

    def _restore_opt_and_bn_state(self, state):
        mm, vv, b1, b2, bn = state; L = _lib.lib()
        _check(L.md2_opt_state_set(self._h, mm.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), mm.size, float(b1), float(b2)), self._h, "opt_state_set")
        self.load_bn_state(bn)

    # -- Flux BatchNorm state: running mean / variance per BatchNorm (what `@save model_host` keeps besides the weights) -----------------------
    def bn_state(self):
        n = _lib.lib().md2_bn_state_count(self._h); out = np.empty(n, np.float32)
        _check(_lib.lib().md2_bn_state_get(self._h, out.ctypes.data_as(C.c_void_p), n), self._h, "bn_state_get"); return out

    def load_bn_state(self, st):
        st = np.ascontiguousarray(st, np.float32)
        _check(_lib.lib().md2_bn_state_set(self._h, st.ctypes.data_as(C.c_void_p), st.size), self._h, "bn_state_set")

    def close(self):
        if self._h is not None:
            _lib.lib().md2_destroy(self._h); self._h = None
        self._comm_spec = None

    # -- data parallelism: the library's own RCCL communicator (include/md2hip.h, md2_comm_*) -------------------------------------------
    def comm_init(self, rank, world, unique_id):
        """Attach an RCCL communicator: `unique_id` = the 128 bytes rank 0 got from `comm_unique_id()`, passed around by the host."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        _check(_lib.lib().md2_comm_init(self._h, int(rank), int(world), C.cast(buf, C.c_void_p)), self._h, "comm_init")
        self._comm_spec = (rank, world)

    def comm_world(self):
        """(rank, world) of the library's communicator as the library sees it (md2_comm_world); (0, 1) without one."""
        r = C.c_int32(0); w = C.c_int32(1)
        if self._h is not None:
            _check(_lib.lib().md2_comm_world(self._h, C.byref(r), C.byref(w)), self._h, "comm_world")
        return int(r.value), int(w.value)

    def comm_destroy(self):
        if self._h is not None and self._comm_spec is not None:
            _check(_lib.lib().md2_comm_destroy(self._h), self._h, "comm_destroy")
        self._comm_spec = None

    def broadcast_state(self, root=0):
        _lib.lib().md2_set_stream(self._h, _stream())
        _check(_lib.lib().md2_broadcast_state(self._h, int(root)), self._h, "broadcast_state")

    def allreduce_grads(self):
        _lib.lib().md2_set_stream(self._h, _stream())
        _check(_lib.lib().md2_allreduce_grads(self._h), self._h, "allreduce_grads")

    def comm_allreduce(self, t):
        _lib.lib().md2_set_stream(self._h, _stream())
        _check(_lib.lib().md2_comm_allreduce(self._h, _dev(t), t.numel()), self._h, "comm_allreduce"); return t

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_intrinsics(self, K, invK):
        self._K = (np.asarray(K, np.float64), np.asarray(invK, np.float64))
        if self._h is not None:
            k, ik = _colmajor(K), _colmajor(invK)
            _check(_lib.lib().md2_set_intrinsics(self._h, k.ctypes.data_as(C.c_void_p), ik.ctypes.data_as(C.c_void_p)), self._h)

    # -- Flux.params(model) -------------------------------------------------------------------
    def load_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        self._params_host = flat
        if self._h is not None:
            _lib.lib().md2_set_stream(self._h, _stream())
            _check(_lib.lib().md2_params_set(self._h, flat.ctypes.data_as(C.c_void_p), flat.size), self._h, "params_set")

    def params(self):
        if self._h is None:
            return self._params_host
        n = _lib.lib().md2_params_count(self._h); out = np.empty(n, np.float32)
        _check(_lib.lib().md2_params_get(self._h, out.ctypes.data_as(C.c_void_p), n), self._h, "params_get"); return out

    def grads(self):
        n = _lib.lib().md2_params_count(self._h); out = np.empty(n, np.float32)
        _check(_lib.lib().md2_grads_get(self._h, out.ctypes.data_as(C.c_void_p), n), self._h, "grads_get"); return out

    def grad_buffer(self):
        """The library's flat fp32 gradient buffer as a torch tensor (no copy) for the DP all-reduce."""
        p = C.c_void_p(); n = C.c_int64()
        _check(_lib.lib().md2_grad_buffer(self._h, C.byref(p), C.byref(n)), self._h)
        return _wrap_device_buffer(p.value, n.value, self._shape[-1] or 0)

    def param_buffer(self):
        p = C.c_void_p(); n = C.c_int64()
        _check(_lib.lib().md2_param_buffer(self._h, C.byref(p), C.byref(n)), self._h)
        return _wrap_device_buffer(p.value, n.value, self._shape[-1] or 0)

    # -- forward --------------------------------------------------------------------------------
    def __call__(self, x, source_ids, target_id):
        x = _f32(x); self._ensure(x, frames=(source_ids, target_id))
        N, _, _, H, W = x.shape
        L = _lib.lib(); L.md2_set_stream(self._h, _stream())
        ds = [torch.empty(N, 1, H >> (5 - l), W >> (5 - l), device=x.device) for l in self.depth_decoder.scale_levels]
        rvec = torch.empty(2, N, 3, device=x.device); tvec = torch.empty(2, N, 3, device=x.device)
        _check(L.md2_model_forward(self._h, _dev(x), *_pad5(ds), _dev(rvec), _dev(tvec)), self._h, "model_forward")
        return ds, [Pose(rvec[0], tvec[0]), Pose(rvec[1], tvec[1])]


def _pad5(tensors):
    """The C ABI takes one pointer per possible scale level (five); the ones past the configured levels are NULL."""
    return [_dev(t) if t is not None else None for t in tensors] + [None] * (5 - len(tensors))


def eval_poses(m: Model, x, source_ids=(1, 3), target_id=2):
    """eval_poses(m, features, source_ids, target_id), src/model.jl:22-35, from the frames x (N,3,C,H,W): the encoder and the pose decoder only (the stage features
    stay in the handle).  Returns [Pose(source 1 -> target), Pose(target -> source 3)] like the model call."""
    x = _f32(x); m._ensure(x, frames=(source_ids, target_id)); N = x.shape[0]
    L = _lib.lib(); L.md2_set_stream(m._h, _stream())
    rvec = torch.empty(2, N, 3, device=x.device); tvec = torch.empty(2, N, 3, device=x.device)
    _check(L.md2_eval_poses(m._h, _dev(x), _dev(rvec), _dev(tvec)), m._h, "eval_poses")
    return [Pose(rvec[0], tvec[0]), Pose(rvec[1], tvec[1])]


def eval_disparity(m: Model, x):
    """eval_disparity(m, x) = m.depth_decoder(m.encoder(x, Val(:stages))), src/model.jl:28.  x (n,C,H,W) single frames; the model's
    handle must exist (n <= its batch) or is created for batch n.  Returns the disparities of the decoder's scale levels, coarse -> fine."""
    x = _f32(x); n, Cc, H, W = x.shape
    if m._h is None or m._shape[1:4] != (Cc, H, W) or m._shape[0] < n:
        m._ensure(x.new_zeros((n, 3, Cc, H, W)))
    L = _lib.lib(); L.md2_set_stream(m._h, _stream())
    ds = [torch.empty(n, 1, H >> (5 - l), W >> (5 - l), device=x.device) for l in m.depth_decoder.scale_levels]
    _check(L.md2_eval_disparity(m._h, _dev(x), n, *_pad5(ds)), m._h, "eval_disparity")
    return ds


def save_checkpoint(m: Model, path):
    """Weights (what the reference's `@save ... model_host` keeps, src/Monodepth.jl:191-194) PLUS the ADAM moments and beta powers
    the reference drops, so training can resume exactly.  numpy .npz, parameters in Flux layout / documented flat order."""
    n = _lib.lib().md2_params_count(m._h)
    mm = np.empty(n, np.float32); vv = np.empty(n, np.float32); b1 = C.c_double(); b2 = C.c_double()
    _check(_lib.lib().md2_opt_state_get(m._h, mm.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), n, C.byref(b1), C.byref(b2)), m._h, "opt_state_get")
    np.savez(path, params=m.params(), adam_m=mm, adam_v=vv, beta1p=b1.value, beta2p=b2.value, bn_state=m.bn_state(), in_channels=m.encoder.in_channels, encoder_depth=m.encoder.depth, lr=m.lr)


def load_checkpoint(m: Model, path):
    ck = np.load(path)
    if int(ck["in_channels"]) != m.encoder.in_channels:
        raise ValueError("checkpoint was saved for in_channels=%d" % int(ck["in_channels"]))
    if "encoder_depth" in ck and int(ck["encoder_depth"]) != m.encoder.depth:
        raise ValueError("checkpoint was saved for ResidualNetwork(%d)" % int(ck["encoder_depth"]))
    m.load_params(ck["params"])
    if m._h is None:
        raise RuntimeError("load_checkpoint: run the model once (or call model._ensure) before restoring the optimiser state")
    mm = np.ascontiguousarray(ck["adam_m"], np.float32); vv = np.ascontiguousarray(ck["adam_v"], np.float32)
    _check(_lib.lib().md2_opt_state_set(m._h, mm.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), mm.size, float(ck["beta1p"]), float(ck["beta2p"])), m._h, "opt_state_set")
    if "bn_state" in ck:                  # BatchNorm running statistics: needed to reproduce test-mode inference (eval_image / eval_video, src/Monodepth.jl:209,237)
        m.load_bn_state(ck["bn_state"])


def testmode(m: Model, on=True):
    """Flux.testmode!(model) / trainmode!(model): BatchNorm uses its tracked running statistics (src/Monodepth.jl:126,209,237)."""
    m._testmode = bool(on)
    if m._h is not None:
        _check(_lib.lib().md2_set_testmode(m._h, int(bool(on))), m._h, "set_testmode")


def trainmode(m: Model):
    testmode(m, False)


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it; every rank passes the same bytes to Model.comm_init)."""
    buf = (C.c_char * 128)()
    _check(_lib.lib().md2_comm_unique_id(C.cast(buf, C.c_void_p)), None, "comm_unique_id")
    return bytes(buf)


def _need_handle(model, what):
    """The handle is created by the first call that sees a batch (`_ensure`); entry points that only take the handle must not pass NULL across the C ABI."""
    if model._h is None:
        raise RuntimeError("%s: the model has no device handle yet (run a forward / training step first)" % what)
    return model._h


def set_deterministic(on=True):
    """Bit-reproducible (ordered reductions, no split-K) mode of the whole library: for bisecting parity failures, much slower."""
    _check(_lib.lib().md2_set_deterministic(int(bool(on))), None, "set_deterministic")


def set_op_storage(bf16=False):
    """Test switch: the op-level convolutions (`conv2d`, `conv2d_pullback`, `conv2d_bn`) run on bf16 tensors with bf16 weight copies -- the kernels of precision='bf16_storage'."""
    _check(_lib.lib().md2_set_op_storage(int(bool(bf16))), None, "set_op_storage")


def set_winograd(on=True, min_blocks=-1):
    """Algorithm choice for the 3x3 / stride-1 convolutions: fused Winograd F(2x2,3x3) for launches of at least `min_blocks` workgroups (default policy: 48; -1 keeps the current value, -2 restores the default), or the direct
    implicit GEMM everywhere (on=False).  What cuDNN's heuristics decide for the reference's Flux `Conv` layers.  on = 2 / 3 pin one of the two Winograd kernels (tests).
    How the products are formed is `set_conv_products`' business (on = 4 / 5 remain as aliases of set_winograd(1) + set_conv_products(2 / 3))."""
    _check(_lib.lib().md2_set_winograd(int(on), int(min_blocks)), None, "set_winograd")


def set_conv_products(fwd_dgrad=-1, wgrad=-1):
    """How the convolutions form their fp32 products (md2_set_conv_products): 0 = the fp32 MFMA, 3 = the exact three-bf16-piece split of every operand (fp32-MFMA accuracy on
    the bf16 matrix pipe), 1 = (default) whichever of those two is faster per launch, 2 = two pieces (~16-bit operand mantissas: REDUCED precision, opt-in).  `fwd_dgrad`: Winograd forward / input-gradient kernels; `wgrad`: the weight
    gradients; -1 keeps.  The defaults carry full fp32 operand accuracy, like the reference's precision = f32 (src/Monodepth.jl:72)."""
    _check(_lib.lib().md2_set_conv_products(int(fwd_dgrad), int(wgrad)), None, "set_conv_products")


def get_conv_products():
    f = C.c_int32(0); w = C.c_int32(0)
    _check(_lib.lib().md2_get_conv_products(C.byref(f), C.byref(w)), None, "get_conv_products")
    return f.value, w.value


class _DevBuf:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def _wrap_device_buffer(ptr, n, device_index):
    return torch.as_tensor(_DevBuf(ptr, n), device="cuda:%d" % device_index)


# --------------------------------------------------------------------------------------------
# train_loss + the loop body (src/training.jl:21-78, src/Monodepth.jl:156-176)
# --------------------------------------------------------------------------------------------
def train_loss(model: Model, x, auto_loss, cache: TrainCache, parameters: Params, do_visualization=False):
    """-> (loss, vis_disparity, vis_warped, vis_loss); keeps the activations for `gradient`."""
    x = _f32(x); model._ensure(x, parameters, cache)
    L = _lib.lib(); L.md2_set_stream(model._h, _stream())
    if parameters.automasking and auto_loss is None:
        raise ValueError("parameters.automasking is set but auto_loss is None")
    vis_disparity = vis_warped = vis_loss = None
    if do_visualization:                      # src/training.jl:34-37, 72-75: host copies of the last scale's disparity / warps / loss map
        bn = model.bn_state()                                         # this extra train-mode forward must not track the BatchNorm statistics a second time with the same batch
        ds, poses = model(x, cache.source_ids, cache.target_id)       # BEFORE the training forward: it reuses the activation buffers
        model.load_bn_state(bn)
        dlast = ds[-1] if tuple(ds[-1].shape[-2:]) == tuple(x.shape[-2:]) else upsample_bilinear(ds[-1], (x.shape[-1], x.shape[-2]))     # src/training.jl:45
        vis_disparity = dlast.cpu()
        target_x = x[:, cache.target_id - 1].contiguous()
        warped = []
        for pose, sid in zip(poses, cache.source_ids):
            R, t = composeT(pose.rvec, pose.tvec, sid < cache.target_id)
            warped.append(warp(dlast, x[:, sid - 1].contiguous(), cache.K, cache.invK, R, t, parameters.min_depth, parameters.max_depth))
        wl = prediction_loss(cache.ssim, warped, target_x)
        if parameters.automasking and auto_loss is not None:
            wl = _apply_mask(_f32(auto_loss), wl)
        vis_warped, vis_loss = [w.cpu() for w in warped], wl.cpu()
    loss = torch.empty((), device=x.device)
    al = _f32(auto_loss) if auto_loss is not None else None
    model._keepalive = (x, al)
    _check(L.md2_train_loss(model._h, _dev(x), _dev(al), _dev(loss)), model._h, "train_loss")
    return loss, vis_disparity, vis_warped, vis_loss


def gradient(model: Model):
    """Zygote `gradient(theta) do train_loss(...) end` for the last train_loss call; result via model.grads()."""
    _lib.lib().md2_set_stream(model._h, _stream())
    _check(_lib.lib().md2_backward(model._h), model._h, "backward")


class ADAM:
    """Flux.Optimise.ADAM(eta) — the moments live in the model's handle."""

    def __init__(self, eta=1e-4):
        self.eta = eta


def update(optimizer: ADAM, model: Model, grad_scale=1.0):
    """Flux.Optimise.update!(optimizer, theta, grads) — the step size is the optimiser's eta."""
    _lib.lib().md2_set_stream(model._h, _stream())
    if optimizer is not None and optimizer.eta != model.lr:
        model.lr = optimizer.eta                      # (the property setter calls md2_set_lr)
    _check(_lib.lib().md2_adam_step(model._h, float(grad_scale)), model._h, "adam_step")


def train_step(model: Model, x, auto_loss, cache: TrainCache, parameters: Params, do_update=True):
    """One iteration of the loop body src/Monodepth.jl:156-176 (forward + backward [+ ADAM]); returns the device loss."""
    x = _f32(x); model._ensure(x, parameters, cache)
    L = _lib.lib(); L.md2_set_stream(model._h, _stream())
    loss = torch.empty((), device=x.device)
    al = _f32(auto_loss) if auto_loss is not None else None
    model._keepalive = (x, al)
    _check(L.md2_train_step(model._h, _dev(x), _dev(al), _dev(loss), int(do_update)), model._h, "train_step")
    return loss


def loss_fwd_bwd(model: Model, x, disps, rvec, tvec, auto_loss=None, want_grads=True):
    """Loss half alone (src/training.jl:28-77) on given disparities/poses; for parity tests."""
    x = _f32(x); L = _lib.lib(); L.md2_set_stream(model._h, _stream())
    ds = [_f32(d) for d in disps]; rvec, tvec = _f32(rvec), _f32(tvec)
    loss = torch.empty((), device=x.device)
    dds = [torch.empty_like(d) for d in ds] if want_grads else [None] * len(ds)
    drv = torch.empty_like(rvec) if want_grads else None; dtv = torch.empty_like(tvec) if want_grads else None
    al = _f32(auto_loss) if auto_loss is not None else None
    _check(L.md2_loss_fwd_bwd(model._h, _dev(x), *_pad5(ds), _dev(rvec), _dev(tvec), _dev(al), _dev(loss), *_pad5(dds), _dev(drv), _dev(dtv)), model._h, "loss_fwd_bwd")
    return loss, dds, drv, dtv


def simple_depth_loss(model: Model, x, disp, rvec, tvec, want_grads=True):
    """slow_depth's objective and pullbacks (src/simple_depth.jl:25-42): disp (N,1,H,W); rvec,tvec (2,N,3)."""
    x, disp, rvec, tvec = _f32(x), _f32(disp), _f32(rvec), _f32(tvec)
    L = _lib.lib(); L.md2_set_stream(model._h, _stream())
    loss = torch.empty((), device=x.device)
    dd = torch.empty_like(disp) if want_grads else None; dr = torch.empty_like(rvec) if want_grads else None; dt = torch.empty_like(tvec) if want_grads else None
    _check(L.md2_simple_depth_loss(model._h, _dev(x), _dev(disp), _dev(rvec), _dev(tvec), _dev(loss), _dev(dd), _dev(dr), _dev(dt)), model._h, "simple_depth_loss")
    return loss, dd, dr, dt


def slow_depth(x, K, invK, target_id=2, source_ids=(1, 3), min_depth=0.1, max_depth=100.0, iters=500, lr=3e-4, log_every=0):
    """slow_depth (src/simple_depth.jl:1-62): optimise a raw disparity map (init 0.5) and two poses (rvec=[0,0,0.01], tvec=0) for ONE
    triplet with ADAM(3e-4).  The reference calls an undefined `warp`; the sequence it stands for is fused in the loss kernel.
    x (1,3,C,H,W).  Returns (disp (1,1,H,W), [Pose, Pose], losses)."""
    x = _f32(x); N, _, Cc, H, W = x.shape
    enc = ResidualNetwork(18, in_channels=Cc)
    model = Model(enc, DepthDecoder(list(enc.stages), [2, 3, 4, 5]), PoseDecoder(512))
    params = Params(target_size=(W, H), batch_size=N, min_depth=min_depth, max_depth=max_depth, automasking=False)
    model._ensure(x, params, frames=(list(source_ids), target_id)); model.set_intrinsics(K, invK)
    dev = x.device
    disp = torch.full((N, 1, H, W), 0.5, device=dev)
    rvec = torch.zeros(2, N, 3, device=dev); rvec[:, :, 2] = 0.01
    tvec = torch.zeros(2, N, 3, device=dev)
    state = [(p, torch.zeros_like(p), torch.zeros_like(p)) for p in (disp, rvec, tvec)]
    losses = []
    for it in range(1, iters + 1):
        loss, dd, dr, dt = simple_depth_loss(model, x, disp, rvec, tvec)
        for (p, m, v), g in zip(state, (dd, dr, dt)):
            adam_raw(p, g, m, v, it, lr=lr)
        if log_every and (it % log_every == 0 or it == 1):
            losses.append(float(loss))
    model.close()
    return disp, [Pose(rvec[0], tvec[0]), Pose(rvec[1], tvec[1])], losses


def simple_depth(x, K=None, invK=None, **kw):
    """simple_depth() (src/simple_depth.jl:64-97) without the dataset / PNG plumbing: Depth10k intrinsics by default."""
    if K is None:
        from . import synthetic
        K, invK = synthetic.depth10k_intrinsics(x.shape[-1], x.shape[-2])
    return slow_depth(x, K, invK, **kw)


def set_two_streams(model: Model, on=True):
    """Overlap the weight-gradient kernels with the rest of the backward pass on a second stream (default on)."""
    _check(_lib.lib().md2_set_two_streams(_need_handle(model, "set_two_streams"), int(on)), model._h)


def profile(model: Model, on=True):
    _check(_lib.lib().md2_profile_enable(_need_handle(model, "profile"), int(on)), model._h)
    if on:
        _check(_lib.lib().md2_profile_reset(model._h), model._h)


def profile_dump(model: Model, path):
    _check(_lib.lib().md2_profile_dump(_need_handle(model, "profile_dump"), str(path).encode()), model._h)


def profile_get(model: Model, cls):
    ms = C.c_double(); n = C.c_int64(); fl = C.c_double(); by = C.c_double()
    _check(_lib.lib().md2_profile_get(_need_handle(model, "profile_get"), cls, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), model._h)
    ex = C.c_double()
    _check(_lib.lib().md2_profile_get_executed(model._h, cls, C.byref(ex)), model._h)
    return dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value, flops_executed=ex.value)


# --------------------------------------------------------------------------------------------
# layer-level ops used by the parity tests (Flux Conv / BatchNorm / MaxPool semantics on NCHW tensors)
# --------------------------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, pad=0, pad_mode="zeros", act="none"):
    x, w = _f32(x), _f32(w); N, Cin, H, W = x.shape; Cout, _, k, _ = w.shape
    pm = 1 if pad_mode == "reflect" else 0; pe = 1 if pm else pad
    Ho = (H + 2 * pe - k) // stride + 1; Wo = (W + 2 * pe - k) // stride + 1
    y = torch.empty(N, Cout, Ho, Wo, device=x.device)
    a = {"none": 0, "relu": 1, "elu": 2, "sigmoid": 3}[act]
    _check(_lib.lib().md2_conv2d_fwd(_dev(x), _dev(w), _dev(_f32(b)) if b is not None else None, _dev(y), N, Cin, H, W, Cout, k, stride, pad, pm, a, _stream()), None, "conv2d_fwd")
    return y


def conv2d_pullback(x, w, dy, stride=1, pad=0, pad_mode="zeros"):
    x, w, dy = _f32(x), _f32(w), _f32(dy); N, Cin, H, W = x.shape; Cout, _, k, _ = w.shape
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(Cout, device=x.device)
    _check(_lib.lib().md2_conv2d_bwd(_dev(x), _dev(w), _dev(dy), _dev(dx), _dev(dw), _dev(db), N, Cin, H, W, Cout, k, stride, pad, 1 if pad_mode == "reflect" else 0, _stream()), None, "conv2d_bwd")
    return dx, dw, db


def batchnorm_train(x, gamma, beta, relu=False):
    x = _f32(x); N, Cc, H, W = x.shape; y = torch.empty_like(x)
    _check(_lib.lib().md2_bn_train_fwd(_dev(x), _dev(_f32(gamma)), _dev(_f32(beta)), _dev(y), N, Cc, H, W, int(relu), _stream()), None, "bn_fwd"); return y


def batchnorm_train_pullback(x, gamma, dy):
    x, dy = _f32(x), _f32(dy); N, Cc, H, W = x.shape; dx = torch.empty_like(x); dg = torch.empty(Cc, device=x.device); db = torch.empty(Cc, device=x.device)
    _check(_lib.lib().md2_bn_train_bwd(_dev(x), _dev(_f32(gamma)), _dev(dy), _dev(dx), _dev(dg), _dev(db), N, Cc, H, W, _stream()), None, "bn_bwd"); return dx, dg, db


def maxpool3s2(x):
    x = _f32(x); N, Cc, H, W = x.shape; y = torch.empty(N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=x.device)
    _check(_lib.lib().md2_maxpool3s2_fwd(_dev(x), _dev(y), N, Cc, H, W, _stream()), None, "maxpool_fwd"); return y


def maxpool3s2_pullback(x, dy):
    x, dy = _f32(x), _f32(dy); N, Cc, H, W = x.shape; dx = torch.empty_like(x)
    _check(_lib.lib().md2_maxpool3s2_bwd(_dev(x), _dev(dy), _dev(dx), N, Cc, H, W, _stream()), None, "maxpool_bwd"); return dx


def adam_raw(p, g, m, v, step, lr=1e-4, grad_scale=1.0):
    _check(_lib.lib().md2_adam_raw(_dev(p), _dev(g), _dev(m), _dev(v), p.numel(), int(step), float(lr), float(grad_scale), _stream()), None, "adam_raw")


def conv2d_bnstats(x, w, stride=1, pad=0):
    """Encoder conv (no bias) + the training-BatchNorm statistics its epilogue accumulates: -> (y, mean[Cout], invstd[Cout])."""
    x, w = _f32(x), _f32(w); N, Cin, H, W = x.shape; Cout, _, k, _ = w.shape
    Ho = (H + 2 * pad - k) // stride + 1; Wo = (W + 2 * pad - k) // stride + 1
    y = torch.empty(N, Cout, Ho, Wo, device=x.device); mean = torch.empty(Cout, device=x.device); invstd = torch.empty(Cout, device=x.device)
    _check(_lib.lib().md2_conv2d_bnstats_fwd(_dev(x), _dev(w), _dev(y), _dev(mean), _dev(invstd), N, Cin, H, W, Cout, k, stride, pad, _stream()), None, "conv2d_bnstats_fwd")
    return y, mean, invstd


def stem_conv(x, w, want_stats=True):
    """The ResNet stem through its direct kernel: x (N,C,H,W), w (64,C,7,7) Flux layout -> (z (N,64,H/2,W/2), mean, invstd)."""
    x, w = _f32(x), _f32(w); N, Cc, H, W = x.shape
    z = torch.empty(N, 64, H // 2, W // 2, device=x.device); mean = torch.empty(64, device=x.device) if want_stats else None; invstd = torch.empty(64, device=x.device) if want_stats else None
    _check(_lib.lib().md2_stem_conv_fwd(_dev(x), _dev(w), _dev(z), _dev(mean), _dev(invstd), N, Cc, H, W, _stream()), None, "stem_conv_fwd")
    return z, mean, invstd


def stem_conv_wgrad(x, dz):
    x, dz = _f32(x), _f32(dz); N, Cc, H, W = x.shape; dw = torch.empty(64, Cc, 7, 7, device=x.device)
    _check(_lib.lib().md2_stem_conv_wgrad(_dev(x), _dev(dz), _dev(dw), N, Cc, H, W, _stream()), None, "stem_conv_wgrad"); return dw


def head_conv(x, w, b):
    """Disparity head: sigmoid(conv3x3(pad_reflect(x)) + b), x (N,Cin,H,W), w (1,Cin,3,3), b (1,) -> (N,1,H,W)."""
    x, w, b = _f32(x), _f32(w), _f32(b); N, Cin, H, W = x.shape; y = torch.empty(N, 1, H, W, device=x.device)
    _check(_lib.lib().md2_head_conv_fwd(_dev(x), _dev(w), _dev(b), _dev(y), N, Cin, H, W, _stream()), None, "head_conv_fwd"); return y


def head_conv_pullback(x, w, y, dy, x_act="none"):
    x, w, y, dy = _f32(x), _f32(w), _f32(y), _f32(dy); N, Cin, H, W = x.shape
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(1, device=x.device)
    _check(_lib.lib().md2_head_conv_bwd(_dev(x), _dev(w), _dev(y), _dev(dy), _dev(dx), _dev(dw), _dev(db), N, Cin, H, W, {"none": 0, "elu": 2}[x_act], _stream()), None, "head_conv_bwd")
    return dx, dw, db
