"""ctypes binding of libmd2hip.so (include/md2hip.h).  There is NO fallback: if the HIP library is
missing or fails to load, importing the product API raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmd2hip.so")
_lib = None


class Md2Config(C.Structure):
    _fields_ = [("batch", C.c_int32), ("channels", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("automasking", C.c_int32), ("device", C.c_int32),
                ("min_depth", C.c_double), ("max_depth", C.c_double), ("disparity_smoothness", C.c_double), ("lr", C.c_double),
                ("encoder_depth", C.c_int32), ("precision", C.c_int32), ("target_id", C.c_int32), ("source_ids", C.c_int32 * 2),
                ("scale_levels", C.c_int32 * 5), ("scales", C.c_double * 5)]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree library (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"])
    subprocess.check_call(["make", "-C", src, "-j4", "-s"])
    return LIB_PATH


P = C.c_void_p
I32 = C.c_int32
I64 = C.c_int64
F64 = C.c_double

_SIGS = {
    "md2_create": (I32, [C.POINTER(Md2Config), C.POINTER(P)]),
    "md2_destroy": (I32, [P]),
    "md2_last_error": (C.c_char_p, [P]),
    "md2_set_stream": (I32, [P, P]),
    "md2_synchronize": (I32, [P]),
    "md2_set_intrinsics": (I32, [P, P, P]),
    "md2_params_count": (I64, [P]),
    "md2_params_set": (I32, [P, P, I64]),
    "md2_params_get": (I32, [P, P, I64]),
    "md2_grads_get": (I32, [P, P, I64]),
    "md2_grad_buffer": (I32, [P, C.POINTER(P), C.POINTER(I64)]),
    "md2_param_buffer": (I32, [P, C.POINTER(P), C.POINTER(I64)]),
    "md2_model_forward": (I32, [P] * 9),
    "md2_train_loss": (I32, [P, P, P, P]),
    "md2_backward": (I32, [P]),
    "md2_adam_step": (I32, [P, C.c_float]),
    "md2_train_step": (I32, [P, P, P, P, I32]),
    "md2_automasking_loss": (I32, [P, P, P]),
    "md2_eval_disparity": (I32, [P, P, I32, P, P, P, P, P]),
    "md2_opt_state_get": (I32, [P, P, P, I64, C.POINTER(F64), C.POINTER(F64)]),
    "md2_opt_state_set": (I32, [P, P, P, I64, F64, F64]),
    "md2_eval_poses": (I32, [P, P, P, P]),
    "md2_loss_fwd_bwd": (I32, [P] * 18),
    "md2_simple_depth_loss": (I32, [P] * 9),
    "md2_set_two_streams": (I32, [P, I32]),
    "md2_profile_enable": (I32, [P, I32]),
    "md2_profile_reset": (I32, [P]),
    "md2_profile_get": (I32, [P, I32, C.POINTER(F64), C.POINTER(I64), C.POINTER(F64), C.POINTER(F64)]),
    "md2_profile_get_executed": (I32, [P, I32, C.POINTER(F64)]),
    "md2_profile_dump": (I32, [P, C.c_char_p]),
    "md2_ssim_fwd": (I32, [P, P, P, I32, I32, I32, P]),
    "md2_ssim_bwd": (I32, [P, P, P, P, P, I32, I32, I32, P]),
    "md2_photometric_loss_fwd": (I32, [P, P, P, I32, I32, I32, I32, P]),
    "md2_photometric_loss_bwd": (I32, [P, P, P, P, I32, I32, I32, I32, P]),
    "md2_prediction_loss_bwd": (I32, [P, I32, P, P, P, I32, I32, I32, I32, P]),
    "md2_warp_bwd": (I32, [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, F64, F64, P]),
    "md2_smooth_loss_bwd": (I32, [P, P, P, P, I32, I32, I32, I32, P]),
    "md2_alloc": (I32, [P, I64]),
    "md2_free": (I32, [P]),
    "md2_prediction_loss_fwd": (I32, [P, I32, P, P, I32, I32, I32, I32, P]),
    "md2_apply_mask": (I32, [P, P, P, I64, P]),
    "md2_disparity_to_depth": (I32, [P, P, I64, F64, F64, P]),
    "md2_backproject": (I32, [P, P, P, I32, I32, I32, P]),
    "md2_project": (I32, [P, P, P, P, P, I32, I32, I32, P]),
    "md2_grid_sample_fwd": (I32, [P, P, P, I32, I32, I32, I32, I32, I32, I32, P]),
    "md2_grid_sample_bwd_grid": (I32, [P, P, P, P, I32, I32, I32, I32, I32, I32, I32, P]),
    "md2_warp_fwd": (I32, [P, P, P, P, P, P, P, I32, I32, I32, I32, F64, F64, P]),
    "md2_upsample_bilinear_fwd": (I32, [P, P, I32, I32, I32, I32, I32, P]),
    "md2_upsample_bilinear_bwd": (I32, [P, P, I32, I32, I32, I32, I32, P]),
    "md2_hat": (I32, [P, P, I32, P]),
    "md2_so3_exp_map": (I32, [P, P, I32, P]),
    "md2_so3_exp_map_bwd": (I32, [P, P, P, I32, P]),
    "md2_composeT": (I32, [P, P, I32, P, P, I32, P]),
    "md2_smooth_loss": (I32, [P, P, P, I32, I32, I32, I32, P]),
    "md2_conv2d_fwd": (I32, [P, P, P, P] + [I32] * 10 + [P]),
    "md2_conv2d_bwd": (I32, [P, P, P, P, P, P] + [I32] * 9 + [P]),
    "md2_bn_train_fwd": (I32, [P, P, P, P, I32, I32, I32, I32, I32, P]),
    "md2_bn_train_bwd": (I32, [P, P, P, P, P, P, I32, I32, I32, I32, P]),
    "md2_maxpool3s2_fwd": (I32, [P, P, I32, I32, I32, I32, P]),
    "md2_maxpool3s2_bwd": (I32, [P, P, P, I32, I32, I32, I32, P]),
    "md2_adam_raw": (I32, [P, P, P, P, I64, I32, F64, C.c_float, P]),
    "md2_set_lr": (I32, [P, F64]),
    "md2_set_testmode": (I32, [P, I32]),
    "md2_bn_state_count": (I64, [P]),
    "md2_bn_state_get": (I32, [P, P, I64]),
    "md2_bn_state_set": (I32, [P, P, I64]),
    "md2_set_deterministic": (I32, [I32]),
    "md2_set_op_storage": (I32, [I32]),
    "md2_set_winograd": (I32, [I32, I32]),
    "md2_set_conv_products": (I32, [I32, I32]),
    "md2_get_conv_products": (I32, [C.POINTER(I32), C.POINTER(I32)]),
    "md2_comm_unique_id": (I32, [P]),
    "md2_comm_init": (I32, [P, I32, I32, P]),
    "md2_comm_destroy": (I32, [P]),
    "md2_comm_world": (I32, [P, C.POINTER(I32), C.POINTER(I32)]),
    "md2_comm_bucket_plan": (I32, [C.POINTER(Md2Config), C.POINTER(I64), C.POINTER(I64), C.POINTER(I64)]),
    "md2_allreduce_grads": (I32, [P]),
    "md2_comm_allreduce": (I32, [P, P, I64]),
    "md2_broadcast_state": (I32, [P, I32]),
    "md2_conv2d_bnstats_fwd": (I32, [P, P, P, P, P] + [I32] * 8 + [P]),
    "md2_stem_conv_fwd": (I32, [P, P, P, P, P, I32, I32, I32, I32, P]),
    "md2_stem_conv_wgrad": (I32, [P, P, P, I32, I32, I32, I32, P]),
    "md2_head_conv_fwd": (I32, [P, P, P, P, I32, I32, I32, I32, P]),
    "md2_head_conv_bwd": (I32, [P] * 7 + [I32] * 5 + [P]),
}

EXPORTS = sorted(_SIGS)


def lib():
    """Load libmd2hip.so; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmd2hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(make -C monodepth2.jl_amd/csrc). There is no CPU fallback for the product path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)          # AttributeError here == missing export: fail loudly
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
