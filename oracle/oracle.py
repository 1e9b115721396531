"""ctypes front-end of the CPU oracle (oracle/libmd2oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product package.  Arrays are numpy, in the REFERENCE layout: a Julia (W,H,C,N)
column-major array is passed as a C-contiguous numpy array of shape (N,C,H,W).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libmd2oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("md2_oracle.cpp", "md2_oracle.h", "md2_oracle_ops.hpp", "md2_oracle_net.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmd2oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.o_smooth_loss_f32.restype = C.c_double
        _LIB.o_smooth_loss_f64.restype = C.c_double
        _LIB.o_smooth_loss_bwd_f32.restype = C.c_double
        _LIB.o_smooth_loss_bwd_f64.restype = C.c_double
        _LIB.o_net_create.restype = C.c_void_p
        _LIB.o_net_create_depth.restype = C.c_void_p
        _LIB.o_net_create_levels.restype = C.c_void_p
        _LIB.o_net_nparams.restype = C.c_long
        _LIB.o_net_bn_state_count.restype = C.c_long
        _LIB.o_net_train_loss.restype = C.c_float
        _LIB.o_loss_fwd_bwd.restype = C.c_float
    return _LIB


class OCfg(C.Structure):
    _fields_ = [("N", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("automasking", C.c_int), ("simple", C.c_int),
                ("min_depth", C.c_double), ("max_depth", C.c_double), ("smoothness", C.c_double),
                ("K", C.c_float * 9), ("invK", C.c_float * 9), ("target_id", C.c_int), ("source_ids", C.c_int * 2),
                ("scale_levels", C.c_int * 5), ("scales", C.c_double * 5)]


def make_cfg(N, Cc, H, W, K, invK, automasking=False, min_depth=0.1, max_depth=100.0, smoothness=1e-3, simple=False, target_id=2, source_ids=(1, 3), scale_levels=None, scales=None):
    """K, invK: 3x3 numpy matrices (mathematical row/col indexing); stored column-major like Julia.  target_id / source_ids: TrainCache's, 1-based.
    scale_levels: the DepthDecoder's (None = 2:5); scales: TrainCache.scales (None = 1 / 2^(5 - level))."""
    c = OCfg()
    for i, l in enumerate(scale_levels or ()):
        c.scale_levels[i] = int(l)
    if scales is not None and not any(scales):          # all zeros: every smoothness weight smoothness * scale_i (src/training.jl:67) is 0 -- said through the other factor (an all-zero array means "not given")
        scales, smoothness = None, 0.0
    for i, v in enumerate(scales or ()):
        c.scales[i] = float(v)
    c.target_id = int(target_id); c.source_ids[0], c.source_ids[1] = int(source_ids[0]), int(source_ids[1])
    c.N, c.C, c.H, c.W, c.automasking, c.simple = N, Cc, H, W, int(automasking), int(simple)
    c.min_depth, c.max_depth, c.smoothness = min_depth, max_depth, smoothness
    Kc = np.asarray(K, np.float32).T.reshape(-1)       # column-major flatten
    iKc = np.asarray(invK, np.float32).T.reshape(-1)
    for i in range(9):
        c.K[i] = float(Kc[i]); c.invK[i] = float(iKc[i])
    return c


def _sfx(a):
    return "f64" if a.dtype == np.float64 else "f32"


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt=None):
    return np.ascontiguousarray(a, dtype=dt if dt is not None else a.dtype)


def colmajor3(M):
    """(…,3,3) matrices -> flat column-major (…,9) as Julia stores them."""
    M = np.asarray(M)
    return np.ascontiguousarray(np.swapaxes(M, -1, -2)).reshape(M.shape[:-2] + (9,))


def from_colmajor3(f):
    f = np.asarray(f)
    return np.swapaxes(f.reshape(f.shape[:-1] + (3, 3)), -1, -2)


# ---------------- L2 ops ----------------
def hat(r):
    r = _c(r); N = r.shape[0]; S = np.zeros((N, 9), r.dtype)
    getattr(lib(), "o_hat_" + _sfx(r))(_p(r), _p(S), N); return from_colmajor3(S)


def hat_pullback(d):
    d = _c(colmajor3(d)); N = d.shape[0]; dv = np.zeros((N, 3), d.dtype)
    getattr(lib(), "o_hat_pullback_" + _sfx(d))(_p(d), _p(dv), N); return dv


def so3_exp_map(r):
    r = _c(r); N = r.shape[0]; R = np.zeros((N, 9), r.dtype)
    getattr(lib(), "o_so3_exp_map_" + _sfx(r))(_p(r), _p(R), N); return from_colmajor3(R)


def so3_exp_map_bwd(r, dR):
    r = _c(r); N = r.shape[0]; g = _c(colmajor3(dR).astype(r.dtype)); dr = np.zeros((N, 3), r.dtype)
    getattr(lib(), "o_so3_exp_map_bwd_" + _sfx(r))(_p(r), _p(g), _p(dr), N); return dr


def composeT(r, t, invert):
    r = _c(r); t = _c(t.reshape(-1, 3).astype(r.dtype)); N = r.shape[0]
    R = np.zeros((N, 9), r.dtype); to = np.zeros((N, 3), r.dtype)
    getattr(lib(), "o_composeT_" + _sfx(r))(_p(r), _p(t), int(invert), _p(R), _p(to), N); return from_colmajor3(R), to


def composeT_bwd(r, t, invert, dR, dto):
    r = _c(r); t = _c(t.reshape(-1, 3).astype(r.dtype)); N = r.shape[0]
    g = _c(colmajor3(dR).astype(r.dtype)); dto = _c(dto.astype(r.dtype)); dr = np.zeros((N, 3), r.dtype); dt = np.zeros((N, 3), r.dtype)
    getattr(lib(), "o_composeT_bwd_" + _sfx(r))(_p(r), _p(t), int(invert), _p(g), _p(dto), _p(dr), _p(dt), N); return dr, dt


def ssim(x, y):
    x = _c(x); y = _c(y.astype(x.dtype)); N, Cc, H, W = x.shape; out = np.zeros_like(x)
    getattr(lib(), "o_ssim_" + _sfx(x))(_p(x), _p(y), _p(out), N * Cc, H, W); return out


def ssim_bwd(x, y, dout):
    x = _c(x); y = _c(y.astype(x.dtype)); dout = _c(dout.astype(x.dtype)); N, Cc, H, W = x.shape
    dx = np.zeros_like(x); dy = np.zeros_like(x)
    getattr(lib(), "o_ssim_bwd_" + _sfx(x))(_p(x), _p(y), _p(dout), _p(dx), _p(dy), N * Cc, H, W); return dx, dy


def photometric_loss(pred, tgt):
    pred = _c(pred); tgt = _c(tgt.astype(pred.dtype)); N, Cc, H, W = pred.shape; out = np.zeros((N, 1, H, W), pred.dtype)
    getattr(lib(), "o_photometric_" + _sfx(pred))(_p(pred), _p(tgt), _p(out), N, Cc, H, W); return out


def photometric_loss_bwd(pred, tgt, dout):
    pred = _c(pred); tgt = _c(tgt.astype(pred.dtype)); dout = _c(dout.astype(pred.dtype)); N, Cc, H, W = pred.shape; d = np.zeros_like(pred)
    getattr(lib(), "o_photometric_bwd_" + _sfx(pred))(_p(pred), _p(tgt), _p(dout), _p(d), N, Cc, H, W); return d


def prediction_loss(preds, tgt):
    stack = _c(np.stack([photometric_loss(p, tgt) for p in preds], 0)); k = stack.shape[0]; n = stack[0].size
    out = np.zeros_like(stack[0]); arg = np.zeros(stack[0].shape, np.int32)
    getattr(lib(), "o_min_stack_" + _sfx(stack))(_p(stack), k, C.c_long(n), _p(out), _p(arg)); return out, arg


def smooth_loss(d, img):
    d = _c(d); img = _c(img.astype(d.dtype)); N, H, W = d.shape; Cc = img.shape[1]
    return getattr(lib(), "o_smooth_loss_" + _sfx(d))(_p(d), _p(img), N, Cc, H, W)


def smooth_loss_bwd(d, img, dl=1.0):
    d = _c(d); img = _c(img.astype(d.dtype)); N, H, W = d.shape; Cc = img.shape[1]; dd = np.zeros_like(d)
    getattr(lib(), "o_smooth_loss_bwd_" + _sfx(d))(_p(d), _p(img), C.c_double(dl), _p(dd), N, Cc, H, W); return dd


def disparity_to_depth(d, mn, mx):
    d = _c(d); out = np.zeros_like(d)
    getattr(lib(), "o_disparity_to_depth_" + _sfx(d))(_p(d), _p(out), C.c_long(d.size), C.c_double(mn), C.c_double(mx)); return out


def backproject(depth, invK, H, W):
    """depth (N, P); invK 3x3 -> (N, P, 3)"""
    depth = _c(depth); N = depth.shape[0]; iK = _c(colmajor3(np.asarray(invK, depth.dtype))); out = np.zeros((N, H * W, 3), depth.dtype)
    getattr(lib(), "o_backproject_" + _sfx(depth))(_p(depth), _p(iK), _p(out), N, H, W); return out


def project(pts, K, R, t, H, W):
    """pts (N,P,3), K 3x3, R (N,3,3), t (N,3) -> uv (N,P,2)"""
    pts = _c(pts); N = pts.shape[0]; dt = pts.dtype
    Kc = _c(colmajor3(np.asarray(K, dt))); Rc = _c(colmajor3(np.asarray(R, dt))); t = _c(np.asarray(t, dt).reshape(N, 3)); uv = np.zeros((N, H * W, 2), dt)
    getattr(lib(), "o_project_" + _sfx(pts))(_p(pts), _p(Kc), _p(Rc), _p(t), _p(uv), N, H, W); return uv


def project_bwd(pts, K, R, t, duv, H, W):
    pts = _c(pts); N = pts.shape[0]; dt = pts.dtype
    Kc = _c(colmajor3(np.asarray(K, dt))); Rc = _c(colmajor3(np.asarray(R, dt))); t = _c(np.asarray(t, dt).reshape(N, 3)); duv = _c(duv.astype(dt))
    dp = np.zeros_like(pts); dR = np.zeros((N, 9), dt); dtt = np.zeros((N, 3), dt)
    getattr(lib(), "o_project_bwd_" + _sfx(pts))(_p(pts), _p(Kc), _p(Rc), _p(t), _p(duv), _p(dp), _p(dR), _p(dtt), N, H, W)
    return dp, from_colmajor3(dR), dtt


def grid_sample(inp, grid, border=True):
    """inp (N,C,H,W); grid (N,Ho,Wo,2) -> (N,C,Ho,Wo)"""
    inp = _c(inp); grid = _c(grid.astype(inp.dtype)); N, Cc, H, W = inp.shape; Ho, Wo = grid.shape[1:3]; out = np.zeros((N, Cc, Ho, Wo), inp.dtype)
    getattr(lib(), "o_grid_sample_" + _sfx(inp))(_p(inp), _p(grid), _p(out), N, Cc, H, W, Ho, Wo, int(border)); return out


def grid_sample_bwd(inp, grid, dout, border=True):
    inp = _c(inp); grid = _c(grid.astype(inp.dtype)); dout = _c(dout.astype(inp.dtype)); N, Cc, H, W = inp.shape; Ho, Wo = grid.shape[1:3]
    dgrid = np.zeros_like(grid); din = np.zeros_like(inp)
    getattr(lib(), "o_grid_sample_bwd_" + _sfx(inp))(_p(inp), _p(grid), _p(dout), _p(dgrid), _p(din), N, Cc, H, W, Ho, Wo, int(border)); return dgrid, din


def upsample_bilinear(x, Ho, Wo):
    x = _c(x); N, Cc, H, W = x.shape; out = np.zeros((N, Cc, Ho, Wo), x.dtype)
    getattr(lib(), "o_upsample_bilinear_" + _sfx(x))(_p(x), _p(out), N * Cc, H, W, Ho, Wo); return out


def upsample_bilinear_bwd(dout, H, W):
    dout = _c(dout); N, Cc, Ho, Wo = dout.shape; dx = np.zeros((N, Cc, H, W), dout.dtype)
    getattr(lib(), "o_upsample_bilinear_bwd_" + _sfx(dout))(_p(dout), _p(dx), N * Cc, H, W, Ho, Wo); return dx


def pad_reflect(x):
    x = _c(x); N, Cc, H, W = x.shape; out = np.zeros((N, Cc, H + 2, W + 2), x.dtype)
    getattr(lib(), "o_pad_reflect_" + _sfx(x))(_p(x), _p(out), N * Cc, H, W); return out


# ---------------- layer ops (fp32; weights in Flux layout (Cout,Cin,KH,KW) row-major, TRUE convolution) ----------------
def conv2d_fwd(x, w, b, stride, pad):
    x = _c(x, np.float32); w = _c(w, np.float32); N, Cin, H, W = x.shape; Cout, _, k, _ = w.shape
    Ho = (H + 2 * pad - k) // stride + 1; Wo = (W + 2 * pad - k) // stride + 1; y = np.zeros((N, Cout, Ho, Wo), np.float32)
    b = None if b is None else _c(b, np.float32)
    lib().o_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), N, Cin, H, W, Cout, k, stride, pad); return y


def conv2d_bwd(x, w, dy, stride, pad, need_dx=True):
    x = _c(x, np.float32); w = _c(w, np.float32); dy = _c(dy, np.float32); N, Cin, H, W = x.shape; Cout, _, k, _ = w.shape
    dx = np.zeros_like(x) if need_dx else None; dw = np.zeros_like(w); db = np.zeros(Cout, np.float32)
    lib().o_conv2d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, Cin, H, W, Cout, k, stride, pad); return dx, dw, db


def bn_train_fwd(x, gamma, beta):
    x = _c(x, np.float32); N, Cc, H, W = x.shape; y = np.zeros_like(x); mean = np.zeros(Cc, np.float32); var = np.zeros(Cc, np.float32)
    lib().o_bn_train_fwd(_p(x), _p(_c(gamma, np.float32)), _p(_c(beta, np.float32)), _p(y), _p(mean), _p(var), N, Cc, H, W); return y, mean, var


def bn_train_bwd(x, gamma, dy):
    x = _c(x, np.float32); dy = _c(dy, np.float32); N, Cc, H, W = x.shape; dx = np.zeros_like(x); dg = np.zeros(Cc, np.float32); db = np.zeros(Cc, np.float32)
    lib().o_bn_train_bwd(_p(x), _p(_c(gamma, np.float32)), _p(dy), _p(dx), _p(dg), _p(db), N, Cc, H, W); return dx, dg, db


def maxpool3s2_fwd(x):
    x = _c(x, np.float32); N, Cc, H, W = x.shape; y = np.zeros((N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), np.float32)
    lib().o_maxpool3s2_fwd(_p(x), _p(y), N, Cc, H, W); return y


def maxpool3s2_bwd(x, dy):
    x = _c(x, np.float32); dy = _c(dy, np.float32); N, Cc, H, W = x.shape; dx = np.zeros_like(x)
    lib().o_maxpool3s2_bwd(_p(x), _p(dy), _p(dx), N, Cc, H, W); return dx


# ---------------- network handle ----------------
def _pad5(arrays):
    return [_p(a) for a in arrays] + [None] * (5 - len(arrays))


class OracleNet:
    def __init__(self, channels=1, depth=18, scale_levels=(2, 3, 4, 5)):
        self.levels = [int(l) for l in scale_levels]
        self.h = C.c_void_p(lib().o_net_create_levels(channels, depth, (C.c_int * 6)(*self.levels))); self.C = channels; self.depth = depth
        if not self.h:
            raise ValueError("encoder depth must be 18 or 50")
        self.nparams = lib().o_net_nparams(self.h)

    def __del__(self):
        try:
            lib().o_net_destroy(self.h)
        except Exception:
            pass

    def layout(self):
        cap = 256; offs = (C.c_long * cap)(); cnts = (C.c_long * cap)(); kinds = (C.c_int * cap)()
        n = lib().o_net_layout(self.h, offs, cnts, kinds, cap)
        return [(offs[i], cnts[i], kinds[i]) for i in range(n)]

    def convs(self):
        cap = 128; d = (C.c_int * (cap * 6))(); n = lib().o_net_convs(self.h, d, cap)
        return [tuple(d[i * 6 + j] for j in range(6)) for i in range(n)]

    def set_params(self, p):
        p = _c(p, np.float32); assert p.size == self.nparams; lib().o_net_params_set(self.h, _p(p))

    def get_params(self):
        p = np.zeros(self.nparams, np.float32); lib().o_net_params_get(self.h, _p(p)); return p

    def get_grads(self):
        p = np.zeros(self.nparams, np.float32); lib().o_net_grads_get(self.h, _p(p)); return p

    def set_grads(self, g):
        g = _c(g, np.float32); lib().o_net_grads_set(self.h, _p(g))

    def model_forward(self, x, cfg):
        x = _c(x, np.float32); N, H, W = cfg.N, cfg.H, cfg.W
        ds = [np.zeros((N, 1, H >> (5 - l), W >> (5 - l)), np.float32) for l in self.levels]
        rvec = np.zeros((2, N, 3), np.float32); tvec = np.zeros((2, N, 3), np.float32)
        lib().o_net_model_forward(self.h, _p(x), C.byref(cfg), *_pad5(ds), _p(rvec), _p(tvec))
        return ds, rvec, tvec

    def train_loss(self, x, cfg, auto_loss=None, do_bwd=False):
        x = _c(x, np.float32); al = None if auto_loss is None else _c(auto_loss, np.float32)
        return float(lib().o_net_train_loss(self.h, _p(x), C.byref(cfg), _p(al), int(do_bwd)))

    def eval_disparity(self, x):
        x = _c(x, np.float32); N, Cc, H, W = x.shape
        ds = [np.zeros((N, 1, H >> (5 - l), W >> (5 - l)), np.float32) for l in self.levels]
        lib().o_net_eval_disparity(self.h, _p(x), N, Cc, H, W, *_pad5(ds)); return ds

    def opt_state(self):
        m = np.zeros(self.nparams, np.float32); v = np.zeros(self.nparams, np.float32); b1 = C.c_double(); b2 = C.c_double()
        lib().o_net_opt_state_get(self.h, _p(m), _p(v), C.byref(b1), C.byref(b2)); return m, v, b1.value, b2.value

    def adam_step(self, eta=1e-4, grad_scale=1.0):
        lib().o_net_adam_step(self.h, C.c_double(eta), C.c_double(grad_scale))

    def set_testmode(self, on=True):
        """Flux.testmode!: BatchNorm normalises with its tracked running statistics."""
        lib().o_net_set_testmode(self.h, int(bool(on)))

    def bn_state(self):
        n = lib().o_net_bn_state_count(self.h); p = np.zeros(n, np.float32); lib().o_net_bn_state_get(self.h, _p(p)); return p

    def set_bn_state(self, st):
        st = _c(st, np.float32); assert st.size == lib().o_net_bn_state_count(self.h); lib().o_net_bn_state_set(self.h, _p(st))


def automasking_loss(x, cfg):
    x = _c(x, np.float32); out = np.zeros((cfg.N, 1, cfg.H, cfg.W), np.float32)
    lib().o_automasking_loss(_p(x), C.byref(cfg), _p(out)); return out


def loss_fwd_bwd(x, cfg, disps, rvec, tvec, auto_loss=None, do_bwd=True, want_warped=False):
    """disps: one array (N,1,h,w) per scale level, coarse->fine (cfg.simple: ONE full-resolution array); rvec,tvec (2,N,3).  Returns dict."""
    x = _c(x, np.float32); ds = [_c(d, np.float32) for d in disps]; rvec = _c(rvec, np.float32); tvec = _c(tvec, np.float32)
    nreal = len(ds)
    while len(ds) < 5:
        ds.append(np.zeros((1,), np.float32))
    al = None if auto_loss is None else _c(auto_loss, np.float32)
    dds = [np.zeros_like(d) for d in ds]; drv = np.zeros_like(rvec); dtv = np.zeros_like(tvec)
    N, Cc, H, W = cfg.N, cfg.C, cfg.H, cfg.W
    warped = np.zeros((5, 2, N, Cc, H, W), np.float32) if want_warped else None
    wl = np.zeros((5, N, H, W), np.float32) if want_warped else None
    loss = lib().o_loss_fwd_bwd(_p(x), C.byref(cfg), _p(ds[0]), _p(ds[1]), _p(ds[2]), _p(ds[3]), _p(ds[4]), _p(rvec), _p(tvec), _p(al), int(do_bwd),
                                _p(dds[0]), _p(dds[1]), _p(dds[2]), _p(dds[3]), _p(dds[4]), _p(drv), _p(dtv), _p(warped), _p(wl))
    return dict(loss=float(loss), ddisp=dds[:nreal], drvec=drv, dtvec=dtv, warped=warped, warp_loss=wl)


def num_threads():
    return lib().o_num_threads()
