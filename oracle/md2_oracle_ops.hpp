// md2_oracle_ops.hpp — op-by-op CPU restatement of src/utils.jl and src/training.jl (+ the NNlib
// ops they call), templated on the scalar type so the reference's Float64 unit tests can be
// replayed exactly.  TEST INFRASTRUCTURE ONLY (see md2_oracle.h).
//
// Unfused on purpose: each function mirrors one reference function / NNlib op, in the same order
// of arithmetic, so that it can double as the "CPU restatement" baseline.
#pragma once
#include <cmath>
#include <vector>
#include <cstring>
#include <algorithm>
#include <cstdint>

namespace md2o {

template <class T> using Vec = std::vector<T>;

// ---------------------------------------------------------------------------------------------
// NNlib.pad_reflect(x, 1) on dims 1,2 (call sites src/utils.jl:26-27, src/depth_decoder.jl:5).
// Mirror EXCLUDING the border sample: out[-1] = x[1], out[W] = x[W-2] (0-based).
// x: [planes][H][W] -> out: [planes][H+2][W+2]
// ---------------------------------------------------------------------------------------------
inline int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

template <class T>
void pad_reflect1(const T* x, T* out, int planes, int H, int W) {
    const int Hp = H + 2, Wp = W + 2;
#pragma omp parallel for
    for (int p = 0; p < planes; ++p)
        for (int h = 0; h < Hp; ++h) {
            const int sh = refl(h - 1, H);
            for (int w = 0; w < Wp; ++w) out[((size_t)p * Hp + h) * Wp + w] = x[((size_t)p * H + sh) * W + refl(w - 1, W)];
        }
}
// adjoint of pad_reflect1: dx[p][refl(h-1)][refl(w-1)] += dout[p][h][w]
template <class T>
void pad_reflect1_bwd(const T* dout, T* dx, int planes, int H, int W) {
    const int Hp = H + 2, Wp = W + 2;
#pragma omp parallel for
    for (int p = 0; p < planes; ++p) {
        T* d = dx + (size_t)p * H * W;
        for (int i = 0; i < H * W; ++i) d[i] = T(0);
        for (int h = 0; h < Hp; ++h) {
            const int sh = refl(h - 1, H);
            for (int w = 0; w < Wp; ++w) d[sh * W + refl(w - 1, W)] += dout[((size_t)p * Hp + h) * Wp + w];
        }
    }
}

// Flux MeanPool((3,3); stride=1), no padding, divisor 9 (src/utils.jl:19). [planes][Hp][Wp] -> [planes][Hp-2][Wp-2]
template <class T>
void meanpool3(const T* x, T* out, int planes, int Hp, int Wp) {
    const int H = Hp - 2, W = Wp - 2;
#pragma omp parallel for
    for (int p = 0; p < planes; ++p)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                T s = 0;
                for (int dh = 0; dh < 3; ++dh)
                    for (int dw = 0; dw < 3; ++dw) s += x[((size_t)p * Hp + h + dh) * Wp + w + dw];
                out[((size_t)p * H + h) * W + w] = s / T(9);
            }
}
template <class T>
void meanpool3_bwd(const T* dout, T* dx, int planes, int Hp, int Wp) {  // dx += adjoint
    const int H = Hp - 2, W = Wp - 2;
#pragma omp parallel for
    for (int p = 0; p < planes; ++p)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const T g = dout[((size_t)p * H + h) * W + w] / T(9);
                for (int dh = 0; dh < 3; ++dh)
                    for (int dw = 0; dw < 3; ++dw) dx[((size_t)p * Hp + h + dh) * Wp + w + dw] += g;
            }
}

// ---------------------------------------------------------------------------------------------
// SSIM functor, src/utils.jl:25-39.  x = predicted, y = target, both [planes][H][W].
// Output in [0,1]; 0 == identical.  c1 = 0.01^2, c2 = 0.03^2 (Float64 cast to T, :30).
// If dx != nullptr also computes the pullback of `dout` wrt x (and dy if non-null).
// ---------------------------------------------------------------------------------------------
template <class T>
void ssim_fwd_bwd(const T* x, const T* y, T* out, const T* dout, T* dx, T* dy, int planes, int H, int W) {
    const int Hp = H + 2, Wp = W + 2;
    const size_t np = (size_t)planes * Hp * Wp, n = (size_t)planes * H * W;
    Vec<T> xr(np), yr(np), xx(np), yy(np), xy(np);
    pad_reflect1(x, xr.data(), planes, H, W);
    pad_reflect1(y, yr.data(), planes, H, W);
    for (size_t i = 0; i < np; ++i) { xx[i] = xr[i] * xr[i]; yy[i] = yr[i] * yr[i]; xy[i] = xr[i] * yr[i]; }
    Vec<T> mx(n), my(n), pxx(n), pyy(n), pxy(n);
    meanpool3(xr.data(), mx.data(), planes, Hp, Wp);
    meanpool3(yr.data(), my.data(), planes, Hp, Wp);
    meanpool3(xx.data(), pxx.data(), planes, Hp, Wp);
    meanpool3(yy.data(), pyy.data(), planes, Hp, Wp);
    meanpool3(xy.data(), pxy.data(), planes, Hp, Wp);
    const T two = T(2.0), c1 = T(0.01 * 0.01), c2 = T(0.03 * 0.03);
    Vec<T> gmx, gmy, gxx, gyy, gxy;
    const bool bwd = dout && (dx || dy);
    if (bwd) { gmx.assign(n, 0); gmy.assign(n, 0); gxx.assign(n, 0); gyy.assign(n, 0); gxy.assign(n, 0); }
    for (size_t i = 0; i < n; ++i) {
        const T sx = pxx[i] - mx[i] * mx[i], sy = pyy[i] - my[i] * my[i], sxy = pxy[i] - mx[i] * my[i];
        const T n1 = two * mx[i] * my[i] + c1, n2 = two * sxy + c2;
        const T d1 = mx[i] * mx[i] + my[i] * my[i] + c1, d2 = sx + sy + c2;
        const T sn = n1 * n2, sd = d1 * d2;
        const T v = (T(1) - sn / sd) * T(0.5);
        if (out) out[i] = std::min(std::max(v, T(0)), T(1));
        if (bwd) {
            // clamp pullback: 1 inside [0,1] (inclusive), 0 outside.
            const T g = (v >= T(0) && v <= T(1)) ? dout[i] : T(0);
            const T gsn = -T(0.5) * g / sd, gsd = T(0.5) * g * sn / (sd * sd);
            const T gn1 = gsn * n2, gn2 = gsn * n1, gd1 = gsd * d2, gd2 = gsd * d1;
            // through sigma_x = pxx - mx^2 etc.
            const T gsx = gd2, gsy = gd2, gsxy = two * gn2;
            gxx[i] = gsx; gyy[i] = gsy; gxy[i] = gsxy;
            gmx[i] = gn1 * two * my[i] + gd1 * two * mx[i] - gsx * two * mx[i] - gsxy * my[i];
            gmy[i] = gn1 * two * mx[i] + gd1 * two * my[i] - gsy * two * my[i] - gsxy * mx[i];
        }
    }
    if (!bwd) return;
    Vec<T> gxr(np, 0), gyr(np, 0), t1(np, 0), t2(np, 0), t3(np, 0);
    meanpool3_bwd(gmx.data(), gxr.data(), planes, Hp, Wp);
    meanpool3_bwd(gmy.data(), gyr.data(), planes, Hp, Wp);
    meanpool3_bwd(gxx.data(), t1.data(), planes, Hp, Wp);
    meanpool3_bwd(gyy.data(), t2.data(), planes, Hp, Wp);
    meanpool3_bwd(gxy.data(), t3.data(), planes, Hp, Wp);
    for (size_t i = 0; i < np; ++i) {
        gxr[i] += two * xr[i] * t1[i] + yr[i] * t3[i];
        gyr[i] += two * yr[i] * t2[i] + xr[i] * t3[i];
    }
    if (dx) pad_reflect1_bwd(gxr.data(), dx, planes, H, W);
    if (dy) pad_reflect1_bwd(gyr.data(), dy, planes, H, W);
}

// ---------------------------------------------------------------------------------------------
// photometric_loss, src/training.jl:1-7.   predicted, target: [N][C][H][W] -> out [N][1][H][W]
// alpha = T(0.85); (one(T) - alpha) for the L1 weight (computed in T like the reference).
// Pullback wrt predicted only (target is data).
// ---------------------------------------------------------------------------------------------
template <class T>
void photometric_fwd_bwd(const T* pred, const T* target, T* out, const T* dout, T* dpred, int N, int C, int H, int W) {
    const size_t P = (size_t)H * W, n = (size_t)N * C * P;
    const T alpha = T(0.85), beta = T(1) - alpha;
    Vec<T> s(n);
    ssim_fwd_bwd<T>(pred, target, s.data(), nullptr, nullptr, nullptr, N * C, H, W);
    if (out)
        for (int b = 0; b < N; ++b)
            for (size_t p = 0; p < P; ++p) {
                T l1 = 0, ss = 0;
                for (int c = 0; c < C; ++c) {
                    const size_t i = ((size_t)b * C + c) * P + p;
                    l1 += std::abs(target[i] - pred[i]);
                    ss += s[i];
                }
                out[(size_t)b * P + p] = alpha * (ss / T(C)) + beta * (l1 / T(C));
            }
    if (dout && dpred) {
        Vec<T> ds(n);
        for (int b = 0; b < N; ++b)
            for (int c = 0; c < C; ++c)
                for (size_t p = 0; p < P; ++p) ds[((size_t)b * C + c) * P + p] = alpha * dout[(size_t)b * P + p] / T(C);
        ssim_fwd_bwd<T>(pred, target, nullptr, ds.data(), dpred, nullptr, N * C, H, W);
        for (int b = 0; b < N; ++b)
            for (int c = 0; c < C; ++c)
                for (size_t p = 0; p < P; ++p) {
                    const size_t i = ((size_t)b * C + c) * P + p;
                    const T d = target[i] - pred[i];
                    const T sg = d > T(0) ? T(1) : (d < T(0) ? T(-1) : T(0));  // sign(0) = 0
                    dpred[i] += -sg * beta * dout[(size_t)b * P + p] / T(C);
                }
    }
}

// minimum(cat(a_1..a_k; dims=3); dims=3), src/training.jl:10,14,18.  Each a_j: [N][1][H][W] stored
// back-to-back in `stack` as [k][N*P].  argmin = FIRST minimum (Zygote's minimum pullback).
template <class T>
void min_stack(const T* stack, int k, size_t n, T* out, int32_t* arg) {
    for (size_t i = 0; i < n; ++i) {
        T best = stack[i]; int a = 0;
        for (int j = 1; j < k; ++j) if (stack[(size_t)j * n + i] < best) { best = stack[(size_t)j * n + i]; a = j; }
        out[i] = best; if (arg) arg[i] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// disparity_to_depth, src/utils.jl:175-179
// ---------------------------------------------------------------------------------------------
template <class T>
void disparity_to_depth(const T* d, T* out, size_t n, double min_depth, double max_depth) {
    const T min_disp = T(1.0 / max_depth), max_disp = T(1.0 / min_depth);
    for (size_t i = 0; i < n; ++i) out[i] = T(1) / (d[i] * (max_disp - min_disp) + min_disp);
}
template <class T>
void disparity_to_depth_bwd(const T* d, const T* ddepth, T* dd, size_t n, double min_depth, double max_depth) {
    const T min_disp = T(1.0 / max_depth), max_disp = T(1.0 / min_depth);
    for (size_t i = 0; i < n; ++i) {
        const T den = d[i] * (max_disp - min_disp) + min_disp;
        dd[i] = -ddepth[i] * (max_disp - min_disp) / (den * den);
    }
}

// ---------------------------------------------------------------------------------------------
// Backproject, src/utils.jl:41-65.  coordinates column p = w + W*(h-1) holds (w, h, 1), 1-BASED.
// depth [N][P], invK Julia 3x3 column-major -> out [N][P][3]
// ---------------------------------------------------------------------------------------------
template <class T>
void backproject(const T* depth, const T* invK, T* out, int N, int H, int W) {
    const size_t P = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const size_t p = (size_t)h * W + w;
                const T cx = T(w + 1), cy = T(h + 1);
                for (int i = 0; i < 3; ++i) {
                    const T ray = invK[0 * 3 + i] * cx + invK[1 * 3 + i] * cy + invK[2 * 3 + i] * T(1);
                    out[((size_t)n * P + p) * 3 + i] = depth[(size_t)n * P + p] * ray;
                }
            }
}
template <class T>
void backproject_bwd(const T* dout, const T* invK, T* ddepth, int N, int H, int W) {
    const size_t P = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const size_t p = (size_t)h * W + w;
                const T cx = T(w + 1), cy = T(h + 1);
                T g = 0;
                for (int i = 0; i < 3; ++i) g += dout[((size_t)n * P + p) * 3 + i] * (invK[0 * 3 + i] * cx + invK[1 * 3 + i] * cy + invK[2 * 3 + i]);
                ddepth[(size_t)n * P + p] = g;
            }
}

// ---------------------------------------------------------------------------------------------
// Project, src/utils.jl:67-99.  points [N][P][3]; K 3x3 col-major (shared); R [N] 3x3 col-major;
// t [N][3] -> uv [N][P][2] normalised to [-1,1]:  ((pix - 1) / (W-1, H-1) - 0.5) * 2
// ---------------------------------------------------------------------------------------------
template <class T>
void project(const T* pts, const T* K, const T* R, const T* t, T* uv, int N, int H, int W) {
    const size_t P = (size_t)H * W;
    const T nx = T(W - 1.0), ny = T(H - 1.0);
    for (int n = 0; n < N; ++n) {
        const T* Rn = R + n * 9; const T* tn = t + n * 3;
        for (size_t p = 0; p < P; ++p) {
            const T* q = pts + ((size_t)n * P + p) * 3;
            T r[3], c[3];
            for (int i = 0; i < 3; ++i) r[i] = (Rn[0 * 3 + i] * q[0] + Rn[1 * 3 + i] * q[1] + Rn[2 * 3 + i] * q[2]) + tn[i];
            for (int i = 0; i < 3; ++i) c[i] = K[0 * 3 + i] * r[0] + K[1 * 3 + i] * r[1] + K[2 * 3 + i] * r[2];
            const T denom = T(1) / (c[2] + T(1e-7));
            uv[((size_t)n * P + p) * 2 + 0] = (((c[0] * denom - T(1)) / nx) - T(0.5)) * T(2.0);
            uv[((size_t)n * P + p) * 2 + 1] = (((c[1] * denom - T(1)) / ny) - T(0.5)) * T(2.0);
        }
    }
}
// pullback: duv -> dpts [N][P][3], dR [N][9] (col-major), dt [N][3]
template <class T>
void project_bwd(const T* pts, const T* K, const T* R, const T* t, const T* duv, T* dpts, T* dR, T* dt, int N, int H, int W) {
    const size_t P = (size_t)H * W;
    const T nx = T(W - 1.0), ny = T(H - 1.0);
    for (int n = 0; n < N; ++n) {
        const T* Rn = R + n * 9; const T* tn = t + n * 3;
        double aR[9] = {0}, at[3] = {0};
        for (size_t p = 0; p < P; ++p) {
            const T* q = pts + ((size_t)n * P + p) * 3;
            T r[3], c[3];
            for (int i = 0; i < 3; ++i) r[i] = (Rn[0 * 3 + i] * q[0] + Rn[1 * 3 + i] * q[1] + Rn[2 * 3 + i] * q[2]) + tn[i];
            for (int i = 0; i < 3; ++i) c[i] = K[0 * 3 + i] * r[0] + K[1 * 3 + i] * r[1] + K[2 * 3 + i] * r[2];
            const T denom = T(1) / (c[2] + T(1e-7));
            const T gu = duv[((size_t)n * P + p) * 2 + 0] * T(2.0) / nx, gv = duv[((size_t)n * P + p) * 2 + 1] * T(2.0) / ny;
            T gc[3] = {gu * denom, gv * denom, -(gu * c[0] + gv * c[1]) * denom * denom};
            T gr[3];
            for (int j = 0; j < 3; ++j) gr[j] = K[j * 3 + 0] * gc[0] + K[j * 3 + 1] * gc[1] + K[j * 3 + 2] * gc[2];
            for (int j = 0; j < 3; ++j) {
                if (dpts) dpts[((size_t)n * P + p) * 3 + j] = Rn[j * 3 + 0] * gr[0] + Rn[j * 3 + 1] * gr[1] + Rn[j * 3 + 2] * gr[2];
                for (int i = 0; i < 3; ++i) aR[j * 3 + i] += (double)(gr[i] * q[j]);
            }
            for (int i = 0; i < 3; ++i) at[i] += (double)gr[i];
        }
        if (dR) for (int i = 0; i < 9; ++i) dR[n * 9 + i] = T(aR[i]);
        if (dt) for (int i = 0; i < 3; ++i) dt[n * 3 + i] = T(at[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// NNlib.grid_sample(input, grid; padding_mode) — call sites src/training.jl:56 (:border) and
// test/runtests.jl:116 (:zeros).  input [N][C][H][W]; grid [N][Ho][Wo][2] (x -> W axis, y -> H axis);
// unnormalise ((g+1)/2)*(size-1) (align-corners).  :border clamps the coordinate into [0,size-1]
// before corner selection; corners that fall outside contribute 0.
// Pullback wrt grid: d/dcoord * (size-1)/2, zero where the coordinate was clamped (<=0 or >=size-1).
// ---------------------------------------------------------------------------------------------
template <class T>
void grid_sample_fwd_bwd(const T* in, const T* grid, T* out, const T* dout, T* dgrid, T* din,
                         int N, int C, int H, int W, int Ho, int Wo, int border) {
    const size_t Po = (size_t)Ho * Wo, Pi = (size_t)H * W;
    if (din) std::memset(din, 0, sizeof(T) * (size_t)N * C * Pi);
#pragma omp parallel for if (!din)
    for (int n = 0; n < N; ++n)
        for (size_t p = 0; p < Po; ++p) {
            const T gx = grid[((size_t)n * Po + p) * 2 + 0], gy = grid[((size_t)n * Po + p) * 2 + 1];
            T x = ((gx + T(1)) * T(0.5)) * T(W - 1), y = ((gy + T(1)) * T(0.5)) * T(H - 1);
            T mx = T(0.5) * T(W - 1), my = T(0.5) * T(H - 1);
            if (border) {
                if (x <= T(0)) { x = 0; mx = 0; } else if (x >= T(W - 1)) { x = T(W - 1); mx = 0; }
                if (y <= T(0)) { y = 0; my = 0; } else if (y >= T(H - 1)) { y = T(H - 1); my = 0; }
            }
            const T fx = std::floor(x), fy = std::floor(y);
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const T wx1 = x - fx, wx0 = T(1) - wx1, wy1 = y - fy, wy0 = T(1) - wy1;
            const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
            T ggx = 0, ggy = 0;
            for (int c = 0; c < C; ++c) {
                const T* im = in + ((size_t)n * C + c) * Pi;
                const T v00 = (vy0 && vx0) ? im[(size_t)y0 * W + x0] : T(0), v01 = (vy0 && vx1) ? im[(size_t)y0 * W + x1] : T(0);
                const T v10 = (vy1 && vx0) ? im[(size_t)y1 * W + x0] : T(0), v11 = (vy1 && vx1) ? im[(size_t)y1 * W + x1] : T(0);
                if (out) out[((size_t)n * C + c) * Po + p] = v00 * wy0 * wx0 + v01 * wy0 * wx1 + v10 * wy1 * wx0 + v11 * wy1 * wx1;
                if (dout) {
                    const T g = dout[((size_t)n * C + c) * Po + p];
                    ggx += g * ((v01 - v00) * wy0 + (v11 - v10) * wy1);
                    ggy += g * ((v10 - v00) * wx0 + (v11 - v01) * wx1);
                    if (din) {
                        T* di = din + ((size_t)n * C + c) * Pi;
                        if (vy0 && vx0) di[(size_t)y0 * W + x0] += g * wy0 * wx0;
                        if (vy0 && vx1) di[(size_t)y0 * W + x1] += g * wy0 * wx1;
                        if (vy1 && vx0) di[(size_t)y1 * W + x0] += g * wy1 * wx0;
                        if (vy1 && vx1) di[(size_t)y1 * W + x1] += g * wy1 * wx1;
                    }
                }
            }
            if (dout && dgrid) { dgrid[((size_t)n * Po + p) * 2 + 0] = ggx * mx; dgrid[((size_t)n * Po + p) * 2 + 1] = ggy * my; }
        }
}

// ---------------------------------------------------------------------------------------------
// NNlib.upsample_bilinear (src/training.jl:45 `size=`, src/depth_decoder.jl:18-19 `(2,2)`).
// Align-corners mapping: src = dst * (in-1)/(out-1).   [planes][H][W] -> [planes][Ho][Wo]
// ---------------------------------------------------------------------------------------------
template <class T>
inline void up_coord(int o, int in, int out, int& i0, int& i1, T& w1) {
    const T r = out > 1 ? T(in - 1) / T(out - 1) : T(0);
    const T s = r * T(o);
    i0 = std::min((int)std::floor(s), in - 1);
    i1 = std::min(i0 + 1, in - 1);
    w1 = s - T(i0);
}
template <class T>
void upsample_bilinear(const T* x, T* out, int planes, int H, int W, int Ho, int Wo) {
#pragma omp parallel for
    for (int p = 0; p < planes; ++p)
        for (int ho = 0; ho < Ho; ++ho) {
            int h0, h1; T wh; up_coord<T>(ho, H, Ho, h0, h1, wh);
            for (int wo = 0; wo < Wo; ++wo) {
                int w0, w1; T ww; up_coord<T>(wo, W, Wo, w0, w1, ww);
                const T* xp = x + (size_t)p * H * W;
                out[((size_t)p * Ho + ho) * Wo + wo] = (T(1) - wh) * ((T(1) - ww) * xp[h0 * W + w0] + ww * xp[h0 * W + w1]) +
                                                       wh * ((T(1) - ww) * xp[h1 * W + w0] + ww * xp[h1 * W + w1]);
            }
        }
}
template <class T>
void upsample_bilinear_bwd(const T* dout, T* dx, int planes, int H, int W, int Ho, int Wo) {
#pragma omp parallel for
    for (int p = 0; p < planes; ++p) {
        T* d = dx + (size_t)p * H * W;
        for (int i = 0; i < H * W; ++i) d[i] = 0;
        for (int ho = 0; ho < Ho; ++ho) {
            int h0, h1; T wh; up_coord<T>(ho, H, Ho, h0, h1, wh);
            for (int wo = 0; wo < Wo; ++wo) {
                int w0, w1; T ww; up_coord<T>(wo, W, Wo, w0, w1, ww);
                const T g = dout[((size_t)p * Ho + ho) * Wo + wo];
                d[h0 * W + w0] += g * (T(1) - wh) * (T(1) - ww); d[h0 * W + w1] += g * (T(1) - wh) * ww;
                d[h1 * W + w0] += g * wh * (T(1) - ww);          d[h1 * W + w1] += g * wh * ww;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// hat / rrule(hat) / so3_exp_map / composeT, src/utils.jl:101-141,181-188.
// rvec [N][3]; 3x3 matrices column-major per batch item: M[n][j*3+i] = M_ij.
// ---------------------------------------------------------------------------------------------
template <class T>
void hat(const T* r, T* S, int N) {
    for (int n = 0; n < N; ++n) {
        T* s = S + n * 9; const T* v = r + n * 3;
        for (int i = 0; i < 9; ++i) s[i] = 0;
        s[0 * 3 + 1] = v[2];   // S[2,1] =  r3
        s[1 * 3 + 0] = -v[2];  // S[1,2] = -r3
        s[0 * 3 + 2] = -v[1];  // S[3,1] = -r2
        s[2 * 3 + 0] = v[1];   // S[1,3] =  r2
        s[1 * 3 + 2] = v[0];   // S[3,2] =  r1
        s[2 * 3 + 1] = -v[0];  // S[2,3] = -r1
    }
}
template <class T>
void hat_pullback(const T* d, T* dv, int N) {  // src/utils.jl:130-141
    for (int n = 0; n < N; ++n) {
        const T* D = d + n * 9;
        dv[n * 3 + 0] = D[1 * 3 + 2] - D[2 * 3 + 1];   // d[3,2] - d[2,3]
        dv[n * 3 + 1] = -D[0 * 3 + 2] + D[2 * 3 + 0];  // -d[3,1] + d[1,3]
        dv[n * 3 + 2] = D[0 * 3 + 1] - D[1 * 3 + 0];   // d[2,1] - d[1,2]
    }
}
template <class T>
inline void mat3mul(const T* A, const T* B, T* Cm) {  // column-major 3x3
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) Cm[j * 3 + i] = A[0 * 3 + i] * B[j * 3 + 0] + A[1 * 3 + i] * B[j * 3 + 1] + A[2 * 3 + i] * B[j * 3 + 2];
}
template <class T>
void so3_exp_map(const T* r, T* R, int N) {
    Vec<T> S((size_t)N * 9); hat(r, S.data(), N);
    for (int n = 0; n < N; ++n) {
        const T* s = S.data() + n * 9; T s2[9]; mat3mul(s, s, s2);
        const T* v = r + n * 3;
        const T th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const T thi = T(1) / std::max(th, T(1e-4));
        const T f1 = thi * std::sin(th), f2 = thi * thi * (T(1) - std::cos(th));
        for (int i = 0; i < 9; ++i) R[n * 9 + i] = f1 * s[i] + f2 * s2[i] + ((i % 4 == 0) ? T(1) : T(0));
    }
}
// Pullback dR -> dr.  At theta == 0 the reference's sqrt pullback is NaN (README.md:47-51); the
// oracle (and the HIP path) use the 0 sub-gradient there — the one documented deviation.
template <class T>
void so3_exp_map_bwd(const T* r, const T* dR, T* dr, int N) {
    Vec<T> S((size_t)N * 9); hat(r, S.data(), N);
    for (int n = 0; n < N; ++n) {
        const T* s = S.data() + n * 9; T s2[9]; mat3mul(s, s, s2);
        const T* v = r + n * 3; const T* g = dR + n * 9;
        const T th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const T thi = T(1) / std::max(th, T(1e-4));
        const T sn = std::sin(th), cs = std::cos(th);
        const T f1 = thi * sn, f2 = thi * thi * (T(1) - cs);
        T df1 = 0, df2 = 0;
        for (int i = 0; i < 9; ++i) { df1 += g[i] * s[i]; df2 += g[i] * s2[i]; }
        // dS = f1*G + f2*(G*S^T + S^T*G)
        T st[9]; for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) st[j * 3 + i] = s[i * 3 + j];
        T a[9], b[9], dS[9]; mat3mul(g, st, a); mat3mul(st, g, b);
        for (int i = 0; i < 9; ++i) dS[i] = f1 * g[i] + f2 * (a[i] + b[i]);
        const T dthi = df1 * sn + df2 * T(2) * thi * (T(1) - cs);
        T dth = df1 * thi * cs + df2 * thi * thi * sn;
        if (th > T(1e-4)) dth += -dthi * thi * thi;   // max(theta, 1e-4): gradient to theta only when it wins
        T hv[3]; hat_pullback(dS, hv, 1);
        for (int i = 0; i < 3; ++i) dr[n * 3 + i] = hv[i] + (th > T(0) ? dth * v[i] / th : T(0));
    }
}
// composeT(rvec, t, invert): R = so3_exp_map(rvec); if invert: R <- R^T, t <- R^T * (-t).  t: [N][3]
template <class T>
void composeT(const T* r, const T* t, int invert, T* R, T* tout, int N) {
    so3_exp_map(r, R, N);
    for (int n = 0; n < N; ++n) {
        T* Rn = R + n * 9;
        if (invert) {
            std::swap(Rn[0 * 3 + 1], Rn[1 * 3 + 0]); std::swap(Rn[0 * 3 + 2], Rn[2 * 3 + 0]); std::swap(Rn[1 * 3 + 2], Rn[2 * 3 + 1]);
            for (int i = 0; i < 3; ++i) tout[n * 3 + i] = Rn[0 * 3 + i] * -t[n * 3 + 0] + Rn[1 * 3 + i] * -t[n * 3 + 1] + Rn[2 * 3 + i] * -t[n * 3 + 2];
        } else
            for (int i = 0; i < 3; ++i) tout[n * 3 + i] = t[n * 3 + i];
    }
}
template <class T>
void composeT_bwd(const T* r, const T* t, int invert, const T* dR, const T* dtout, T* dr, T* dt, int N) {
    Vec<T> R((size_t)N * 9), g((size_t)N * 9);
    so3_exp_map(r, R.data(), N);
    for (int n = 0; n < N; ++n) {
        const T* Rn = R.data() + n * 9; T* gn = g.data() + n * 9;
        if (invert) {
            // Rt = Rn^T (the returned R), tout = Rt * (-t):  dRt = dR + dtout * (-t)^T ; dt = -Rt^T dtout = -Rn dtout
            T gRt[9];
            for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) gRt[j * 3 + i] = dR[n * 9 + j * 3 + i] + dtout[n * 3 + i] * -t[n * 3 + j];
            for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) gn[j * 3 + i] = gRt[i * 3 + j];   // transpose back
            for (int i = 0; i < 3; ++i) dt[n * 3 + i] = -(Rn[0 * 3 + i] * dtout[n * 3 + 0] + Rn[1 * 3 + i] * dtout[n * 3 + 1] + Rn[2 * 3 + i] * dtout[n * 3 + 2]);
        } else {
            for (int i = 0; i < 9; ++i) gn[i] = dR[n * 9 + i];
            for (int i = 0; i < 3; ++i) dt[n * 3 + i] = dtout[n * 3 + i];
        }
    }
    so3_exp_map_bwd(r, g.data(), dr, N);
}

// ---------------------------------------------------------------------------------------------
// smooth_loss, src/utils.jl:159-173.  disparity [N][H][W], image [N][C][H][W] -> scalar.
// Forward differences, no padding; mean over ALL (W-1)*H*N resp. W*(H-1)*N elements.
// Pullback wrt disparity (scaled by dloss).
// ---------------------------------------------------------------------------------------------
template <class T>
T smooth_loss_fwd_bwd(const T* d, const T* img, T dloss, T* dd, int N, int C, int H, int W) {
    const size_t P = (size_t)H * W;
    double sx = 0, sy = 0;
    const double cntx = (double)(W - 1) * H * N, cnty = (double)W * (H - 1) * N;
    if (dd) std::memset(dd, 0, sizeof(T) * N * P);
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const size_t i = (size_t)n * P + (size_t)h * W + w;
                if (w + 1 < W) {
                    const T gd = d[i] - d[i + 1];
                    T gi = 0;
                    for (int c = 0; c < C; ++c) { const size_t j = ((size_t)n * C + c) * P + (size_t)h * W + w; gi += std::abs(img[j] - img[j + 1]); }
                    if (C > 1) gi /= T(C);
                    const T e = std::exp(-gi);
                    sx += (double)(std::abs(gd) * e);
                    if (dd) { const T s = gd > 0 ? T(1) : (gd < 0 ? T(-1) : T(0)); const T g = dloss * s * e / T(cntx); dd[i] += g; dd[i + 1] -= g; }
                }
                if (h + 1 < H) {
                    const T gd = d[i] - d[i + W];
                    T gi = 0;
                    for (int c = 0; c < C; ++c) { const size_t j = ((size_t)n * C + c) * P + (size_t)h * W + w; gi += std::abs(img[j] - img[j + W]); }
                    if (C > 1) gi /= T(C);
                    const T e = std::exp(-gi);
                    sy += (double)(std::abs(gd) * e);
                    if (dd) { const T s = gd > 0 ? T(1) : (gd < 0 ? T(-1) : T(0)); const T g = dloss * s * e / T(cnty); dd[i] += g; dd[i + W] -= g; }
                }
            }
    return T(sx / cntx + sy / cnty);
}

}  // namespace md2o
