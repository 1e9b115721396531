// fp32 vector issue-rate calibration on gfx950: does v_pk_fma_f32 (two fp32 FMAs per lane and instruction) buy anything over v_fma_f32?
// ("pack the two sources into float2 lanes".)  hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak
// Each wave runs `iters` x 64 independent FMAs (16 accumulators, inline asm so that the instruction is the one named); blocks of 256 threads, `wpc` waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    f2 acc[16]; f2 a = {s, s + 1.f}, b = {0.999f, 1.001f};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f2){(float)threadIdx.x + i, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "v"(b.x));
            }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int wpc = 1; wpc <= 4; wpc *= 2)
        for (int pk = 0; pk < 2; ++pk) {
            const int blocks = 256 * wpc;          // one 256-thread block = one wave per SIMD of a CU
            for (int w = 0; w < 2; ++w) { if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)blocks * 4 * iters * 64, fl = insts * 64 * 2 * (pk ? 2 : 1);
            printf("%-13s %d wave(s)/SIMD: %.3f ms  %.2f G wave-instr/s  %.1f TFLOP/s  (%.2f cycles per instruction and SIMD at 2.4 GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32", wpc, ms, insts / ms / 1e6, fl / ms / 1e9,
                   2.4e9 * 1024 * ms * 1e-3 / insts);
        }
    return 0;
}
