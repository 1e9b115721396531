// Calibration: bare v_mfma_f32_16x16x4_f32 / 32x32x2 issue rate on this device (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024, 2048}) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("16x16x4 x8acc blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, (double)blocks * 4 * iters * 8 * 2048 / (ms * 1e-3) / 1e12);
            hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("32x32x2 x4acc blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, (double)blocks * 4 * iters * 4 * 4096 / (ms * 1e-3) / 1e12);
        }
    }
    // sustained: ~3 s of back-to-back launches on bit-rich per-lane operands; the chip lowers its clock under a dense MFMA load
    // (MI355X_MICROARCH.md, "DVFS give-back"), so this -- not the first launch -- is the rate a long MFMA-bound kernel can expect
    {
        const int blocks = 1024, iters = 200000;
        double last = 0;
        for (int rep = 0; rep < 24; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.2345678f + rep * 0.37f, 0.7654321f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            last = (double)blocks * 4 * iters * 4 * 4096 / (ms * 1e-3) / 1e12;
            if (rep % 4 == 3) printf("sustained 32x32x2 launch %d: %.1f ms  %.1f TFLOP/s\n", rep, ms, last);
        }
    }
    return 0;
}
