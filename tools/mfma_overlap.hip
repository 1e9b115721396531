// Do two waves on one SIMD overlap one wave's non-MFMA phase with the other's fp32 MFMAs?  Each wave loops { phase P (stall of ~1000 cycles: s_sleep, or 64 dependent
// v_fma_f32, or a dependent ds_read chain); 16 v_mfma_f32_32x32x2_f32 in 4 dependent chains }.  1 block of 256 threads per CU = 1 wave per SIMD, 2 blocks = 2 per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int PRIO>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, float s, const float* gbuf) {
    __shared__ float lds[4096];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = s + threadIdx.x, b = 1.0001f, v = s;
    lds[threadIdx.x] = threadIdx.x & 1023; lds[threadIdx.x + 256] = 7.f;
    __syncthreads();
    int idx = threadIdx.x;
    if (PRIO == 1) {          // static, asymmetric priority by hardware wave slot
        const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));      // HW_REG_HW_ID.WAVE_ID
        if (slot & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    }
    if (PRIO == 2 && (blockIdx.x & 1)) __builtin_amdgcn_s_sleep(8);                  // start-up stagger, rule a: odd block ids
    if (PRIO == 3 && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_sleep(8);           // rule b: second half of the grid
    if (PRIO == 4) { const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)); if (slot & 1) __builtin_amdgcn_s_sleep(8); }      // rule c: odd hardware slot
    if (PRIO == 5) { const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)); out[blockIdx.x * 256 + threadIdx.x] = slot; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) __builtin_amdgcn_s_sleep(16);                       // ~1024 cycles
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 128; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(a));      // dependent chain: ~128 x 8 cycles
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) idx = (int)gbuf[(idx & 1023) * 16];             // dependent global-load chain (L2 hits): 2 x ~500 cycles
            a += (float)idx;
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) idx = (int)lds[idx & 1023];             // dependent LDS chain: 8 x ~130 cycles
            a += (float)idx;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float t = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE, int PRIO> float run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, PRIO>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, out + 300000);
    hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, PRIO>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, out + 300000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 2 * 256 * 1024 * 4); hipMemset(out, 0, 2 * 256 * 1024 * 4);
    const int iters = 2000;
    const char* names[5] = {"MFMAs only", "s_sleep(16) + MFMAs", "128 dependent v_fma + MFMAs", "8 dependent ds_read + MFMAs", "2 dependent global loads + MFMAs"};
    for (int mode = 0; mode < 5; ++mode)
        for (int wps = 1; wps <= 4; ++wps) {
            float ms = mode == 0 ? run<0, 0>(256 * wps, iters, out) : mode == 1 ? run<1, 0>(256 * wps, iters, out) : mode == 2 ? run<2, 0>(256 * wps, iters, out) : mode == 3 ? run<3, 0>(256 * wps, iters, out) : run<4, 0>(256 * wps, iters, out);
            float mp = mode == 1 ? run<1, 1>(256 * wps, iters, out) : 0, ma = mode == 1 ? run<1, 2>(256 * wps, iters, out) : 0, mb = mode == 1 ? run<1, 3>(256 * wps, iters, out) : 0, mc = mode == 1 ? run<1, 4>(256 * wps, iters, out) : 0;
            printf("%-30s %d wave(s)/SIMD: %.0f ns per iteration | prio %.0f | stagger odd-block %.0f | stagger grid-half %.0f | stagger odd-slot %.0f\n", names[mode], wps, ms * 1e6 / iters, mp * 1e6 / iters, ma * 1e6 / iters, mb * 1e6 / iters, mc * 1e6 / iters);
        }
    {   // which hardware slots do two co-resident blocks get?
        hipLaunchKernelGGL((k<1, 5>), dim3(512), dim3(256), 0, 0, out, 10, 1.f, out + 300000); hipDeviceSynchronize();
        static float h[512 * 256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("slots of wave 0 of blocks 0..15: "); for (int b = 0; b < 16; ++b) printf("%d ", (int)h[b * 256]); printf("| blocks 256..263: "); for (int b = 256; b < 264; ++b) printf("%d ", (int)h[b * 256]); printf("\n");
    }
    return 0;
}
