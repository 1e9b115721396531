// Test infrastructure (not part of the product library): a kernel that fills its workgroups' LDS with NaN bit patterns and exits.  Launched in a loop on a second stream while a
// training step runs, it makes every LDS region a step's workgroups inherit hold NaN -- LDS is not cleared between workgroups -- so a kernel that reads LDS it never wrote, or
// past its allocation, and multiplies what it finds by zero, returns NaN instead of silently working (the out-of-bounds read round 5 found in the row-packed Winograd weight
// gradient).  Built by __graft_entry__.build() into tests/native/liblds_poison.so; used by tests/test_gpu_model.py::test_step_is_immune_to_foreign_lds_contents.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ __launch_bounds__(64) void lds_poison_kernel(int dwords, unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned sm[];
    for (int i = threadIdx.x; i < dwords; i += 64) sm[i] = pattern;
    __syncthreads();
    if (sm[(threadIdx.x * 97) % dwords] == 0x12345678u) *sink = 1u;          // (keeps the stores)
}
extern "C" int32_t lds_poison_launch(void* stream, int32_t launches, int32_t workgroups, int32_t kilobytes, uint32_t pattern, void* sink) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(lds_poison_kernel, dim3(workgroups), dim3(64), (size_t)kilobytes * 1024, static_cast<hipStream_t>(stream), kilobytes * 256, pattern, static_cast<unsigned*>(sink));
    return (int32_t)hipGetLastError();
}
