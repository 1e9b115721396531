// What buffer_load_dwordx4 ... lds writes for lanes whose offset fails the buffer's range check (gfx950): zeros, or nothing?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o tools/lds_dma_probe && ./tools/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ __launch_bounds__(256) void probe(const float* x, int nbytes, unsigned* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned sm[256 * 4];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) sm[i] = 0x7FC00000u;          // NaN pattern
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, nbytes, 0x00020000);
    unsigned off = tid * 16u;
    if (mode == 1 && (tid & 1)) off = 0xFFFFFFF0u;                        // odd lanes out of range
    if (mode == 2 && wave == 1) off = 0xFFFFFFF0u;                        // a whole wave out of range
    if (mode == 3 && (tid & 63) >= 4) off = 0xFFFFFFF0u;                  // all but four lanes of every wave
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sm + wave * 256), 16, (int)off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[blockIdx.x * 1024 + i] = sm[i];
}
// Does s_waitcnt vmcnt(1) after {load A: in range, HBM-cold; load B: every lane out of range} guarantee that A has landed?  (in-order completion of LDS-DMA loads)
__global__ __launch_bounds__(64) void order_probe(const float* x, long nbytes, unsigned* out, int oob_second) {
    __shared__ __attribute__((aligned(16))) unsigned sm[2 * 256];
    const int tid = threadIdx.x;
    for (int i = tid; i < 512; i += 64) sm[i] = 0x7FC00000u;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)nbytes, 0x00020000);
    const unsigned offA = (unsigned)(((long)blockIdx.x * 64 + tid) * 4096u % nbytes);          // one 16-byte piece per 4 KB: every lane its own cold line
    const unsigned offB = oob_second ? 0xFFFFFFF0u : offA + 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sm), 16, (int)offA, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sm + 256), 16, (int)offB, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    const unsigned v = sm[tid * 4];                                                              // A's first dword of this lane
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 64 + tid] = v;
}
// Does the LDS-DMA write land in the workgroup's own allocation when that allocation starts above 64 KB?  72 KB of dynamic LDS per workgroup: the second workgroup on a CU sits at 72 KB.
__global__ __launch_bounds__(64) void base_probe(const float* x, unsigned* out, int spin, int doff) {          // doff: dword offset of the DMA destination inside the allocation
    extern __shared__ __attribute__((aligned(16))) unsigned dsm[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 64) dsm[doff + i] = 0x7FC00000u;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dsm + doff), 16, (int)(tid * 16), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned v = dsm[doff + tid * 4];
    for (int i = 0; i < spin; ++i) v += dsm[doff + ((tid * 4 + i) & 1023)] & 0u;          // stay resident for a while, so that workgroups overlap on a CU
    out[blockIdx.x * 64 + tid] = v;
}
// Two roles on the same CUs: even workgroups keep issuing buffer_load ... lds with out-of-range lanes (zeros into their LDS); odd workgroups issue ORDINARY buffer loads whose
// offset is out of range and count the lanes that did not get zeros back (what the Winograd / implicit-GEMM loaders rely on for padding).
__global__ __launch_bounds__(256) void mix_probe(const float* x, const float* x2, unsigned* bad, int iters, int with_dma) {
    __shared__ __attribute__((aligned(16))) unsigned sm[3 * 128 * 16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if ((blockIdx.x & 1) == 0) {          // the implicit GEMM's loader: two resources (tensor with its span, weights with an open range), scalar offsets, three stages, counted vmcnt, barriers
        if (!with_dma) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 1 << 20, 0x00020000);
        const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x2), 0, 0xFFFFFFE0u, 0x00020000);
        for (int it = 0; it < iters; ++it) {
            const int st = it % 3;
            const unsigned off = ((lane + it + wave) % 3 == 0) ? 0xFFFFFFF0u : (unsigned)(((blockIdx.x * 256 + tid) * 64u) & ((1u << 20) - 64u));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sm + st * 2048 + wave * 256), 16, (int)off, (it & 3) * 64, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_ptr_t)(sm + st * 2048 + 1024 + wave * 256), 16, (int)(tid * 16), (it & 15) * 4096, 0, 0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // the Winograd loaders: ordinary 16-byte buffer loads, a whole patch row out of range (every lane of some waves), scalar offset in use
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x2), 0, 1 << 18, 0x00020000);
    unsigned nb = 0;
    for (int it = 0; it < iters; ++it) {
        const bool oob = wave == 3 || ((lane + it) & 7) == 0;
        const unsigned off = oob ? 0xFFFFFFF0u : (unsigned)(((blockIdx.x * 256 + tid) * 64u) & ((1u << 18) - 256u));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (it & 3) * 32, 0);
        if (oob) nb += (v[0] | v[1] | v[2] | v[3]) != 0u; else nb += v[0] != 0x3f3f3f3fu;
    }
    if (nb) atomicAdd(bad, nb);
}
int main() {
    {
        float* big; hipMalloc(&big, 1 << 20); hipMemset(big, 0x3f, 1 << 20); float* big2; hipMalloc(&big2, 1 << 20); hipMemset(big2, 0x3f, 1 << 20); unsigned* bad; hipMalloc(&bad, 4);
        for (int dma = 0; dma < 2; ++dma) {
            hipMemset(bad, 0, 4);
            hipLaunchKernelGGL(mix_probe, dim3(4096), dim3(256), 0, 0, big, big2, bad, 3000, dma); hipDeviceSynchronize();
            unsigned hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("mix probe, LDS-DMA neighbours %s: %u ordinary buffer loads returned something other than expected (zeros for out-of-range lanes)\n", dma ? "present" : "absent", hb);
        }
    }
    {
        float* xs; hipMalloc(&xs, 4096); std::vector<float> hx(1024); for (int i = 0; i < 1024; ++i) hx[i] = 1.0f + i; hipMemcpy(xs, hx.data(), 4096, hipMemcpyHostToDevice);
        const int nwg = 2048; unsigned* o3; hipMalloc(&o3, nwg * 64 * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&base_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        const int cfg[4][2] = {{72, 0}, {50, 45}, {36, 30}, {24, 20}};          // KB of LDS per workgroup, KB offset of the DMA destination: the last workgroup on a CU writes near 72 / 145 / 138 / 140 KB
        for (int c = 0; c < 4; ++c) {
            hipLaunchKernelGGL(base_probe, dim3(nwg), dim3(64), cfg[c][0] * 1024, 0, xs, o3, 20000, cfg[c][1] * 256); hipDeviceSynchronize();
            std::vector<unsigned> h(nwg * 64); hipMemcpy(h.data(), o3, h.size() * 4, hipMemcpyDeviceToHost);
            long good = 0, stale = 0, other = 0; int badwg = 0;
            for (int g = 0; g < nwg; ++g) { bool bad = false; for (int t = 0; t < 64; ++t) { const unsigned v = h[g * 64 + t]; float f; __builtin_memcpy(&f, &v, 4); if (f == 1.0f + 4 * t) ++good; else { bad = true; if (v == 0x7FC00000u) ++stale; else ++other; } } badwg += bad; }
            printf("base probe (%d KB of LDS per workgroup, destination at +%d KB): lanes with the loaded value %ld, still the fill pattern %ld, anything else %ld; workgroups affected %d of %d\n", cfg[c][0], cfg[c][1], good, stale, other, badwg, nwg);
        }
    }
    {
        const long nb = 1L << 30; float* big; hipMalloc(&big, nb); hipMemset(big, 0x3f, nb);       // 0x3f3f3f3f: finite
        const int nwg = 4096; unsigned* o2; hipMalloc(&o2, nwg * 64 * 4);
        for (int oob = 0; oob < 2; ++oob) {
            hipLaunchKernelGGL(order_probe, dim3(nwg), dim3(64), 0, 0, big, nb, o2, oob); hipDeviceSynchronize();
            std::vector<unsigned> h(nwg * 64); hipMemcpy(h.data(), o2, h.size() * 4, hipMemcpyDeviceToHost);
            long stale = 0; for (unsigned v : h) stale += v == 0x7FC00000u;
            printf("order probe, second load %s: %ld of %d lanes read load A's slot before it had landed after s_waitcnt vmcnt(1)\n", oob ? "entirely out of range" : "in range", stale, nwg * 64);
        }
    }
    const int n = 1024; float* x; unsigned* out; hipMalloc(&x, n * 4); hipMalloc(&out, 64 * 1024 * 4);
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = 1.0f + i; hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe, dim3(64), dim3(256), 0, 0, x, n * 4, out, mode); hipDeviceSynchronize();
        std::vector<unsigned> o(64 * 1024); hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
        long good = 0, zero = 0, untouched = 0, other = 0;
        for (int b = 0; b < 64; ++b) for (int i = 0; i < 1024; ++i) {
            const unsigned v = o[b * 1024 + i]; const int tid = i >> 2; bool oob = (mode == 1 && (tid & 1)) || (mode == 2 && (tid >> 6) == 1) || (mode == 3 && (tid & 63) >= 4);
            float f; __builtin_memcpy(&f, &v, 4);
            if (!oob) { if (f == 1.0f + i) ++good; else ++other; }
            else { if (v == 0) ++zero; else if (v == 0x7FC00000u) ++untouched; else ++other; }
        }
        printf("mode %d: in-range dwords correct %ld, out-of-range dwords zero %ld / left untouched %ld, anything else %ld\n", mode, good, zero, untouched, other);
    }
    return 0;
}
