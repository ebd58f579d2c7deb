// Does a CU-masked stream (hipExtStreamCreateWithCUMask) keep its kernels on its CUs, and do a "chain" of short dependent
// launches on one masked stream and a bulk kernel on the complementary mask overlap without queueing behind each other?
// (The question behind the look-ahead Cholesky: its step chain must not wait for the trailing update's grid.)
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/cumask_probe.hip -o tools/probes/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where(unsigned* out) {          // XCC id and CU id of every workgroup
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
    }
}
__global__ void spin(long long cycles, int* sink) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("CUs %d\n", ncu);
    // mask bit i = CU i in the driver's numbering; try "first 32 bits" and "every 8th"
    const int words = (ncu + 31) / 32;
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<uint32_t> ma(words, 0), mb(words, 0);
        int na = 0;
        for (int i = 0; i < ncu; ++i) {
            const bool a = variant == 0 ? (i < 32) : (i % 8 == 0);
            if (a) { ma[i / 32] |= 1u << (i % 32); ++na; } else mb[i / 32] |= 1u << (i % 32);
        }
        hipStream_t sa, sb;
        hipError_t e = hipExtStreamCreateWithCUMask(&sa, words, ma.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 0; }
        CK(hipExtStreamCreateWithCUMask(&sb, words, mb.data()));
        unsigned* d; CK(hipMalloc(&d, 4096 * 4));
        int* sink; CK(hipMalloc(&sink, 4));
        std::vector<unsigned> h(4096);
        for (int which = 0; which < 2; ++which) {
            CK(hipMemset(d, 0xff, 4096 * 4));
            hipLaunchKernelGGL(where, dim3(2048), dim3(256), 0, which ? sb : sa, d);
            CK(hipStreamSynchronize(which ? sb : sa));
            CK(hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost));
            int per_xcc[16] = {0};
            std::vector<unsigned> uniq;
            for (int i = 0; i < 2048; ++i) {
                per_xcc[(h[i] >> 16) & 0xf]++;
                const unsigned key = (h[i] & 0xffff0000u) | ((h[i] >> 8) & 0xf) | (((h[i] >> 13) & 0x7) << 4) | (((h[i] >> 12) & 1) << 7);   // cu, se, sh
                bool f = false; for (unsigned u : uniq) f |= (u == key);
                if (!f) uniq.push_back(key);
            }
            printf("variant %d stream %c (%d CUs asked): distinct (xcc,se,sh,cu) seen %zu; per XCC:", variant, which ? 'B' : 'A', which ? ncu - na : na, uniq.size());
            for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
            printf("\n");
        }
        // overlap: bulk = 2048 workgroups x 100 us on B; chain = 50 dependent launches of 32 workgroups x 10 us on A
        hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
        const long long us = 2100;        // ~cycles per microsecond
        for (int mode = 0; mode < 3; ++mode) {     // 0: chain alone, 1: bulk alone, 2: both
            CK(hipDeviceSynchronize());
            if (mode != 1) CK(hipEventRecord(a0, sa));
            if (mode != 0) { CK(hipEventRecord(b0, sb)); hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, sb, 100 * us, (int*)nullptr); CK(hipEventRecord(b1, sb)); }
            if (mode != 1) { for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, sa, 10 * us, sink); CK(hipEventRecord(a1, sa)); }
            CK(hipDeviceSynchronize());
            float ta = 0, tb = 0;
            if (mode != 1) CK(hipEventElapsedTime(&ta, a0, a1));
            if (mode != 0) CK(hipEventElapsedTime(&tb, b0, b1));
            printf("variant %d mode %d: chain %.3f ms, bulk %.3f ms\n", variant, mode, ta, tb);
        }
        // same with two plain streams (no mask) for comparison
        hipStream_t pa, pb; CK(hipStreamCreateWithFlags(&pa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&pb, hipStreamNonBlocking));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a0, pa)); CK(hipEventRecord(b0, pb));
        hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, pb, 100 * us, (int*)nullptr); CK(hipEventRecord(b1, pb));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, pa, 10 * us, sink);
        CK(hipEventRecord(a1, pa));
        CK(hipDeviceSynchronize());
        float ta, tb; CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1));
        printf("variant %d plain streams, both: chain %.3f ms, bulk %.3f ms\n", variant, ta, tb);
        CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb)); CK(hipStreamDestroy(pa)); CK(hipStreamDestroy(pb));
        CK(hipFree(d)); CK(hipFree(sink));
    }
    return 0;
}
