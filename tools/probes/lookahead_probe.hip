// Can the panel-step chain of the Cholesky run BESIDE a trailing update without CU masks, if its stream has a higher priority?
// bulk  = the real rank-256 trailing update (k_gemm, ~8000 workgroups of 35 KB LDS, four per CU: every LDS slot taken);
// chain = 4 dependent launches of a stand-in for k_potrf_step: 120 workgroups x 256 threads, 67.6 KB of dynamic LDS (so a
//         workgroup only fits on a CU after TWO update workgroups have retired), ~16 us of spinning each.
// Reports the chain's and the update's durations alone and together, for plain streams and for a high-priority chain stream.
// RESIDENT=n (a resident update grid of n workgroups) needs tools/probes/patches/r03_potrf_lookahead_resident.patch applied and
// -DGPT_LOOKAHEAD_PATCH.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/lookahead_probe.hip -o tools/probes/lookahead_probe
#include "../../gaussian_process_transportation_amd/csrc/gpt_fit.hip"
#include <cstdio>
using namespace gpt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void step_standin(long long cycles, int* sink) {
    extern __shared__ double sm[];
    sm[threadIdx.x] = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = (int)sm[1];
}

int main() {
    const int NP = 8192;
    double* K; CK(hipMalloc(&K, (size_t)NP * NP * 8)); CK(hipMemset(K, 0, (size_t)NP * NP * 8));
    int* sink; CK(hipMalloc(&sink, 4));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    const size_t lds = 2 * 64 * 66 * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(step_standin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const long long us = 2400;
#ifdef GPT_LOOKAHEAD_PATCH
    const int resident = getenv("RESIDENT") ? atoi(getenv("RESIDENT")) : 0;
#else
    const int resident = 0;
#endif
    printf("update grid: %s\n", resident ? "resident workgroups" : "one workgroup per tile");
    for (int rem : {7936, 4096}) for (int variant = 0; variant < 3; ++variant) {
        hipStream_t sc, sb;
        if (variant == 0) { CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
        else if (variant == 1) { CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo)); }
        else { CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
        for (int mode = 0; mode < 3; ++mode) {      // 0 chain alone, 1 bulk alone, 2 both (bulk launched first)
            float best_a = 1e9, best_b = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                if (mode != 0) { CK(hipEventRecord(b0, sb)); 
#ifdef GPT_LOOKAHEAD_PATCH
                    syrk_update(sb, K, NP, NP - rem, rem, NP - rem - 256, 256, resident);
#else
                    syrk_update(sb, K, NP, NP - rem, rem, NP - rem - 256, 256);
#endif
                    CK(hipEventRecord(b1, sb)); }
                if (mode != 1) {
                    CK(hipEventRecord(a0, sc));
                    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(step_standin, dim3(120), dim3(256), lds, sc, 16 * us, sink);
                    CK(hipEventRecord(a1, sc));
                }
                CK(hipDeviceSynchronize());
                float ta = 0, tb = 0;
                if (mode != 1) { CK(hipEventElapsedTime(&ta, a0, a1)); if (ta < best_a) best_a = ta; }
                if (mode != 0) { CK(hipEventElapsedTime(&tb, b0, b1)); if (tb < best_b) best_b = tb; }
            }
            printf("rem %d, %s, %s: chain %.1f us, update %.1f us\n", rem,
                   variant == 0 ? "plain streams" : (variant == 1 ? "chain high / update low priority" : "chain high / update default"),
                   mode == 0 ? "chain alone" : (mode == 1 ? "update alone" : "both"), mode == 1 ? 0.f : best_a * 1e3f, mode == 0 ? 0.f : best_b * 1e3f);
        }
        CK(hipStreamDestroy(sc)); CK(hipStreamDestroy(sb));
    }
    return 0;
}
