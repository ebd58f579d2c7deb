// Phase timing of k_potrf_step (shader-clock stamps of workgroup 1): load+lazy update | factor | sqrt+L11 | solve | store.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGPT_STEP_TRACE tools/probes/potrf_step_probe.hip -o tools/probes/potrf_step_probe
#include "../../gaussian_process_transportation_amd/csrc/gpt_fit.hip"
#include <cstdio>
#include <vector>
using namespace gpt;
int main() {
    const int NP = 2048, nb = NP / NB;
    std::vector<double> h((size_t)NP * NP);
    unsigned s = 1;
    for (int i = 0; i < NP; ++i)
        for (int j = 0; j <= i; ++j) {
            s = s * 1664525u + 1013904223u;
            double v = ((s >> 8) & 0xffff) / 65536.0 - 0.5;
            h[(size_t)i * NP + j] = h[(size_t)j * NP + i] = (i == j) ? NP : v;
        }
    double *K, *W; int* info; long long* trace;
    hipMalloc(&K, h.size() * 8); hipMalloc(&W, h.size() * 8); hipMalloc(&info, 4); hipMalloc(&trace, 64 * 8);
    hipMemcpy(K, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemset(W, 0, h.size() * 8); hipMemset(info, 0, 4);
    constexpr size_t step_lds = (size_t)(2 * NB * PS) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kb = 0; kb < 8; ++kb) {
        const int p0 = kb & ~3;
        hipMemset(trace, 0, 64 * 8);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_potrf_step, dim3(nb - kb), dim3(256), step_lds, 0, K, W, NP, kb, p0, info, trace);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long t[8]; hipMemcpy(t, trace, sizeof(t), hipMemcpyDeviceToHost);
        printf("kb=%d (j blocks %d): event %.1f us | cycles: load+update %lld  factor %lld  sqrt+L11 %lld  solve %lld  store %lld  total %lld\n", kb,
               kb - p0, ms * 1e3, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
    }
    return 0;
}
