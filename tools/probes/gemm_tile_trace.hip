// Where a k_gemm tile spends its time: shader-clock stamps (s_memtime) of thread 0 of the first workgroups of a rank-K
// trailing update, for problem sizes from one wave of tiles (latency-bound) to many.  Diagnostic build (GPT_GEMM_TRACE).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGPT_GEMM_TRACE tools/probes/gemm_tile_trace.hip -o tools/probes/gemm_tile_trace
#include "../../gaussian_process_transportation_amd/csrc/gpt_fit.hip"
#include <cstdio>
#include <vector>
using namespace gpt;
int main(int argc, char** argv) {
    const int NP = 8192;
    double* K; hipMalloc(&K, (size_t)NP * NP * 8); hipMemset(K, 0, (size_t)NP * NP * 8);
    long long* tr; hipMalloc(&tr, 16 * 64 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &tr, sizeof(tr));
    hipStream_t s; hipStreamCreate(&s);
    for (int rem : {256, 2048}) for (int kw : {128, 256}) {
        std::vector<long long> h(16 * 64);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(tr, 0, 16 * 64 * 8, s);
            // dirty the panel from another kernel first (as the factorisation does): a memset of the panel region
            hipMemsetAsync(K + (size_t)(NP - rem) * NP, 0, (size_t)rem * NP * 8, s);
            syrk_update(s, K, NP, NP - rem, rem, NP - rem - kw, kw);
            hipStreamSynchronize(s);
        }
        hipMemcpy(h.data(), tr, 16 * 64 * 8, hipMemcpyDeviceToHost);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        for (int r = 0; r < 20; ++r) syrk_update(s, K, NP, NP - rem, rem, NP - rem - kw, kw);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rem=%d K=%d: %.1f us per launch (back to back)\n", rem, kw, ms * 1e3 / 20);
        for (int b : {0, 1, 9}) {
            const long long* t = &h[b * 64];
            if (!t[0]) continue;
            const int nch = kw / 32;
            if (gemm_pipelined()) {
                printf("  wg %d: prologue (fetch0, C, stage0, fetch1) %lld  barrier+scaleC %lld |", b, t[1] - t[0], t[2] - t[1]);
                long long prev = t[2];
                for (int c = 0; c < nch; ++c) {
                    printf(" [2 ksteps %lld stage+fetch %lld 6 ksteps %lld]", t[3 + 3 * c] - prev, t[4 + 3 * c] - t[3 + 3 * c], t[5 + 3 * c] - t[4 + 3 * c]);
                    prev = t[5 + 3 * c];
                }
                printf(" | tail-sync %lld store %lld | total %lld cycles\n", t[60] - prev, t[61] - t[60], t[61] - t[0]);
                continue;
            }
            printf("  wg %d: fetch0 issue %lld |", b, t[1] - t[0]);
            long long prev = t[1];
            for (int c = 0; c < nch; ++c) {
                printf(" [stage %lld sync+fetch %lld mfma %lld]", t[2 + 3 * c] - prev, t[3 + 3 * c] - t[2 + 3 * c], t[4 + 3 * c] - t[3 + 3 * c]);
                prev = t[4 + 3 * c];
            }
            printf(" | tail-sync %lld store %lld | total %lld cycles\n", t[60] - prev, t[61] - t[60], t[61] - t[0]);
        }
    }
    return 0;
}
