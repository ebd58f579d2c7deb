// Probe of k_gemm's rate: full GEMMs (leading dimension power of two vs padded) and the rank-K trailing update of
// the Cholesky (K = 256, in-place C, lower triangle), with its features switched off one at a time.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm_ld_probe.hip -o tools/probes/gemm_ld_probe
#include "../../gaussian_process_transportation_amd/csrc/gpt_fit.hip"
#include <cstdio>
#include <vector>
using namespace gpt;
template <bool BT, int TS = 128>
static double run(int n, long ld, int reps) {
    double *A, *B, *C;
    size_t bytes = (size_t)n * ld * sizeof(double);
    hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&C, bytes);
    hipMemset(A, 0, bytes); hipMemset(B, 0, bytes); hipMemset(C, 0, bytes);
    GemmArgs g{};
    g.A = A; g.lda = ld; g.B = B; g.ldb = ld; g.C = C; g.ldc = ld;
    g.M = g.M_last = n; g.N = n; g.K = g.K_last = n; g.nbatch = 1; g.alpha = 1.0; g.beta = 0.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch_gemm_ts<BT, false, TS>(0, g);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch_gemm_ts<BT, false, TS>(0, g);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(A); hipFree(B); hipFree(C);
    return 2.0 * n * (double)n * n * reps / (ms * 1e-3) / 1e12;
}
// trailing update: C[rem x rem] (+)= -P P^T with P = rem x kw panel inside an ld x ld matrix
static void syrk(int rem, int kw, long ld, double beta, int lower, bool same_ab, int reps) {
    double *Kmat, *P2;
    size_t bytes = (size_t)ld * ld * sizeof(double);
    hipMalloc(&Kmat, bytes); hipMalloc(&P2, bytes);
    hipMemset(Kmat, 0, bytes); hipMemset(P2, 0, bytes);
    const long r0 = ld - rem;
    GemmArgs g{};
    g.A = Kmat + r0 * ld + (r0 - kw); g.lda = ld;
    g.B = same_ab ? g.A : P2 + r0 * ld + (r0 - kw); g.ldb = ld;
    g.C = Kmat + r0 * ld + r0; g.ldc = ld;
    g.M = g.M_last = rem; g.N = rem; g.K = g.K_last = kw; g.nbatch = 1; g.alpha = -1.0; g.beta = beta; g.lower_only = lower;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch_gemm<true>(0, g);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch_gemm<true>(0, g);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles = lower ? (rem / 128) * (rem / 128 + 1) / 2.0 : (rem / 128.0) * (rem / 128.0);
    const double flop = tiles * 128.0 * 128.0 * kw * 2.0 * reps;
    printf("update rem=%d K=%d beta=%.0f lower=%d A==B %d: %.1f us  %.1f TF\n", rem, kw, beta, lower, (int)same_ab, ms * 1e3 / reps,
           flop / (ms * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(Kmat); hipFree(P2);
}
int main() {
    for (long pad : {0L, 32L})
        printf("n=4096 ld=8192+%ld: BT=false %.1f TF   BT=true %.1f TF\n", pad, run<false>(4096, 8192 + pad, 5), run<true>(4096, 8192 + pad, 5));
    printf("n=2048: 128-tiles BT=false %.1f TF  BT=true %.1f TF | 64-tiles BT=false %.1f TF  BT=true %.1f TF\n", run<false>(2048, 8192, 20),
           run<true>(2048, 8192, 20), run<false, 64>(2048, 8192, 20), run<true, 64>(2048, 8192, 20));
    printf("n=1024: 128-tiles BT=false %.1f TF  BT=true %.1f TF | 64-tiles BT=false %.1f TF  BT=true %.1f TF\n", run<false>(1024, 8192, 50),
           run<true>(1024, 8192, 50), run<false, 64>(1024, 8192, 50), run<true, 64>(1024, 8192, 50));
    printf("n=4096: 64-tiles BT=false %.1f TF  BT=true %.1f TF\n", run<false, 64>(4096, 8192, 5), run<true, 64>(4096, 8192, 5));
    // rank-K trailing updates through launch_gemm's own choice of tile (env GPT_GEMM_TS64_BELOW)
    syrk(7680, 256, 8192, 1.0, 1, true, 10);
    syrk(7680, 256, 8192, 0.0, 0, true, 10);
    syrk(7680, 128, 8192, 1.0, 1, true, 10);
    syrk(4096, 256, 8192, 1.0, 1, true, 10);
    syrk(4096, 128, 8192, 1.0, 1, true, 10);
    syrk(2048, 256, 8192, 1.0, 1, true, 10);
    syrk(2048, 128, 8192, 1.0, 1, true, 10);
    syrk(1024, 128, 8192, 1.0, 1, true, 10);
    return 0;
}
