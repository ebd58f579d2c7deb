// Trailing update C -= P P^T (lower 128-tiles) with NO LDS: both MFMA operands of v_mfma_f64_16x16x4 are rows of the
// K-contiguous panel P, so a lane can fetch its own fragment elements straight from global memory (16 B = two consecutive
// k per lane; the MFMA pair m = 0, 1 of a k-block then sums k = 2q + m, q = lane / 16 - the same permutation of k on
// both operands).  A wave owns a 64 x 64 tile of C (16 accumulators, like k_var), a workgroup 2 x 2 of them; no barrier.
// Question: does a wave-level software pipeline (NST k-blocks of 8 in flight) beat k_gemm's LDS-staged 64 x 64 tile
// (one barrier pair per 32 k, 69 % MFMA-busy) on the rank-256 update?
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/syrk_direct_probe.hip -o tools/probes/syrk_direct_probe
// usage: syrk_direct_probe [NP=8192] [reps=10] [rem=7936] [K=256]
#include "../../gaussian_process_transportation_amd/csrc/gpt_fit.hip"
#include <cstdio>
#include <vector>
using namespace gpt;

// TILED: the same arithmetic on a matrix stored as contiguous 64 x 64 tiles (tile (I, J) at ((I * ld / 64) + J) * 4096 doubles, rows of 64):
// a wave's C tile is ONE 32 KiB block and its operand rows 512-byte runs inside 32 KiB blocks, instead of 64 row segments 64 KiB apart.
// (Timing only: C / P are then the addresses of tile (row0 / 64, col / 64) of that layout.)
template <int NST, int KB, int OCC, bool TILED = false>     // k-blocks of 8 in flight; k-blocks per call (K = 8 KB); workgroups per CU
__global__ __launch_bounds__(256, OCC) void k_syrk_direct(double* __restrict__ C, const double* __restrict__ P, int ld, int T) {
    // tile (ti, tj), tj <= ti, from the linear index: rows in order, XCD x takes tiles x, x + 8, ... of the list
    int ti = 0, n = blockIdx.x;
    {
        ti = (int)((sqrtf(8.0f * n + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= n) ++ti;
        while (ti * (ti + 1) / 2 > n) --ti;
    }
    const int tj = n - ti * (ti + 1) / 2;
    if (ti >= T) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wi = w >> 1, wj = w & 1;
    if (ti == tj && wj > wi) return;
    const int lc = lane & 15, lk = lane >> 4;
    const int r0 = ti * 128 + wi * 64, c0 = tj * 128 + wj * 64;
    // addressing: uniform 64-bit bases (SGPRs; the row steps are scalar adds) + ONE 32-bit lane offset per operand, so that
    // the 64 C accesses and 8 fragment streams do not each keep a 64-bit address in registers
    const size_t ldb = TILED ? 512 : (size_t)ld * sizeof(double);
    const size_t tile_row = (size_t)(ld / 64) * 32768;                       // TILED: bytes per row of tiles
    const char* Pa = reinterpret_cast<const char*>(P) + (TILED ? (size_t)(r0 / 64) * tile_row : (size_t)r0 * ldb);
    const char* Pb = reinterpret_cast<const char*>(P) + (TILED ? (size_t)(c0 / 64) * tile_row : (size_t)c0 * ldb);
    char* Cw = reinterpret_cast<char*>(C) + (TILED ? (size_t)(r0 / 64) * tile_row + (size_t)(c0 / 64) * 32768 : (size_t)r0 * ldb + (size_t)c0 * sizeof(double));
    const unsigned offp = (unsigned)lc * (unsigned)ldb + 16u * lk;            // row lc, k pair lk
    const unsigned offc = (unsigned)lk * (unsigned)ldb + 8u * lc;             // row lk, column lc
    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const char* row = Cw + (size_t)(16 * a + 4 * e) * ldb;
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b][e] = *reinterpret_cast<const double*>(row + offc + 128 * b);
        }
    d2 fa[NST][4], fb[NST][4];
    auto load = [&](auto sc, int kb) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[s][a] = *reinterpret_cast<const d2*>(Pa + (size_t)(16 * a) * ldb + offp + (TILED ? (kb / 8) * 32768 + (kb % 8) * 64 : 64 * kb));
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[s][b] = *reinterpret_cast<const d2*>(Pb + (size_t)(16 * b) * ldb + offp + (TILED ? (kb / 8) * 32768 + (kb % 8) * 64 : 64 * kb));
    };
    unroll_ints(std::make_integer_sequence<int, NST - 1>{}, [&](auto sc) { load(sc, decltype(sc)::value); });
    unroll_ints(std::make_integer_sequence<int, KB>{}, [&](auto kc) {
        constexpr int kb = decltype(kc)::value;
        __builtin_amdgcn_sched_barrier(0);               // keep the loads where they are written: hipcc hoists them all otherwise
        if constexpr (kb + NST - 1 < KB) load(std::integral_constant<int, (kb + NST - 1) % NST>{}, kb + NST - 1);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int s = kb % NST;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-fa[s][a][m], fb[s][b][m], acc[a][b], 0, 0, 0);
    });
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            char* row = Cw + (size_t)(16 * a + 4 * e) * ldb;
#pragma unroll
            for (int b = 0; b < 4; ++b) *reinterpret_cast<double*>(row + offc + 128 * b) = acc[a][b][e];
        }
}

template <int NST, int KB, int OCC, bool TILED = false>
static float run(hipStream_t s, double* K, int NP, int rem, int kw, int reps, int ld = 0) {
    const int T = rem / 128, ntiles = T * (T + 1) / 2;
    if (!ld) ld = NP;
    double* C = K + (size_t)(NP - rem) * ld + (NP - rem);
    const double* P = K + (size_t)(NP - rem) * ld + (NP - rem - kw);
    if (TILED) {
        const size_t tr = (size_t)(NP / 64) * 4096;
        C = K + (size_t)((NP - rem) / 64) * tr + (size_t)((NP - rem) / 64) * 4096;
        P = K + (size_t)((NP - rem) / 64) * tr + (size_t)((NP - rem - kw) / 64) * 4096;
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_syrk_direct<NST, KB, OCC, TILED>), dim3(ntiles), dim3(256), 0, s, C, P, ld, T);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_syrk_direct<NST, KB, OCC, TILED>), dim3(ntiles), dim3(256), 0, s, C, P, ld, T);
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int NP = argc > 1 ? atoi(argv[1]) : 8192;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int rem = argc > 3 ? atoi(argv[3]) : 7936;
    const int kw = argc > 4 ? atoi(argv[4]) : 256;
    if (rem % 128 || kw != 256 || rem + kw > NP) { printf("rem must be a multiple of 128, K = 256\n"); return 1; }
    const size_t bytes = (size_t)NP * NP * sizeof(double);
    double *K1, *K2;
    hipMalloc(&K1, bytes); hipMalloc(&K2, bytes + (size_t)NP * 1024);       // room for a padded row pitch
    std::vector<double> h((size_t)NP * NP);
    unsigned sd = 12345;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = ((sd >> 8) & 0xffff) / 65536.0 * 1e-3; }
    hipMemcpy(K1, h.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(K2, h.data(), bytes, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    // correctness: one update each, compare the lower block triangle
    syrk_update(s, K1, NP, NP - rem, rem, NP - rem - kw, kw);
    {
        const int T = rem / 128;
        hipLaunchKernelGGL((k_syrk_direct<2, 32, 2>), dim3(T * (T + 1) / 2), dim3(256), 0, s, K2 + (size_t)(NP - rem) * NP + (NP - rem),
                           K2 + (size_t)(NP - rem) * NP + (NP - rem - kw), NP, T);
    }
    hipStreamSynchronize(s);
    std::vector<double> r1((size_t)NP * NP), r2((size_t)NP * NP);
    hipMemcpy(r1.data(), K1, bytes, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), K2, bytes, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int i = NP - rem; i < NP; ++i)
        for (int j = NP - rem; j <= i; ++j) {
            const double d = fabs(r1[(size_t)i * NP + j] - r2[(size_t)i * NP + j]);
            if (d > worst) worst = d;
            if (fabs(r1[(size_t)i * NP + j]) > scale) scale = fabs(r1[(size_t)i * NP + j]);
        }
    printf("max |k_gemm - direct| on the lower triangle: %.3e (scale %.3e)\n", worst, scale);
    const double flop = (double)rem * (rem + 64) * kw;           // lower 64-tiles (the upper wave tile of a diagonal 128-tile is skipped)
    auto report = [&](const char* name, float us) { printf("%-34s %7.1f us  %5.1f TF\n", name, us, flop / us / 1e6); };
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        for (int r = 0; r < reps; ++r) syrk_update(s, K1, NP, NP - rem, rem, NP - rem - kw, kw);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, a, b);
        report("k_gemm (shipped)", ms * 1e3f / reps);
    }
    report("direct, 1 block ahead, 2 WG/CU", run<2, 32, 2>(s, K2, NP, rem, kw, reps));
    report("direct, 2 blocks ahead, 2 WG/CU", run<3, 32, 2>(s, K2, NP, rem, kw, reps));
    report("direct, 1 block ahead, 1 WG/CU", run<2, 32, 1>(s, K2, NP, rem, kw, reps));
    report("direct, 2 blocks ahead, 1 WG/CU", run<3, 32, 1>(s, K2, NP, rem, kw, reps));
    report("direct, 3 blocks ahead, 1 WG/CU", run<4, 32, 1>(s, K2, NP, rem, kw, reps));
    // the same row-major matrix with a row pitch that is NOT a multiple of 4 KiB: rows of a tile then spread over the L2 channels
    for (int pad : {16, 32, 64}) {
        char name[64];
        snprintf(name, sizeof name, "direct, 1 ahead, 1 WG/CU, ld+%d", pad);
        report(name, run<2, 32, 1>(s, K2, NP, rem, kw, reps, NP + pad));
        snprintf(name, sizeof name, "direct, 1 ahead, 2 WG/CU, ld+%d", pad);
        report(name, run<2, 32, 2>(s, K2, NP, rem, kw, reps, NP + pad));
        GemmArgs c{};
        const int ld = NP + pad;
        c.A = K2 + (size_t)(NP - rem) * ld + (NP - rem - kw); c.lda = ld; c.B = c.A; c.ldb = ld;
        c.C = K2 + (size_t)(NP - rem) * ld + (NP - rem); c.ldc = ld;
        c.M = c.M_last = rem; c.N = rem; c.K = c.K_last = kw; c.nbatch = 1; c.alpha = -1.0; c.beta = 1.0; c.lower_only = 1;
        launch_gemm<true>(s, c); hipStreamSynchronize(s);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        for (int r = 0; r < reps; ++r) launch_gemm<true>(s, c);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, a, b);
        snprintf(name, sizeof name, "k_gemm, ld+%d", pad);
        report(name, ms * 1e3f / reps);
    }
    report("tiled layout, 1 ahead, 2 WG/CU", run<2, 32, 2, true>(s, K2, NP, rem, kw, reps));
    report("tiled layout, 1 ahead, 1 WG/CU", run<2, 32, 1, true>(s, K2, NP, rem, kw, reps));
    report("tiled layout, 2 ahead, 1 WG/CU", run<3, 32, 1, true>(s, K2, NP, rem, kw, reps));
    report("tiled layout, 3 ahead, 1 WG/CU", run<4, 32, 1, true>(s, K2, NP, rem, kw, reps));
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        for (int r = 0; r < reps; ++r) syrk_update(s, K1, NP, NP - rem, rem, NP - rem - kw, kw);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, a, b);
        report("k_gemm (shipped), again", ms * 1e3f / reps);
    }
    return 0;
}
