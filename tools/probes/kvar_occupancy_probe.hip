// VERDICT r1 item 6 asks for a k_var variant with ONE wave per SIMD holding two or three row groups' accumulators, so that a
// B chunk in LDS serves more rows and the scratch image is reloaded less often.  Before rebuilding the kernel around that,
// this probe measures the premise: does one wave per SIMD (4 waves per workgroup, R = 2 or 3 row groups of 64 rows each)
// sustain the fp64 MFMA rate that two waves per SIMD (8 waves, R = 1: the shipped shape) reach, when — as in k_var's reload
// sweeps — the A fragments stream from global memory / L2 (one stream that all workgroups walk in step, 2 KiB per wave, row
// group and k-step, requested two steps ahead), the B fragments come from LDS (one ds_read_b128 pair per k-step, shared by
// the R row groups), and the workgroup meets at a barrier every 32 k-steps?  No B fill, no triangular skipping, no epilogue:
// only the steady state.  Build: hipcc -O3 --offload-arch=gfx950 tools/probes/kvar_occupancy_probe.hip -o tools/probes/kvar_occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

struct AF { d2 lo, hi; };

// ABL (bit mask, timing only): 1 = A fragments not reloaded, 2 = B fragments not re-read from LDS, 4 = no barrier
template <int WAVES, int R, int UNR = 8, int ABL = 0, int PF = 2>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void k_probe(const double* __restrict__ A, double* __restrict__ out, int ksteps, int passes) {
    extern __shared__ __attribute__((aligned(16))) double Bs[];          // [32 k-steps][64 lanes][4]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32 * 64 * 4; i += WAVES * 64) Bs[i] = 1e-3 * (double)((i * 7) % 13);
    __syncthreads();
    d4 acc[R][4][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[r][i][t] = d4{0, 0, 0, 0};
    // stream element (k-step s, wave w, row group r): 128 d2 pairs... 2 KiB = 64 lanes x 2 x d2
    constexpr size_t STEP = (size_t)WAVES * R * 128;                     // d2 units per k-step
    const d2* base = reinterpret_cast<const d2*>(A) + (size_t)w * R * 128 + lane;
    auto lda = [&](AF& a, size_t s, int r) { const d2* p = base + s * STEP + (size_t)r * 128; a.lo = p[0]; a.hi = p[64]; };
    for (int pass = 0; pass < passes; ++pass) {
        AF ring[PF][R];                                                  // A fragments of the next PF k-steps (PF divides UNR)
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) lda(ring[i][r], i, r);
        for (int k0 = 0; k0 < ksteps; k0 += 32) {
          for (int s8 = 0; s8 < 32; s8 += UNR) {                             // unrolled by 8 (as k_var's sub-chunks), 4 per barrier
#pragma unroll
            for (int su = 0; su < UNR; ++su) {
                const int s = s8 + su;
                const int k4 = k0 + s;
                const d4 b = *reinterpret_cast<const d4*>(&Bs[(((ABL & 2) ? 0 : s) * 64 + lane) * 4]);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const AF a = ring[su % PF][r];
                    const size_t sn = (k4 + PF < ksteps) ? (size_t)(k4 + PF) : (size_t)(ksteps - 1);
                    if (!(ABL & 1)) lda(ring[su % PF][r], sn, r);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[r][0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.lo[0], b[t], acc[r][0][t], 0, 0, 0);
                        acc[r][1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.lo[1], b[t], acc[r][1][t], 0, 0, 0);
                        acc[r][2][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.hi[0], b[t], acc[r][2][t], 0, 0, 0);
                        acc[r][3][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.hi[1], b[t], acc[r][3][t], 0, 0, 0);
                    }
                }
            }
          }
            if (!(ABL & 4)) __syncthreads();
        }
    }
    double sum = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) sum += acc[r][i][t][0] + acc[r][i][t][1] + acc[r][i][t][2] + acc[r][i][t][3];
    out[(size_t)blockIdx.x * WAVES * 64 + threadIdx.x] = sum;
}

template <int WAVES, int R, int UNR = 8, int ABL = 0, int PF = 2>
void run(const double* A, double* out, int blocks, int ksteps, int passes, size_t lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<WAVES, R, UNR, ABL, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<WAVES, R, UNR, ABL, PF>), dim3(blocks), dim3(WAVES * 64), lds, 0, A, out, ksteps, 1);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<WAVES, R, UNR, ABL, PF>), dim3(blocks), dim3(WAVES * 64), lds, 0, A, out, ksteps, passes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_probe<WAVES, R, UNR, ABL, PF>)));
    const double flops = 2.0 * 16 * 16 * 4 * 16.0 * R * WAVES * (double)ksteps * passes * blocks;
    if (ABL) printf("  [ablation %d%s%s%s] ", ABL, (ABL & 1) ? " no A loads" : "", (ABL & 2) ? " no B reads" : "", (ABL & 4) ? " no barrier" : "");
    if (PF != 2) printf("  [A fragments %d steps ahead] ", PF);
    printf("waves/WG %d (%d per SIMD), row groups per wave %d (%4d rows per B pass), unrolled by %d: %8.3f ms  %6.2f TFLOP/s   [%d registers, %zu B of scratch per lane]\n",
           WAVES, WAVES / 4, R, WAVES * R * 64, UNR, best, flops / best * 1e-9, fa.numRegs, fa.localSizeBytes);
}

int main(int argc, char** argv) {
    const int ksteps = argc > 1 ? atoi(argv[1]) : 2048;      // one i-block row of W at N = 8192
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t abytes = (size_t)ksteps * 8 * 3 * 2048 + 4096;          // the largest stream: 4 waves x 3 groups = 12 < 24
    double *A, *out; CK(hipMalloc(&A, abytes)); CK(hipMalloc(&out, (size_t)cus * 512 * sizeof(double)));
    CK(hipMemset(A, 0, abytes));
    printf("device %s, %d CUs; A stream %d k-steps; one workgroup per CU (128 KiB of LDS asked for)\n", p.gcnArchName, cus, ksteps);
    const size_t lds = 128 * 1024;
    const int passes8 = 24;                                               // equal MFMA work in every variant
    run<8, 1>(A, out, cus, ksteps, passes8, lds);
    run<4, 2>(A, out, cus, ksteps, passes8, lds);
    run<4, 3>(A, out, cus, ksteps, passes8 * 2 / 3, lds);
    run<4, 3, 2>(A, out, cus, ksteps, passes8 * 2 / 3, lds);
    run<4, 1>(A, out, cus, ksteps, passes8 * 2, lds);
    run<8, 1>(A, out, cus, ksteps, passes8, lds);
    // where the shipped shape's steady state loses its 5 % against the bare MFMA rate (77.7 TFLOP/s, mfma_f64_probe)
    run<8, 1, 8, 1>(A, out, cus, ksteps, passes8, lds);
    run<8, 1, 8, 2>(A, out, cus, ksteps, passes8, lds);
    run<8, 1, 8, 4>(A, out, cus, ksteps, passes8, lds);
    run<8, 1, 8, 3>(A, out, cus, ksteps, passes8, lds);
    run<8, 1, 8, 7>(A, out, cus, ksteps, passes8, lds);
    run<8, 1>(A, out, cus, ksteps, passes8, lds);
    // how far ahead the A fragments have to be requested
    run<8, 1, 8, 0, 1>(A, out, cus, ksteps, passes8, lds);
    run<8, 1, 8, 0, 4>(A, out, cus, ksteps, passes8, lds);
    run<8, 1, 8, 0, 8>(A, out, cus, ksteps, passes8, lds);
    run<4, 2, 8, 0, 4>(A, out, cus, ksteps, passes8, lds);
    run<8, 1>(A, out, cus, ksteps, passes8, lds);
    return 0;
}
