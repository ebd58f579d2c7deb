// the steady-state ceiling of k_var<float>'s loop shape — the fp32 twin of kvar_occupancy_probe.hip.
// As in the reload sweeps of the fp32 variance kernel (configs[4]: SVGP exact conversion, v_mfma_f32_16x16x4_f32, one MFMA per
// 32 cycles and SIMD, a 16-MFMA block = 512 cycles): a workgroup of 8 waves (2 per SIMD), each wave a 64 x 64 accumulator (16
// MFMA tiles); the A fragments stream from global memory / L2 — ONE stream all workgroups walk in step, 1 KiB per wave and k-step
// (one 16-byte load per lane), requested PF = 4 steps ahead; the B fragments come from LDS (one ds_read_b128 per lane and
// k-step), the workgroup meets at a barrier every CH = 32 k-steps.  No generating sweep, no fill of the next chunk, no triangle,
// no epilogue: what is left is the rate this loop shape can reach on this chip, to hold against k_var<float>'s 126-128 TFLOP/s
// (0.80-0.81 of the 157.3 peak) and against the fp64 probe's 74.0-74.7 (0.94-0.95).
// Variants: A prefetch depth, chunk length, one wave per SIMD, and the ablations (no A loads / no B reads).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/kvar_f32_ceiling_probe.hip -o tools/probes/kvar_f32_ceiling_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// ABL (bit mask, timing only): 1 = A fragments not reloaded, 2 = B fragments not re-read from LDS, 4 = no barrier
template <int WAVES, int PF, int CH, int ABL>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void k_probe(const float* __restrict__ A, float* __restrict__ out, int ksteps, int passes) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];           // [CH k-steps][64 lanes][4]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < CH * 64 * 4; i += WAVES * 64) Bs[i] = 1e-3f * (float)((i * 7) % 13);
    __syncthreads();
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = f4{0, 0, 0, 0};
    constexpr size_t STEP = (size_t)WAVES * 64;                          // f4 units per k-step: 1 KiB per wave
    const f4* base = reinterpret_cast<const f4*>(A) + (size_t)w * 64 + lane;
    for (int pass = 0; pass < passes; ++pass) {
        f4 ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = base[(size_t)i * STEP];
        for (int k0 = 0; k0 < ksteps; k0 += CH) {
            for (int s8 = 0; s8 < CH; s8 += 8) {
#pragma unroll
                for (int su = 0; su < 8; ++su) {
                    const int s = s8 + su;
                    const int k4 = k0 + s;
                    const f4 b = *reinterpret_cast<const f4*>(&Bs[(((ABL & 2) ? 0 : s) * 64 + lane) * 4]);
                    const f4 a = ring[su % PF];
                    const size_t sn = (k4 + PF < ksteps) ? (size_t)(k4 + PF) : (size_t)(ksteps - 1);
                    if (!(ABL & 1)) ring[su % PF] = base[sn * STEP];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[t], acc[0][t], 0, 0, 0);
                        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[t], acc[1][t], 0, 0, 0);
                        acc[2][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[t], acc[2][t], 0, 0, 0);
                        acc[3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[t], acc[3][t], 0, 0, 0);
                    }
                }
            }
            if (!(ABL & 4)) __syncthreads();
        }
    }
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) sum += acc[i][t][0] + acc[i][t][1] + acc[i][t][2] + acc[i][t][3];
    out[(size_t)blockIdx.x * WAVES * 64 + threadIdx.x] = sum;
}

template <int WAVES, int PF, int CH, int ABL = 0>
void run(const float* A, float* out, int blocks, int ksteps, int passes) {
    const size_t lds = (size_t)CH * 64 * 4 * sizeof(float) > 64 * 1024 ? (size_t)CH * 64 * 4 * sizeof(float) : 64 * 1024;   // as k_var<float>: 64 KiB
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<WAVES, PF, CH, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<WAVES, PF, CH, ABL>), dim3(blocks), dim3(WAVES * 64), lds, 0, A, out, ksteps, 1);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<WAVES, PF, CH, ABL>), dim3(blocks), dim3(WAVES * 64), lds, 0, A, out, ksteps, passes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_probe<WAVES, PF, CH, ABL>)));
    const double flops = 2.0 * 16 * 16 * 4 * 16.0 * WAVES * (double)ksteps * passes * blocks;
    const double tf = flops / best * 1e-9;
    printf("waves/WG %d (%d per SIMD), A %d steps ahead, barrier every %3d steps%s%s%s: %8.3f ms  %7.2f TFLOP/s = %.3f of 157.3   [%d registers, %zu B scratch]\n",
           WAVES, WAVES / 4, PF, CH, (ABL & 1) ? ", NO A loads" : "", (ABL & 2) ? ", NO B reads" : "", (ABL & 4) ? ", NO barrier" : "", best, tf, tf / 157.3,
           fa.numRegs, fa.localSizeBytes);
}

int main(int argc, char** argv) {
    const int ksteps = argc > 1 ? atoi(argv[1]) : 512;        // one i-block row of W_t at Z = 2048 (nbi = 4: sweeps of 128 .. 512 k-steps)
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t abytes = (size_t)ksteps * 8 * 1024 + 4096;
    float *A, *out; CK(hipMalloc(&A, abytes)); CK(hipMalloc(&out, (size_t)cus * 512 * sizeof(float)));
    CK(hipMemset(A, 0, abytes));
    printf("device %s, %d CUs; A stream %d k-steps (%.1f MB); one workgroup per CU\n", p.gcnArchName, cus, ksteps, abytes / 1e6);
    const int passes = 2048 * 48 / ksteps;                     // equal MFMA work for every stream length
    run<8, 4, 32>(A, out, cus, ksteps, passes);                // the shipped shape
    run<8, 2, 32>(A, out, cus, ksteps, passes);
    run<8, 8, 32>(A, out, cus, ksteps, passes);
    run<8, 4, 64>(A, out, cus, ksteps, passes);
    run<8, 4, 128>(A, out, cus, ksteps, passes);
    run<4, 4, 32>(A, out, cus, ksteps, passes * 2);            // one wave per SIMD
    run<8, 4, 32>(A, out, cus, ksteps, passes);
    run<8, 4, 32, 1>(A, out, cus, ksteps, passes);
    run<8, 4, 32, 2>(A, out, cus, ksteps, passes);
    run<8, 4, 32, 3>(A, out, cus, ksteps, passes);
    run<8, 4, 32, 4>(A, out, cus, ksteps, passes);
    run<8, 4, 32, 7>(A, out, cus, ksteps, passes);             // bare MFMA rate of the loop
    run<8, 4, 32>(A, out, cus, ksteps, passes);
    return 0;
}
