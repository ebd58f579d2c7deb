// Does VALU work overlap with v_mfma_f64_16x16x4_f64 on gfx950?  Per wave: loop { 16 independent
// MFMAs ; NV filler instructions of one kind }, 512-thread workgroups (2 waves/SIMD), 1 WG per CU.
// Prints ns per loop iteration per SIMD-pair and the implied cost per filler instruction.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o /tmp/mvp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

enum { F_NONE, F_FMA64, F_ADD64, F_MUL64, F_FMA32, F_ADDU32, F_LDEXP64, F_RNDNE64, F_CVT, F_DSREAD, F_PKFMA32, F_MOV, F_GLOAD };

template <int KIND, int NV, int NMFMA>
__global__ __launch_bounds__(512, 2) void probe(double* out, int iters, const double* in) {
    __shared__ double lds[1024];
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    const int t = threadIdx.x;
    lds[t] = in[t]; lds[t + 512] = in[t + 512];
    __syncthreads();
    double a = in[t & 255], b = in[(t & 255) + 256];
    double f[8]; float g[8]; unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = in[t + i] * 1e-3; g[i] = (float)f[i]; u[i] = t + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = v & 7;
            if (KIND == F_FMA64) f[i] = __builtin_fma(f[i], 1.0000001, 1e-9);
            if (KIND == F_ADD64) f[i] = f[i] + 1e-9;
            if (KIND == F_MUL64) f[i] = f[i] * 1.0000001;
            if (KIND == F_FMA32) g[i] = __builtin_fmaf(g[i], 1.0001f, 1e-6f);
            if (KIND == F_ADDU32) u[i] = u[i] * 3u + 7u;
            if (KIND == F_LDEXP64) f[i] = __builtin_ldexp(f[i], (int)(u[i] & 1));
            if (KIND == F_RNDNE64) f[i] = __builtin_rint(f[i] * 1.5);
            if (KIND == F_CVT) u[i] = (unsigned)(int)f[i] + u[i];
            if (KIND == F_DSREAD) f[i] += lds[(t + (int)u[i]) & 1023];
            if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %0" : "+v"(u[i]));
            if (KIND == F_GLOAD) f[i] += in[(t + it * 64 + v) & 1023];
        }
        if (NV > 0 && NMFMA > 0) {
            // phase-separated (all MFMA then all filler) is what the compiler emits by default
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + g[i] + u[i];
    out[blockIdx.x * blockDim.x + t] = s;
}

template <int KIND, int NV, int NMFMA>
double run(const char* name, double* out, const double* in, int blocks, double base_ns) {
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<KIND, NV, NMFMA><<<blocks, 512>>>(out, 100, in);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        probe<KIND, NV, NMFMA><<<blocks, 512>>>(out, iters, in);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double ns_iter = best * 1e6 / iters;      // per loop iteration of one wave (2 waves share a SIMD)
    const double tf = 2048.0 * NMFMA * iters * 8.0 * blocks / (best * 1e-3) / 1e12;   // 2048 = 2 * 16 * 16 * 4 flop per MFMA
    printf("%-10s NMFMA=%2d NV=%3d  %8.1f ns/iter  %6.2f TF", name, NMFMA, NV, ns_iter, tf);
    if (base_ns > 0 && NV > 0) printf("   +%.2f ns per filler instr per wave (%.1f cyc @2.4GHz, /2 waves = %.1f)", (ns_iter - base_ns) / NV,
                                      (ns_iter - base_ns) / NV * 2.4, (ns_iter - base_ns) / NV * 2.4 / 2);
    printf("\n");
    return ns_iter;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount;
    double *out, *in; CK(hipMalloc(&out, sizeof(double) * 512 * blocks)); CK(hipMalloc(&in, sizeof(double) * 2048));
    double h[2048]; for (int i = 0; i < 2048; ++i) h[i] = 0.5 + 1e-3 * (i % 97) - 2e-3 * (i % 13);
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    const double base = run<F_NONE, 0, 16>("mfma only", out, in, blocks, 0);
    run<F_FMA64, 32, 16>("fma64", out, in, blocks, base);
    run<F_FMA64, 64, 16>("fma64", out, in, blocks, base);
    run<F_ADD64, 64, 16>("add64", out, in, blocks, base);
    run<F_MUL64, 64, 16>("mul64", out, in, blocks, base);
    run<F_FMA32, 64, 16>("fma32", out, in, blocks, base);
    run<F_ADDU32, 64, 16>("mad_u32", out, in, blocks, base);
    run<F_LDEXP64, 64, 16>("ldexp64", out, in, blocks, base);
    run<F_RNDNE64, 64, 16>("rndne64+mul", out, in, blocks, base);
    run<F_CVT, 64, 16>("cvt+add", out, in, blocks, base);
    run<F_MOV, 64, 16>("v_mov", out, in, blocks, base);
    run<F_DSREAD, 64, 16>("ds_read+add", out, in, blocks, base);
    run<F_GLOAD, 16, 16>("gload+add", out, in, blocks, base);
    // filler alone (no MFMA): native cost of the filler streams
    run<F_FMA64, 64, 0>("fma64 only", out, in, blocks, 0);
    run<F_FMA32, 64, 0>("fma32 only", out, in, blocks, 0);
    run<F_ADDU32, 64, 0>("mad_u32 only", out, in, blocks, 0);
    return 0;
}
