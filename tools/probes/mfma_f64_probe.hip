// Probe for v_mfma_f64_16x16x4_f64 on gfx950: (1) operand/result lane maps checked with
// asymmetric data against a host loop, (2) sustained issue rate (TFLOP/s) with
// NACC independent accumulators per wave, 1 or 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void layout_kernel(const double* A, const double* B, double* C) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];   // A[i=l&15][k=l>>4], A row-major 16x4
    double b = B[(l >> 4) * 16 + (l & 15)];  // B[k=l>>4][j=l&15], B row-major 4x16
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // row=(l>>4)+4r, col=l&15
}

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run_rate(int threads, int blocks, int iters) {
    double* out; CK(hipMalloc(&out, sizeof(double) * threads * blocks));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<NACC><<<blocks, threads>>>(out, 10, 1.0, 0.5);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        rate_kernel<NACC><<<blocks, threads>>>(out, iters, 1.0, 0.5);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * (threads / 64) * blocks;
    printf("NACC=%2d threads=%4d blocks=%5d iters=%d  %.3f ms  %.2f TFLOP/s\n", NACC, threads, blocks, iters, best, flops / best * 1e-9);
    CK(hipFree(out));
}

int main() {
    // ---- layout check
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1.0 + 0.37 * i + 0.01 * i * i; B[i] = -2.0 + 0.11 * i * i - 0.5 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dC);
    CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(C[i] - R[i]) / (fabs(R[i]) + 1));
    printf("layout check (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C row=(l>>4)+4r col=l&15): max rel err %.3e %s\n", err, err < 1e-13 ? "OK" : "MISMATCH");
    // ---- rate
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    int cus = p.multiProcessorCount;
    run_rate<1>(256, cus * 4, 20000);
    run_rate<2>(256, cus * 4, 10000);
    run_rate<4>(256, cus * 4, 5000);
    run_rate<8>(256, cus * 4, 4000);
    run_rate<32>(256, cus, 4000);
    run_rate<32>(256, cus * 2, 2000);
    run_rate<16>(512, cus, 4000);
    run_rate<32>(256, cus * 8, 1000);
    return err < 1e-13 ? 0 : 1;
}
