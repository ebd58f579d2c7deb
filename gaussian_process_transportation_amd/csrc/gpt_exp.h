// fp64 exp for the kernel-generation paths, built to spend as few DP-unit instructions as possible
// (on gfx950 every fp64 VALU instruction takes ~4.5 cycles away from the fp64 MFMA stream of its SIMD):
//   exp(t) = 2^(n/256) * e^r,  n = rint(t * 256/ln2),  r = t - n*ln2/256  (|r| <= 1.36e-3)
//   2^(n/256) = 2^(n>>8) * T[n & 255]  (256-entry table in LDS),  e^r by a degree-4 Taylor polynomial
//   (truncation 3.8e-17 relative).  9 fp64 VALU ops + cvt + ldexp; max relative error 3.5e-16 over
//   [-745, 3] (tools/gen_exp_table.py, checked against 50-digit decimals).
#pragma once
#include "exp2_table.h"

namespace gpt {

__device__ const double g_exp2_table[256] = GPT_EXP2_TABLE_INIT;

// t must lie in [-1e5, 700]; values below -745 return 0.
__device__ __forceinline__ double exp_tab(double t, const double* __restrict__ T /* LDS copy of g_exp2_table */) {
    t = fmax(t, -800.0);
    const double n = rint(t * GPT_EXP_INV_STEP);
    double r = fma(n, -GPT_EXP_STEP_HI, t);
    r = fma(n, -GPT_EXP_STEP_LO, r);
    const int ni = (int)n;
    const double tj = T[ni & 255];
    double pz = fma(r, 1.0 / 24.0, 1.0 / 6.0);
    pz = fma(pz, r, 0.5);
    pz = fma(pz, r, 1.0);
    pz = fma(pz, r, 1.0);
    return ldexp(tj * pz, ni >> 8);
}

// Four independent exps, stage by stage: the chain of one exp_tab is ~25 dependent instructions around an LDS look-up, and
// four of them written one after the other are scheduled one after the other (one s_waitcnt lgkmcnt(0) each: 750 clocks per
// exp at the opening of a generating sweep, where no MFMA hides them — profiles/r04_small_n.txt).  Written by stages, pinned with
// sched_barrier (left alone hipcc re-serialises the chains to save registers), the four chains interleave and the four table
// reads share one wait.  Same operations per element as exp_tab: bit-identical results.
__device__ __forceinline__ void exp_tab4(const double (&tin)[4], const double* __restrict__ T, double (&out)[4]) {
    double t[4], n[4], r[4], tj[4], pz[4];
    int ni[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = fmax(tin[e], -800.0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) n[e] = rint(t[e] * GPT_EXP_INV_STEP);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) ni[e] = (int)n[e];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) tj[e] = T[ni[e] & 255];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fma(n[e], -GPT_EXP_STEP_HI, t[e]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fma(n[e], -GPT_EXP_STEP_LO, r[e]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) pz[e] = fma(r[e], 1.0 / 24.0, 1.0 / 6.0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) pz[e] = fma(pz[e], r[e], 0.5);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) pz[e] = fma(pz[e], r[e], 1.0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) pz[e] = fma(pz[e], r[e], 1.0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = ldexp(tj[e] * pz[e], ni[e] >> 8);
}

// Stationary kernels of the path (sklearn/gaussian_process/kernels.py: RBF 1553-1565, Matern 1717-1745):
//   KT 0: RBF  exp(-r^2/2)      KT 1: Matern 1/2  exp(-r)
//   KT 2: Matern 3/2  (1 + sqrt3 r) exp(-sqrt3 r)      KT 3: Matern 5/2  (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)
// with r = |(x - x')/l|.  kernel_libm: c * k(r^2) with the device libm (fit-side kernels).
// kernel_tab: exp(lnc) * k, from h = r^2/2, with the table exp (prediction kernels).
constexpr int KT_RBF = 0, KT_MATERN12 = 1, KT_MATERN32 = 2, KT_MATERN52 = 3;

__device__ __forceinline__ double kernel_libm(const int kt, const double c, const double r2) {
    if (kt == KT_RBF) return c * exp(-0.5 * r2);
    const double r = sqrt(r2);
    if (kt == KT_MATERN12) return c * exp(-r);
    if (kt == KT_MATERN32) { const double t = 1.7320508075688772 * r; return c * (1.0 + t) * exp(-t); }
    const double t = 2.23606797749979 * r;
    return c * (1.0 + t + t * t * (1.0 / 3.0)) * exp(-t);
}

template <int KT>
__device__ __forceinline__ double kernel_tab(const double h, const double lnc, const double* __restrict__ T) {
    if (KT == KT_RBF) return exp_tab(lnc - h, T);
    const double r = sqrt(h + h);
    if (KT == KT_MATERN12) return exp_tab(lnc - r, T);
    if (KT == KT_MATERN32) { const double t = 1.7320508075688772 * r; return (1.0 + t) * exp_tab(lnc - t, T); }
    const double t = 2.23606797749979 * r;
    return (1.0 + t + t * t * (1.0 / 3.0)) * exp_tab(lnc - t, T);
}

// four values at once (the four column tiles of a lane in k_var's generating sweeps): h[e] -> k[e]
template <int KT>
__device__ __forceinline__ void kernel_tab4(const double (&h)[4], const double lnc, const double* __restrict__ T, double (&k)[4]) {
    double a[4], pre[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (KT == KT_RBF) { a[e] = lnc - h[e]; pre[e] = 1.0; }
        else {
            const double r = sqrt(h[e] + h[e]);
            if (KT == KT_MATERN12) { a[e] = lnc - r; pre[e] = 1.0; }
            else if (KT == KT_MATERN32) { const double t = 1.7320508075688772 * r; a[e] = lnc - t; pre[e] = 1.0 + t; }
            else { const double t = 2.23606797749979 * r; a[e] = lnc - t; pre[e] = 1.0 + t + t * t * (1.0 / 3.0); }
        }
    }
    exp_tab4(a, T, k);
    if (KT != KT_RBF && KT != KT_MATERN12) {
#pragma unroll
        for (int e = 0; e < 4; ++e) k[e] = pre[e] * k[e];
    }
}
// fp32 variant (models fitted with GPT_F32): the hardware's v_exp_f32 (2^x, ~1 ulp) — no table, no DP instructions.
template <int KT>
__device__ __forceinline__ float kernel_tab(const float h, const float lnc, const double* __restrict__ /*unused*/) {
    constexpr float L2E = 1.44269504088896341f;
    if (KT == KT_RBF) return __builtin_amdgcn_exp2f((lnc - h) * L2E);
    const float r = __builtin_sqrtf(h + h);
    if (KT == KT_MATERN12) return __builtin_amdgcn_exp2f((lnc - r) * L2E);
    if (KT == KT_MATERN32) { const float t = 1.7320508075688772f * r; return (1.0f + t) * __builtin_amdgcn_exp2f((lnc - t) * L2E); }
    const float t = 2.23606797749979f * r;
    return (1.0f + t + t * t * (1.0f / 3.0f)) * __builtin_amdgcn_exp2f((lnc - t) * L2E);
}

template <int KT>
__device__ __forceinline__ void kernel_tab4(const float (&h)[4], const float lnc, const double* __restrict__ T, float (&k)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = kernel_tab<KT>(h[e], lnc, T);
}

}  // namespace gpt
