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

}  // namespace gpt
