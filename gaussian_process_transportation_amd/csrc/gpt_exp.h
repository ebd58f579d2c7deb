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

}  // namespace gpt
