// Prediction kernels for gfx950 (MI355X), templated on the element type (fp64: the exact-GP path of the reference;
// fp32: its SVGP exact-conversion path, which the reference computes in fp32 torch).
//
//  k_mean_jac : posterior mean  mu[m,o] = sum_n k(x_m, X_n) alpha[n,o]   (sklearn/_gpr.py:443-444)
//               and Jacobian    J[m,o,d] = sum_n (X[n,d]-x[m,d])/l_d^2 k(x_m,X_n) alpha[n,o]
//               (reference models/gaussian_process.py:72-90) as one wavefront-reduced contraction:
//               a wave owns QPW queries, its 64 lanes stride the source points, 16 partial sums per
//               query live in registers and are reduced across the wave once at the end.
//
//  k_var      : posterior variance  c + s^2 - |W k*|^2  (sklearn/_gpr.py:454-485 does L \ k*; here
//               W = L^-1 is explicit so the solve becomes a triangular GEMM), the Jacobian variance
//               c/l_d^2 - |W dk_d|^2 (gaussian_process.py:95-98) and d var/dx_d = -2 (W dk_d).(W k*)
//               (gaussian_process.py:122-125) on the matrix cores; design notes at the kernel.  With ntask > 1 the
//               A operand is the stack of the tasks' inverse factors (each pre-scaled by its outputscale) and
//               the kernel columns are shared by all tasks: the SVGP exact-conversion model,
//               models/torch/stocastic_variational_gaussian_process_derivatives.py:113-153.
#include "gpt_common.h"
#include "gpt_exp.h"
#include "gpt_plan.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace gpt {

// ------------------------------------------------------------------------------------------
// Element-type traits: vector types, the MFMA, and the layout unit of the A stream Wf.
//   fp64: per (k4-step, row group) two d2 per lane (row tiles 0,1 | 2,3): [q 0..2)[lane 0..64)[p 0..2)
//   fp32: per (k4-step, row group) one f4 per lane (row tiles 0..3):       [lane 0..64)[e 0..4)
// 16 B per lane and load either way; the v_mfma_*_16x16x4 A/B lane maps are the same for both types.
// ------------------------------------------------------------------------------------------
template <typename T> struct El;
template <> struct El<double> {
    typedef d4 v4;
    typedef d2 avec;
    static constexpr int A_STEP = 1024, A_GROUP = 128;     // avec per k4-step of a tile / per row group inside it
    static constexpr int SUBS = 4;                         // sub-chunks of 8 k4-steps per LDS chunk: 2 x 32 steps x 2 KiB = 128 KiB
    static constexpr int PF = 2;                           // A fragments are requested this many k4-steps (of 1024 cycles) ahead
    static constexpr bool DIAG_LDS = false;                // the B image of a diagonal tile (256 KiB) does not fit in LDS
    static constexpr bool BATCH_PROLOGUE = false;          // measured in round 1: no gain on the long fp64 sweeps
    struct AF { d2 lo, hi; };
    static __device__ __forceinline__ void lda(AF& a, const avec* __restrict__ p, const int lane) { a.lo = p[lane]; a.hi = p[lane + 64]; }
    static __device__ __forceinline__ void keep(const AF& a, const v4& b) { asm volatile("" :: "v"(a.lo), "v"(a.hi), "v"(b)); }
    static __device__ __forceinline__ void keep1(const v4& b) { asm volatile("" :: "v"(b)); }
    static __device__ __forceinline__ v4 mfma(const double a, const double b, const v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ void mfma16(v4 (&acc)[4][4], const AF& a, const v4& b) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0][t] = mfma(a.lo[0], b[t], acc[0][t]);
            acc[1][t] = mfma(a.lo[1], b[t], acc[1][t]);
            acc[2][t] = mfma(a.hi[0], b[t], acc[2][t]);
            acc[3][t] = mfma(a.hi[1], b[t], acc[3][t]);
        }
    }
    // row tiles R0 .. 3 only (the last k-steps of a wave's diagonal 64 x 64 block: zeros above the diagonal)
    template <int R0> static __device__ __forceinline__ void mfma_from(v4 (&acc)[4][4], const AF& a, const v4& b) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (R0 <= 0) acc[0][t] = mfma(a.lo[0], b[t], acc[0][t]);
            if (R0 <= 1) acc[1][t] = mfma(a.lo[1], b[t], acc[1][t]);
            if (R0 <= 2) acc[2][t] = mfma(a.hi[0], b[t], acc[2][t]);
            acc[3][t] = mfma(a.hi[1], b[t], acc[3][t]);
        }
    }
};
template <> struct El<float> {
    typedef f4 v4;
    typedef f4 avec;
    static constexpr int A_STEP = 512, A_GROUP = 64;
#ifndef GPT_F32_SUBS
#define GPT_F32_SUBS 4
#endif
#ifndef GPT_F32_PF
#define GPT_F32_PF 4
#endif
    static constexpr int SUBS = GPT_F32_SUBS;              // 2 x 32 steps x 1 KiB = 64 KiB (64-step chunks measured 4 % slower: profiles/r02_svgp_variants.txt)
    static constexpr int PF = GPT_F32_PF;                  // an fp32 MFMA block lasts 512 cycles, less than an L2 round trip under load
#ifndef GPT_F32_DIAG_LDS
#define GPT_F32_DIAG_LDS 1
#endif
    static constexpr bool DIAG_LDS = GPT_F32_DIAG_LDS != 0;   // diagonal tiles of the reload sweeps: B image (128 KiB) staged in LDS
#ifndef GPT_F32_BATCH_PROLOGUE
#define GPT_F32_BATCH_PROLOGUE 1
#endif
    static constexpr bool BATCH_PROLOGUE = GPT_F32_BATCH_PROLOGUE != 0;
    struct AF { f4 v; };
    static __device__ __forceinline__ void lda(AF& a, const avec* __restrict__ p, const int lane) { a.v = p[lane]; }
    static __device__ __forceinline__ void keep(const AF& a, const v4& b) { asm volatile("" :: "v"(a.v), "v"(b)); }
    static __device__ __forceinline__ void keep1(const v4& b) { asm volatile("" :: "v"(b)); }
    static __device__ __forceinline__ v4 mfma(const float a, const float b, const v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ void mfma16(v4 (&acc)[4][4], const AF& a, const v4& b) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0][t] = mfma(a.v[0], b[t], acc[0][t]);
            acc[1][t] = mfma(a.v[1], b[t], acc[1][t]);
            acc[2][t] = mfma(a.v[2], b[t], acc[2][t]);
            acc[3][t] = mfma(a.v[3], b[t], acc[3][t]);
        }
    }
    template <int R0> static __device__ __forceinline__ void mfma_from(v4 (&acc)[4][4], const AF& a, const v4& b) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (R0 <= 0) acc[0][t] = mfma(a.v[0], b[t], acc[0][t]);
            if (R0 <= 1) acc[1][t] = mfma(a.v[1], b[t], acc[1][t]);
            if (R0 <= 2) acc[2][t] = mfma(a.v[2], b[t], acc[2][t]);
            acc[3][t] = mfma(a.v[3], b[t], acc[3][t]);
        }
    }
};

// ------------------------------------------------------------------------------------------
// OC = outputs handled by this pass (1..4): only their partial sums are accumulated.
// DW = coordinates carried per point: 3 (source rows of 4: the tuned D <= 3 layout), WIDE_D (rows of 8, D = 4..8) or MAX_D (rows of 16, D = 9..15).
// WJ = false: the mean alone (predict without derivative: configs[1]) — no Jacobian sums, a third fewer vector instructions.
template <typename T, int QPW, int OC, int KT, int DW, bool WJ = true>
__global__ __launch_bounds__(256) void k_mean_jac(KernelParams p, const T* __restrict__ Xs,
                                                  const T* __restrict__ A4, const T* __restrict__ Xq,
                                                  int64_t M, int o_base, T* __restrict__ mean,
                                                  T* __restrict__ J) {
    typedef typename El<T>::v4 v4;
    constexpr int XS = DW == 3 ? 4 : DW;            // elements per source row
    __shared__ double Tt[256];
    if (std::is_same<T, double>::value) {
        Tt[threadIdx.x] = g_exp2_table[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * QPW;
    if (m0 >= M) return;
    const int D = p.D;
    constexpr T RS2 = (T)0.70710678118654752440;    // coordinates scaled by 1/sqrt(2): k = exp(ln c - |d'|^2)
    // (branch-free: written as `(d < D) ? Xq[..] : 0` every load sat in its own basic block behind its own wait — QPW x DW memory
    // latencies in a row before the first source, as long as the whole contraction at N = 1024; a missing coordinate reads
    // coordinate 0 and is scaled by zero)
    T q[QPW][DW], qsc[DW];
#pragma unroll
    for (int d = 0; d < DW; ++d) qsc[d] = (d < D) ? (T)(p.inv_ls[d] * 0.70710678118654752440) : (T)0;
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int64_t m = (m0 + i < M) ? (m0 + i) : (M - 1);
        const T* qp = Xq + m * D;
#pragma unroll
        for (int d = 0; d < DW; ++d) q[i][d] = qp[d < D ? d : 0];
    }
#pragma unroll
    for (int i = 0; i < QPW; ++i)
#pragma unroll
        for (int d = 0; d < DW; ++d) q[i][d] *= qsc[d];
    T acc[QPW][OC][1 + DW];
#pragma unroll
    for (int i = 0; i < QPW; ++i)
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int e = 0; e < 1 + DW; ++e) acc[i][o][e] = (T)0;

    const T lnc = (T)p.lnc;
    for (int n = lane; n < p.N; n += 64) {
        T x[DW];
        if constexpr (DW == 3) {
            const v4 xs = *reinterpret_cast<const v4*>(Xs + (size_t)n * 4);
            x[0] = xs[0] * RS2; x[1] = xs[1] * RS2; x[2] = xs[2] * RS2;
        } else {
#pragma unroll
            for (int v = 0; v < DW / 4; ++v) {
                const v4 xs = *reinterpret_cast<const v4*>(Xs + (size_t)n * XS + 4 * v);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[4 * v + e] = xs[e] * RS2;
            }
        }
        const v4 al = *reinterpret_cast<const v4*>(A4 + (size_t)n * 4);
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            T df[DW];
#pragma unroll
            for (int d = 0; d < DW; ++d) df[d] = x[d] - q[i][d];
            T hh = df[0] * df[0];
#pragma unroll
            for (int d = 1; d < DW; ++d) hh = fma(df[d], df[d], hh);
            const T kv = kernel_tab<KT>(hh, lnc, Tt);
#pragma unroll
            for (int o = 0; o < OC; ++o) {
                const T t = kv * al[o];
                acc[i][o][0] += t;
                if (WJ) {
#pragma unroll
                    for (int d = 0; d < DW; ++d) acc[i][o][1 + d] += t * df[d];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QPW; ++i)
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int e = 0; e < (WJ ? 1 + DW : 1); ++e) {
                T v = acc[i][o][e];
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
                acc[i][o][e] = v;
            }
    if (lane == 0) {
        const int O = p.O;
        constexpr double S2 = 1.41421356237309504880;     // undo the 1/sqrt(2) on the (X - x) factor
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int64_t m = m0 + i;
            if (m >= M) break;
#pragma unroll
            for (int o = 0; o < OC; ++o) {
                const int oo = o_base + o;
                if (mean) mean[m * O + oo] = acc[i][o][0];
                if (WJ && J) {
#pragma unroll
                    for (int d = 0; d < DW; ++d)
                        if (d < D) J[(m * O + oo) * D + d] = acc[i][o][1 + d] * (T)(p.inv_ls[d] * S2);
                }
            }
        }
    }
}

template <typename T, int DW>
static void launch_mean_jac_t(hipStream_t s, const KernelParams& p, const T* Xs, const T* A4,
                              const T* Xq, int64_t M, T* mean, T* J) {
#ifndef GPT_MJ_QPW
#define GPT_MJ_QPW 2
#endif
    constexpr int QPW = DW == 3 ? GPT_MJ_QPW : 1;          // queries per wave; 500k queries at N = 8192: 1 -> 9.17 ms, 2 -> 6.19-6.26, 4 -> 6.11 (r3mj)
#ifndef GPT_MJ_QPW_MEAN
#define GPT_MJ_QPW_MEAN 4
#endif
    // the mean alone (no Jacobian sums: 3 accumulators per query instead of 12): more queries per wave when there are waves enough
    // (M >= 32 768: 8 per SIMD), so that a small model's short source loop (16 iterations at N = 1024) is not dwarfed by the wave's
    // prologue and its cross-lane reductions.  N = 1024, M = 50 000: 0.071 -> 0.064 ms; at M = 10^4 four per wave LOSE 7 - 10 % (r4s38)
    constexpr int QPW_M = DW == 3 ? GPT_MJ_QPW_MEAN : 1;
    const bool many = !J && M >= 32768 && QPW_M != QPW;
    const int qpw = many ? QPW_M : QPW;
    const int64_t waves = (M + qpw - 1) / qpw;
    const int64_t blocks = (waves + 3) / 4;
    for (int ob = 0; ob < p.O; ob += 4) {
        const int cnt = (p.O - ob) < 4 ? (p.O - ob) : 4;
        const T* a4 = A4 + (size_t)(ob / 4) * p.NP * 4;
        const dim3 grid((unsigned)blocks);
#define GPT_MJ(OC_, KT_)                                                                                                          \
        do {                                                                                                                      \
            if (J) hipLaunchKernelGGL((k_mean_jac<T, QPW, OC_, KT_, DW, true>), grid, dim3(256), 0, s, p, Xs, a4, Xq, M, ob, mean, J); \
            else if (many) hipLaunchKernelGGL((k_mean_jac<T, QPW_M, OC_, KT_, DW, false>), grid, dim3(256), 0, s, p, Xs, a4, Xq, M, ob, mean, J); \
            else hipLaunchKernelGGL((k_mean_jac<T, QPW, OC_, KT_, DW, false>), grid, dim3(256), 0, s, p, Xs, a4, Xq, M, ob, mean, J);  \
        } while (0)
#define GPT_MJ_K(OC_)                                     \
        switch (p.ktype) {                                 \
            case KT_MATERN12: GPT_MJ(OC_, KT_MATERN12); break; \
            case KT_MATERN32: GPT_MJ(OC_, KT_MATERN32); break; \
            case KT_MATERN52: GPT_MJ(OC_, KT_MATERN52); break; \
            default: GPT_MJ(OC_, KT_RBF);                  \
        }
        switch (cnt) {
            case 1: GPT_MJ_K(1); break;
            case 2: GPT_MJ_K(2); break;
            case 3: GPT_MJ_K(3); break;
            default: GPT_MJ_K(4);
        }
#undef GPT_MJ_K
#undef GPT_MJ
    }
}

void launch_mean_jac(hipStream_t s, const KernelParams& p, const void* Xs, const void* A4,
                     const void* Xq, int64_t M, void* mean, void* J) {
    if (M <= 0 || (!mean && !J)) return;
    if (p.D <= 3) {
        if (p.dtype == DT_F32)
            launch_mean_jac_t<float, 3>(s, p, (const float*)Xs, (const float*)A4, (const float*)Xq, M, (float*)mean, (float*)J);
        else
            launch_mean_jac_t<double, 3>(s, p, (const double*)Xs, (const double*)A4, (const double*)Xq, M, (double*)mean, (double*)J);
    } else if (p.D <= WIDE_D) {
        if (p.dtype == DT_F32)
            launch_mean_jac_t<float, WIDE_D>(s, p, (const float*)Xs, (const float*)A4, (const float*)Xq, M, (float*)mean, (float*)J);
        else
            launch_mean_jac_t<double, WIDE_D>(s, p, (const double*)Xs, (const double*)A4, (const double*)Xq, M, (double*)mean, (double*)J);
    } else {
        if (p.dtype == DT_F32)
            launch_mean_jac_t<float, MAX_D>(s, p, (const float*)Xs, (const float*)A4, (const float*)Xq, M, (float*)mean, (float*)J);
        else
            launch_mean_jac_t<double, MAX_D>(s, p, (const double*)Xs, (const double*)A4, (const double*)Xq, M, (double*)mean, (double*)J);
    }
}

// ------------------------------------------------------------------------------------------
// Variance kernel.
//
// What the hardware does (measured on MI355X, profiles/r01_*):
//   * v_mfma_f64_16x16x4_f64 issues every 64 cycles per SIMD: 77.7 TFLOP/s with 2 waves/SIMD at 2.4 GHz
//     (v_mfma_f32_16x16x4_f32 every 32: 155 TFLOP/s);
//   * every fp64 VALU instruction of a SIMD takes ~4.5 cycles away from its fp64 MFMA stream (the fp64
//     matrix and vector paths share the DP units), so B values must be generated once, not per wave;
//   * a workgroup barrier every 8 k-steps costs ~4 %; once the matrix pipe is >90 % busy the chip lowers
//     its clock (2.38 -> 2.18 GHz), so what is left is energy per MFMA: operands must come from close by
//     (A from L2 with all workgroups walking W in step, B from LDS), not from HBM.
// Design:
//   * a workgroup (512 threads = 8 waves, 2 per SIMD) owns a 64-column block and sweeps the 512-row
//     i-blocks of every task, longest sweep first; wave w accumulates the 64x64 product of its 64-row group (16 MFMA
//     tiles) and folds it into per-column sums when the sweep ends, so V = W K*^T never touches memory
//     (a sweep that the work split cut along k stores its partial product instead: gpt_plan.h);
//   * the B operand of a sweep reaches the waves through a double-buffered LDS image in MFMA lane order,
//     2 x 32 k-steps, one barrier per 32 k-steps.  Wave w fills k-steps w, w+8, w+16,
//     w+24 of the next chunk from the middle of its own MFMA run (the two waves of a SIMD staggered).
//     In a GENERATING sweep the fragments are computed (k* / dk_d columns, one exp each: table-driven in fp64,
//     gpt_exp.h, v_exp_f32 in fp32) and a copy goes to this workgroup's scratch image in HBM/L2
//     ([k-step][lane][4]); the other sweeps of the block reload them from there — so a
//     block pays N exps per column, not N*(N/512+1)/2;
//   * the A operand streams from the fragment-ordered image Wf with 16-byte loads per lane,
//     two k-steps ahead; in the diagonal tile a wave skips the k-steps where its row group is
//     entirely above the diagonal (wave g has 16 (g + 1) of 128), row groups paired (0,7)(1,6)(2,5)(3,4) on
//     the SIMDs.  In the reload sweeps the diagonal tile runs OUTSIDE the lock-step LDS pipeline (every wave
//     on its own, B straight from the scratch image), so the pairing balances it: 0.56 of a full tile
//     instead of 0.75 (+2.7 %) — since round 4 in every sweep, and 0.52: the last 16 k-steps of a wave's range are its own lower-
//     triangular 64 x 64 block, whose zero 16 x 16 blocks get no MFMA (GPT_DIAG_TRIANGLE);
//   * work split: gpt_plan.h (rounds of whole blocks, then an explicit item list cut at quarter tiles — inside diagonal tiles
//     too where a workgroup's share is small).
// Alternatives measured and dropped (profiles/r01_kvar_variant_ab.txt): per-wave B generation (v1, 53 TF),
// 8-step chunks (-3.7 %), barrier-free sweeps with every wave reading B from the scratch image (-3 %: L2
// hit rate 97 % -> 56 %, 3.4 TB/s from beyond L2, clock 2.18 GHz), deeper A prefetch (0 %).
// NCOMP = 1: one column per query (k*).  NCOMP = 4: four columns per query (k*, dk_0, dk_1, dk_2).
// NCOMP = 3: D columns per query (dk_0 .. dk_{D-1}) — the Jacobian variance without the variance.
// DW = 3 is all of the above: D <= 3, source rows of 4 elements, query coordinates in registers.  DW = WIDE_D / MAX_D is the
// wide path for D = 4 .. 8 (rows of 8) / 9 .. 15 (rows of 16): only the generating sweep differs — the block's query coordinates sit in LDS
// (qs[d][query]), distances are coordinate loops — with NCOMP = 1, or NCOMP = 8 / 16: k*, dk_0 .. dk_{D-1} and zero
// columns up to 8 (D <= 7) or 16 per query, so that a query's columns stay inside one 16-column MFMA tile.  KSTAR = false
// (wide path, NCOMP = 4 for D = 4 and 8 for D = 8): the Jacobian variance alone, dk_0 .. dk_{D-1} without the k* column —
// half the columns of the fused layout at exactly those two dimensions.
// ------------------------------------------------------------------------------------------
// Timing-only ablation builds (results are wrong unless 0): -DGPT_ABL=1 no per-chunk barrier, 2 no A-operand
// loads, 3 diagonal tile skipped, 4 no B fill (LDS image left as is), 5 no MFMAs, 6 every wave 72 steps in the diagonal
// tile of a reload sweep (the work of a SIMD's pair split evenly).  tools/gpu_ablate.sh,
// profiles/r01_final_ablation.txt.
#ifndef GPT_ABL
#define GPT_ABL 0
#endif
// Further timing-only ablations (round 4, small models: tools/small_n_probe.py): 7 generating sweep without the exp (the
// squared distance itself is stored), 8 no copy of the generated fragments to the scratch image, 9 no fold of the
// accumulators into the column sums.
// -DGPT_VAR_TRACE (make trace): shader-clock stamps (s_memtime) of every wave of two workgroups at the phase boundaries of
// their first VT_ITEMS items, written to the buffer set through gpt_debug_set_var_trace.
#ifndef GPT_DIAG_TRIANGLE
#define GPT_DIAG_TRIANGLE 1      // diagonal tiles: no MFMAs for the 16 x 16 blocks above the diagonal of a wave's own 64 x 64 block
#endif
#ifndef GPT_GEN_DIAG_FREE
#define GPT_GEN_DIAG_FREE 1      // generating sweeps: diagonal tile barrier-free after its fragments went to the scratch image
#endif
#ifndef GPT_GEN_ROLLED
#define GPT_GEN_ROLLED 1         // openings of a generating sweep: one rolled copy of the generating code (instruction cache)
#endif
#ifndef GPT_GEN_BATCH_PROLOGUE
#define GPT_GEN_BATCH_PROLOGUE 1 // generating sweeps: the first chunk's four source loads in flight together
#endif
#ifndef GPT_GEN_STAGED_EXP
#define GPT_GEN_STAGED_EXP 1     // openings of a generating sweep: the four exps of a lane stage by stage
#endif
#ifndef GPT_DIAG_RING
#define GPT_DIAG_RING 4          // fp64 barrier-free diagonal tile: operand ring depth (2 = the loop of rounds 1-3)
#endif
#ifdef GPT_VAR_TRACE
constexpr int VT_STAMPS = 16, VT_ITEMS = 24, VT_WGS = 2;
__device__ long long* g_var_trace = nullptr;
// (inline asm with AMDGPU constraints has to sit in a __device__ function: in the body of a __global__ template the host
// pass rejects the constraint, silently drops the kernel's host stub and the library no longer loads)
static __device__ __forceinline__ long long vt_clock() {
    long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    return t_;
}
// the chip-wide 100 MHz counter (s_memtime counts per XCD: workgroups on different XCDs cannot be compared with it)
static __device__ __forceinline__ long long vt_realtime() {
    long long t_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    return t_;
}
#define GPT_VT(i) do { if (vt_base && lane == 0 && it < VT_ITEMS) vt_base[(size_t)it * VT_STAMPS + (i)] = vt_clock(); } while (0)
#else
#define GPT_VT(i) do { } while (0)
#endif
constexpr int VAR_SUB = 8;          // k4-steps per sub-chunk (= waves per workgroup: wave w fills step w of each)
// chunk double buffer (fp64: 2 x 32 steps x 2 KiB = 128 KiB; fp32: 64 KiB) or, fp32, the B image of a diagonal tile (128 steps x 1 KiB)
template <typename T> constexpr size_t var_lds_bytes() {
    constexpr size_t chunks = (size_t)2 * VAR_SUB * El<T>::SUBS * 64 * 4 * sizeof(T), image = (size_t)WT_K4 * 64 * 4 * sizeof(T);
    return (El<T>::DIAG_LDS && image > chunks) ? image : chunks;
}

template <typename T, int NCOMP, bool CROSS, int KT, int DW = 3, bool KSTAR = true, bool HALF = false>
__global__ __launch_bounds__(512, 2) void k_var(KernelParams p, VarPlanDev pl, const T* __restrict__ Xs,
                                                const T* __restrict__ Wf, const T* __restrict__ Xq,
                                                int64_t M, T* __restrict__ slab, T* __restrict__ vslab, T* __restrict__ bscratch) {
    typedef typename El<T>::v4 v4;
    typedef typename El<T>::avec avec;
    typedef typename El<T>::AF AF;
    constexpr size_t A_STEP = El<T>::A_STEP;
    constexpr int VAR_SUBS = El<T>::SUBS;               // sub-chunks per LDS chunk
    constexpr int VAR_CH = VAR_SUB * VAR_SUBS;          // k4-steps per LDS chunk, one barrier each (32 fp64 / 64 fp32; divides 128)
    extern __shared__ __attribute__((aligned(16))) unsigned char Bs_raw[];       // [buffer][k4-step][lane][column tile]
    T* const Bs_dyn = reinterpret_cast<T*>(Bs_raw);
    constexpr bool WIDE = DW != 3;
    constexpr int XS = WIDE ? DW : 4;                   // elements per source row
    constexpr int CPQ = NCOMP >= 4 ? NCOMP : 1;         // columns per query when they sit side by side (a power of two)
    static_assert(!WIDE || (DW == 8 && (NCOMP == 1 || NCOMP == 8 || NCOMP == 16 || (!KSTAR && NCOMP == 4))) || (DW == 16 && KSTAR && (NCOMP == 1 || NCOMP == 16)),
                  "wide path: rows of 8, NCOMP 1 / 8 / 16 (4 / 8 without k*); rows of 16, NCOMP 1 / 16");
    static_assert(WIDE || NCOMP == 1 || NCOMP == 3 || NCOMP == 4, "D <= 3: NCOMP 1 / 3 / 4");
    static_assert(KSTAR || (WIDE && !CROSS && (NCOMP == 4 || NCOMP == 8)), "no k* column: wide path, Jacobian variance alone");
    __shared__ T red[2][8][VAR_COLS];
    __shared__ double Tt[256];
    __shared__ T qs[WIDE ? DW : 1][VAR_COLS];           // wide path: scaled coordinates of this block's queries, [d][query]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lc = lane & 15, lk = lane >> 4;
    const int g = (w < 4) ? w : (11 - w);     // row group of this wave: 0,1,2,3,7,6,5,4
    const int D = p.D;
    auto Bs = [&](const int buf, const int step) -> T* { return Bs_dyn + ((size_t)(buf * VAR_CH + step) * 64 + lane) * 4; };
    if (std::is_same<T, double>::value && threadIdx.x < 256) Tt[threadIdx.x] = g_exp2_table[threadIdx.x];
#ifdef GPT_VAR_TRACE
    long long* vt_base = nullptr;
    if (g_var_trace && (blockIdx.x == 0 || blockIdx.x == 37))
        vt_base = g_var_trace + ((size_t)(blockIdx.x == 0 ? 0 : 1) * 8 + w) * VT_ITEMS * VT_STAMPS;
    // begin / end time (100 MHz, chip-wide) of every workgroup (wave 0), behind the phase stamps: [VT_WGS * 8 * VT_ITEMS * VT_STAMPS + 2 * blockIdx.x]
    long long* const vt_wg = (g_var_trace && threadIdx.x == 0 && blockIdx.x < 1024) ? g_var_trace + (size_t)VT_WGS * 8 * VT_ITEMS * VT_STAMPS + 2 * blockIdx.x : nullptr;
    if (vt_wg) vt_wg[0] = vt_realtime();
#endif

    constexpr T RS2 = (T)0.70710678118654752440;    // coordinates are pre-scaled by 1/sqrt(2): t = ln c - |d'|^2
    // NCOMP=4: column = 4 query + comp, comp = lc & 3 in every tile: b = kv * (cbv + sum_d cd[d] * d'_d).
    // NCOMP=3 (Jacobian variance alone, no k* column): column = D query + d, D columns per query; for D = 3 the d of a
    // lane's column changes from tile to tile and from block to block (16 = 64 = 1 mod 3), selected in `produce`.
    const int comp = (CPQ > 1) ? (lc & (CPQ - 1)) + (KSTAR ? 0 : 1) : 0;      // 0: the k* column, 1 + d: dk_d
    const T cbv = (comp == 0) ? (T)1 : (T)0;
    T cd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) cd[d] = (!WIDE && comp == d + 1 && d < D) ? (T)(p.inv_ls[d] * 1.41421356237309504880) : (T)0;
    // wide path: a derivative column needs one coordinate beyond the distance, its own: dimension own_d, factor sc_own
    // (zero in the k* column and in the zero columns)
    const int own_d = (WIDE && comp >= 1 && comp <= D) ? comp - 1 : 0;
    T sc_own = (T)0;
    if (WIDE) {
#pragma unroll
        for (int d = 0; d < DW; ++d) if (comp == d + 1 && d < D) sc_own = (T)(p.inv_ls[d] * 1.41421356237309504880);
    }
    const T lnc = (T)p.lnc;
    const int nbi = pl.nbi;
    // A stream: element (step S, group g, ...) — uniform base + per-lane 32-bit offset (scalar-base addressing: no
    // 64-bit VALU address arithmetic, and no VALU writes into registers that loads are still in flight to)
    const avec* const wuni = reinterpret_cast<const avec*>(Wf) + (size_t)g * El<T>::A_GROUP;
    // this workgroup's B image: k-step s, lane l at s*64 + l (v4 units)
    v4* const buni = reinterpret_cast<v4*>(bscratch) + (size_t)blockIdx.x * ((size_t)p.NP * 16);

    const int per_block = pl.ntask * nbi;                       // sweeps of a whole block
    const int64_t n_implicit = (pl.rnd_end - pl.rnd_begin) * per_block;     // this launch's rounds
    const int it_begin = pl.item_begin[blockIdx.x], it_end = pl.item_begin[blockIdx.x + 1];

    T ssq[4] = {0, 0, 0, 0}, crs[4] = {0, 0, 0, 0};
    for (int64_t it = 0;; ++it) {
        // ---- next item: derived (rounds of whole blocks, in step with every other workgroup) or listed (tail)
        int64_t cb; int task, ib, k_lo, k_hi, flags, slot, vslot;
        if (it < n_implicit) {
            const int64_t rl = it / per_block;
            const int r = (int)(it - rl * per_block);
            const int64_t rnd = pl.rnd_begin + rl;
            task = r / nbi;
            ib = nbi - 1 - (r - task * nbi);
            cb = rnd * pl.P + blockIdx.x;
            k_lo = 0; k_hi = VAR_KQ * (ib + 1);
            flags = (r == 0 ? (VI_FIRST | VI_GEN) : 0) | (ib == nbi - 1 ? VI_ZERO : 0);
            slot = (ib == 0) ? (int)(cb * pl.ntask + task) : -1;
            vslot = -1;
        } else {
            const int64_t idx = it_begin + (it - n_implicit);
            if (!pl.with_tail || idx >= it_end) break;
            const VarItem item = pl.items[idx];
            cb = item.cb; task = item.task; ib = item.ib; k_lo = item.k_lo; k_hi = item.k_hi;
            flags = item.flags; slot = item.slot; vslot = item.vslot;
        }
        GPT_VT(0);
        if (flags & VI_FIRST) {
            __syncthreads();                               // LDS (Bs, red, Tt) free / ready
            // The scratch image changes owner: block n's fragments overwrite block n - 1's.  Writers and readers of a workgroup's
            // image are the waves of THAT workgroup, i.e. of one CU, and its vector L1 is coherent among them (a store through the
            // L1 updates or drops the line it hits; AMDGPU memory model, non-tgsplit mode: "no special action is required for
            // coherence between wavefronts in the same work-group") — so workgroup scope is the scope that is needed: ordering, no
            // cache invalidation.  Until round 4 this was an AGENT-scope acquire (buffer_inv sc1): 20 000 - 35 000 clocks at the
            // opening of every block, 3 % of the kernel at N = 1024 (ablation 10 in profiles/r04_small_n.txt).
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        } else if (flags & VI_GEN) {
            __syncthreads();                               // nobody may still be reading the part of the image rewritten now
        }
        if (WIDE && (flags & VI_FIRST)) {
            // the block's queries (64 for NCOMP = 1, 64 / NCOMP otherwise), coordinate w (and w + 8 in rows of 16) by wave w, scaled as the sources are
            constexpr int NQ = VAR_COLS / (CPQ > 1 ? CPQ : 1);
            const int64_t m = cb * NQ + lane;
            const int64_t mm = (m < M) ? m : (M - 1);
#pragma unroll
            for (int c0 = 0; c0 < DW; c0 += 8) {
                const int cw = c0 + w;
                double il = 0.0;
#pragma unroll
                for (int d = 0; d < DW; ++d) if (d == cw) il = p.inv_ls[d];
                if (lane < NQ) qs[cw][lane] = (cw < D) ? Xq[mm * D + cw] * (T)(il * 0.70710678118654752440) : (T)0;
            }
            __syncthreads();
        }
        if (flags & VI_ZERO) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { ssq[t] = (T)0; crs[t] = (T)0; }
        }
        GPT_VT(1);
        const int base3 = (NCOMP == 3) ? (int)((cb * VAR_COLS + lc) % D) : 0;
        const T sc3[3] = {(T)(p.inv_ls[0] * 1.41421356237309504880), (T)(p.inv_ls[1] * 1.41421356237309504880),
                          (T)(p.inv_ls[2] * 1.41421356237309504880)};

        // One sweep.  GEN = true: B fragments are generated and a copy is kept in the scratch image; GEN = false: they
        // are reloaded from it.  Two instantiations, so that the query coordinates and exp temporaries of the generating
        // sweep do not occupy registers in the others.
        auto sweep = [&](auto gen_tag) {
            constexpr bool GEN = decltype(gen_tag)::value;
            // this lane's four columns (one per MFMA column tile): scaled query coordinates
            GPT_VT(11);
            T q[4][3];
            if (GEN && !WIDE) {
                // twelve loads, no branch between them (a missing coordinate reads coordinate 0 and is scaled by zero): written as
                // `(d < D) ? Xq[..] : 0` each load sat in its own basic block behind its own s_waitcnt vmcnt(0) — twelve memory
                // latencies in a row at the opening of every generating sweep (r04_small_n.txt; a prefetch of the next block's
                // coordinates into LDS by the waves that finish the last diagonal tile early was also built: with the loads batched it
                // saved 1 500 clocks per block and cost the 3-column kernel spill code in its hot loop — removed)
                T raw[4][3], qsc[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) qsc[d] = (d < D) ? (T)(p.inv_ls[d] * 0.70710678118654752440) : (T)0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int64_t col = cb * VAR_COLS + 16 * t + lc;
                    const int64_t m = (NCOMP == 1) ? col : ((NCOMP == 4) ? (col >> 2) : (col / D));
                    const int64_t mm = (m < M) ? m : (M - 1);
                    const T* qp = Xq + mm * D;
#pragma unroll
                    for (int d = 0; d < 3; ++d) raw[t][d] = qp[d < D ? d : 0];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int d = 0; d < 3; ++d) q[t][d] = raw[t][d] * qsc[d];
            }
#ifdef GPT_VAR_TRACE
            if (GEN) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }      // (trace builds: the query coordinates have arrived)
            GPT_VT(9);
#endif
            // rows of 16 (and the wide kernels with cross terms): the source's coordinates are read where they are used (produce_to), four
            // at a time — carried across the MFMA steps like the narrower rows' they are 16 - 32 more registers than the fp64 kernel has, and
            // it spilled inside its loops
            constexpr bool LATE_X = GEN && WIDE && (DW == 16 || CROSS);
            T gx[LATE_X ? 1 : DW];                         // coordinates of the source this wave generates next
            T gx_own = (T)0;                               // (wide) and the one a derivative column multiplies by
            v4 bl;                                         // or the fragments it reloads next
            auto load_x_to = [&](const T* xp, auto& ox, T& oown) {
                if constexpr (LATE_X) {
                    (void)xp; (void)ox; (void)oown;
                } else if constexpr (WIDE) {
#pragma unroll
                    for (int v = 0; v < DW / 4; ++v) {
                        const v4 xv = *reinterpret_cast<const v4*>(xp + 4 * v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ox[4 * v + e] = xv[e];
                    }
                    if (NCOMP != 1) oown = xp[own_d];
                } else {
                    ox[0] = xp[0]; ox[1] = xp[1]; ox[2] = xp[2];
                }
            };
            auto load_x = [&](const T* xp) { load_x_to(xp, gx, gx_own); };
            auto fetch = [&](const int k4) {
                if (GEN) load_x(Xs + (size_t)(k4 * 4 + lk) * XS);
                else bl = (buni + (size_t)k4 * 64)[lane];
            };
            // B fragments of k-step k4 -> LDS chunk buffer `buf` (to_lds) and, when generated, the scratch image
            // staged: the four exps of a lane stage by stage (gpt_exp.h kernel_tab4: their latencies overlap — the openings of a
            // sweep, where no MFMA hides them); not staged: one after the other, as few live registers as possible (inside the
            // MFMA loop, where the staged form spills)
            auto produce_to = [&](auto lds_tag, auto staged_tag, const int buf, const int k4, const bool to_scr = true, const bool lds_on = true) {
                constexpr bool to_lds = decltype(lds_tag)::value;
                constexpr bool staged = decltype(staged_tag)::value;
                T* dstl = Bs(buf, k4 % VAR_CH);
                if constexpr (LATE_X) {
                    const T* xp = Xs + (size_t)(k4 * 4 + lk) * XS;
                    const T xo = (NCOMP != 1) ? xp[own_d] * RS2 : (T)0;
                    v4 b;
                    T hh[4] = {(T)0, (T)0, (T)0, (T)0}, kv[4];
#pragma unroll
                    for (int v = 0; v < DW / 4; ++v) {
                        const v4 xv = *reinterpret_cast<const v4*>(xp + 4 * v);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int qi = (16 * t + lc) / (CPQ > 1 ? CPQ : 1);      // this column's query within the block
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const T df = fma(xv[e], RS2, -qs[4 * v + e][qi]); hh[t] = fma(df, df, hh[t]); }
                        }
                    }
                    if constexpr (staged) kernel_tab4<KT>(hh, lnc, Tt, kv);
                    else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) kv[t] = kernel_tab<KT>(hh[t], lnc, Tt);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int qi = (16 * t + lc) / (CPQ > 1 ? CPQ : 1);
                        b[t] = (NCOMP == 1) ? kv[t] : kv[t] * (cbv + sc_own * (xo - qs[own_d][qi]));
                    }
                    if (to_lds && lds_on) *reinterpret_cast<v4*>(dstl) = b;
                    if (to_scr) (buni + (size_t)k4 * 64)[lane] = b;
                } else if constexpr (GEN && WIDE) {
                    T x[DW];
#pragma unroll
                    for (int d = 0; d < DW; ++d) x[d] = gx[d] * RS2;
                    const T xo = gx_own * RS2;
                    v4 b;
                    T hh[4], kv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int qi = (16 * t + lc) / (CPQ > 1 ? CPQ : 1);      // this column's query within the block
                        T df = x[0] - qs[0][qi];
                        hh[t] = df * df;
#pragma unroll
                        for (int d = 1; d < DW; ++d) { df = x[d] - qs[d][qi]; hh[t] = fma(df, df, hh[t]); }
                    }
                    if constexpr (staged) kernel_tab4<KT>(hh, lnc, Tt, kv);
                    else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) kv[t] = kernel_tab<KT>(hh[t], lnc, Tt);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int qi = (16 * t + lc) / (CPQ > 1 ? CPQ : 1);
                        b[t] = (NCOMP == 1) ? kv[t] : kv[t] * (cbv + sc_own * (xo - qs[own_d][qi]));
                    }
                    if (to_lds && lds_on) *reinterpret_cast<v4*>(dstl) = b;
                    if (to_scr) (buni + (size_t)k4 * 64)[lane] = b;
                } else if (GEN) {
                    const T x0 = gx[0] * RS2, x1 = gx[1] * RS2, x2 = gx[2] * RS2;
                    v4 b;
                    auto column = [&](const int t, const T kv, const T d0, const T d1, const T d2_) -> T {
                        if (NCOMP == 3) {
                            int dsel = base3 + ((D == 3) ? t : 0);            // (64 cb + 16 t + lc) mod D, base3 = (64 cb + lc) mod D
                            dsel = (dsel >= D) ? dsel - D : dsel;
                            const T e = (dsel == 0) ? d0 * sc3[0] : ((dsel == 1) ? d1 * sc3[1] : d2_ * sc3[2]);
                            return kv * e;
                        }
                        return (NCOMP == 1) ? kv : kv * (cbv + cd[0] * d0 + cd[1] * d1 + cd[2] * d2_);
                    };
                    if constexpr (staged) {
                        T d0[4], d1[4], d2_[4], hh[4], kv[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            d0[t] = x0 - q[t][0]; d1[t] = x1 - q[t][1]; d2_[t] = x2 - q[t][2];
                            hh[t] = d0[t] * d0[t];
                            hh[t] = fma(d1[t], d1[t], hh[t]);
                            hh[t] = fma(d2_[t], d2_[t], hh[t]);
                        }
                        if (GPT_ABL == 7) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) kv[t] = hh[t];
                        } else {
                            kernel_tab4<KT>(hh, lnc, Tt, kv);
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) b[t] = column(t, kv[t], d0[t], d1[t], d2_[t]);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const T d0 = x0 - q[t][0], d1 = x1 - q[t][1], d2_ = x2 - q[t][2];
                            T hh = d0 * d0;
                            hh = fma(d1, d1, hh);
                            hh = fma(d2_, d2_, hh);
                            const T kv = (GPT_ABL == 7) ? hh : kernel_tab<KT>(hh, lnc, Tt);
                            b[t] = column(t, kv, d0, d1, d2_);
                        }
                    }
                    if (to_lds && lds_on) *reinterpret_cast<v4*>(dstl) = b;
                    if (GPT_ABL != 8 && to_scr) (buni + (size_t)k4 * 64)[lane] = b;
                } else {
                    *reinterpret_cast<v4*>(dstl) = bl;
                }
            };
            auto produce = [&](const int buf, const int k4) { produce_to(std::true_type{}, std::false_type{}, buf, k4); };
            // GEN: `cnt` k-steps k4_0 + j * stride generated with their source loads in flight together (load -> wait -> exp ->
            // store one at a time cost 9.5 k cycles per k-step at the opening of a sweep: profiles/r04_small_n.txt)
            auto generate_batch = [&](auto lds_tag, auto cnt_tag, const int buf, const int k4_0, const int stride, const bool to_scr = true,
                                      const bool lds_on = true) {
                constexpr int cnt = decltype(cnt_tag)::value;
                T bx[cnt][LATE_X ? 1 : DW], bo[cnt];
#pragma unroll
                for (int j = 0; j < cnt; ++j) { bo[j] = (T)0; load_x_to(Xs + (size_t)((k4_0 + j * stride) * 4 + lk) * XS, bx[j], bo[j]); }
#pragma unroll
                for (int j = 0; j < cnt; ++j) {
#pragma unroll
                    for (int d = 0; d < (LATE_X ? 1 : DW); ++d) gx[d] = bx[j][d];
                    gx_own = bo[j];
                    produce_to(lds_tag, std::integral_constant<bool, GPT_GEN_STAGED_EXP != 0>{}, buf, k4_0 + j * stride, to_scr, lds_on);
                }
            };

            const size_t S_ib = (size_t)task * pl.tiles_per_task * WT_K4 + (size_t)64 * ib * (ib + 1);   // stream index of k4-step 0 of this i-block
            static_assert(VAR_CH == VAR_Q_COST && WT_K4 == VAR_KQ * VAR_CH, "an item's k range counts LDS chunks (quarter tiles)");
            // the item's part of the diagonal tile: k4-steps [d_lo, d_hi) of it — the whole tile (0, 128) everywhere except in the lists
            // of small launches, where the plan may cut it at quarters (gpt_plan.h: cut_diag)
            const bool has_diag = k_hi > VAR_KQ * ib;
            const int d_lo = (k_lo > VAR_KQ * ib ? k_lo - VAR_KQ * ib : 0) * VAR_CH, d_hi = (k_hi - VAR_KQ * ib) * VAR_CH;
            const int K0 = k_lo * VAR_CH;                                      // first k4-step of this item
            // The diagonal tile (where wave g only has 16 (g + 1) steps of work) runs OUT of the lock-step LDS pipeline: see
            // below.  A generating sweep first puts that tile's fragments into the scratch image (GEN_DIAG_FREE; until round 4 it
            // kept the tile inside the lock-step part, where it costs 0.75 of a full tile instead of 0.56).
            constexpr bool DIAG_FREE = !GEN || GPT_GEN_DIAG_FREE != 0;
            const int lock_end = ((!DIAG_FREE || !has_diag) ? k_hi : VAR_KQ * ib) * VAR_CH;
            const int ch0 = K0 / VAR_CH, ch1 = lock_end / VAR_CH;              // lock-step chunks [ch0, ch1)
            // fp64, small models (HALF: its own instantiation of the kernel, so that the N = 8192 kernel keeps its code and registers):
            // the first 64 k-steps of a diagonal tile's B image go through LDS — see the tile below
            constexpr bool half = HALF && !El<T>::DIAG_LDS;
            if (GEN && DIAG_FREE && has_diag && GPT_ABL != 3 && !half) {
                // 128 k-steps (of a whole tile), 16 per wave (w, w + 8, ...), VALU only; complete and visible before the barrier below
                const int kd0 = ib * WT_K4;
                if (GPT_GEN_ROLLED != 0) {
#pragma unroll 1
                    for (int j = d_lo / VAR_SUB; j < d_hi / VAR_SUB; ++j)
                        generate_batch(std::false_type{}, std::integral_constant<int, 1>{}, 0, kd0 + w + VAR_SUB * j, VAR_SUB);
                } else {
#pragma unroll 1
                    for (int j0 = d_lo / VAR_SUB; j0 < d_hi / VAR_SUB; j0 += 4)
                        generate_batch(std::false_type{}, std::integral_constant<int, 4>{}, 0, kd0 + w + VAR_SUB * j0, VAR_SUB);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            GPT_VT(10);
            if (ch1 > ch0) {                                  // first chunk: wave w fills steps w, w+8, w+16, w+24
                if (GEN && GPT_GEN_ROLLED != 0) {
                    // ONE copy of the generating code, run VAR_SUBS times: the opening of a sweep is straight-line code that runs once
                    // per block, i.e. from a cold instruction cache — its time followed its LENGTH, not its arithmetic (four k-steps
                    // unrolled and batched: 40 000 clocks; twenty: 60 000; serial or staged exps: no difference — r04_small_n.txt)
#pragma unroll 1
                    for (int j = 0; j < VAR_SUBS; ++j)
                        generate_batch(std::true_type{}, std::integral_constant<int, 1>{}, ch0 & 1, K0 + j * VAR_SUB + w, VAR_SUB);
                } else if (GEN && GPT_GEN_BATCH_PROLOGUE != 0) {
                    generate_batch(std::true_type{}, std::integral_constant<int, VAR_SUBS>{}, ch0 & 1, K0 + w, VAR_SUB);
                } else if (!GEN && El<T>::BATCH_PROLOGUE) {
                    // the four reloads in flight together instead of load -> wait -> write four times (short fp32 sweeps:
                    // 12 per block at configs[4], each opening with this latency)
                    v4 pre[VAR_SUBS];
#pragma unroll
                    for (int j = 0; j < VAR_SUBS; ++j) pre[j] = (buni + (size_t)(K0 + j * VAR_SUB + w) * 64)[lane];
#pragma unroll
                    for (int j = 0; j < VAR_SUBS; ++j) *reinterpret_cast<v4*>(Bs(ch0 & 1, (K0 + j * VAR_SUB + w) % VAR_CH)) = pre[j];
                } else {
#pragma unroll
                    for (int j = 0; j < VAR_SUBS; ++j) {
                        fetch(K0 + j * VAR_SUB + w);
                        produce(ch0 & 1, K0 + j * VAR_SUB + w);
                    }
                }
            }
            GPT_VT(2);
            constexpr int PF = El<T>::PF;                  // divides VAR_SUB, so step s of every sub-chunk uses ring slot s % PF
            AF a_ring[PF];                                 // A fragments of the next PF steps
#pragma unroll
            for (int i = 0; i < PF; ++i)                   // the first 16 steps of an item are active for every group
                El<T>::lda(a_ring[i], wuni + (S_ib + K0 + i) * A_STEP, lane);
            __syncthreads();
            GPT_VT(3);
            v4 acc[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = v4{0, 0, 0, 0};
            const int my_limit = ib * WT_K4 + 16 * (g + 1);     // first k4-step of the i-block with nothing left for this group
            // Sources of the fills inside the loop, as loop-carried per-lane pointers (k-step K0 + VAR_CH + w first, then
            // VAR_SUB further each time): an address recomputed from the k-step lands in whatever registers are free —
            // the previous fill's destination registers — and that write-after-load made hipcc drain vmcnt to 0 (and with
            // it the A fragments in flight) at the top of every sub-chunk.
            const v4* bsrc = buni + (size_t)(K0 + VAR_CH + w) * 64 + lane;
            const T* xsrc = Xs + (size_t)((K0 + VAR_CH + w) * 4 + lk) * XS;
            auto fetch_next = [&]() {
                if (GEN) {
                    load_x(xsrc);
                    xsrc += VAR_SUB * 4 * XS;
                } else {
                    bl = bsrc[0];
                    bsrc += VAR_SUB * 64;
                }
            };
            // the chunk body exists twice — with and without the fill of the following chunk — so that "is there a next
            // chunk" is no branch (and no join in front of the first MFMAs of a sub-chunk) inside it
            auto chunk = [&](auto more_tag, const int ch) {
                constexpr bool more = decltype(more_tag)::value;
                const int cur = ch & 1;
                v4 b_nxt = *reinterpret_cast<const v4*>(Bs(cur, 0));    // B fragments are read one k-step ahead
                for (int sub = 0; sub < VAR_SUBS; ++sub) {
                    const int k0 = ch * VAR_CH + sub * VAR_SUB;                 // first k-step of this sub-chunk
                    const int kn = (ch + 1) * VAR_CH + sub * VAR_SUB + w;       // the k-step this wave fills meanwhile
                    if (more) fetch_next();
                    const bool active = (k0 < my_limit) && !(GPT_ABL == 3 && k0 >= ib * WT_K4);   // my_limit is a multiple of 16: all or nothing
                    auto step = [&](const int s) {
                        const int k4 = k0 + s;
                        const AF a = a_ring[s % PF];
                        const v4 b = b_nxt;
                        const int kl = my_limit - 1;
                        const size_t Sn = S_ib + ((k4 + PF < my_limit) ? (k4 + PF) : kl);
                        if (GPT_ABL != 2) El<T>::lda(a_ring[s % PF], wuni + Sn * A_STEP, lane);
                        const int sn = sub * VAR_SUB + s + 1;
                        if (sn < VAR_CH) b_nxt = *reinterpret_cast<const v4*>(Bs(cur, sn));
                        if (GPT_ABL == 5) { El<T>::keep(a, b); return; }
                        El<T>::mfma16(acc, a, b);
                    };
                    // The fill of the next chunk sits INSIDE the active / idle paths, not behind their join: vmcnt counts in
                    // issue order, and behind a join hipcc has to wait for vmcnt(0) — which also waits for the A fragments
                    // the steps just before have requested (a full L2 round trip per sub-chunk); inside the straight-line
                    // path it waits for the fill's own loads only.
                    if (active) {
                        step(0); step(1);
                        if (more && w < 4 && GPT_ABL != 4) produce(cur ^ 1, kn);
                        step(2); step(3); step(4); step(5);
                        if (more && w >= 4 && GPT_ABL != 4) produce(cur ^ 1, kn);
                        step(6); step(7);
                    } else {
                        if (more && GPT_ABL != 4) produce(cur ^ 1, kn);
                        if (sub + 1 < VAR_SUBS) b_nxt = *reinterpret_cast<const v4*>(Bs(cur, (sub + 1) * VAR_SUB));
                    }
                }
                if (GPT_ABL != 1) __syncthreads();
            };
            for (int ch = ch0; ch + 1 < ch1; ++ch) chunk(std::true_type{}, ch);
            if (ch1 > ch0) chunk(std::false_type{}, ch1 - 1);
            GPT_VT(4);
            if (DIAG_FREE && has_diag && GPT_ABL != 3) {
                // Diagonal tile, barrier-free: every wave runs its own 16 (g + 1) k-steps on its own.  No
                // lock-step, so the waves with g and 7 - g that share a SIMD add up to the same work on every SIMD: the tile
                // costs 0.56 of a full one instead of the 0.75 it costs inside the lock-step pipeline.
                const int kd0 = ib * WT_K4;                              // first k-step of the diagonal tile
                const int lim_g = (GPT_ABL == 6) ? 72 : 16 * (g + 1);    // multiple of 16 (ablation 6: every wave the average, 72: what an even split inside a SIMD would cost)
                const int limit = lim_g < d_hi ? lim_g : d_hi;           // this wave's steps: [d_lo, limit), none if limit <= d_lo
                const avec* ap = wuni + (S_ib + kd0) * A_STEP;
                const v4* bp = buni + (size_t)kd0 * 64;
                auto ldA = [&](AF& a, const int k) {
                    const int kk = k < limit ? k : limit - 1;            // clamped: redundant, in bounds
                    if (GPT_ABL == 12 && k >= d_lo + 8) return;          // (timing-only ablation: the tile without its A stream)
                    El<T>::lda(a, ap + (size_t)kk * A_STEP, lane);
                };
                // The last 16 k-steps of a wave's range are its own 64 x 64 diagonal block of the factor (when the item's range reaches
                // that far): W is lower triangular, so row tile r of the group has nothing but zeros from the block's k-step 4 (r + 1)
                // on, and those MFMAs — 24 of the block's 64 tile-steps, 8 % of the whole tile — are not issued.  Adding 0 x b changes
                // nothing, so results are the same to the bit.  body(first row tile with work, k4) runs RR steps from k4.
                // (Not in the instantiations that sit at the register limit — cross terms, the fp64 3-column kernel, wide Matern: the
                // peeled steps cost them a spilled pointer inside the ring loop, tools/check_isa_spills.py.)
                constexpr bool TRI_OK = GPT_DIAG_TRIANGLE != 0 && !CROSS && !(std::is_same<T, double>::value && NCOMP == 3) && !(WIDE && KT != KT_RBF);
                const bool tri = TRI_OK && limit == lim_g && limit > d_lo && GPT_ABL != 6;
                auto tri_loop = [&](auto rtag, const int k_begin, const int k_end, const bool tri_end, auto&& body) {
                    constexpr int RR = decltype(rtag)::value;
                    static_assert(RR == 2 || RR == 4, "ring depth of the diagonal tile");
                    const int k_main = tri_end ? k_end - 12 : k_end;
                    for (int k4 = k_begin; k4 < k_main; k4 += RR) body(std::integral_constant<int, 0>{}, k4);
                    if (tri_end) {
                        if constexpr (RR == 4) {
                            body(std::integral_constant<int, 1>{}, k_end - 12); body(std::integral_constant<int, 2>{}, k_end - 8);
                            body(std::integral_constant<int, 3>{}, k_end - 4);
                        } else {
                            body(std::integral_constant<int, 1>{}, k_end - 12); body(std::integral_constant<int, 1>{}, k_end - 10);
                            body(std::integral_constant<int, 2>{}, k_end - 8); body(std::integral_constant<int, 2>{}, k_end - 6);
                            body(std::integral_constant<int, 3>{}, k_end - 4); body(std::integral_constant<int, 3>{}, k_end - 2);
                        }
                    }
                };
                if constexpr (El<T>::DIAG_LDS) {
                    // fp32: B through LDS.  With each wave re-reading its 16 (g + 1) steps of the B image from L2 / Infinity Cache
                    // (as the fp64 path below does) the tile cost 0.75 of a full one after all: 576 KiB per tile and workgroup
                    // in half the time an fp64 tile gives (profiles/r02_svgp_variants.txt).  The whole image of the tile — 128
                    // k-steps x 1 KiB — fits in LDS now that the chunk buffers are free (the last chunk's barrier has passed):
                    // the 8 waves copy it ONCE, 16 k-steps each, then run barrier-free with B from LDS.
                    v4* const img = reinterpret_cast<v4*>(Bs_dyn);       // [k-step 0..128)[lane]
                    constexpr int DP = El<T>::PF;
                    AF a[DP];
#pragma unroll
                    for (int i = 0; i < DP; ++i) ldA(a[i], d_lo + i);
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if (d_lo >= 64 * (half + 1) || d_hi <= 64 * half) continue;      // (workgroup-uniform: not a k-step of this item)
                        v4 r[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) r[j] = (bp + (size_t)(w + 8 * (8 * half + j)) * 64)[lane];
#pragma unroll
                        for (int j = 0; j < 8; ++j) img[(w + 8 * (8 * half + j)) * 64 + lane] = r[j];
                    }
                    __syncthreads();
                    GPT_VT(12);
                    v4 b_nxt = img[d_lo * 64 + lane];
                    tri_loop(std::integral_constant<int, DP>{}, d_lo, limit > d_lo ? limit : d_lo, tri, [&](auto jtag, const int k4) {
#pragma unroll
                        for (int i = 0; i < DP; ++i) {
                            const v4 b = b_nxt;
                            const int kn = (k4 + i + 1 < limit) ? (k4 + i + 1) : (limit - 1);
                            if (GPT_ABL != 13) b_nxt = img[kn * 64 + lane];          // (13: timing-only, the tile without its B reads)
                            El<T>::template mfma_from<decltype(jtag)::value>(acc, a[i], b);
                            ldA(a[i], k4 + i + DP);
                        }
                    });
                    GPT_VT(14);
                    __syncthreads();                                     // the image is free again (next sweep's first fill)
                } else {
                    // fp64: A from Wf and B straight from the scratch image, both one MFMA block (1024 cycles) ahead; program
                    // order pinned with sched_barrier so hipcc keeps the loads away from their first use.  (Staging the first
                    // 64 k-steps of the image in LDS, as far as 128 KiB go, was 1.1 % slower — two more barriers per tile — for 4 %
                    // fewer fetched bytes: profiles/r02_kvar_diag_image_fp64.txt.)
                    auto ldB = [&](v4& b, const int k) {
                        const int kk = k < limit ? k : limit - 1;
                        if (GPT_ABL == 13 && k >= d_lo + 8) return;
                        b = (bp + (size_t)kk * 64)[lane];
                    };
                    // (2: the round-1 .. 3 loop, kept for A/B — and for the 3-column kernel, whose generating side keeps more state alive:
                    // with the deeper ring hipcc reloads a spilled pointer inside the loop, and that reload's wait drains the ring)
                    constexpr int R = (NCOMP == 3) ? 2 : GPT_DIAG_RING;
                    if constexpr (half) {
                        // Small models (N <= 2560): the B images of an XCD's 32 workgroups (0.5 MB each at N = 1024) do not stay in its
                        // 4 MB of L2, a diagonal tile read straight from the scratch image is 576 wave-steps x 2 KiB from beyond L2, the
                        // wave that is alone on its SIMD runs at the latency of those reads, and the 256 KB a generating sweep writes
                        // for its own diagonal tile leave 256 workgroups at the same moment (60 000 clocks per opening:
                        // profiles/r04_small_n.txt).  Here the first 64 k-steps of the tile's image — all that waves 0..3 need, half of
                        // what waves 4..7 need — sit in LDS (the chunk buffers are free: the last lock-step barrier has passed): a
                        // generating sweep produces them there (and in the scratch image only when a later sweep will reload them), a
                        // reload sweep copies them from the scratch image ONCE, 8 k-steps per wave; steps 64.. still come from the
                        // scratch image, R blocks ahead.  Two barriers per tile (at N = 8192, where the image is L2-resident, that was
                        // a loss of 1.1 %: r02_kvar_diag_image_fp64.txt — hence by size).
                        v4* const img = reinterpret_cast<v4*>(Bs_dyn);    // [k-step 0..64)[lane] = Bs(buf = k / 32, k % 32)
                        if constexpr (GEN) {
                            const bool priv = it < n_implicit && pl.ntask == 1;      // the top sweep's own tile: nobody reloads these k-steps
#pragma unroll 1
                            for (int j = d_lo / VAR_SUB; j < d_hi / VAR_SUB; ++j) {      // k-steps kd0 + w + 8 j: below 64 -> LDS (+ scratch unless private), the rest -> scratch
                                const int k = kd0 + w + VAR_SUB * j;
                                const bool low = j < 8;
                                generate_batch(std::true_type{}, std::integral_constant<int, 1>{}, low ? (k - kd0) / VAR_CH : 0, k, VAR_SUB, !(priv && low), low);
                            }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        } else {
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
                                if (d_lo >= 32 * (hh + 1) || d_hi <= 32 * hh) continue;      // (workgroup-uniform)
                                v4 r[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) r[j] = (bp + (size_t)(w + 8 * (4 * hh + j)) * 64)[lane];
#pragma unroll
                                for (int j = 0; j < 4; ++j) img[(w + 8 * (4 * hh + j)) * 64 + lane] = r[j];
                            }
                        }
                        __syncthreads();
                        GPT_VT(12);
                        AF a[R];
                        v4 b[R];
#pragma unroll
                        for (int i = 0; i < R; ++i) ldA(a[i], d_lo + i);
                        const int l1 = limit < 64 ? limit : 64;
                        const int p2 = d_lo > 64 ? d_lo : 64;             // first step that comes from the scratch image
                        v4 b_nxt = img[(d_lo < 64 ? d_lo : 63) * 64 + lane];
                        tri_loop(std::integral_constant<int, R>{}, d_lo, l1 > d_lo ? l1 : d_lo, tri && limit <= 64, [&](auto jtag, const int k4) {
#pragma unroll
                            for (int i = 0; i < R; ++i) {
                                const v4 bb = b_nxt;
                                const int kn = (k4 + i + 1 < l1) ? (k4 + i + 1) : (l1 - 1);
                                b_nxt = img[kn * 64 + lane];
                                __builtin_amdgcn_sched_barrier(0);
                                El<T>::template mfma_from<decltype(jtag)::value>(acc, a[i], bb);
                                __builtin_amdgcn_sched_barrier(0);
                                ldA(a[i], k4 + i + R);
                            }
                        });
                        GPT_VT(13);
                        if (limit > p2) {
#pragma unroll
                            for (int i = 0; i < R; ++i) ldB(b[i], p2 + i);
                        }
                        tri_loop(std::integral_constant<int, R>{}, p2, limit > p2 ? limit : p2, tri && limit > 64, [&](auto jtag, const int k4) {
#pragma unroll
                            for (int i = 0; i < R; ++i) {
                                __builtin_amdgcn_sched_barrier(0);
                                El<T>::template mfma_from<decltype(jtag)::value>(acc, a[i], b[i]);
                                __builtin_amdgcn_sched_barrier(0);
                                ldA(a[i], k4 + i + R); ldB(b[i], k4 + i + R);
                            }
                        });
                        GPT_VT(14);
                        __syncthreads();                                  // the image is free again (next sweep's first fill)
                    } else if constexpr (R == 2) {
                        AF a0, a1;
                        v4 b0, b1;
                        ldA(a0, d_lo); ldA(a1, d_lo + 1); ldB(b0, d_lo);
                        tri_loop(std::integral_constant<int, 2>{}, d_lo, limit > d_lo ? limit : d_lo, tri, [&](auto jtag, const int k4) {
                            ldB(b1, k4 + 1);
                            __builtin_amdgcn_sched_barrier(0);
                            El<T>::template mfma_from<decltype(jtag)::value>(acc, a0, b0);
                            __builtin_amdgcn_sched_barrier(0);
                            ldA(a0, k4 + 2); ldB(b0, k4 + 2);
                            __builtin_amdgcn_sched_barrier(0);
                            El<T>::template mfma_from<decltype(jtag)::value>(acc, a1, b1);
                            __builtin_amdgcn_sched_barrier(0);
                            ldA(a1, k4 + 3);
                        });
                    } else {
                        // Both operands R - 1 MFMA blocks ahead.  A wave alone on its SIMD (g = 7 for 112 of its 128 steps) has
                        // 1024 cycles per block, and the B image of a small model's block comes from beyond L2 (32 workgroups x
                        // 0.5 MB per XCD at N = 1024): one block ahead such a wave ran at 1490 cycles per step (r04_small_n.txt).
                        AF a[R];
                        v4 b[R];
#pragma unroll
                        for (int i = 0; i < R; ++i) { ldA(a[i], d_lo + i); ldB(b[i], d_lo + i); }
                        tri_loop(std::integral_constant<int, R>{}, d_lo, limit > d_lo ? limit : d_lo, tri, [&](auto jtag, const int k4) {     // d_lo and limit are multiples of 16, R divides 16
#pragma unroll
                            for (int i = 0; i < R; ++i) {
                                __builtin_amdgcn_sched_barrier(0);
                                El<T>::template mfma_from<decltype(jtag)::value>(acc, a[i], b[i]);
                                __builtin_amdgcn_sched_barrier(0);
                                ldA(a[i], k4 + i + R); ldB(b[i], k4 + i + R);
                            }
                        });
                    }
                }
            }
            GPT_VT(5);
            if (vslot >= 0) {
                // cut sweep: this part's 512 x 64 partial product goes to vslab, [vslot][wave][r*4+t][lane] (k_var_combine)
                v4* dst = reinterpret_cast<v4*>(vslab) + ((size_t)vslot * 8 + w) * (16 * 64) + lane;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t) dst[(r * 4 + t) * 64] = acc[r][t];
            } else if (GPT_ABL != 9) {
                // whole sweep: fold this wave's 64 rows of V into the per-column sums
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const T v = acc[r][t][e];
                            ssq[t] += v * v;
                            if (CROSS) crs[t] += v * __shfl(v, lane & ~(CPQ - 1));
                        }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t) El<T>::keep1(acc[r][t]);
            }
            GPT_VT(6);
        };

        if (flags & VI_GEN) {
            sweep(std::true_type{});
            // every wave's part of the scratch image must have reached L2 before another wave reloads it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            sweep(std::false_type{});
        }
        GPT_VT(7);

        if (slot >= 0) {
            // rows of a column are spread over the 4 lane groups lk = 0..3 and over the 8 waves
            T s2[4], cr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s2[t] = ssq[t]; cr[t] = crs[t];
                s2[t] += __shfl_xor(s2[t], 16); s2[t] += __shfl_xor(s2[t], 32);
                if (CROSS) { cr[t] += __shfl_xor(cr[t], 16); cr[t] += __shfl_xor(cr[t], 32); }
            }
            if (lk == 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    red[0][w][16 * t + lc] = s2[t];
                    red[1][w][16 * t + lc] = CROSS ? cr[t] : (T)0;
                }
            }
            __syncthreads();
            if (threadIdx.x < VAR_SLOT) {
                const int which = threadIdx.x >> 6, cl = threadIdx.x & 63;
                T v = (T)0;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) v += red[which][ww][cl];
                slab[(size_t)slot * VAR_SLOT + threadIdx.x] = v;
            }
        }
        GPT_VT(8);
    }
#ifdef GPT_VAR_TRACE
    if (vt_wg) vt_wg[1] = vt_realtime();
#endif
}

// A sweep the work split cut along k: add its parts' partial products in part order, then square / reduce as the
// kernel's own epilogue does.  VAR_SPLIT_SLOTS (8) workgroups per cut sweep, one per row group (= wave of k_var), each with a slab
// slot of its own; wave r of the workgroup takes the group's row tile r, same lane -> element map as k_var.  (One workgroup
// per cut sweep until round 4: 14 workgroups x 1.3 MB at configs[1], 18 us at the rate of 14 CUs.)
template <typename T, bool CROSS, int CPQ = 4>
__global__ __launch_bounds__(256) void k_var_combine(VarPlanDev pl, const T* __restrict__ vslab, T* __restrict__ slab) {
    typedef typename El<T>::v4 v4;
    __shared__ T red[2][4][VAR_COLS];
    const VarSplit sp = pl.splits[blockIdx.x / VAR_SPLIT_SLOTS];
    const int grp = blockIdx.x % VAR_SPLIT_SLOTS;
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    v4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = v4{0, 0, 0, 0};
    for (int v = sp.v_begin; v < sp.v_end; ++v) {
        const v4* src = reinterpret_cast<const v4*>(vslab) + ((size_t)v * 8 + grp) * (16 * 64) + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += src[(r * 4 + t) * 64];
    }
    T ssq[4] = {0, 0, 0, 0}, crs[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const T v = acc[t][e];
            ssq[t] += v * v;
            if (CROSS) crs[t] += v * __shfl(v, lane & ~(CPQ - 1));
        }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ssq[t] += __shfl_xor(ssq[t], 16); ssq[t] += __shfl_xor(ssq[t], 32);
        if (CROSS) { crs[t] += __shfl_xor(crs[t], 16); crs[t] += __shfl_xor(crs[t], 32); }
    }
    if (lk == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            red[0][r][16 * t + lc] = ssq[t];
            red[1][r][16 * t + lc] = CROSS ? crs[t] : (T)0;
        }
    }
    __syncthreads();
    if (threadIdx.x < VAR_SLOT) {
        const int which = threadIdx.x >> 6, cl = threadIdx.x & 63;
        T v = (T)0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v += red[which][rr][cl];
        slab[((size_t)sp.slot + grp) * VAR_SLOT + threadIdx.x] = v;
    }
}

// Adds the partial sums of a (column block, task) in slot order and turns them into outputs:
// var (M, ntask), Jvar (M, ntask, D), dvar (D, M) (ntask = 1 only).
template <typename T, int NCOMP, bool KSTAR = true>
__global__ __launch_bounds__(64) void k_var_finalize(KernelParams p, VarPlanDev pl, const T* __restrict__ slab, int64_t M,
                                                     const double* __restrict__ hdr, T* __restrict__ var,
                                                     T* __restrict__ Jvar, T* __restrict__ dvar) {
    const int64_t cb = blockIdx.x;
    const int cl = threadIdx.x;
    const int D = p.D, NT = pl.ntask;
    const int64_t col = cb * VAR_COLS + cl;
    const int64_t m = (NCOMP == 1) ? col : ((NCOMP >= 4) ? (col / NCOMP) : (col / D));
    const int cmp = (NCOMP == 1) ? 0 : ((NCOMP >= 4) ? (int)(col & (NCOMP - 1)) + (KSTAR ? 0 : 1) : 1 + (int)(col % D));
    for (int task = 0; task < NT; ++task) {
        T s2 = (T)0, cr = (T)0;
        int s_begin, s_end;
        if (cb < pl.nfull) { s_begin = (int)(cb * NT + task); s_end = s_begin + 1; }      // handled whole by one workgroup
        else {
            const int64_t e = (cb - pl.nfull) * NT + task;
            s_begin = pl.fin[2 * e]; s_end = pl.fin[2 * e + 1];
        }
        for (int s = s_begin; s < s_end; ++s) {
            const T* sl = slab + (size_t)s * VAR_SLOT;
            s2 += sl[cl];
            cr += sl[VAR_COLS + cl];
        }
        if (m >= M) continue;
        const T c = (T)hdr[16 + task];
        if (cmp == 0) {
            if (var) { const T v = c + (T)p.noise - s2; var[m * NT + task] = v < (T)0 ? (T)0 : v; }
        } else {
            const int d = cmp - 1;
            if (d < D) {
                if (Jvar) Jvar[(m * NT + task) * D + d] = c * (T)(p.inv_ls[d] * p.inv_ls[d]) - s2;
                if (dvar) dvar[(int64_t)d * M + m] = (T)-2 * cr;
            }
        }
    }
}

// Dynamic LDS beyond the default limit is an opt-in per kernel and device.
template <typename T>
static void var_kernel_setup() {
    static PerDeviceOnce once;
    once.run([] {
    const void* fns[] = {reinterpret_cast<const void*>(k_var<T, 1, false, KT_RBF>), reinterpret_cast<const void*>(k_var<T, 4, true, KT_RBF>),
                         reinterpret_cast<const void*>(k_var<T, 4, false, KT_RBF>), reinterpret_cast<const void*>(k_var<T, 3, false, KT_RBF>),
                         reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN12>),
                         reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN32>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN52>),
                         // D = 4 .. 8
                         reinterpret_cast<const void*>(k_var<T, 1, false, KT_RBF, WIDE_D>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN12, WIDE_D>),
                         reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN32, WIDE_D>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN52, WIDE_D>),
                         reinterpret_cast<const void*>(k_var<T, 8, true, KT_RBF, WIDE_D>), reinterpret_cast<const void*>(k_var<T, 8, false, KT_RBF, WIDE_D>),
                         reinterpret_cast<const void*>(k_var<T, 16, true, KT_RBF, WIDE_D>), reinterpret_cast<const void*>(k_var<T, 16, false, KT_RBF, WIDE_D>),
                         reinterpret_cast<const void*>(k_var<T, 4, false, KT_RBF, WIDE_D, false>), reinterpret_cast<const void*>(k_var<T, 8, false, KT_RBF, WIDE_D, false>),
                         reinterpret_cast<const void*>(k_var<T, 1, false, KT_RBF, MAX_D>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN12, MAX_D>),
                         reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN32, MAX_D>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN52, MAX_D>),
                         reinterpret_cast<const void*>(k_var<T, 16, true, KT_RBF, MAX_D>), reinterpret_cast<const void*>(k_var<T, 16, false, KT_RBF, MAX_D>)};
    for (const void* f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)var_lds_bytes<T>());
    if constexpr (std::is_same<T, double>::value) {       // the small-model instantiations (HALF): launch_var_t
        const void* hf[] = {reinterpret_cast<const void*>(k_var<T, 1, false, KT_RBF, 3, true, true>), reinterpret_cast<const void*>(k_var<T, 4, true, KT_RBF, 3, true, true>),
                            reinterpret_cast<const void*>(k_var<T, 4, false, KT_RBF, 3, true, true>),
                            reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN12, 3, true, true>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN32, 3, true, true>),
                            reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN52, 3, true, true>),
                            reinterpret_cast<const void*>(k_var<T, 1, false, KT_RBF, WIDE_D, true, true>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN12, WIDE_D, true, true>),
                            reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN32, WIDE_D, true, true>), reinterpret_cast<const void*>(k_var<T, 1, false, KT_MATERN52, WIDE_D, true, true>),
                            reinterpret_cast<const void*>(k_var<T, 8, true, KT_RBF, WIDE_D, true, true>), reinterpret_cast<const void*>(k_var<T, 8, false, KT_RBF, WIDE_D, true, true>),
                            reinterpret_cast<const void*>(k_var<T, 16, true, KT_RBF, WIDE_D, true, true>), reinterpret_cast<const void*>(k_var<T, 16, false, KT_RBF, WIDE_D, true, true>)};
        for (const void* f : hf) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)var_lds_bytes<T>());
    }
    });
}

// rounds of whole column blocks per launch of k_var (0: all in one launch); see launch_var_t
static int var_rounds_per_launch() {
    const char* e = getenv("GPT_VAR_ROUNDS_PER_LAUNCH");          // read per call: tests compare settings in one process
    const int v = e ? atoi(e) : 16;
    return v < 0 ? 0 : v;
}

template <typename T>
static void launch_var_t(hipStream_t s, const KernelParams& p, const VarWorkspace& ws, const T* Xs, const T* Wf,
                         const T* Xq, int64_t M, int ncomp, T* var, T* Jvar, T* dvar, const double* hdr) {
    const VarPlanDev& pl_all = ws.plan->d;
    constexpr size_t lds = var_lds_bytes<T>();
    var_kernel_setup<T>();
    T* slab = static_cast<T*>(ws.slab);
    T* vslab = static_cast<T*>(ws.vslab);
    T* bscr = static_cast<T*>(ws.bscratch);
    const dim3 grid((unsigned)pl_all.P), fgrid((unsigned)pl_all.ncb), cgrid((unsigned)pl_all.n_splits * VAR_SPLIT_SLOTS);
    const bool wide = p.D > 3, wide16 = p.D > WIDE_D;
    const bool cross = ncomp >= 4 && dvar != nullptr;
    // The persistent workgroups are not synchronised between rounds and drift apart; once they are further apart than a W
    // tile stays in L2 each fetches its own copy of the W stream (mode J+Jvar, 122 rounds in one launch: 155 MB fetched per
    // column block against 68 MB for the same kernel over 31 rounds, L2 hit rate 54 % against 78 %).  A kernel boundary is the
    // one barrier that needs no co-residency: the rounds go out `var_rounds_per_launch()` at a time.  16 per launch: the same
    // run time (269.9-270.5 k/s in one launch, 270.9-271.3 k/s at 8-16), HBM-side bytes 4.99 -> 1.40 TB per 500k queries,
    // L2 hit rate 54 -> 86 % (profiles/r03_kvar_round_drift.txt).
    const int64_t rounds = pl_all.nfull / pl_all.P;
    // (the setting is in rounds of the shape it was measured on — 136 tiles per block, N = 8192; a launch of a smaller model
    // gets as many rounds as make the same number of tiles, so that the boundaries stay ~0.1 % of the run time)
    const int rpl_set = var_rounds_per_launch();
    const int64_t tiles_per_round = (int64_t)pl_all.ntask * pl_all.tiles_per_task;
    const int64_t rpl_scaled = tiles_per_round > 0 ? ((int64_t)rpl_set * 136 + tiles_per_round - 1) / tiles_per_round : rpl_set;
    int64_t rpl = rpl_set > 0 ? (rpl_scaled > rpl_set ? rpl_scaled : rpl_set) : (rounds > 0 ? rounds : 1);
    if (const char* e = getenv("GPT_VAR_ROUNDS_EXACT")) { if (atoi(e) > 0) rpl = atoi(e); }      // tests: this many, whatever the shape
    // fp64 diagonal tiles with the first half of their B image in LDS: for models whose scratch images do not stay in L2 (k_var);
    // GPT_VAR_DIAG_HALF = 0 / 1 forces it off / on (read per call: tests compare both in one process)
    constexpr bool HALF_OK = std::is_same<T, double>::value;          // (fp32 stages the whole image of a diagonal tile in LDS: DIAG_LDS)
    bool diag_half = HALF_OK && pl_all.nbi <= 5;
    if (const char* e = getenv("GPT_VAR_DIAG_HALF")) { if (atoi(e) >= 0) diag_half = HALF_OK && atoi(e) != 0; }
    for (int64_t r0 = 0; r0 == 0 || r0 < rounds; r0 += rpl) {
    VarPlanDev pl = pl_all;
    pl.rnd_begin = r0;
    pl.rnd_end = r0 + rpl < rounds ? r0 + rpl : rounds;
    pl.with_tail = pl.rnd_end >= rounds ? 1 : 0;
#define GPT_KVAR(NC_, CR_, KT_, DW_)                                                                                                      \
    do {                                                                                                                                 \
        if (diag_half) hipLaunchKernelGGL((k_var<T, NC_, CR_, KT_, DW_, true, HALF_OK>), grid, dim3(512), lds, s, p, pl, Xs, Wf, Xq, M, slab, vslab, bscr); \
        else hipLaunchKernelGGL((k_var<T, NC_, CR_, KT_, DW_>), grid, dim3(512), lds, s, p, pl, Xs, Wf, Xq, M, slab, vslab, bscr);       \
    } while (0)
#define GPT_KVAR1(DW_)                                                  \
        switch (p.ktype) {                                               \
            case KT_MATERN12: GPT_KVAR(1, false, KT_MATERN12, DW_); break; \
            case KT_MATERN32: GPT_KVAR(1, false, KT_MATERN32, DW_); break; \
            case KT_MATERN52: GPT_KVAR(1, false, KT_MATERN52, DW_); break; \
            default: GPT_KVAR(1, false, KT_RBF, DW_); break;             \
        }
    if (ncomp == VAR_NCOMP_DERIV4) {          // D = 4, Jacobian variance alone: dk_0 .. dk_3
        hipLaunchKernelGGL((k_var<T, 4, false, KT_RBF, WIDE_D, false>), grid, dim3(512), lds, s, p, pl, Xs, Wf, Xq, M, slab, vslab, bscr);
    } else if (ncomp == VAR_NCOMP_DERIV8) {   // D = 8
        hipLaunchKernelGGL((k_var<T, 8, false, KT_RBF, WIDE_D, false>), grid, dim3(512), lds, s, p, pl, Xs, Wf, Xq, M, slab, vslab, bscr);
    } else if (ncomp == 1) {
        if (wide16) {             // D = 9 .. 15: rows of 16 (no HALF instantiations)
#define GPT_KVAR16(NC_, CR_, KT_) hipLaunchKernelGGL((k_var<T, NC_, CR_, KT_, MAX_D>), grid, dim3(512), lds, s, p, pl, Xs, Wf, Xq, M, slab, vslab, bscr)
            switch (p.ktype) {
                case KT_MATERN12: GPT_KVAR16(1, false, KT_MATERN12); break;
                case KT_MATERN32: GPT_KVAR16(1, false, KT_MATERN32); break;
                case KT_MATERN52: GPT_KVAR16(1, false, KT_MATERN52); break;
                default: GPT_KVAR16(1, false, KT_RBF); break;
            }
        } else if (wide) { GPT_KVAR1(WIDE_D) } else { GPT_KVAR1(3) }
    } else if (ncomp == 3) {      // Jacobian variance alone: D columns per query (D <= 3)
        // (no HALF instantiation: at the register limit it keeps a spilled pointer inside the lock-step loop)
        hipLaunchKernelGGL((k_var<T, 3, false, KT_RBF, 3>), grid, dim3(512), lds, s, p, pl, Xs, Wf, Xq, M, slab, vslab, bscr);
    } else if (ncomp == 4) {      // Jacobian variance / d var: RBF only (the API refuses other kernels)
        if (cross) GPT_KVAR(4, true, KT_RBF, 3);
        else GPT_KVAR(4, false, KT_RBF, 3);
    } else if (ncomp == 8) {      // D = 4 .. 7
        if (cross) GPT_KVAR(8, true, KT_RBF, WIDE_D);
        else GPT_KVAR(8, false, KT_RBF, WIDE_D);
    } else if (wide16) {          // D = 9 .. 15
        if (cross) GPT_KVAR16(16, true, KT_RBF);
        else GPT_KVAR16(16, false, KT_RBF);
    } else {                      // D = 8
        if (cross) GPT_KVAR(16, true, KT_RBF, WIDE_D);
        else GPT_KVAR(16, false, KT_RBF, WIDE_D);
    }
    }
#undef GPT_KVAR16
#undef GPT_KVAR1
#undef GPT_KVAR
    const VarPlanDev& pl = pl_all;
    if (pl.n_splits > 0) {
        if (!cross) hipLaunchKernelGGL((k_var_combine<T, false>), cgrid, dim3(256), 0, s, pl, vslab, slab);
        else if (ncomp == 4) hipLaunchKernelGGL((k_var_combine<T, true, 4>), cgrid, dim3(256), 0, s, pl, vslab, slab);
        else if (ncomp == 8) hipLaunchKernelGGL((k_var_combine<T, true, 8>), cgrid, dim3(256), 0, s, pl, vslab, slab);
        else hipLaunchKernelGGL((k_var_combine<T, true, 16>), cgrid, dim3(256), 0, s, pl, vslab, slab);
    }
    switch (ncomp) {
        case VAR_NCOMP_DERIV4: hipLaunchKernelGGL((k_var_finalize<T, 4, false>), fgrid, dim3(64), 0, s, p, pl, slab, M, hdr, var, Jvar, dvar); break;
        case VAR_NCOMP_DERIV8: hipLaunchKernelGGL((k_var_finalize<T, 8, false>), fgrid, dim3(64), 0, s, p, pl, slab, M, hdr, var, Jvar, dvar); break;
        case 1: hipLaunchKernelGGL((k_var_finalize<T, 1>), fgrid, dim3(64), 0, s, p, pl, slab, M, hdr, var, Jvar, dvar); break;
        case 3: hipLaunchKernelGGL((k_var_finalize<T, 3>), fgrid, dim3(64), 0, s, p, pl, slab, M, hdr, var, Jvar, dvar); break;
        case 4: hipLaunchKernelGGL((k_var_finalize<T, 4>), fgrid, dim3(64), 0, s, p, pl, slab, M, hdr, var, Jvar, dvar); break;
        case 8: hipLaunchKernelGGL((k_var_finalize<T, 8>), fgrid, dim3(64), 0, s, p, pl, slab, M, hdr, var, Jvar, dvar); break;
        default: hipLaunchKernelGGL((k_var_finalize<T, 16>), fgrid, dim3(64), 0, s, p, pl, slab, M, hdr, var, Jvar, dvar);
    }
}

void launch_var(hipStream_t s, const KernelParams& p, const VarWorkspace& ws, const void* Xs, const void* Wf,
                const void* Xq, int64_t M, int ncomp, void* var, void* Jvar, void* dvar, const double* hdr) {
    if (M <= 0 || !ws.plan) return;
    if (p.dtype == DT_F32)
        launch_var_t<float>(s, p, ws, (const float*)Xs, (const float*)Wf, (const float*)Xq, M, ncomp, (float*)var, (float*)Jvar, (float*)dvar, hdr);
    else
        launch_var_t<double>(s, p, ws, (const double*)Xs, (const double*)Wf, (const double*)Xq, M, ncomp, (double*)var, (double*)Jvar, (double*)dvar, hdr);
}

}  // namespace gpt

#ifdef GPT_VAR_TRACE
// trace builds only (make trace): dev_buf holds VT_WGS * 8 * VT_ITEMS * VT_STAMPS int64, or NULL to switch the trace off
extern "C" int gpt_debug_set_var_trace(void* dev_buf) {
    long long* p = static_cast<long long*>(dev_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(gpt::g_var_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
