// Prediction kernels for gfx950 (MI355X).
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
//               (gaussian_process.py:122-125) on the fp64 matrix cores; design notes at the kernel.
#include "gpt_common.h"
#include "gpt_exp.h"
#include <cstdlib>
#include <type_traits>

namespace gpt {

// ------------------------------------------------------------------------------------------
// OC = outputs handled by this pass (1..4): only their partial sums are accumulated.
template <int QPW, int OC, int KT>
__global__ __launch_bounds__(256) void k_mean_jac(KernelParams p, const double* __restrict__ Xs,
                                                  const double* __restrict__ A4, const double* __restrict__ Xq,
                                                  int64_t M, int o_base, double* __restrict__ mean,
                                                  double* __restrict__ J) {
    __shared__ double Tt[256];
    Tt[threadIdx.x] = g_exp2_table[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * QPW;
    if (m0 >= M) return;
    const int D = p.D;
    constexpr double RS2 = 0.70710678118654752440;    // coordinates scaled by 1/sqrt(2): k = exp(ln c - |d'|^2)
    double q[QPW][3];
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int64_t m = (m0 + i < M) ? (m0 + i) : (M - 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) q[i][d] = (d < D) ? Xq[m * D + d] * (p.inv_ls[d] * RS2) : 0.0;
    }
    double acc[QPW][OC][4];
#pragma unroll
    for (int i = 0; i < QPW; ++i)
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][o][e] = 0.0;

    const double lnc = p.lnc;
    for (int n = lane; n < p.N; n += 64) {
        const d4 xs = *reinterpret_cast<const d4*>(Xs + (size_t)n * 4);
        const d4 al = *reinterpret_cast<const d4*>(A4 + (size_t)n * 4);
        const double x0 = xs[0] * RS2, x1 = xs[1] * RS2, x2 = xs[2] * RS2;
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const double d0 = x0 - q[i][0], d1 = x1 - q[i][1], d2 = x2 - q[i][2];
            double hh = d0 * d0;
            hh = fma(d1, d1, hh);
            hh = fma(d2, d2, hh);
            const double kv = kernel_tab<KT>(hh, lnc, Tt);
#pragma unroll
            for (int o = 0; o < OC; ++o) {
                const double t = kv * al[o];
                acc[i][o][0] += t;
                acc[i][o][1] += t * d0;
                acc[i][o][2] += t * d1;
                acc[i][o][3] += t * d2;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QPW; ++i)
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double v = acc[i][o][e];
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
                if (J) {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        if (d < D) J[(m * O + oo) * D + d] = acc[i][o][1 + d] * (p.inv_ls[d] * S2);
                }
            }
        }
    }
}

void launch_mean_jac(hipStream_t s, const KernelParams& p, const double* Xs, const double* A4,
                     const double* Xq, int64_t M, double* mean, double* J) {
    if (M <= 0 || (!mean && !J)) return;
    constexpr int QPW = 2;
    const int64_t waves = (M + QPW - 1) / QPW;
    const int64_t blocks = (waves + 3) / 4;
    for (int ob = 0; ob < p.O; ob += 4) {
        const int cnt = (p.O - ob) < 4 ? (p.O - ob) : 4;
        const double* a4 = A4 + (size_t)(ob / 4) * p.NP * 4;
        const dim3 grid((unsigned)blocks);
#define GPT_MJ(OC_, KT_) hipLaunchKernelGGL((k_mean_jac<QPW, OC_, KT_>), grid, dim3(256), 0, s, p, Xs, a4, Xq, M, ob, mean, J)
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

// ------------------------------------------------------------------------------------------
// Variance kernel.
//
// What the hardware does (measured on MI355X, profiles/r01_*):
//   * v_mfma_f64_16x16x4_f64 issues every 64 cycles per SIMD: 77.7 TFLOP/s with 2 waves/SIMD at 2.4 GHz;
//   * every fp64 VALU instruction of a SIMD takes ~4.5 cycles away from its fp64 MFMA stream (the fp64
//     matrix and vector paths share the DP units), so B values must be generated once, not per wave;
//   * a workgroup barrier every 8 k-steps costs ~4 %; once the matrix pipe is >90 % busy the chip lowers
//     its clock (2.38 -> 2.18 GHz), so what is left is energy per MFMA: operands must come from close by
//     (A from L2 with all workgroups walking W in step, B from LDS), not from HBM.
// Design:
//   * a workgroup (512 threads = 8 waves, 2 per SIMD) owns a 64-column block and sweeps the 512-row
//     i-blocks, longest sweep first; wave w accumulates the 64x64 product of its 64-row group (16 MFMA
//     tiles, 128 VGPRs) and folds it into per-column sums when the sweep ends, so V = W K*^T never
//     touches memory;
//   * the B operand of a sweep reaches the waves through a double-buffered LDS image in MFMA lane order,
//     2 x 32 k-steps x 2 KiB = 128 KiB, one barrier per 32 k-steps.  Wave w fills k-steps w, w+8, w+16,
//     w+24 of the next chunk from the middle of its own MFMA run (the two waves of a SIMD staggered).
//     In the FIRST sweep of a block the fragments are generated (k* / dk_d columns, one table-driven fp64
//     exp each, gpt_exp.h) and a copy goes to this workgroup's scratch image in HBM/L2 ([k-step][lane][4],
//     4 MB at N=8192); the other sweeps reload them from there (2 KiB per k-step per workgroup) — so a
//     block pays N exps per column, not N*(N/512+1)/2;
//   * the A operand streams from the fragment-ordered image Wf with two 16-byte loads per lane and
//     k-step, two steps ahead; in the diagonal tile a wave skips the k-steps where its row group is
//     entirely above the diagonal (wave g has 16 (g + 1) of 128), row groups paired (0,7)(1,6)(2,5)(3,4) on
//     the SIMDs.  In the reload sweeps the diagonal tile runs OUTSIDE the lock-step LDS pipeline (every wave
//     on its own, B straight from the scratch image), so the pairing balances it: 0.56 of a full tile
//     instead of 0.75 (+2.7 %);
//   * work split: rounds of whole blocks (all workgroups in step => W tiles are shared through L2), then
//     a stream-K split of the leftover blocks; partial column sums go to unique slab slots and
//     k_var_finalize adds them in fixed order (deterministic, no atomics).
// Alternatives measured and dropped (profiles/r01_kvar_variant_ab.txt): per-wave B generation (v1, 53 TF),
// 8-step chunks (-3.7 %), barrier-free sweeps with every wave reading B from the scratch image (-3 %: L2
// hit rate 97 % -> 56 %, 3.4 TB/s from beyond L2, clock 2.18 GHz), deeper A prefetch (0 %).
// NCOMP = 1: one column per query (k*).  NCOMP = 4: four columns per query (k*, dk_0, dk_1, dk_2).
// NCOMP = 3: D columns per query (dk_0 .. dk_{D-1}) — the Jacobian variance without the variance.
// ------------------------------------------------------------------------------------------
// Timing-only ablation builds (results are wrong unless 0): -DGPT_ABL=1 no per-chunk barrier, 2 no A-operand
// loads, 3 diagonal tile skipped, 4 no B fill (LDS image left as is), 5 no MFMAs.  tools/gpu_ablate.sh,
// profiles/r01_final_ablation.txt.
#ifndef GPT_ABL
#define GPT_ABL 0
#endif
constexpr int VAR_COLS = 64;        // columns per column block
constexpr int VAR_SUB = 8;          // k4-steps per sub-chunk (= waves per workgroup: wave w fills step w of each)
constexpr int VAR_SUBS = 4;         // sub-chunks per LDS chunk
constexpr int VAR_CH = VAR_SUB * VAR_SUBS;   // k4-steps per LDS chunk, one barrier each (32)
constexpr size_t VAR_LDS_BYTES = (size_t)2 * VAR_CH * 64 * 4 * sizeof(double);   // 128 KiB
constexpr int VAR_DIAG_COST = 72;   // k4-steps a diagonal tile costs a SIMD: waves g and 7-g, 16 (g+1) + 16 (8-g) of 2 x 128, halved
constexpr int VAR_SLOT = 2 * VAR_COLS;   // doubles per slab slot: ssq[64], crs[64]

// Work split.  Rounds 0..R-1: workgroup p takes the whole column block r*P + p — all workgroups then walk
// the same W tiles at the same time, which is what keeps the W stream in L2 (each XCD's 32 workgroups
// share one copy).  The ncb - R*P blocks left over ("tail") are laid end to end, costed per i-block, and
// cut into P equal ranges (stream-K) so that every CU finishes together.
struct VarPlan {
    int64_t ncb;     // column blocks
    int64_t nfull;   // R * P blocks handled whole, round-robin
    int64_t ncb_t;   // tail blocks = ncb - nfull
    int64_t T;       // cost of one column block (all i-blocks)
    int64_t U;       // tail cost = ncb_t * T
    int P;           // workgroups
    int nbi;         // i-blocks
};

__host__ __device__ inline int64_t var_cost_prefix(int ib) {          // sum_{i<ib} (128 i + VAR_DIAG_COST)
    return (int64_t)64 * ib * (ib - 1) + (int64_t)VAR_DIAG_COST * ib;
}

// first work unit (tail column block, i-block) of tail range p; p = P gives the end of the tail
__host__ __device__ inline void var_boundary(const VarPlan& pl, int p, int64_t& cb, int& ib) {
    if (p >= pl.P) { cb = pl.ncb_t; ib = 0; return; }
    const int64_t B = pl.U / pl.P * p + (pl.U % pl.P) * p / pl.P;
    cb = B / pl.T;
    const int64_t off = B % pl.T;
    int lo = 0, hi = pl.nbi;                                            // smallest ib with prefix(ib) >= off
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (var_cost_prefix(mid) >= off) hi = mid; else lo = mid + 1;
    }
    ib = lo;
    if (ib >= pl.nbi) { cb += 1; ib = 0; }
}

#define GPT_MFMA16(acc, a01, a23, b)                                                              \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                            \
        acc[0][t_] = __builtin_amdgcn_mfma_f64_16x16x4f64((a01)[0], (b)[t_], acc[0][t_], 0, 0, 0); \
        acc[1][t_] = __builtin_amdgcn_mfma_f64_16x16x4f64((a01)[1], (b)[t_], acc[1][t_], 0, 0, 0); \
        acc[2][t_] = __builtin_amdgcn_mfma_f64_16x16x4f64((a23)[0], (b)[t_], acc[2][t_], 0, 0, 0); \
        acc[3][t_] = __builtin_amdgcn_mfma_f64_16x16x4f64((a23)[1], (b)[t_], acc[3][t_], 0, 0, 0); \
    }

template <int NCOMP, bool CROSS, int KT>
__global__ __launch_bounds__(512, 2) void k_var(KernelParams p, VarPlan pl, const double* __restrict__ Xs,
                                                const double* __restrict__ Wf, const double* __restrict__ Xq,
                                                int64_t M, double* __restrict__ slab, double* __restrict__ bscratch) {
    extern __shared__ __attribute__((aligned(16))) double Bs_dyn[];       // [buffer][k4-step][lane][column tile]
    __shared__ double red[2][8][VAR_COLS];
    __shared__ double Tt[256];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lc = lane & 15, lk = lane >> 4;
    const int g = (w < 4) ? w : (11 - w);     // row group of this wave: 0,1,2,3,7,6,5,4
    const int D = p.D;
    auto Bs = [&](const int buf, const int step) -> double* { return Bs_dyn + ((size_t)(buf * VAR_CH + step) * 64 + lane) * 4; };
    if (threadIdx.x < 256) Tt[threadIdx.x] = g_exp2_table[threadIdx.x];

    int64_t cb0, cb1; int ib0, ib1;            // this workgroup's range of the tail
    var_boundary(pl, blockIdx.x, cb0, ib0);
    var_boundary(pl, blockIdx.x + 1, cb1, ib1);
    const int64_t rounds = pl.nfull / pl.P;

    constexpr double RS2 = 0.70710678118654752440;    // coordinates are pre-scaled by 1/sqrt(2): t = ln c - |d'|^2
    // NCOMP=4: column = 4 query + comp, comp = lc & 3 in every tile: b = kv * (cbv + sum_d cd[d] * d'_d).
    // NCOMP=3 (Jacobian variance alone, no k* column): column = D query + d, D columns per query; for D = 3 the d of a
    // lane's column changes from tile to tile and from block to block (16 = 64 = 1 mod 3), selected in `produce`.
    const int comp = (NCOMP == 4) ? (lc & 3) : 0;
    const double cbv = (comp == 0) ? 1.0 : 0.0;
    double cd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) cd[d] = (comp == d + 1 && d < D) ? p.inv_ls[d] * 1.41421356237309504880 : 0.0;
    const double lnc = p.lnc;
    const int nbi = pl.nbi;
    // A stream in d2 units: element (step S, group g, q, lane) at ((S*8 + g)*2 + q)*64 + lane
    // uniform base + per-lane 32-bit offset (scalar-base addressing: no 64-bit VALU address arithmetic, and no VALU
    // writes into registers that loads are still in flight to)
    const d2* const wuni = reinterpret_cast<const d2*>(Wf) + (size_t)g * 128;
    const int wlane = lane;
    constexpr size_t STEP_D2 = WT_STEP_DOUBLES / 2;   // 1024
    // this workgroup's B image in d2 units: k-step s, lane l at (s*64 + l)*2 (+1)
    d2* const buni = reinterpret_cast<d2*>(bscratch) + (size_t)blockIdx.x * ((size_t)p.NP * 32);
    const int blane = lane * 2;

    for (int64_t piece = 0;; ++piece) {
        int64_t cb, slot; int lo, hi;
        if (piece < rounds) {                          // whole block, in step with every other workgroup
            cb = piece * pl.P + blockIdx.x; slot = cb; lo = 0; hi = nbi;
        } else {                                       // this workgroup's share of the tail
            const int64_t cbt = cb0 + (piece - rounds);
            if (cbt > cb1 || cbt >= pl.ncb_t) break;
            lo = (cbt == cb0) ? ib0 : 0;
            hi = (cbt == cb1) ? ib1 : nbi;
            if (lo >= hi) continue;
            cb = pl.nfull + cbt; slot = pl.nfull + blockIdx.x + cbt;
        }
        __syncthreads();                               // LDS (Bs, red, Tt) free / ready
        // the scratch image is about to be rewritten: drop the L1 lines of it this CU may still hold
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

        double ssq[4] = {0.0, 0.0, 0.0, 0.0}, crs[4] = {0.0, 0.0, 0.0, 0.0};
        const int base3 = (NCOMP == 3) ? (int)((cb * VAR_COLS + lc) % D) : 0;
        const double sc3[3] = {p.inv_ls[0] * 1.41421356237309504880, p.inv_ls[1] * 1.41421356237309504880,
                               p.inv_ls[2] * 1.41421356237309504880};

        // One sweep of i-block ib.  GEN = true (first sweep of the block): B fragments are generated and a copy
        // is kept in the scratch image; GEN = false: they are reloaded from it.  Two instantiations, so that the
        // query coordinates and exp temporaries of the generating sweep do not occupy registers in the others.
        auto sweep = [&](auto gen_tag, const int ib) {
            constexpr bool GEN = decltype(gen_tag)::value;
            // this lane's four columns (one per MFMA column tile): scaled query coordinates
            double q[4][3];
            if (GEN) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int64_t col = cb * VAR_COLS + 16 * t + lc;
                    const int64_t m = (NCOMP == 1) ? col : ((NCOMP == 4) ? (col >> 2) : (col / D));
                    const int64_t mm = (m < M) ? m : (M - 1);
#pragma unroll
                    for (int d = 0; d < 3; ++d) q[t][d] = (d < D) ? Xq[mm * D + d] * (p.inv_ls[d] * RS2) : 0.0;
                }
            }
            double gx[3];                                  // coordinates of the source this wave generates next
            d2 bl[2];                                      // or the fragments it reloads next
            auto fetch = [&](const int k4) {
                if (GEN) {
                    const double* xp = Xs + (size_t)(k4 * 4 + lk) * 4;
                    gx[0] = xp[0]; gx[1] = xp[1]; gx[2] = xp[2];
                } else {
                    const d2* src = buni + (size_t)k4 * 128;
                    bl[0] = src[blane]; bl[1] = src[blane + 1];
                }
            };
            auto produce = [&](const int buf, const int k4) {   // B fragments of k-step k4 -> LDS (+ scratch)
                double* dstl = Bs(buf, k4 % VAR_CH);
                if (GEN) {
                    const double x0 = gx[0] * RS2, x1 = gx[1] * RS2, x2 = gx[2] * RS2;
                    d4 b;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double d0 = x0 - q[t][0], d1 = x1 - q[t][1], d2_ = x2 - q[t][2];
                        double hh = d0 * d0;
                        hh = fma(d1, d1, hh);
                        hh = fma(d2_, d2_, hh);
                        const double kv = kernel_tab<KT>(hh, lnc, Tt);
                        if (NCOMP == 3) {
                            int dsel = base3 + ((D == 3) ? t : 0);            // (64 cb + 16 t + lc) mod D, base3 = (64 cb + lc) mod D
                            dsel = (dsel >= D) ? dsel - D : dsel;
                            const double e = (dsel == 0) ? d0 * sc3[0] : ((dsel == 1) ? d1 * sc3[1] : d2_ * sc3[2]);
                            b[t] = kv * e;
                        } else {
                            b[t] = (NCOMP == 1) ? kv : kv * (cbv + cd[0] * d0 + cd[1] * d1 + cd[2] * d2_);
                        }
                    }
                    *reinterpret_cast<d4*>(dstl) = b;
                    d2* dst = buni + (size_t)k4 * 128 + blane;
                    dst[0] = d2{b[0], b[1]};
                    dst[1] = d2{b[2], b[3]};
                } else {
                    *reinterpret_cast<d2*>(dstl) = bl[0];
                    *reinterpret_cast<d2*>(dstl + 2) = bl[1];
                }
            };

            const size_t S_ib = (size_t)64 * ib * (ib + 1);   // stream index of the first k4-step of this sweep
            if (GEN || ib > 0) {
#pragma unroll
                for (int j = 0; j < VAR_SUBS; ++j) {          // chunk 0: wave w fills steps w, w+8, w+16, w+24
                    fetch(j * VAR_SUB + w);
                    produce(0, j * VAR_SUB + w);
                }
            }
            d2 a_nxt[2], a_nx2[2];                         // A fragments of the next step and of the one after it
            a_nxt[0] = (wuni + S_ib * STEP_D2)[wlane]; a_nxt[1] = (wuni + S_ib * STEP_D2)[wlane + 64];   // step 0 is active for every group
            a_nx2[0] = (wuni + (S_ib + 1) * STEP_D2)[wlane]; a_nx2[1] = (wuni + (S_ib + 1) * STEP_D2)[wlane + 64];   // and so is step 1
            __syncthreads();
            d4 acc[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = d4{0, 0, 0, 0};
            const int nk4 = (ib + 1) * WT_K4;
            const int my_limit = ib * WT_K4 + 16 * (g + 1);     // first k4-step of the sweep with nothing left for this group
            // Reload sweeps take the diagonal tile (last 128 k-steps, where wave g only has 16 (g + 1) steps of work)
            // OUT of the lock-step LDS pipeline: see below.  The generating sweep keeps it in (its fragments are
            // not in the scratch image yet).
            const int nchunks = (GEN ? nk4 : ib * WT_K4) / VAR_CH;
            // Sources of the fills inside the loop, as loop-carried per-lane pointers (k-step VAR_CH + w first, then
            // VAR_SUB further each time): an address recomputed from the k-step lands in whatever registers are free —
            // the previous fill's destination registers — and that write-after-load made hipcc drain vmcnt to 0 (and with
            // it the A fragments in flight) at the top of every sub-chunk.
            const d2* bsrc = buni + (size_t)(VAR_CH + w) * 128 + blane;
            const double* xsrc = Xs + (size_t)((VAR_CH + w) * 4 + lk) * 4;
            auto fetch_next = [&]() {
                if (GEN) {
                    gx[0] = xsrc[0]; gx[1] = xsrc[1]; gx[2] = xsrc[2];
                    xsrc += VAR_SUB * 16;
                } else {
                    bl[0] = bsrc[0]; bl[1] = bsrc[1];
                    bsrc += VAR_SUB * 128;
                }
            };
            // the chunk body exists twice — with and without the fill of the following chunk — so that "is there a next
            // chunk" is no branch (and no join in front of the first MFMAs of a sub-chunk) inside it
            auto chunk = [&](auto more_tag, const int ch) {
                constexpr bool more = decltype(more_tag)::value;
                const int cur = ch & 1;
                d4 b_nxt = *reinterpret_cast<const d4*>(Bs(cur, 0));    // B fragments are read one k-step ahead
                for (int sub = 0; sub < VAR_SUBS; ++sub) {
                    const int k0 = ch * VAR_CH + sub * VAR_SUB;                 // first k-step of this sub-chunk
                    const int kn = (ch + 1) * VAR_CH + sub * VAR_SUB + w;       // the k-step this wave fills meanwhile
                    if (more) fetch_next();
                    const bool active = (k0 < my_limit) && !(GPT_ABL == 3 && k0 >= ib * WT_K4);   // my_limit is a multiple of 16: all or nothing
                    auto step = [&](const int s) {
                        const int k4 = k0 + s;
                        const d2 a01 = a_nxt[0], a23 = a_nxt[1];
                        const d4 b = b_nxt;
                        const int kl = my_limit - 1;
                        const size_t Sn = S_ib + ((k4 + 2 < my_limit) ? (k4 + 2) : kl);
                        a_nxt[0] = a_nx2[0]; a_nxt[1] = a_nx2[1];
                        if (GPT_ABL != 2) { a_nx2[0] = (wuni + Sn * STEP_D2)[wlane]; a_nx2[1] = (wuni + Sn * STEP_D2)[wlane + 64]; }
                        const int sn = sub * VAR_SUB + s + 1;
                        if (sn < VAR_CH) b_nxt = *reinterpret_cast<const d4*>(Bs(cur, sn));
                        if (GPT_ABL == 5) { asm volatile("" :: "v"(a01), "v"(a23), "v"(b)); return; }
                        GPT_MFMA16(acc, a01, a23, b);
                    };
                    // The fill of the next chunk sits INSIDE the active / idle paths, not behind their join: vmcnt counts in
                    // issue order, and behind a join hipcc has to wait for vmcnt(0) — which also waits for the A fragments
                    // the steps just before have requested (a full L2 round trip per sub-chunk); inside the straight-line
                    // path it waits for the fill's own loads only (vmcnt(4) / vmcnt(12)).
                    if (active) {
                        step(0); step(1);
                        if (more && w < 4 && GPT_ABL != 4) produce(cur ^ 1, kn);
                        step(2); step(3); step(4); step(5);
                        if (more && w >= 4 && GPT_ABL != 4) produce(cur ^ 1, kn);
                        step(6); step(7);
                    } else {
                        if (more && GPT_ABL != 4) produce(cur ^ 1, kn);
                        if (sub + 1 < VAR_SUBS) b_nxt = *reinterpret_cast<const d4*>(Bs(cur, (sub + 1) * VAR_SUB));
                    }
                }
                if (GPT_ABL != 1) __syncthreads();
            };
            for (int ch = 0; ch + 1 < nchunks; ++ch) chunk(std::true_type{}, ch);
            if (nchunks > 0) chunk(std::false_type{}, nchunks - 1);
            if (!GEN && GPT_ABL != 3) {
                // Diagonal tile of a reload sweep, barrier-free: every wave runs its own 16 (g + 1) k-steps with A from
                // Wf and B straight from the scratch image (both one MFMA block ahead; program order pinned with
                // sched_barrier so hipcc keeps the loads away from their first use).  No lock-step, so the waves with
                // g and 7 - g that share a SIMD add up to the same work on every SIMD: the tile costs 0.56 of a full one
                // instead of 0.75.
                const int kd0 = ib * WT_K4;                              // first k-step of the diagonal tile
                const int limit = 16 * (g + 1);                          // even
                const d2* ap = wuni + (S_ib + kd0) * STEP_D2;
                const d2* bp = buni + (size_t)kd0 * 128;
                auto ldA = [&](d2 (&a)[2], const int k) {
                    const int kk = k < limit ? k : limit - 1;            // clamped: redundant, in bounds
                    a[0] = (ap + (size_t)kk * STEP_D2)[wlane]; a[1] = (ap + (size_t)kk * STEP_D2)[wlane + 64];
                };
                auto ldB = [&](d2 (&b)[2], const int k) {
                    const int kk = k < limit ? k : limit - 1;
                    b[0] = (bp + (size_t)kk * 128)[blane]; b[1] = (bp + (size_t)kk * 128)[blane + 1];
                };
                d2 a0[2], a1[2], b0[2], b1[2];
                ldA(a0, 0); ldA(a1, 1); ldB(b0, 0);
                for (int k4 = 0; k4 < limit; k4 += 2) {
                    ldB(b1, k4 + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    { const d4 bb = d4{b0[0][0], b0[0][1], b0[1][0], b0[1][1]}; GPT_MFMA16(acc, a0[0], a0[1], bb); }
                    __builtin_amdgcn_sched_barrier(0);
                    ldA(a0, k4 + 2); ldB(b0, k4 + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    { const d4 bb = d4{b1[0][0], b1[0][1], b1[1][0], b1[1][1]}; GPT_MFMA16(acc, a1[0], a1[1], bb); }
                    __builtin_amdgcn_sched_barrier(0);
                    ldA(a1, k4 + 3);
                }
            }
            // i-block finished: fold this wave's 64 rows of V into the per-column sums
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double v = acc[r][t][e];
                        ssq[t] += v * v;
                        if (CROSS) crs[t] += v * __shfl(v, lane & ~3);
                    }
        };

        sweep(std::true_type{}, hi - 1);                      // longest sweep first: it covers every source the others need
        if (hi - 2 >= lo) {
            // every wave's part of the scratch image must have reached L2 before another wave reloads it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int ib = hi - 2; ib >= lo; --ib) {
                sweep(std::false_type{}, ib);
            }
        }

        // rows of a column are spread over the 4 lane groups lk = 0..3 and over the 8 waves
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ssq[t] += __shfl_xor(ssq[t], 16); ssq[t] += __shfl_xor(ssq[t], 32);
            if (CROSS) { crs[t] += __shfl_xor(crs[t], 16); crs[t] += __shfl_xor(crs[t], 32); }
        }
        if (lk == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                red[0][w][16 * t + lc] = ssq[t];
                red[1][w][16 * t + lc] = CROSS ? crs[t] : 0.0;
            }
        }
        __syncthreads();
        if (threadIdx.x < VAR_SLOT) {
            const int which = threadIdx.x >> 6, cl = threadIdx.x & 63;
            double v = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) v += red[which][ww][cl];
            slab[(size_t)slot * VAR_SLOT + threadIdx.x] = v;
        }
    }
}

// Adds the partial rows of a column block in range order and turns them into outputs.
template <int NCOMP>
__global__ __launch_bounds__(64) void k_var_finalize(KernelParams p, VarPlan pl, const double* __restrict__ slab, int64_t M,
                                                     double* __restrict__ var, double* __restrict__ Jvar,
                                                     double* __restrict__ dvar) {
    const int64_t cb = blockIdx.x;
    const int cl = threadIdx.x;
    double s2 = 0.0, cr = 0.0;
    if (cb < pl.nfull) {                                  // handled whole by one workgroup: one slot
        const double* sl = slab + (size_t)cb * VAR_SLOT;
        s2 = sl[cl]; cr = sl[VAR_COLS + cl];
    } else {                                              // tail: add the ranges that touched it, in range order
        const int64_t cbt = cb - pl.nfull;
        int64_t pa64 = (cbt * pl.T) * pl.P / pl.U - 2;    // B_p <= U p / P, so this p starts at or before the block
        if (pa64 < 0) pa64 = 0;
        if (pa64 > pl.P - 1) pa64 = pl.P - 1;
        for (int pp = (int)pa64; pp < pl.P; ++pp) {
            int64_t cbs, cbe; int ibs, ibe;
            var_boundary(pl, pp, cbs, ibs);
            if (cbs > cbt) break;
            var_boundary(pl, pp + 1, cbe, ibe);
            const int lo = (cbs == cbt) ? ibs : 0;                      // cbs <= cbt here
            const int hi = (cbe > cbt) ? pl.nbi : ((cbe == cbt) ? ibe : 0);
            if (lo >= hi) continue;
            const double* sl = slab + (size_t)(pl.nfull + pp + cbt) * VAR_SLOT;
            s2 += sl[cl];
            cr += sl[VAR_COLS + cl];
        }
    }
    const int D = p.D;
    const int64_t col = cb * VAR_COLS + cl;
    const int64_t m = (NCOMP == 1) ? col : ((NCOMP == 4) ? (col >> 2) : (col / D));
    const int cmp = (NCOMP == 1) ? 0 : ((NCOMP == 4) ? (int)(col & 3) : 1 + (int)(col % D));
    if (m >= M) return;
    if (cmp == 0) {
        if (var) { const double v = p.c + p.noise - s2; var[m] = v < 0.0 ? 0.0 : v; }
    } else {
        const int d = cmp - 1;
        if (d < D) {
            if (Jvar) Jvar[m * D + d] = p.c * p.inv_ls[d] * p.inv_ls[d] - s2;
            if (dvar) dvar[(int64_t)d * M + m] = -2.0 * cr;
        }
    }
}

static int var_workgroups() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        const char* e = getenv("GPT_VAR_WGS");
        if (e && atoi(e) > 0) n = atoi(e);
    }
    return n;
}

static VarPlan make_plan(const KernelParams& p, int64_t M, int ncomp) {
    VarPlan pl;
    pl.nbi = p.NP / WT;
    const int cpq = (ncomp == 3) ? p.D : ncomp;          // ncomp 3 = Jacobian variance alone: D columns per query
    pl.ncb = (M * cpq + VAR_COLS - 1) / VAR_COLS;
    pl.P = var_workgroups();
    pl.T = var_cost_prefix(pl.nbi);
    pl.nfull = pl.ncb / pl.P * pl.P;
    pl.ncb_t = pl.ncb - pl.nfull;
    pl.U = pl.ncb_t * pl.T;
    return pl;
}

size_t var_slab_doubles(int64_t M, int ncomp) {
    const int64_t ncb = (M * ncomp + VAR_COLS - 1) / VAR_COLS;
    return (size_t)(ncb + var_workgroups() + 1) * VAR_SLOT;
}

size_t var_bscratch_doubles(int NP) { return (size_t)var_workgroups() * (size_t)NP * VAR_COLS; }

void launch_var(hipStream_t s, const KernelParams& p, const double* Xs, const double* Wf,
                const double* Xq, int64_t M, int ncomp, double* var, double* Jvar, double* dvar, double* slab,
                double* bscratch) {
    if (M <= 0) return;
    const VarPlan pl = make_plan(p, M, ncomp);
    static bool attr_set = false;
    if (!attr_set) {      // 128 KiB of dynamic LDS per workgroup
        const void* fns[] = {reinterpret_cast<const void*>(k_var<1, false, KT_RBF>), reinterpret_cast<const void*>(k_var<4, true, KT_RBF>),
                             reinterpret_cast<const void*>(k_var<4, false, KT_RBF>), reinterpret_cast<const void*>(k_var<3, false, KT_RBF>),
                             reinterpret_cast<const void*>(k_var<1, false, KT_MATERN12>),
                             reinterpret_cast<const void*>(k_var<1, false, KT_MATERN32>), reinterpret_cast<const void*>(k_var<1, false, KT_MATERN52>)};
        for (const void* f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VAR_LDS_BYTES);
        attr_set = true;
    }
    const dim3 grid((unsigned)pl.P), fgrid((unsigned)pl.ncb);
    if (ncomp == 1) {
        switch (p.ktype) {
            case KT_MATERN12: hipLaunchKernelGGL((k_var<1, false, KT_MATERN12>), grid, dim3(512), VAR_LDS_BYTES, s, p, pl, Xs, Wf, Xq, M, slab, bscratch); break;
            case KT_MATERN32: hipLaunchKernelGGL((k_var<1, false, KT_MATERN32>), grid, dim3(512), VAR_LDS_BYTES, s, p, pl, Xs, Wf, Xq, M, slab, bscratch); break;
            case KT_MATERN52: hipLaunchKernelGGL((k_var<1, false, KT_MATERN52>), grid, dim3(512), VAR_LDS_BYTES, s, p, pl, Xs, Wf, Xq, M, slab, bscratch); break;
            default: hipLaunchKernelGGL((k_var<1, false, KT_RBF>), grid, dim3(512), VAR_LDS_BYTES, s, p, pl, Xs, Wf, Xq, M, slab, bscratch);
        }
        hipLaunchKernelGGL((k_var_finalize<1>), fgrid, dim3(64), 0, s, p, pl, slab, M, var, Jvar, dvar);
    } else if (ncomp == 3) {      // Jacobian variance alone: three columns per query
        hipLaunchKernelGGL((k_var<3, false, KT_RBF>), grid, dim3(512), VAR_LDS_BYTES, s, p, pl, Xs, Wf, Xq, M, slab, bscratch);
        hipLaunchKernelGGL((k_var_finalize<3>), fgrid, dim3(64), 0, s, p, pl, slab, M, var, Jvar, dvar);
    } else {      // Jacobian variance / d var: RBF only (the API refuses other kernels)
        if (dvar) hipLaunchKernelGGL((k_var<4, true, KT_RBF>), grid, dim3(512), VAR_LDS_BYTES, s, p, pl, Xs, Wf, Xq, M, slab, bscratch);
        else hipLaunchKernelGGL((k_var<4, false, KT_RBF>), grid, dim3(512), VAR_LDS_BYTES, s, p, pl, Xs, Wf, Xq, M, slab, bscratch);
        hipLaunchKernelGGL((k_var_finalize<4>), fgrid, dim3(64), 0, s, p, pl, slab, M, var, Jvar, dvar);
    }
}

}  // namespace gpt
