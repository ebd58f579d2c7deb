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

namespace gpt {

// ------------------------------------------------------------------------------------------
template <int QPW>
__global__ __launch_bounds__(256) void k_mean_jac(KernelParams p, const double* __restrict__ Xs,
                                                  const double* __restrict__ A4, const double* __restrict__ Xq,
                                                  int64_t M, int o_base, int o_cnt, double* __restrict__ mean,
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
    double acc[QPW][4][4];
#pragma unroll
    for (int i = 0; i < QPW; ++i)
#pragma unroll
        for (int o = 0; o < 4; ++o)
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
            double tt = fma(-d0, d0, lnc);
            tt = fma(-d1, d1, tt);
            tt = fma(-d2, d2, tt);
            const double kv = exp_tab(tt, Tt);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
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
        for (int o = 0; o < 4; ++o)
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
            for (int o = 0; o < 4; ++o) {
                if (o >= o_cnt) break;
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
        hipLaunchKernelGGL(k_mean_jac<QPW>, dim3((unsigned)blocks), dim3(256), 0, s, p, Xs,
                           A4 + (size_t)(ob / 4) * p.NP * 4, Xq, M, ob, cnt, mean, J);
    }
}

// ------------------------------------------------------------------------------------------
// Variance kernel.
//
// Measured on MI355X (profiles/r01_mfma_f64_vs_valu_probe.txt): v_mfma_f64_16x16x4_f64 issues every
// 64 cycles per SIMD (77.7 TFLOP/s with 2 waves/SIMD), but every fp64 VALU instruction of the same
// SIMD costs ~4.5 of those cycles — the fp64 matrix and vector paths share the DP units.  The design
// therefore minimises fp64 VALU work per MFMA and keeps the rest off the critical path:
//   * a workgroup (512 threads = 8 waves, 2 per SIMD) owns 64 columns and a whole 512-row i-block:
//     wave w accumulates the 64x64 product of its 64-row group (16 MFMA tiles, 128 VGPRs);
//   * the B operand (k* / dk_d columns, one fp64 exp each — table-driven, gpt_exp.h) is generated ONCE
//     per workgroup per k-step: wave w produces the fragments of k-step w of the next 8-step chunk into a
//     double-buffered 2 x 16 KiB LDS image already in MFMA lane order (one barrier per chunk), so a wave
//     pays 4 exps per 128 MFMAs.  The generation sits in the MIDDLE of the wave's MFMA run and the two
//     waves of a SIMD are staggered (after step 2 / after step 6), so a barrier release is followed by
//     MFMAs at once and one wave's exp latency hides under its partner's MFMAs;
//   * the A operand streams from the fragment-ordered image Wf with two 16-byte loads per lane and
//     k-step, prefetched one step ahead; in the diagonal tile a wave skips the k-steps where its row
//     group is entirely above the diagonal, and row groups are paired (0,7)(1,6)(2,5)(3,4) on the SIMDs
//     so the skipped work is balanced;
//   * when an i-block is finished its V rows are squared and folded into per-column sums, so
//     V = W K*^T never touches memory;
//   * work is split stream-K style: the (column block, i-block) units, costed 128*ib + 72 k-steps, are
//     laid end to end and cut into P equal ranges, one persistent workgroup per CU; a range writes one
//     partial row of column sums per column block it touches into a slab slot (p + cb, unique), and
//     k_var_finalize adds the slots of a block in fixed order (deterministic, no atomics).
// NCOMP = 1: one column per query (k*).  NCOMP = 4: four columns per query (k*, dk_0, dk_1, dk_2).
// ------------------------------------------------------------------------------------------
constexpr int VAR_COLS = 64;        // columns per column block
constexpr int VAR_CH = 8;           // k4-steps per LDS chunk (= waves per workgroup)
constexpr int VAR_DIAG_COST = 72;   // k4-steps a diagonal tile costs its slowest SIMD (16 + 128 of 2 x 128)
constexpr int VAR_SLOT = 2 * VAR_COLS;   // doubles per slab slot: ssq[64], crs[64]

struct VarPlan {
    int64_t ncb;     // column blocks
    int64_t T;       // cost of one column block (all i-blocks)
    int64_t U;       // total cost = ncb * T
    int P;           // workgroups (ranges)
    int nbi;         // i-blocks
};

__host__ __device__ inline int64_t var_cost_prefix(int ib) {          // sum_{i<ib} (128 i + 72)
    return (int64_t)64 * ib * (ib - 1) + (int64_t)VAR_DIAG_COST * ib;
}

// first work unit (column block, i-block) of range p; p = P gives the end of the work
__host__ __device__ inline void var_boundary(const VarPlan& pl, int p, int64_t& cb, int& ib) {
    if (p >= pl.P) { cb = pl.ncb; ib = 0; return; }
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

template <int NCOMP, bool CROSS>
__global__ __launch_bounds__(512, 2) void k_var(KernelParams p, VarPlan pl, const double* __restrict__ Xs,
                                                const double* __restrict__ Wf, const double* __restrict__ Xq,
                                                int64_t M, double* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) double Bs[2][VAR_CH][64][4];   // [buffer][k4-step][lane][column tile]
    __shared__ double red[2][8][VAR_COLS];
    __shared__ double Tt[256];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lc = lane & 15, lk = lane >> 4;
    const int g = (w < 4) ? w : (11 - w);     // row group of this wave: 0,1,2,3,7,6,5,4
    const int D = p.D;
    if (threadIdx.x < 256) Tt[threadIdx.x] = g_exp2_table[threadIdx.x];

    int64_t cb0, cb1; int ib0, ib1;
    var_boundary(pl, blockIdx.x, cb0, ib0);
    var_boundary(pl, blockIdx.x + 1, cb1, ib1);

    constexpr double RS2 = 0.70710678118654752440;    // coordinates are pre-scaled by 1/sqrt(2): t = ln c - |d'|^2
    const int comp = (NCOMP == 1) ? 0 : (lc & 3);     // NCOMP=4: b = kv * (cb + sum_d cd[d] * d'_d)
    const double cbv = (comp == 0) ? 1.0 : 0.0;
    double cd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) cd[d] = (comp == d + 1 && d < D) ? p.inv_ls[d] * 1.41421356237309504880 : 0.0;
    const double lnc = p.lnc;
    const int nbi = pl.nbi;
    // A stream in d2 units: element (step S, group g, q, lane) at ((S*8 + g)*2 + q)*64 + lane
    const d2* wbase = reinterpret_cast<const d2*>(Wf) + (size_t)g * 128 + lane;
    constexpr size_t STEP_D2 = WT_STEP_DOUBLES / 2;   // 1024

    for (int64_t cb = cb0; cb <= cb1; ++cb) {
        const int lo = (cb == cb0) ? ib0 : 0;
        const int hi = (cb == cb1) ? ib1 : nbi;
        if (cb >= pl.ncb || lo >= hi) continue;        // uniform over the workgroup
        __syncthreads();                               // LDS (Bs, red, Tt) free / ready

        // this lane's four columns (one per MFMA column tile): scaled query coordinates
        double q[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t col = cb * VAR_COLS + 16 * t + lc;
            const int64_t m = (NCOMP == 1) ? col : (col >> 2);
            const int64_t mm = (m < M) ? m : (M - 1);
#pragma unroll
            for (int d = 0; d < 3; ++d) q[t][d] = (d < D) ? Xq[mm * D + d] * (p.inv_ls[d] * RS2) : 0.0;
        }
        double gx[3];                                  // coordinates of the source this wave generates next
        auto fetch = [&](const int k4base) {
            const double* xp = Xs + (size_t)((k4base + w) * 4 + lk) * 4;
            gx[0] = xp[0]; gx[1] = xp[1]; gx[2] = xp[2];
        };
        auto generate = [&](const int buf) {           // B fragments of k-step (k4base + w) -> LDS
            const double x0 = gx[0] * RS2, x1 = gx[1] * RS2, x2 = gx[2] * RS2;
            d4 b;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double d0 = x0 - q[t][0], d1 = x1 - q[t][1], d2_ = x2 - q[t][2];
                double tt = fma(-d0, d0, lnc);
                tt = fma(-d1, d1, tt);
                tt = fma(-d2_, d2_, tt);
                const double kv = exp_tab(tt, Tt);
                b[t] = (NCOMP == 1) ? kv : kv * (cbv + cd[0] * d0 + cd[1] * d1 + cd[2] * d2_);
            }
            *reinterpret_cast<d4*>(&Bs[buf][w][lane][0]) = b;
        };

        double ssq[4] = {0.0, 0.0, 0.0, 0.0}, crs[4] = {0.0, 0.0, 0.0, 0.0};
        size_t S_ib = (size_t)64 * lo * (lo + 1);      // stream index of the first k4-step of i-block lo
        fetch(0);
        generate(0);
        d2 a_nxt[2];
        a_nxt[0] = wbase[S_ib * STEP_D2]; a_nxt[1] = wbase[S_ib * STEP_D2 + 64];   // step 0 is active for every group
        __syncthreads();

        int it = 0;            // chunk counter (LDS buffer parity)
        for (int ib = lo; ib < hi; ++ib) {
            d4 acc[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = d4{0, 0, 0, 0};
            const int nk4 = (ib + 1) * WT_K4;
            const int my_limit = ib * WT_K4 + 16 * (g + 1);     // first k4-step of the sweep with nothing left for this group
            const int nchunks = nk4 / VAR_CH;
            for (int ch = 0; ch < nchunks; ++ch) {
                const int cur = it & 1;
                const bool more = (ch + 1 < nchunks) || (ib + 1 < hi);
                if (more) fetch((ch + 1 < nchunks) ? (ch + 1) * VAR_CH : 0);
                const bool active = ch * VAR_CH < my_limit;    // my_limit is a multiple of 16: all or nothing
                auto step = [&](const int s) {
                    const int k4 = ch * VAR_CH + s;
                    const d2 a01 = a_nxt[0], a23 = a_nxt[1];
                    const size_t Sn = (k4 + 1 < my_limit) ? (S_ib + k4 + 1) : (S_ib + nk4);
                    a_nxt[0] = wbase[Sn * STEP_D2]; a_nxt[1] = wbase[Sn * STEP_D2 + 64];
                    const d4 b = *reinterpret_cast<const d4*>(&Bs[cur][s][lane][0]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[0], b[t], acc[0][t], 0, 0, 0);
                        acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[1], b[t], acc[1][t], 0, 0, 0);
                        acc[2][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[0], b[t], acc[2][t], 0, 0, 0);
                        acc[3][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[1], b[t], acc[3][t], 0, 0, 0);
                    }
                };
                if (active) { step(0); step(1); }
                if (more && w < 4) generate(cur ^ 1);
                if (active) { step(2); step(3); step(4); step(5); }
                if (more && w >= 4) generate(cur ^ 1);
                if (active) { step(6); step(7); }
                __syncthreads();
                ++it;
            }
            S_ib += nk4;
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
            slab[((size_t)blockIdx.x + (size_t)cb) * VAR_SLOT + threadIdx.x] = v;
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
    // first range that can touch this block: boundaries are monotone in p and spaced U/P apart
    int64_t pa64 = (cb * pl.T) * pl.P / pl.U - 2;       // B_p <= U p / P, so this p starts at or before the block
    if (pa64 < 0) pa64 = 0;
    if (pa64 > pl.P - 1) pa64 = pl.P - 1;
    const int pa = (int)pa64;
    double s2 = 0.0, cr = 0.0;
    for (int pp = pa; pp < pl.P; ++pp) {
        int64_t cbs, cbe; int ibs, ibe;
        var_boundary(pl, pp, cbs, ibs);
        if (cbs > cb) break;
        var_boundary(pl, pp + 1, cbe, ibe);
        const int lo = (cbs == cb) ? ibs : 0;                       // cbs <= cb here
        const int hi = (cbe > cb) ? pl.nbi : ((cbe == cb) ? ibe : 0);
        if (lo >= hi) continue;
        const double* sl = slab + ((size_t)pp + (size_t)cb) * VAR_SLOT;
        s2 += sl[cl];
        cr += sl[VAR_COLS + cl];
    }
    const int D = p.D;
    const int64_t col = cb * VAR_COLS + cl;
    const int64_t m = (NCOMP == 1) ? col : (col >> 2);
    const int cmp = (NCOMP == 1) ? 0 : (int)(col & 3);
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
    pl.ncb = (M * ncomp + VAR_COLS - 1) / VAR_COLS;
    pl.T = var_cost_prefix(pl.nbi);
    pl.U = pl.ncb * pl.T;
    pl.P = var_workgroups();
    return pl;
}

size_t var_slab_doubles(int64_t M, int ncomp) {
    const int64_t ncb = (M * ncomp + VAR_COLS - 1) / VAR_COLS;
    return (size_t)(ncb + var_workgroups() + 1) * VAR_SLOT;
}

void launch_var(hipStream_t s, const KernelParams& p, const double* Xs, const double* Wf,
                const double* Xq, int64_t M, int ncomp, double* var, double* Jvar, double* dvar, double* slab) {
    if (M <= 0) return;
    const VarPlan pl = make_plan(p, M, ncomp);
    const dim3 grid((unsigned)pl.P), fgrid((unsigned)pl.ncb);
    if (ncomp == 1) {
        hipLaunchKernelGGL((k_var<1, false>), grid, dim3(512), 0, s, p, pl, Xs, Wf, Xq, M, slab);
        hipLaunchKernelGGL((k_var_finalize<1>), fgrid, dim3(64), 0, s, p, pl, slab, M, var, Jvar, dvar);
    } else {
        if (dvar) hipLaunchKernelGGL((k_var<4, true>), grid, dim3(512), 0, s, p, pl, Xs, Wf, Xq, M, slab);
        else hipLaunchKernelGGL((k_var<4, false>), grid, dim3(512), 0, s, p, pl, Xs, Wf, Xq, M, slab);
        hipLaunchKernelGGL((k_var_finalize<4>), fgrid, dim3(64), 0, s, p, pl, slab, M, var, Jvar, dvar);
    }
}

}  // namespace gpt
