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
//               (gaussian_process.py:122-125).  The B operand (k* and dk_d columns) is generated in
//               registers directly in v_mfma_f64_16x16x4_f64 lane order (one exp per lane per MFMA
//               column tile); the A operand streams from the fragment-ordered tile image Wf with
//               16-byte loads, one k-step ahead; products accumulate in 16 MFMA tiles per wave and are
//               squared and summed per column when an i-block is finished, so V = W K*^T never
//               touches memory.
#include "gpt_common.h"
#include <cstdlib>

namespace gpt {

// ------------------------------------------------------------------------------------------
template <int QPW>
__global__ __launch_bounds__(256) void k_mean_jac(KernelParams p, const double* __restrict__ Xs,
                                                  const double* __restrict__ A4, const double* __restrict__ Xq,
                                                  int64_t M, int o_base, int o_cnt, double* __restrict__ mean,
                                                  double* __restrict__ J) {
    const int lane = threadIdx.x & 63;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * QPW;
    if (m0 >= M) return;
    const int D = p.D;
    double q[QPW][3];
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
        const int64_t m = (m0 + i < M) ? (m0 + i) : (M - 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) q[i][d] = (d < D) ? Xq[m * D + d] * p.inv_ls[d] : 0.0;
    }
    double acc[QPW][4][4];
#pragma unroll
    for (int i = 0; i < QPW; ++i)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][o][e] = 0.0;

    const double c = p.c;
    for (int n = lane; n < p.N; n += 64) {
        const d4 xs = *reinterpret_cast<const d4*>(Xs + (size_t)n * 4);
        const d4 al = *reinterpret_cast<const d4*>(A4 + (size_t)n * 4);
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const double d0 = xs[0] - q[i][0], d1 = xs[1] - q[i][1], d2 = xs[2] - q[i][2];
            const double kv = c * exp(-0.5 * (d0 * d0 + d1 * d1 + d2 * d2));
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
                        if (d < D) J[(m * O + oo) * D + d] = acc[i][o][1 + d] * p.inv_ls[d];
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
// Variance kernel.  512 threads = 8 waves (2 per SIMD); the workgroup owns 256 columns, each wave
// 32 of them (2 MFMA column tiles) against all WT=128 rows of the current i-block (8 row tiles).
// NCOMP = 1: one column per query (k*).  NCOMP = 4: four columns per query (k*, dk_0, dk_1, dk_2).
//
// SCHED selects how the B-operand generation (fp64 exp on the VALU) is overlapped with the MFMAs:
//   0  generate b for step k, then its 16 MFMAs (two phases per step)
//   1  as 0, with s_setprio 1 around the MFMA phase so the two waves of a SIMD alternate phases
//      instead of drifting into lockstep (both in the VALU phase = idle matrix pipe)
//   2  software-pipelined: b for step k+1 is generated while the MFMAs of step k issue; the two
//      instruction streams are interleaved with sched_group_barrier (1 MFMA : 5 VALU)
//   3  as 2 without the explicit interleave (compiler order) but with s_setprio around the MFMAs
// ------------------------------------------------------------------------------------------
constexpr int VAR_COLS = 256;

template <int NCOMP>
__device__ __forceinline__ void make_b(const double x0, const double x1, const double x2, const double (&q)[2][3],
                                       const double (&cb)[2], const double (&cd)[2][3], const double c,
                                       double (&b)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const double d0 = x0 - q[t][0], d1 = x1 - q[t][1], d2_ = x2 - q[t][2];
        const double kv = c * exp(-0.5 * (d0 * d0 + d1 * d1 + d2_ * d2_));
        if (NCOMP == 1) b[t] = kv;
        else b[t] = kv * (cb[t] + cd[t][0] * d0 + cd[t][1] * d1 + cd[t][2] * d2_);
    }
}

template <int NCOMP, bool CROSS, int SCHED>
__global__ __launch_bounds__(512, 2) void k_var(KernelParams p, const double* __restrict__ Xs,
                                                const double* __restrict__ Wf, const double* __restrict__ Xq,
                                                int64_t M, double* __restrict__ var, double* __restrict__ Jvar,
                                                double* __restrict__ dvar) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    const int D = p.D;
    const int64_t col0 = (int64_t)blockIdx.x * VAR_COLS + w * 32;

    double q[2][3];          // scaled query coordinates of this lane's two columns
    double cb[2], cd[2][3];  // NCOMP=4: b = kv * (cb + sum_d cd[d] * (xs_d - q_d))
    int64_t qm[2]; int comp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t col = col0 + 16 * t + lc;
        const int64_t m = (NCOMP == 1) ? col : (col >> 2);
        comp[t] = (NCOMP == 1) ? 0 : (int)(col & 3);
        qm[t] = m;
        const int64_t mm = (m < M) ? m : (M - 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) q[t][d] = (d < D) ? Xq[mm * D + d] * p.inv_ls[d] : 0.0;
        cb[t] = (comp[t] == 0) ? 1.0 : 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) cd[t][d] = (comp[t] == d + 1 && d < D) ? p.inv_ls[d] : 0.0;
    }
    double ssq[2] = {0.0, 0.0}, crs[2] = {0.0, 0.0};
    const double c = p.c;
    const int nbi = p.NP / WT;
    constexpr int NQ = WT_RT / 2;                  // 16-byte fragment pairs per k4-step (4)
    const d2* wp = reinterpret_cast<const d2*>(Wf) + lane;   // stream pointer (d2 units)
    constexpr size_t STEP_D2 = WT_STEP_DOUBLES / 2;          // d2 per k4-step (256)
    constexpr bool PIPE_B = (SCHED >= 2);

    // prologue of the software pipeline: fragments + source coords of the first k4-step
    d2 a_nxt[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) a_nxt[u] = wp[u * 64];
    wp += STEP_D2;
    double xs_nxt[3];
    {
        const double* xp = Xs + (size_t)lk * 4;
        xs_nxt[0] = xp[0]; xs_nxt[1] = xp[1]; xs_nxt[2] = xp[2];
    }
    double b_nxt[2] = {0.0, 0.0};
    if (PIPE_B) {   // b of step 0 now, coords of step 1 in flight
        make_b<NCOMP>(xs_nxt[0], xs_nxt[1], xs_nxt[2], q, cb, cd, c, b_nxt);
        const double* xp = Xs + (size_t)(4 + lk) * 4;
        xs_nxt[0] = xp[0]; xs_nxt[1] = xp[1]; xs_nxt[2] = xp[2];
    }

    for (int ib = 0; ib < nbi; ++ib) {
        d4 acc[WT_RT][2];
#pragma unroll
        for (int r = 0; r < WT_RT; ++r) { acc[r][0] = d4{0, 0, 0, 0}; acc[r][1] = d4{0, 0, 0, 0}; }
        const int nk4 = (ib + 1) * WT_K4;
        for (int k4 = 0; k4 < nk4; ++k4) {
            d2 a_cur[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) a_cur[u] = a_nxt[u];
            const double x0 = xs_nxt[0], x1 = xs_nxt[1], x2 = xs_nxt[2];
            // prefetch the next k4-step (the stream is contiguous across i-blocks; the image ends
            // with one spare step so the very last prefetch stays in bounds)
#pragma unroll
            for (int u = 0; u < NQ; ++u) a_nxt[u] = wp[u * 64];
            wp += STEP_D2;
            {
                // coords wanted next: step k4+1 (two-phase) or k4+2 (pipelined), wrapping into the
                // next i-block, whose k-sweep restarts at source 0
                const int ahead = PIPE_B ? 2 : 1;
                const int kn = (k4 + ahead < nk4) ? (k4 + ahead) : (k4 + ahead - nk4);
                const double* xp = Xs + (size_t)(kn * 4 + lk) * 4;
                xs_nxt[0] = xp[0]; xs_nxt[1] = xp[1]; xs_nxt[2] = xp[2];
            }
            double b[2];
            if (PIPE_B) {
                b[0] = b_nxt[0]; b[1] = b_nxt[1];
                make_b<NCOMP>(x0, x1, x2, q, cb, cd, c, b_nxt);     // for step k4+1, independent of the MFMAs below
            } else {
                make_b<NCOMP>(x0, x1, x2, q, cb, cd, c, b);
            }
            if (SCHED == 1 || SCHED == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[2 * u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[u][0], b[t], acc[2 * u][t], 0, 0, 0);
                    acc[2 * u + 1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[u][1], b[t], acc[2 * u + 1][t], 0, 0, 0);
                }
            }
            if (SCHED == 1 || SCHED == 3) __builtin_amdgcn_s_setprio(0);
            if (SCHED == 2) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // 5 VALU
                }
            }
        }
        // i-block finished: fold its 128 rows of V into the per-column sums
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < WT_RT; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double v = acc[r][t][e];
                    ssq[t] += v * v;
                    if (CROSS) crs[t] += v * __shfl(v, lane & ~3);
                }
    }
    // rows of a column are spread over the 4 lane groups lk = 0..3
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ssq[t] += __shfl_xor(ssq[t], 16); ssq[t] += __shfl_xor(ssq[t], 32);
        if (CROSS) { crs[t] += __shfl_xor(crs[t], 16); crs[t] += __shfl_xor(crs[t], 32); }
    }
    if (lk == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t m = qm[t];
            if (m >= M) continue;
            if (comp[t] == 0) {
                if (var) { const double v = c + p.noise - ssq[t]; var[m] = v < 0.0 ? 0.0 : v; }
            } else {
                const int d = comp[t] - 1;
                if (d < D) {
                    if (Jvar) Jvar[m * D + d] = c * p.inv_ls[d] * p.inv_ls[d] - ssq[t];
                    if (CROSS && dvar) dvar[(int64_t)d * M + m] = -2.0 * crs[t];
                }
            }
        }
    }
}

static int var_sched() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GPT_VAR_SCHED");
        v = e ? atoi(e) : 1;
        if (v < 0 || v > 3) v = 1;
    }
    return v;
}

template <int SCHED>
static void launch_var_s(hipStream_t s, const KernelParams& p, const double* Xs, const double* Wf,
                         const double* Xq, int64_t M, int ncomp, double* var, double* Jvar, double* dvar, unsigned blocks) {
    if (ncomp == 1) {
        hipLaunchKernelGGL((k_var<1, false, SCHED>), dim3(blocks), dim3(512), 0, s, p, Xs, Wf, Xq, M, var, Jvar, dvar);
    } else if (dvar) {
        hipLaunchKernelGGL((k_var<4, true, SCHED>), dim3(blocks), dim3(512), 0, s, p, Xs, Wf, Xq, M, var, Jvar, dvar);
    } else {
        hipLaunchKernelGGL((k_var<4, false, SCHED>), dim3(blocks), dim3(512), 0, s, p, Xs, Wf, Xq, M, var, Jvar, dvar);
    }
}

void launch_var(hipStream_t s, const KernelParams& p, const double* Xs, const double* Wf,
                const double* Xq, int64_t M, int ncomp, double* var, double* Jvar, double* dvar) {
    if (M <= 0) return;
    const int64_t cols = M * ncomp;
    const unsigned blocks = (unsigned)((cols + VAR_COLS - 1) / VAR_COLS);
    switch (var_sched()) {
        case 0: launch_var_s<0>(s, p, Xs, Wf, Xq, M, ncomp, var, Jvar, dvar, blocks); break;
        case 2: launch_var_s<2>(s, p, Xs, Wf, Xq, M, ncomp, var, Jvar, dvar, blocks); break;
        case 3: launch_var_s<3>(s, p, Xs, Wf, Xq, M, ncomp, var, Jvar, dvar, blocks); break;
        default: launch_var_s<1>(s, p, Xs, Wf, Xq, M, ncomp, var, Jvar, dvar, blocks); break;
    }
}

}  // namespace gpt
