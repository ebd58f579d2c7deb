// Fit-side kernels for gfx950 (MI355X): RBF / Matern Gram assembly, two-level blocked fp64
// Cholesky (one launch per 64 columns, rank-128 trailing updates by MFMA v_mfma_f64_16x16x4_f64),
// blocked triangular inverse by recursive doubling, alpha = K^-1 Y, and packing of W = L^-1 into the
// MFMA-fragment-ordered tile stream the variance kernel consumes.
//
// What is replaced (reference, all CPU/LAPACK): sklearn/_gpr.py:346-364 (kernel_(X), +alpha on
// the diagonal, cholesky, cho_solve) and models/gaussian_process.py:42-43 (explicit K^-1, here
// kept as its triangular factor W = L^-1 so that k^T K^-1 k = |W k|^2).
#include "gpt_common.h"
#include "gpt_exp.h"
#include "gpt_fit_plan.h"
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

namespace gpt {

struct InvLs { double v[MAX_D]; };       // 1 / length_scale per dimension, by value to the kernels that scale coordinates

// =====================================================================================
// Gram assembly: K[i][j] = c * exp(-0.5 |xs_i - xs_j|^2) (+ diag_add on the diagonal) for the
// block lower triangle; padded rows/cols get the identity so the factorisation stays PD.
// HBM-write bound: one 64x64 tile per workgroup, 16 bytes per lane and store, 8 stores per thread.
// =====================================================================================
// DW = 3: rows of 4 (the tuned D <= 3 layout); DW = 8 / 16: rows of 8 / 16 (3 < D <= 8 / <= 15, unused coordinates are zero).
template <int DW>
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ Xs, int N, int NP, int ktype, double c,
                                              double diag_add, double* __restrict__ K) {
    constexpr int XS = DW == 3 ? 4 : DW;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    __shared__ double xi[64][DW], xj[64][DW];
    const int t = threadIdx.x;
    if (t < 64) {
        const double* p = Xs + (size_t)(bi * 64 + t) * XS;
#pragma unroll
        for (int d = 0; d < DW; ++d) xi[t][d] = p[d];
    } else if (t < 128) {
        const double* p = Xs + (size_t)(bj * 64 + (t - 64)) * XS;
#pragma unroll
        for (int d = 0; d < DW; ++d) xj[t - 64][d] = p[d];
    }
    __syncthreads();
    // lane -> column pair, 2 rows per wave-level store: every store instruction writes two contiguous 512-B row segments
    const int cc = (t & 31) * 2;
    double b[DW][2];
#pragma unroll
    for (int d = 0; d < DW; ++d) { b[d][0] = xj[cc][d]; b[d][1] = xj[cc + 1][d]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = (t >> 5) + 8 * u;
        const int i = bi * 64 + r;
        double a[DW];
#pragma unroll
        for (int d = 0; d < DW; ++d) a[d] = xi[r][d];
        double v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = bj * 64 + cc + e;
            double r2 = 0.0;
#pragma unroll
            for (int d = 0; d < DW; ++d) { const double df = a[d] - b[d][e]; r2 += df * df; }
            double val = kernel_libm(ktype, c, r2);
            if (i == j) val = c + diag_add;            // k(0) = 1 exactly (kernels.py:1562)
            if (i >= N || j >= N) val = (i == j) ? 1.0 : 0.0;
            v[e] = val;
        }
        *reinterpret_cast<d2*>(K + (size_t)i * NP + bj * 64 + cc) = d2{v[0], v[1]};
    }
}

void launch_gram(hipStream_t s, const double* Xs, int D, int N, int NP, int ktype, double c, double diag_add, double* K) {
    dim3 grid(NP / 64, NP / 64);
    if (D <= 3) hipLaunchKernelGGL(k_gram<3>, grid, dim3(256), 0, s, Xs, N, NP, ktype, c, diag_add, K);
    else if (D <= WIDE_D) hipLaunchKernelGGL(k_gram<WIDE_D>, grid, dim3(256), 0, s, Xs, N, NP, ktype, c, diag_add, K);
    else hipLaunchKernelGGL(k_gram<MAX_D>, grid, dim3(256), 0, s, Xs, N, NP, ktype, c, diag_add, K);
}

// Xs[i][d] = X[i][d] / l_d for i < N, d < D, zero elsewhere (rows of 4 or 8: xs_stride): the fp64 image the fit kernels
// read and, when Xm is given, the model's copy in its element type.  X is the raw (N, D) row-major input, kept on the
// device so that a new set of length-scales (every evaluation of the optimizer's objective) costs no host work and no copy.
template <typename TM>
__global__ __launch_bounds__(256) void k_scale_x(const double* __restrict__ X, int N, int NP, int D, int XS, InvLs il,
                                                 double* __restrict__ Xs64, TM* __restrict__ Xm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NP) return;
#pragma unroll
    for (int d = 0; d < MAX_D; ++d) {
        if (d >= XS) break;
        const double v = (i < N && d < D) ? X[(size_t)i * D + d] * il.v[d] : 0.0;
        Xs64[(size_t)i * XS + d] = v;
        if (Xm) Xm[(size_t)i * XS + d] = (TM)v;
    }
}

void launch_scale_x(hipStream_t s, const double* X, int N, int NP, int D, const double* inv_ls, double* Xs64, void* Xm, int dtype) {
    InvLs il;
    for (int d = 0; d < MAX_D; ++d) il.v[d] = inv_ls[d];
    const dim3 grid((NP + 255) / 256);
    if (dtype == DT_F32) hipLaunchKernelGGL(k_scale_x<float>, grid, dim3(256), 0, s, X, N, NP, D, xs_stride(D), il, Xs64, static_cast<float*>(Xm));
    else hipLaunchKernelGGL(k_scale_x<double>, grid, dim3(256), 0, s, X, N, NP, D, xs_stride(D), il, Xs64, static_cast<double*>(Xm));
}

// out[0] = sum_i a[i] b[i], one workgroup, fixed order (deterministic): sum_o y_o^T alpha_o of the LML over the padded
// [pass][NP][4] images (padding is zero on both sides).
__global__ __launch_bounds__(256) void k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* __restrict__ out) {
    __shared__ double red[256];
    double sacc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) sacc = fma(a[i], b[i], sacc);
    red[threadIdx.x] = sacc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

void launch_dot(hipStream_t s, const double* a, const double* b, int64_t n, double* out) {
    hipLaunchKernelGGL(k_dot, dim3(1), dim3(256), 0, s, a, b, n, out);
}

// K[i][j] += S[i][j] on the lower triangle of the first N rows (S row-major N x N, symmetric): the general
// SPD "noise" matrix of the SVGP exact-conversion model (K_uu + Sigma_pseudo).
__global__ __launch_bounds__(256) void k_add_lower(double* __restrict__ K, const double* __restrict__ S, int N, int NP) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)N * N) return;
    const int i = (int)(e / N), j = (int)(e % N);
    if (j <= i) K[(size_t)i * NP + j] += S[e];
}

void launch_add_lower(hipStream_t s, double* K, const double* S, int N, int NP) {
    const int64_t tot = (int64_t)N * N;
    hipLaunchKernelGGL(k_add_lower, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, K, S, N, NP);
}

constexpr int DS = NB + 1;   // LDS row stride (doubles) of the NB x NB block images

// =====================================================================================
// One step of the panel factorisation (left-looking inside an outer panel of `OB` columns).
// Workgroup b of step kb owns block row r = kb + b (b = 0: the diagonal block itself).  Every workgroup
//   1. rebuilds the diagonal block  D = A[kb,kb] - sum_j L[kb,j] L[kb,j]^T  over the panel's earlier block
//      columns j in [p0, kb) (MFMA) and factors it — redundantly, so that a step is ONE launch with no
//      dependency between workgroups (the serial chain of the factorisation is launches, not flops);
//   2. does the same lazy update for its own block  B = A[r,kb] - sum_j L[r,j] L[kb,j]^T, solves
//      L[r,kb] = B L11^-T (blocked by 16 columns, MFMA) and writes it in place.
// Nobody writes A[kb,kb] here: workgroup 0 parks L11 in the diagonal block of W, where k_potrf_finish
// later turns it into inv(L11) and copies L11 into K.  So every block a workgroup reads is either
// final (written by an earlier launch) or its own.
// Inside the 64x64 block both the factor and the solve are blocked by 16: see the two sections of the kernel.
// =====================================================================================
constexpr int PS = NB + 2;   // [row][k] LDS stride of the MFMA operand images (conflict-free ds_read_b64 fragments)

// f(integral_constant<int, 0>{}), f(integral_constant<int, 1>{}), ... : a loop the compiler cannot decline to unroll
template <int... I, class F>
__device__ __forceinline__ void unroll_ints(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// value of lane `src` (wave-uniform index) as a scalar
__device__ __forceinline__ double lane_value(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

#ifdef GPT_STEP_TRACE      // tools/probes/potrf_step_probe.hip: shader-clock stamps of workgroup 1's phases
#define GPT_TRACE_ARG , long long* trace
#define GPT_TRACE_NULL , nullptr
#define GPT_TRACE(i) do { if (trace && blockIdx.x == 1 && threadIdx.x == 0) trace[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GPT_TRACE_ARG
#define GPT_TRACE_NULL
#define GPT_TRACE(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_potrf_step(double* __restrict__ K, double* __restrict__ W, int NP, int kb, int p0,
                                                    int* __restrict__ info GPT_TRACE_ARG) {
    __shared__ __attribute__((aligned(16))) double colp[2][NB];
    __shared__ __attribute__((aligned(16))) double dinv[NB];
    __shared__ double Ib[3][16 * 17];          // inverses of the first three 16x16 diagonal blocks of L11, [n][k], stride 17
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    GPT_TRACE(0);
    double* Ab = smem;               // L[r,j] image [NB][PS]; later the block B / X, stride DS
    double* Bb = smem + NB * PS;     // L[kb,j] image [NB][PS]; later D, then L11, stride DS
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lc = lane & 15, lk = lane >> 4;
    const bool panel = b > 0;
    const size_t k0 = (size_t)kb * NB, r0 = (size_t)(kb + b) * NB;

    // ---- lazy update of D (tile row w) and of the own block, accumulators seeded with A ------------------
    // global loads run one block ahead of the MFMAs: [row t>>2][16 k from (t&3)*16] of L[kb,j] and L[r,j]
    const int rr = t >> 2, h = (t & 3) * 16;
    d2 pfB[8], pfA[8];
    auto fetch = [&](int j) {
        const d2* srcB = reinterpret_cast<const d2*>(K + (k0 + rr) * NP + (size_t)j * NB + h);
        const d2* srcA = reinterpret_cast<const d2*>(K + (r0 + rr) * NP + (size_t)j * NB + h);
#pragma unroll
        for (int u = 0; u < 8; ++u) pfB[u] = srcB[u];
        if (panel) {
#pragma unroll
            for (int u = 0; u < 8; ++u) pfA[u] = srcA[u];
        }
    };
    if (p0 < kb) fetch(p0);
    d4 accD[4], accB[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t i = 16 * w + lk + 4 * e, col = k0 + 16 * c + lc;
            accD[c][e] = K[(k0 + i) * NP + col];
            accB[c][e] = panel ? K[(r0 + i) * NP + col] : 0.0;
        }
    for (int j = p0; j < kb; ++j) {
        {
            d2* dstB = reinterpret_cast<d2*>(&Bb[rr * PS + h]);
            d2* dstA = reinterpret_cast<d2*>(&Ab[rr * PS + h]);
#pragma unroll
            for (int u = 0; u < 8; ++u) dstB[u] = pfB[u];
            if (panel) {
#pragma unroll
                for (int u = 0; u < 8; ++u) dstA[u] = pfA[u];
            }
        }
        __syncthreads();
        if (j + 1 < kb) fetch(j + 1);
        if (panel) {
#pragma unroll
            for (int s4 = 0; s4 < NB / 4; ++s4) {
                const double aD = -Bb[(16 * w + lc) * PS + 4 * s4 + lk];
                const double aB = -Ab[(16 * w + lc) * PS + 4 * s4 + lk];
                double bb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) bb[c] = Bb[(16 * c + lc) * PS + 4 * s4 + lk];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    accD[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aD, bb[c], accD[c], 0, 0, 0);
                    accB[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB, bb[c], accB[c], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int s4 = 0; s4 < NB / 4; ++s4) {
                const double aD = -Bb[(16 * w + lc) * PS + 4 * s4 + lk];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    accD[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aD, Bb[(16 * c + lc) * PS + 4 * s4 + lk], accD[c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 16 * w + lk + 4 * e, col = 16 * c + lc;
            Bb[i * DS + col] = accD[c][e];
            Ab[i * DS + col] = accB[c][e];
        }
    __syncthreads();
    GPT_TRACE(1);

    // ---- factor D, blocked: four panels of 16 columns.  A panel is factored by wave 0 alone (lane = row, the 16
    // columns in registers, pivots and multipliers by v_readlane: no barrier inside a panel), normalised and written
    // back to LDS; the rank-16 update of the rows and columns behind it is 4 MFMAs per 16x16 tile, shared by the four
    // waves.  8 barriers instead of one per column.
    int first_bad = 0;               // 1-based column of the first pivot <= 0 (or NaN); uniform over the workgroup and the grid
    auto factor_panel = [&](auto pc) {
        constexpr int p = decltype(pc)::value, c0 = 16 * p;
        if (w == 0) {
            double pa[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) pa[jj] = Bb[lane * DS + c0 + jj];
            double dmine = 1.0;                                  // lane j: pivot j
            // the current column goes through LDS (colp, two buffers): the wave reads pivot and multipliers back with
            // wave-uniform loads — only this wave uses LDS here and a wave's LDS operations execute in order, so no
            // barrier; cheaper than v_readlane pairs (FMAs then take plain register operands)
            colp[0][lane] = pa[0];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = c0 + jj;
                const double* col = colp[jj & 1];
                const double dj = col[j];
                double mk[16];
#pragma unroll
                for (int kk = 0; kk < 16; kk += 2)
                    if (kk + 1 > jj) {
                        const d2 v = *reinterpret_cast<const d2*>(&col[c0 + kk]);
                        mk[kk] = v[0]; mk[kk + 1] = v[1];
                    }
                const bool ok = dj > 0.0;                        // branch-free: a failed pivot is replaced by 1
                first_bad = (first_bad == 0 && !ok) ? j + 1 : first_bad;
                const double d = ok ? dj : 1.0;
                dmine = (lane == j) ? d : dmine;
                double r = __builtin_amdgcn_rcp(d);              // 1/d: v_rcp_f64 + two Newton steps (<= 2 ulp)
                r = fma(fma(-d, r, 1.0), r, r);
                r = fma(fma(-d, r, 1.0), r, r);
                const double f = (lane > j) ? pa[jj] * r : 0.0;
                if (jj + 1 < 16) {
                    pa[jj + 1] = fma(-f, mk[jj + 1], pa[jj + 1]);
                    colp[(jj + 1) & 1][lane] = pa[jj + 1];
                }
#pragma unroll
                for (int kk = jj + 2; kk < 16; ++kk) pa[kk] = fma(-f, mk[kk], pa[kk]);
            }
            // 1/sqrt(pivot) by v_rsq_f64 + two Newton steps, sqrt(pivot) = pivot * that + one correction (both <= 1 ulp off
            // the IEEE results; a tenth of their instruction count)
            double y = __builtin_amdgcn_rsq(dmine);
            y = fma(y * fma(-dmine * y, y, 1.0), 0.5, y);
            y = fma(y * fma(-dmine * y, y, 1.0), 0.5, y);
            double sq = dmine * y;
            sq = fma(fma(-sq, sq, dmine) * 0.5, y, sq);
            const double dsqv = sq, dinvv = y;
            if (lane >= c0 && lane < c0 + 16) dinv[lane] = dinvv;
#pragma unroll
            for (int jj = 0; jj < 16; jj += 2) {
                const d2 sj = *reinterpret_cast<const d2*>(&dinv[c0 + jj]);      // wave-uniform, in order behind the store above
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = c0 + jj + u;
                    Bb[lane * DS + j] = (lane > j) ? pa[jj + u] * sj[u] : ((lane == j) ? dsqv : 0.0);
                }
            }
        }
        __syncthreads();
        if constexpr (p < 3) {
            // A22 -= L21 L21^T on the lower 16x16 tiles (ti, tj), p < tj <= ti <= 3; tile n of the list goes to wave n % 4
            constexpr int nt = 3 - p;
#pragma unroll
            for (int n = 0; n < nt * (nt + 1) / 2; ++n) {
                if ((n & 3) != w) continue;
                int ti = 0, tj = n;
                while (tj > ti) { tj -= ti + 1; ++ti; }
                ti += p + 1; tj += p + 1;
                d4 acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = Bb[(16 * ti + lk + 4 * e) * DS + 16 * tj + lc];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bb[(16 * ti + lc) * DS + c0 + 4 * s4 + lk],
                                                               Bb[(16 * tj + lc) * DS + c0 + 4 * s4 + lk], acc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) Bb[(16 * ti + lk + 4 * e) * DS + 16 * tj + lc] = acc[e];
            }
            __syncthreads();
            // while wave 0 factors the next panel, wave p+1 inverts the finished diagonal block L_pp for the substitution
            // below (lane c: column c of the inverse by forward substitution; L entries by wave-uniform reads)
            if (w == p + 1) {
                const int c = lane & 15;
                double xi[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    double sp = (i == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < i; ++k) sp = fma(-Bb[(c0 + i) * DS + c0 + k], xi[k], sp);
                    xi[i] = (i < c) ? 0.0 : sp * dinv[c0 + i];
                }
                if (lane < 16) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) Ib[p][i * 17 + c] = xi[i];
                }
            }
        }
    };
    unroll_ints(std::make_integer_sequence<int, 4>{}, factor_panel);
    if (b == 0 && t == 0 && first_bad) atomicCAS(info, 0, (int)k0 + first_bad);
    GPT_TRACE(2);
    if (!panel) {                                                // parked in W until k_potrf_finish
        for (int e = t; e < NB * NB; e += 256) {
            const int r = e / NB, c = e % NB;
            W[(k0 + r) * NP + k0 + c] = Bb[r * DS + c];
        }
        return;
    }

    // ---- X = B L11^-T, blocked by 16 columns; wave w owns rows 16w .. 16w+15 of B and needs no barrier:
    //   X_p = (B_p - sum_{q<p} X_q L_pq^T) L_pp^-T.   The sum is 4 MFMAs per earlier block (X_q is read back from LDS, where
    // it replaced B_q); the 16x16 triangular solve is 4 more MFMAs with inv(L_pp) (computed by idle waves during the
    // factor) for p < 3 and a substitution for the last block.
    GPT_TRACE(3);
    {
        const int row = lane >> 2, qd = lane & 3;
        double* const Aw = Ab + (size_t)16 * w * DS;          // this wave's 16 rows
        double* const Kw = K + (r0 + 16 * w) * NP + k0;
        auto solve_block = [&](auto pc) {
            constexpr int p = decltype(pc)::value, c0 = 16 * p;
            if constexpr (p > 0) {
                d4 acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = Aw[(lk + 4 * e) * DS + c0 + lc];
#pragma unroll
                for (int k4 = 0; k4 < 4 * p; ++k4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Aw[lc * DS + 4 * k4 + lk], Bb[(c0 + lc) * DS + 4 * k4 + lk], acc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) Aw[(lk + 4 * e) * DS + c0 + lc] = acc[e];
            }
            if constexpr (p < 3) {
                // X_p = T inv(L_pp)^T: 4 MFMAs, T read back from LDS as the A operand
                d4 xa{0, 0, 0, 0};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
                    xa = __builtin_amdgcn_mfma_f64_16x16x4f64(Aw[lc * DS + c0 + 4 * k4 + lk], Ib[p][lc * 17 + 4 * k4 + lk], xa, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Aw[(lk + 4 * e) * DS + c0 + lc] = xa[e];                       // operand of the later blocks
                    Kw[(size_t)(lk + 4 * e) * NP + c0 + lc] = xa[e];
                }
            } else {
                // last block: substitution with 4 lanes per row — lane (row, qd) keeps x[row][qd + 4m], column jj's owner
                // scales its entry and the quad shares it (DPP)
                double x[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) x[m] = Aw[row * DS + c0 + qd + 4 * m];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int oq = jj & 3, om = jj >> 2;
                    double lcol[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (m >= om) lcol[m] = Bb[(c0 + qd + 4 * m) * DS + c0 + jj];
                    const double xo = x[om] * dinv[c0 + jj];
                    double xj;
                    switch (oq) {
                        case 0: xj = quad_bcast<0>(xo); break;
                        case 1: xj = quad_bcast<1>(xo); break;
                        case 2: xj = quad_bcast<2>(xo); break;
                        default: xj = quad_bcast<3>(xo); break;
                    }
                    if (qd <= oq) lcol[om] = 0.0;
                    x[om] = (qd == oq) ? xj : x[om];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (m >= om) x[m] = fma(-xj, lcol[m], x[m]);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) Kw[(size_t)row * NP + c0 + qd + 4 * m] = x[m];
            }
        };
        unroll_ints(std::make_integer_sequence<int, 4>{}, solve_block);
    }
    GPT_TRACE(4);
    GPT_TRACE(5);
}

// After the last step: for every diagonal block, move the parked L11 from W into K (zeros above the diagonal)
// and leave inv(L11) in W (the seed of trinv_levels).  One workgroup per block; blocks [b0, b0 + gridDim.x).
__global__ __launch_bounds__(256) void k_potrf_finish(double* __restrict__ K, double* __restrict__ W, int NP, int b0) {
    __shared__ double dinv[NB];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ls = smem;                      // L11, row-major [NB][DS]
    double* Xs = smem + NB * DS;            // L11^-1
    const int t = threadIdx.x;
    const size_t k0 = (size_t)(b0 + blockIdx.x) * NB;
    for (int e = t; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        const double v = (c <= r) ? W[(k0 + r) * NP + k0 + c] : 0.0;
        Ls[r * DS + c] = v;
        K[(k0 + r) * NP + k0 + c] = v;
    }
    __syncthreads();
    if (t < NB) dinv[t] = 1.0 / Ls[t * DS + t];
    __syncthreads();
    // lanes 4c..4c+3 own column c of X = L^-1; lane p keeps x[k], k = p (mod 4), in registers and sums those
    // terms of each row's dot product; the row result is shared by two lane exchanges
    {
        const int c = t >> 2, p = t & 3;
        double xr[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) xr[m] = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double sp = 0.0;
#pragma unroll
            for (int m = 0; m < (i + 3) / 4; ++m) {
                const int k = p + 4 * m;
                if (k < i) sp = fma(Ls[i * DS + k], xr[m], sp);       // x[k] = 0 for k < c
            }
            sp += __shfl_xor(sp, 1);
            sp += __shfl_xor(sp, 2);
            const double x = (i >= c) ? (((i == c) ? 1.0 : 0.0) - sp) * dinv[i] : 0.0;
            if (p == (i & 3)) xr[i >> 2] = x;
            if (p == 0) Xs[i * DS + c] = x;
        }
    }
    __syncthreads();
    for (int e = t; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        W[(k0 + r) * NP + k0 + c] = Xs[r * DS + c];
    }
}

// =====================================================================================
// Generic fp64 MFMA GEMM:  C = alpha * A * op(B) + beta * C,  one TS x TS output tile per workgroup
// (TS = 128 or 64; 4 waves as 2 x 2, (TS/2)^2 each), K consumed in LDS-staged chunks of 32.
//   BT = true : op(B) = B^T, B stored [n][k] row-major (SYRK / TRSM-by-inverse)
//   BT = false: op(B) = B,   B stored [k][n] row-major (triangular-inverse products)
//   AT = true : A is given transposed, A(i,k) = Amem[k*lda + i] (K^-1 = W^T W)
// All dims are multiples of 64; partial 128-tiles are guarded per 64x64 wave quadrant.
// LDS strides: [row][k] images use 34 doubles per row, [k][n] images TS + 16 — both make the
// 16x4 fragment reads (ds_read_b64, two 32-lane halves) bank-conflict free.
// Batched with element strides; the last batch entry may have its own M / K.
// TS = 64 is for problems with few tiles: a tile's K loop is a serial chain (a 128-tile with K = 2048 lives
// 0.5 ms however idle the chip is), so when there are not enough 128-tiles to fill 2 x 256 workgroup slots the
// quarter-size tiles cut that chain by four and fill the chip (the lower levels of the triangular inverse, the late
// trailing updates of the Cholesky, everything at N <= 4096).
// =====================================================================================
struct GemmArgs {
    const double* A; long lda; long sA;
    const double* B; long ldb; long sB;
    double* C; long ldc; long sC;
    int M, N, K;
    int M_last, K_last;      // dims of the last batch entry (trinv partial pair)
    int nbatch;
    double alpha, beta;
    int lower_only;          // skip tiles strictly above the block diagonal (SYRK)
    int a_lower;             // A lower triangular: k < i0 + TS
    int b_lower;             // B lower triangular (BT=false): k >= j0
    int bt_lower;            // B lower triangular, used transposed (BT=true, B stored [n][k]): k < j0 + TS
    int k_from_ij;           // both operands vanish for k < max(i0, j0) (W^T W with W lower triangular)
    int stagger;             // diagnostic: < 0 = rows of a folded triangle dealt round-robin over the XCDs (the round-2 order)
};

constexpr int GA_S = 34;     // [row][k] stride

// One MFMA operand from LDS as a plain ds_read_b64 that hipcc cannot fuse with its neighbour.  Left alone it merges the two
// reads of a k-step into ds_read2_b64 / ds_read2st64_b64, which the LDS serves in 16-lane groups over 32 banks: 8 cycles
// instead of 2 x 2, and 2-way conflicts on images laid out for the 64-bank ds_read_b64 (measured on the rank-256 trailing
// update: SQ_LDS_BANK_CONFLICT 31.6 M of 79 M LDS cycles, the LDS as busy as the matrix pipe; profiles/r03_fit_gemm_pmc.txt).
typedef __attribute__((address_space(3))) const double* lds_cptr;
__device__ __forceinline__ double lds_ld(const double* p) {
    unsigned off = (unsigned)(size_t)(lds_cptr)p;
    asm("" : "+v"(off));
    return *(lds_cptr)(size_t)off;
}

// Tile -> workgroup mapping.  Workgroups are dealt round-robin over the 8 XCDs (linear id mod 8) and each XCD has
// its own L2, so the 1-D grid is decoded as (xcd = id % 8, s = id / 8).  Tile rows of all batch entries are
// numbered R = batch * TM + ti and dealt to the XCDs in groups of 8, boustrophedon (0..7, 7..0, ...), so that
// triangular problems stay balanced between XCDs; an XCD then walks its G rows x TN columns in the order that
// starts the longest tiles first (greedy longest-processing-time: the K range of a tile shrinks with tj when B is
// triangular, grows with ti when A is, shrinks with ti for W^T W), and the ~64 workgroups it runs at a time share
// a few row and column panels in that XCD's L2.
#ifndef GPT_GEMM_ABL       // timing-only ablation builds of k_gemm (tools/probes/syrk_seq_probe.hip): 1 no operand loads, 2 no LDS stores,
#define GPT_GEMM_ABL 0     // 3 no C traffic, 4 no C store, 5 no MFMAs, 6 no barriers.  Results are wrong unless 0.
#endif
#ifdef GPT_GEMM_TRACE      // tools/probes/gemm_tile_trace.hip: shader-clock stamps of the phases of one tile (diagnostic builds only)
__device__ long long* g_gemm_trace = nullptr;
#define GPT_GT(i) do { if (g_gemm_trace && threadIdx.x == 0 && blockIdx.x < 16) { long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_gemm_trace[blockIdx.x * 64 + (i)] = t_; } } while (0)
#else
#define GPT_GT(i) do { } while (0)
#endif
// ---- tile -> workgroup decode shared by the tile bodies (see the comment above gemm_tile) -------------------------
struct TilePos { int b, ti, tj; bool ok; };
__device__ __forceinline__ TilePos gemm_decode(const GemmArgs& g, int TM, int TN, int G, int fold_tm, const int vid) {
    TilePos t{0, 0, 0, false};
    const int xcd = vid & 7, sidx = vid >> 3;
    int gi;
    if (g.a_lower)        { gi = G - 1 - sidx / TN; t.tj = sidx % TN; }     // long rows first
    else if (g.k_from_ij) { gi = sidx / TN;         t.tj = sidx % TN; }     // short offsets first
    else if (g.bt_lower)  { t.tj = TN - 1 - sidx / G; gi = sidx % G; }      // column by column, long columns first
    else                  { t.tj = sidx / G;        gi = sidx % G; }        // column by column
    int R = 8 * gi + ((gi & 1) ? 7 - xcd : xcd);
    if (fold_tm && g.stagger >= 0 && !g.k_from_ij) {
        // folded triangle: every row of the fold has the same number of tiles, so an XCD can take CONTIGUOUS rows
        // [xcd TM / 8, (xcd + 1) TM / 8) and stay balanced.  Its G rows then share their column panels: in the first part of
        // the walk (tj <= ti) all rows read column panel tj, in the second part (tile (T-1-u, tj-u-1)) a window of G
        // consecutive panels that slides by one per step — where rows dealt round-robin over the XCDs each stream a
        // diagonal of their own (36 % of the panel requests of a rank-256 update missed L2: profiles/r03_fit_gemm_pmc.txt).
        const int lo = xcd * TM / 8, hi = (xcd + 1) * TM / 8;
        R = lo + gi;
        if (R >= hi) return t;
    }
    t.b = R / TM;
    t.ti = R - t.b * TM;
    if (t.b >= g.nbatch) return t;
    if (fold_tm) {
        const int u = t.ti, v = t.tj;
        if (v <= u) { t.ti = u; t.tj = v; }
        else {
            t.ti = fold_tm - 1 - u; t.tj = v - u - 1;
            if (t.ti == u) return t;
        }
    }
    t.ok = true;
    return t;
}

template <bool BT, bool AT, int TS>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int TM, int TN, int G, int fold_tm, const int vid, double* smem) {
    constexpr int WS = TS / 2;              // wave tile edge
    constexpr int RT = WS / 16;             // MFMA tiles per wave tile edge
    constexpr int GB_S = TS + 16;           // [k][n] stride
    constexpr int NP_ = TS / 16;            // staging passes per operand and chunk (one 16-byte load per thread each)
    double* As = smem;                                       // [TS][GA_S]   (AT: [32][GB_S])
    double* Bs = smem + (AT ? 32 * GB_S : TS * GA_S);        // BT: [TS][GA_S]   else: [32][GB_S]
    const TilePos tp = gemm_decode(g, TM, TN, G, fold_tm, vid);
    if (!tp.ok) return;
    const int b = tp.b, ti = tp.ti, tj = tp.tj;
    const bool last = (b == g.nbatch - 1);
    const int M = last ? g.M_last : g.M;
    const int K = last ? g.K_last : g.K;
    const int N = g.N;
    const int i0 = ti * TS, j0 = tj * TS;
    if (i0 >= M || j0 >= N) return;
    if (g.lower_only && j0 > i0) return;
    const double* A = g.A + (size_t)b * g.sA;
    const double* B = g.B + (size_t)b * g.sB;
    double* C = g.C + (size_t)b * g.sC;
    int kbeg = 0, kend = K;
    if (g.a_lower) kend = min(K, i0 + TS);
    if (g.b_lower) kbeg = j0;
    if (g.bt_lower) kend = min(K, j0 + TS);
    if (g.k_from_ij) kbeg = i0 > j0 ? i0 : j0;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int lc = lane & 15, lk = lane >> 4;
    const bool active = (i0 + WS * wr < M) && (j0 + WS * wc < N) && !(g.lower_only && i0 == j0 && wc > wr);
    // beta != 0: the accumulators start from (beta/alpha) C, so the read of C flies under the first chunk's
    // staging instead of sitting in the epilogue (the rank-k updates of the Cholesky have only 4-8 chunks per tile)
    d4 acc[RT][RT];
    const double cscale = g.beta != 0.0 ? g.beta / g.alpha : 0.0;
    double* const ctile = C + (size_t)(i0 + WS * wr + lk) * g.ldc + j0 + WS * wc + lc;     // this lane's first element
    if (active && cscale != 0.0 && GPT_GEMM_ABL != 3) {          // one uniform branch, then all the loads in flight together
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < RT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][c][e] = ctile[(size_t)(16 * r + 4 * e) * g.ldc + 16 * c];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < RT; ++c) acc[r][c] *= cscale;
    } else {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < RT; ++c) acc[r][c] = d4{0, 0, 0, 0};
    }

    // global -> registers one chunk ahead of the MFMAs (the loads of chunk k+1 fly under the MFMAs of chunk k),
    // registers -> LDS between two barriers.  Every wave-level load covers whole 128-B lines:
    //   [row][k] sources (A, and B when BT): 16 lanes x 16 B along k, 4 rows per instruction;
    //   [k][n] sources (B when !BT, A when AT): TS/2 lanes x 16 B along n/i per k row.
    const int rk_r = t >> 4, rk_k = (t & 15) * 2;                 // [row][k] staging: row rk_r + 16u, k offset rk_k
    constexpr int KN_L = TS / 2, KN_R = 256 / KN_L;               // [k][n] staging: lanes per k row, k rows per pass
    const int kn_k = t / KN_L, kn_n = (t % KN_L) * 2;             //                  k row kn_k + KN_R u, n offset kn_n
    // uniform (scalar) base per instruction + one per-lane 32-bit offset per operand keeps the address registers few
    const int offA = AT ? kn_k * (int)g.lda + kn_n : rk_r * (int)g.lda + rk_k;
    const int offB = BT ? rk_r * (int)g.ldb + rk_k : kn_k * (int)g.ldb + kn_n;
    const bool okA_n = (i0 + kn_n) < M, okB_n = (j0 + kn_n) < N;     // all dims are multiples of 64
    d2 pa[NP_], pb[NP_];
    auto fetch = [&](int kc) {
        if (GPT_GEMM_ABL == 1) return;               // timing-only ablation: no operand loads
#pragma unroll
        for (int u = 0; u < NP_; ++u) {
            if (!AT) {
                const double* base = A + (size_t)(i0 + 16 * u) * g.lda + kc;
                pa[u] = (i0 + 16 * u) < M ? *reinterpret_cast<const d2*>(base + offA) : d2{0, 0};
            } else {
                const double* base = A + (size_t)(kc + KN_R * u) * g.lda + i0;
                pa[u] = okA_n ? *reinterpret_cast<const d2*>(base + offA) : d2{0, 0};
            }
            if (BT) {
                const double* base = B + (size_t)(j0 + 16 * u) * g.ldb + kc;
                pb[u] = (j0 + 16 * u) < N ? *reinterpret_cast<const d2*>(base + offB) : d2{0, 0};
            } else {
                const double* base = B + (size_t)(kc + KN_R * u) * g.ldb + j0;
                pb[u] = okB_n ? *reinterpret_cast<const d2*>(base + offB) : d2{0, 0};
            }
        }
    };
    auto stage = [&]() {
        if (GPT_GEMM_ABL == 2) { asm volatile("" :: "v"(pa[0]), "v"(pb[0]), "v"(pa[NP_ - 1]), "v"(pb[NP_ - 1])); return; }   // no LDS stores
#pragma unroll
        for (int u = 0; u < NP_; ++u) {
            if (!AT) *reinterpret_cast<d2*>(&As[(rk_r + 16 * u) * GA_S + rk_k]) = pa[u];
            else     *reinterpret_cast<d2*>(&As[(kn_k + KN_R * u) * GB_S + kn_n]) = pa[u];
            if (BT)  *reinterpret_cast<d2*>(&Bs[(rk_r + 16 * u) * GA_S + rk_k]) = pb[u];
            else     *reinterpret_cast<d2*>(&Bs[(kn_k + KN_R * u) * GB_S + kn_n]) = pb[u];
        }
    };
    GPT_GT(0);
    if (kbeg < kend) fetch(kbeg);
    GPT_GT(1);
    for (int kc = kbeg; kc < kend; kc += 32) {
        stage();
        GPT_GT(2 + 3 * ((kc - kbeg) / 32));
        if (GPT_GEMM_ABL != 6) __syncthreads();
        if (kc + 32 < kend) fetch(kc + 32);
        GPT_GT(3 + 3 * ((kc - kbeg) / 32));
        if (active) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                double a[RT], bb[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    a[r] = lds_ld(AT ? &As[(4 * s + lk) * GB_S + WS * wr + 16 * r + lc]
                                     : &As[(WS * wr + 16 * r + lc) * GA_S + 4 * s + lk]);
#pragma unroll
                for (int c = 0; c < RT; ++c)
                    bb[c] = lds_ld(BT ? &Bs[(WS * wc + 16 * c + lc) * GA_S + 4 * s + lk]
                                      : &Bs[(4 * s + lk) * GB_S + WS * wc + 16 * c + lc]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < RT; ++c) {
                        if (GPT_GEMM_ABL == 5) { asm volatile("" :: "v"(a[r]), "v"(bb[c])); continue; }      // no MFMAs
                        acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], bb[c], acc[r][c], 0, 0, 0);
                    }
            }
        }
        GPT_GT(4 + 3 * ((kc - kbeg) / 32));
        if (GPT_GEMM_ABL != 6) __syncthreads();
    }
    GPT_GT(60);
    if (!active) return;
    if (GPT_GEMM_ABL == 3 || GPT_GEMM_ABL == 4) {          // no C traffic at all (3) / no store (4): keep the accumulators alive
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < RT; ++c) asm volatile("" :: "v"(acc[r][c]));
        return;
    }
    const double alpha = g.alpha;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < RT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) ctile[(size_t)(16 * r + 4 * e) * g.ldc + 16 * c] = alpha * acc[r][c][e];
    GPT_GT(61);
}

template <bool BT, bool AT, int TS>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g, int TM, int TN, int G, int fold_tm) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gemm_tile<BT, AT, TS>(g, TM, TN, G, fold_tm, blockIdx.x, smem);
}

struct GemmGrid { int TM, TN, G, fold_tm, nvid; };
static GemmGrid gemm_grid(const GemmArgs& g, int TS) {
    GemmGrid q{};
    // rows to cover: a single batch entry IS the last one.  (Sizing the grid by M there — the top level of the triangular
    // inverse at N = 2500 is 512 x 2048 with M = 2048 — left 3 of 4 workgroups empty, and since workgroups are dealt to the
    // CUs in order, the live ones sat four to a CU on a quarter of the chip: 179 us for that product.  A second chunk of
    // operand prefetch, tried first on the theory that the tile's K loop was latency-bound, changed nothing.)
    const int Mmax = g.nbatch == 1 ? g.M_last : (g.M > g.M_last ? g.M : g.M_last);
    q.TM = (Mmax + TS - 1) / TS; q.TN = (g.N + TS - 1) / TS;
    static const bool nofold = [] { const char* e = getenv("GPT_GEMM_NOFOLD"); return e && atoi(e) != 0; }();
    if (!nofold && g.lower_only && g.nbatch == 1 && q.TM == q.TN && q.TM > 1) { q.fold_tm = q.TM; q.TM = (q.fold_tm + 1) / 2; q.TN = q.fold_tm + 1; }
    q.G = (g.nbatch * q.TM + 7) / 8;            // tile rows (over all batch entries) per XCD
    q.nvid = 8 * q.G * q.TN;
    return q;
}

// Tile body: 0 = the shipped one (k_gemm); probe builds (-DGPT_GEMM_VARIANTS) also know 1 = two LDS buffers + register
// staging, 2 = LDS-DMA ring, 3 = two chunks of loads in flight (tools/probes/gemm_bodies.h)
static int gemm_body() {
#ifdef GPT_GEMM_VARIANTS
    static const int v = [] { const char* e = getenv("GPT_GEMM_BODY"); return e ? atoi(e) : 0; }();
    return v;
#else
    return 0;
#endif
}

#ifdef GPT_GEMM_VARIANTS      // probe builds: the alternative tile bodies that were measured and not shipped
#include "../../tools/probes/gemm_bodies.h"
#endif

template <bool BT, bool AT, int TS>
static void launch_gemm_ts(hipStream_t s, const GemmArgs& g) {
    const GemmGrid q = gemm_grid(g, TS);
    constexpr size_t lds = (size_t)((AT ? 32 * (TS + 16) : TS * GA_S) + (BT ? TS * GA_S : 32 * (TS + 16))) * sizeof(double);
#ifdef GPT_GEMM_VARIANTS
    if constexpr (TS != 32) {
        if (gemm_body() != 0) { launch_gemm_variant<BT, AT, TS>(s, g, q, lds); return; }
    }
#endif
    static PerDeviceOnce once;
    once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<BT, AT, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((k_gemm<BT, AT, TS>), dim3(q.nvid), dim3(256), lds, s, g, q.TM, q.TN, q.G, q.fold_tm);
}

// 64-tiles below this many 128-tiles (measured: equal within noise from 2500 up, tools/gpu_fit_ab.sh; a 4096^3 product runs at 64.6 vs 66.3 TFLOP/s)
static int gemm_tile_threshold() {
    static const int thr = [] {                      // (initialised once, thread-safe: handles may be driven from several threads)
        const char* e = getenv("GPT_GEMM_TS64_BELOW");
        return e ? atoi(e) : 2500;
    }();
    return thr;
}

// 32-tiles below this many 64-tiles: a tile's K loop is a serial chain on ONE CU (a 64-tile of K = 256 lasts 17 us however
// idle the chip is), so a product with fewer 64-tiles than CUs is bound by that chain; quarter-size tiles cut it by four and
// put the other CUs to work (the thin updates on the Cholesky's chain, the late trailing updates, everything at N <= 2500)
static int gemm_tile32_threshold() {
    static const int thr = [] {
        const char* e = getenv("GPT_GEMM_TS32_BELOW");
        return e ? atoi(e) : 1024;      // measured: N = 2500 fit 1.52 -> 1.34 ms, objective 1.85 -> 1.60 ms; 512 .. 2048 equal within noise
    }();
    return thr;
}

template <bool BT, bool AT = false>
static void launch_gemm(hipStream_t s, const GemmArgs& g) {
    const int Mmax = g.nbatch == 1 ? g.M_last : (g.M > g.M_last ? g.M : g.M_last);
    double tiles = (double)g.nbatch * ((Mmax + 127) / 128) * ((g.N + 127) / 128);
    if (g.lower_only) tiles *= 0.5;
    // (round 4 tried 64-tiles for every product with K >= 1024 — on the theory that a deep 64-tile is 25 us of work and a few hundred
    // fill the chip: the N = 2500 inverse went 0.23 -> 0.32 ms and its K^-1 = W^T W 149 -> 209 us.  Few tiles, whatever their depth,
    // want the small tile: profiles/r04_fit_summary.txt)
    const double thr32 = (double)gemm_tile32_threshold();
    if (4.0 * tiles < thr32 && gemm_body() == 0) launch_gemm_ts<BT, AT, 32>(s, g);
    else if (tiles < gemm_tile_threshold()) launch_gemm_ts<BT, AT, 64>(s, g);
    else launch_gemm_ts<BT, AT, 128>(s, g);
}

// =====================================================================================
// Blocked Cholesky (lower), in place in K; inv(L_kk) of every diagonal block is left in W.
// =====================================================================================
// Two-level blocking: panels of OB = `ob_blocks` x NB columns are factored by k_potrf_step (one launch per NB
// columns, left-looking inside the panel); the trailing matrix then gets ONE rank-OB update per panel (MFMA GEMM on
// the folded block lower triangle).  Everything runs in the caller's stream: running the update beside the next
// panel's steps was tried twice (a second stream with events; update tiles riding in the step launches) and lost
// both times — profiles/r01_fit_cholesky_ab.log.
// (potrf_outer_blocks / potrf_group — panels of OB columns, `grp` panels per trailing update — live in gpt_fit_plan.h: the plan
// needs them for the split of form 1)
// Trailing update of the Cholesky: A[row0.., row0..row0+ncols) -= P P^T on the block lower triangle, P = A[row0.., kcol0..kcol0+kw)
// (rows [row0, row_end): the whole matrix behind row0, or the diagonal block a leaf of the recursive form is confined to)
static void syrk_update(hipStream_t s, double* K, int NP, int row0, int ncols, int kcol0, int kw, int row_end = -1) {
    const int rem = (row_end < 0 ? NP : row_end) - row0;
    if (rem <= 0 || kw <= 0) return;
    GemmArgs c{};
    c.A = K + (size_t)row0 * NP + kcol0; c.lda = NP;
    c.B = c.A; c.ldb = NP;
    c.C = K + (size_t)row0 * NP + row0; c.ldc = NP;
    c.M = c.M_last = rem; c.N = ncols < rem ? ncols : rem; c.K = c.K_last = kw; c.nbatch = 1;
    c.alpha = -1.0; c.beta = 1.0; c.lower_only = 1;
    static const int stagger = [] { const char* e = getenv("GPT_GEMM_FOLD_INTERLEAVED"); return e && atoi(e) ? -1 : 0; }();
    c.stagger = stagger;
    launch_gemm<true>(s, c);
}

// row_blk_end: block rows [.., row_blk_end) take part (NP / NB for the whole matrix; blk_end for a leaf of the recursive form)
static void potrf_groups(hipStream_t s, double* K, double* W, int NP, int* info, int blk_begin, int blk_end, int row_blk_end, int grp_override = 0) {
    const int nb = row_blk_end;
    const int ob = potrf_outer_blocks();
    constexpr size_t step_lds = (size_t)(2 * NB * PS) * sizeof(double);
    constexpr size_t fin_lds = (size_t)(2 * NB * DS) * sizeof(double);
    static PerDeviceOnce once;
    once.run([&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_finish), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds);
    });
    // Third level: panels are grouped by `grp`.  Inside a group a panel's own columns receive the group's earlier panels
    // just before its steps (a thin GEMM, K = up to (grp-1) panels); everything behind the group is updated once per
    // group with K = grp panels — the read-modify-write of the trailing matrix, which bounds the rank-128 update
    // (42 TFLOP/s against 50-55 at rank 256-512), happens grp times less often.
    // (measured: groups of 2 take 3 % off the Cholesky at N = 8192 and add 3-5 % at N <= 2500, where the extra thin GEMM
    // on the chain costs more than the trailing matrix's traffic; tools/gpu_fit_ab.sh with GPT_POTRF_GROUP)
    const int grp = grp_override > 0 ? grp_override : potrf_group(NP);
    auto syrk = [&](int row0, int ncols, int kcol0, int kw) { syrk_update(s, K, NP, row0, ncols, kcol0, kw, nb * NB); };
    const int gw = grp * ob;                                        // blocks per group
    for (int g0 = blk_begin; g0 < blk_end; g0 += gw) {
        const int gend = g0 + gw < nb ? g0 + gw : nb;
        for (int p0 = g0; p0 < gend; p0 += ob) {
            const int pend = p0 + ob < gend ? p0 + ob : gend;
            if (p0 > g0) syrk(p0 * NB, (pend - p0) * NB, g0 * NB, (p0 - g0) * NB);   // this panel's columns <- earlier panels of the group
            for (int kb = p0; kb < pend; ++kb)
                hipLaunchKernelGGL(k_potrf_step, dim3(nb - kb), dim3(256), step_lds, s, K, W, NP, kb, p0, info GPT_TRACE_NULL);
        }
        syrk(gend * NB, nb * NB - gend * NB, g0 * NB, (gend - g0) * NB);             // everything behind the group
    }
}

// diagonal blocks [b0, b1): parked L_kk from W into K, inv(L_kk) into W (the seeds of the triangular inverse)
static void potrf_finish(hipStream_t s, double* K, double* W, int NP, int b0, int b1) {
    constexpr size_t fin_lds = (size_t)(2 * NB * DS) * sizeof(double);
    if (b1 > b0) hipLaunchKernelGGL(k_potrf_finish, dim3(b1 - b0), dim3(256), fin_lds, s, K, W, NP, b0);
}

// =====================================================================================
// W = L^-1 by recursive doubling.  The diagonal NB blocks of W already hold inv(L_kk).
// Level s (s = NB, 2NB, ...): for each pair of adjacent s-blocks
//     T   = L21 * W11          (W11 lower triangular: k >= j0)
//     W21 = -W22 * T           (W22 lower triangular: k <  i0 + 128)
// T lives in `scratch` (>= n*n/4 doubles).  Inverts the diagonal sub-block [off, off + n) of a matrix with leading dimension NP.
// =====================================================================================
static void trinv_levels(hipStream_t s, const double* L, double* W, int NP, int off, int n, double* scratch) {
    L += (size_t)off * NP + off;
    W += (size_t)off * NP + off;
    for (long sz = NB; sz < n; sz *= 2) {
        const int npairs = (int)((n + 2 * sz - 1) / (2 * sz));
        // last pair: rows available for the second block
        const long r0_last = (long)(npairs - 1) * 2 * sz;
        long m_last = n - r0_last - sz;
        int nbp = npairs;
        if (m_last <= 0) { nbp = npairs - 1; m_last = sz; }
        if (nbp <= 0) continue;
        if (m_last > sz) m_last = sz;
        const long pair_stride = 2 * sz * (long)NP + 2 * sz;
        GemmArgs a{};
        a.A = L + sz * (long)NP; a.lda = NP; a.sA = pair_stride;            // L21
        a.B = W; a.ldb = NP; a.sB = pair_stride;                            // W11
        a.C = scratch; a.ldc = sz; a.sC = sz * sz;                          // T
        a.M = (int)sz; a.M_last = (int)m_last; a.N = (int)sz; a.K = a.K_last = (int)sz; a.nbatch = nbp;
        a.alpha = 1.0; a.beta = 0.0; a.b_lower = 1;
        launch_gemm<false>(s, a);
        GemmArgs c{};
        c.A = W + sz * (long)NP + sz; c.lda = NP; c.sA = pair_stride;       // W22
        c.B = scratch; c.ldb = sz; c.sB = sz * sz;                          // T
        c.C = W + sz * (long)NP; c.ldc = NP; c.sC = pair_stride;            // W21
        c.M = (int)sz; c.M_last = (int)m_last; c.N = (int)sz; c.K = (int)sz; c.K_last = (int)m_last; c.nbatch = nbp;
        c.alpha = -1.0; c.beta = 0.0; c.a_lower = 1;
        launch_gemm<false>(s, c);
    }
}

void fit_aux_release(FitAux& aux) {
    for (auto& e : aux.events) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (aux.side) (void)hipStreamDestroy(aux.side);
    if (aux.chain) (void)hipStreamDestroy(aux.chain);
    aux.side = aux.chain = nullptr;
    aux.tried = aux.ok = false;
}

// The streams of the blocked form (gpt_fit_plan.h): `side` on the first `eighths`/8 of the CUs (look-ahead updates and the
// inverse), `chain` on the others (the leaves' launch chains).  Bit i of a CU mask is CU i in the driver's numbering, dealt
// round-robin over the XCDs, so both sets span all 8 XCDs (7/8: every XCD keeps 4 CUs for the chain).
// tools/probes/cumask_probe.hip: a launch chain on one mask and a bulk kernel on the other run side by side without delaying
// each other, which plain streams do not (a k_potrf_step queues behind a GEMM's grid: 19 -> 68-83 us per step, r04_fit_summary.txt).
static bool fit_aux_init(FitAux& aux, int n_events, int eighths) {
    if (n_events > FIT_MAX_EVENTS) return false;
    if (aux.tried && (!aux.ok || aux.side_eighths == eighths)) return aux.ok;
    if (aux.tried) fit_aux_release(aux);             // a handle that changes size class (or form) gets new streams
    aux.tried = true;
    aux.side_eighths = eighths;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, current_device()) != hipSuccess) return false;
    const int ncu = prop.multiProcessorCount;
    const int n_side = ncu * eighths / 8;
    if (n_side < 16 || ncu - n_side < 8) return false;
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> ms(words, 0), mc(words, 0);
    for (int i = 0; i < ncu; ++i) (i < n_side ? ms : mc)[i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(&aux.side, words, ms.data()) != hipSuccess) { aux.side = nullptr; (void)hipGetLastError(); return false; }
    if (hipExtStreamCreateWithCUMask(&aux.chain, words, mc.data()) != hipSuccess) {
        (void)hipStreamDestroy(aux.side); aux.side = aux.chain = nullptr; (void)hipGetLastError();
        return false;
    }
    bool ok = true;
    for (auto& e : aux.events) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    aux.ok = ok;
    return ok;
}

size_t factor_scratch_doubles(int NP) { return factor_scratch_doubles_of(NP); }

// dst[r][c] = src[r][c], r < rows, c < cols (cols a multiple of 2): the bounce buffer of L21 back into K
__global__ __launch_bounds__(256) void k_copy2d(const double* __restrict__ src, long lds_, double* __restrict__ dst, long ldd, int rows, int cols) {
    const int c2 = cols / 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)rows * c2; e += (long)gridDim.x * 256) {
        const long r = e / c2, c = (e % c2) * 2;
        *reinterpret_cast<d2*>(dst + r * ldd + c) = *reinterpret_cast<const d2*>(src + r * lds_ + c);
    }
}

static_assert(NB == FIT_NB && FIT_AUX_EVENTS == FIT_MAX_EVENTS, "event pool of a handle = what a plan may ask for");
// Executes the plan of gpt_fit_plan.h (see there for the forms, the streams and the events).
void launch_factor_inverse(hipStream_t s, double* K, double* W, int NP, int* info, double* scratch, FitAux* aux, hipEvent_t ev_factored) {
    FitPlan pl = fit_plan(NP);
    bool multi = pl.multi_stream();
    if (multi && !(aux && fit_aux_init(*aux, pl.n_events, pl.side_eighths))) { pl = fit_plan(NP, 0); multi = false; }     // no masked streams: one leaf
    for (const FitOp& op : pl.ops) {
        const int off = op.off, b = op.n1, r = op.n2;
        hipStream_t st = !multi ? s : (op.stream == FS_SIDE ? aux->side : (op.stream == FS_CHAIN ? aux->chain : s));
        if (multi)
            for (int e : op.wait) if (e >= 0) hipStreamWaitEvent(st, aux->events[e], 0);
        double* const Wpp = W + (size_t)off * NP + off;
        switch (op.kind) {
        case FOP_POTRF:
            potrf_groups(st, K, W, NP, info, off / NB, (off + b) / NB, op.row_end / NB, op.grp);
            break;
        case FOP_FINISH:
            potrf_finish(st, K, W, NP, off / NB, (off + b) / NB);
            break;
        case FOP_TRINV:
            trinv_levels(st, K, W, NP, off, b, scratch + op.r0);
            break;
        case FOP_FACTORED:
            if (ev_factored) hipEventRecord(ev_factored, st);
            break;
        case FOP_UPDATE: {
            GemmArgs c{};
            c.A = K + (size_t)off * NP + op.k0; c.lda = NP;              // L[off:, k0:k0+kw]
            c.B = c.A; c.ldb = NP;                                        // its first b rows, used transposed
            c.C = K + (size_t)off * NP + off; c.ldc = NP;
            c.M = c.M_last = NP - off; c.N = b; c.K = c.K_last = op.kw; c.nbatch = 1;
            c.alpha = -1.0; c.beta = 1.0; c.lower_only = 1;
            launch_gemm<true>(st, c);
        } break;
        case FOP_TRSM: {
            GemmArgs a{};
            a.A = K + (size_t)(off + b) * NP + off; a.lda = NP;           // A21 (r x b), up to date
            a.B = Wpp; a.ldb = NP;                                        // W_pp (b x b, lower), used transposed
            a.C = scratch + op.r0; a.ldc = b;
            a.M = a.M_last = r; a.N = b; a.K = a.K_last = b; a.nbatch = 1;
            a.alpha = 1.0; a.beta = 0.0; a.bt_lower = 1;
            launch_gemm<true>(st, a);
        } break;
        case FOP_COPY_L21: {
            const long tot = (long)r * b / 2;
            const unsigned grid = (unsigned)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
            hipLaunchKernelGGL(k_copy2d, dim3(grid), dim3(256), 0, st, scratch + op.r0, (long)b, K + (size_t)(off + b) * NP + off, (long)NP, r, b);
        } break;
        case FOP_T: {
            GemmArgs a{};
            a.A = K + (size_t)off * NP; a.lda = NP;                       // L[off:off+b, 0:off]
            a.B = W; a.ldb = NP;                                          // W[0:off, 0:off] (lower)
            a.C = scratch + op.r1; a.ldc = off;
            a.M = a.M_last = b; a.N = off; a.K = a.K_last = off; a.nbatch = 1;
            a.alpha = 1.0; a.beta = 0.0; a.b_lower = 1;
            launch_gemm<false>(st, a);
        } break;
        case FOP_WFIN: {
            GemmArgs c{};
            c.A = Wpp; c.lda = NP;                                        // W_pp (lower)
            c.B = scratch + op.r1; c.ldb = off;                           // T
            c.C = W + (size_t)off * NP; c.ldc = NP;                       // W[off:off+b, 0:off]
            c.M = c.M_last = b; c.N = off; c.K = c.K_last = b; c.nbatch = 1;
            c.alpha = -1.0; c.beta = 0.0; c.a_lower = 1;
            launch_gemm<false>(st, c);
        } break;
        }
        if (multi && op.record >= 0) hipEventRecord(aux->events[op.record], st);
    }
}

// =====================================================================================
// alpha = K^-1 Y = W^T (W Y).   Y4 / tmp4 / A4 are [NP][4] (up to 4 outputs per pass).
// fwd: one wave per row (coalesced row of W).  bwd: workgroups on 64-column x 512-row pieces, lanes on columns
// (coalesced 512-B row segments), the 4 waves split the rows and reduce through LDS; `scratch` holds the
// (NP/512) x NP x 4 partial sums.
// =====================================================================================
__global__ __launch_bounds__(256) void k_alpha_fwd(const double* __restrict__ W, const double* __restrict__ Y4, int NP,
                                                   double* __restrict__ T4) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NP) return;
    const double* wrow = W + (size_t)i * NP;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int j = lane; j <= i; j += 64) {
        const double wv = wrow[j];
        const d4 y = *reinterpret_cast<const d4*>(Y4 + (size_t)j * 4);
        s0 += wv * y[0]; s1 += wv * y[1]; s2 += wv * y[2]; s3 += wv * y[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o);
    }
    if (lane == 0) *reinterpret_cast<d4*>(T4 + (size_t)i * 4) = d4{s0, s1, s2, s3};
}

// bwd, first pass: workgroup (strip, chunk) sums rows [chunk*ALPHA_ROWS, +ALPHA_ROWS) of a 64-column strip into
// part[chunk][j][4]; second pass adds the chunks in order (deterministic; no atomics).
constexpr int ALPHA_ROWS = 512;

__global__ __launch_bounds__(256) void k_alpha_bwd(const double* __restrict__ W, const double* __restrict__ T4, int NP,
                                                   double* __restrict__ part) {
    __shared__ double red[4][64][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * ALPHA_ROWS;
    if (i0 + ALPHA_ROWS <= j0) return;                  // strictly above the diagonal: zeros (never read back either)
    const int j = j0 + lane;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const int ibeg = (i0 > j0 ? i0 : j0) + w;
#pragma unroll 4
    for (int i = ibeg; i < i0 + ALPHA_ROWS; i += 4) {
        const double wv = W[(size_t)i * NP + j];        // zero above the diagonal
        const d4 tv = *reinterpret_cast<const d4*>(T4 + (size_t)i * 4);
        s0 += wv * tv[0]; s1 += wv * tv[1]; s2 += wv * tv[2]; s3 += wv * tv[3];
    }
    red[w][lane][0] = s0; red[w][lane][1] = s1; red[w][lane][2] = s2; red[w][lane][3] = s3;
    __syncthreads();
    if (w == 0) {
        d4 r;
#pragma unroll
        for (int o = 0; o < 4; ++o) r[o] = red[0][lane][o] + red[1][lane][o] + red[2][lane][o] + red[3][lane][o];
        *reinterpret_cast<d4*>(part + ((size_t)blockIdx.y * NP + j) * 4) = r;
    }
}

__global__ __launch_bounds__(256) void k_alpha_sum(const double* __restrict__ part, int N, int NP, double* __restrict__ A4) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= NP) return;
    d4 r{0, 0, 0, 0};
    for (int ch = j / ALPHA_ROWS; ch < NP / ALPHA_ROWS; ++ch) r += *reinterpret_cast<const d4*>(part + ((size_t)ch * NP + j) * 4);
    if (j >= N) r = d4{0, 0, 0, 0};
    *reinterpret_cast<d4*>(A4 + (size_t)j * 4) = r;
}

void launch_alpha(hipStream_t s, const double* W, const double* Y4, int N, int NP, double* tmp4, double* A4, double* scratch) {
    hipLaunchKernelGGL(k_alpha_fwd, dim3(NP / 4), dim3(256), 0, s, W, Y4, NP, tmp4);
    hipLaunchKernelGGL(k_alpha_bwd, dim3(NP / 64, NP / ALPHA_ROWS), dim3(256), 0, s, W, tmp4, NP, scratch);
    hipLaunchKernelGGL(k_alpha_sum, dim3((NP + 255) / 256), dim3(256), 0, s, scratch, N, NP, A4);
}

// =====================================================================================
// Pack W (row-major, lower) into the fragment-ordered tile stream Wf.
// Tile (ib, kb), kb <= ib, index ib(ib+1)/2 + kb, WT x WT doubles each, laid out
//   [k4 = 0..WT/4)[g = 0..8)[q = 0..2)[lane = 0..64)[p = 0..2)
// holding W[ib*WT + 64*g + 16*(2q+p) + (lane&15)][kb*WT + 4*k4 + (lane>>4)]  — i.e. for row group g
// (the 64 rows one wave owns) the A operands of v_mfma_f64_16x16x4_f64 for its four 16-row tiles,
// 16 B per lane and 2 KiB contiguous per wave and k4-step.
// Rows/cols >= N (padding) are zeroed so padded sources never reach a variance.
// One workgroup transposes a 64-row x 128-col sub-block through LDS (coalesced on both sides).
// =====================================================================================
size_t wf_elems(int NP) {
    const size_t nb = NP / WT;
    return nb * (nb + 1) / 2 * WT_TILE_DOUBLES;
}
size_t wf_overrun_elems() { return WT_STEP_DOUBLES; }     // one k4-step of prefetch overrun behind the last tile set

constexpr int PKR = 64;          // sub-block: one 64-row group ...
constexpr int PKC = 128;         // ... x 128 columns (32 k4-steps); 65 KB of LDS, two workgroups per CU
constexpr int PK_S = PKC + 1;

// TO = double: [k4][g][q 0..2)[lane][p 0..2) (row tile 2q + p);  TO = float: [k4][g][lane][e 0..4) (row tile e).
template <typename TO>
__global__ __launch_bounds__(256) void k_pack_w(const double* __restrict__ W, int N, int NP, TO* __restrict__ Wf, double scale) {
    extern __shared__ __attribute__((aligned(16))) double tile[];   // [PKR][PK_S]
    constexpr int SUBR = WT / PKR, SUBC = WT / PKC;                  // sub-blocks per tile edge (8 x 4)
    const int ib = blockIdx.y / SUBR, g = blockIdx.y % SUBR;        // g: row group of the tile
    const int kb = blockIdx.x / SUBC, kc = blockIdx.x % SUBC;
    if (kb > ib) return;
    const int t = threadIdx.x;
    const int r0 = ib * WT + g * PKR, c0 = kb * WT + kc * PKC;
    // thread -> row t>>2 ... (4 passes of 16 rows), 32 consecutive columns of that row in 16-byte loads
    for (int e = t; e < PKR * PKC / 2; e += 256) {
        const int r = e / (PKC / 2), c = (e % (PKC / 2)) * 2;
        const int gr = r0 + r, gc = c0 + c;
        d2 v{0.0, 0.0};
        if (gr < N && gc <= gr) {                       // gc even: the pair (gc, gc+1) straddles the diagonal at most at gc+1
            v = *reinterpret_cast<const d2*>(W + (size_t)gr * NP + gc);
            if (gc + 1 > gr || gc + 1 >= N) v[1] = 0.0;
            if (gc >= N) v[0] = 0.0;
        }
        tile[r * PK_S + c] = v[0] * scale;
        tile[r * PK_S + c + 1] = v[1] * scale;
    }
    __syncthreads();
    TO* out = Wf + ((size_t)ib * (ib + 1) / 2 + kb) * WT_TILE_DOUBLES;
    // this sub-block covers k4 in [32 kc, 32 kc + 32) of row group g
    if (std::is_same<TO, double>::value) {
        d2* o2 = reinterpret_cast<d2*>(out);
        for (int e = t; e < (PKC / 4) * 2 * 64; e += 256) {
            const int lane = e & 63, q = (e >> 6) & 1, k4l = e >> 7;
            const int lc = lane & 15, lk = lane >> 4;
            const int col = 4 * k4l + lk;
            const int rowb = 16 * (2 * q) + lc;
            const double v0 = tile[rowb * PK_S + col];
            const double v1 = tile[(rowb + 16) * PK_S + col];
            const int k4 = (PKC / 4) * kc + k4l;
            o2[((size_t)(k4 * WT_GROUPS + g) * 2 + q) * 64 + lane] = d2{v0, v1};
        }
    } else {
        f4* o4 = reinterpret_cast<f4*>(out);
        for (int e = t; e < (PKC / 4) * 64; e += 256) {
            const int lane = e & 63, k4l = e >> 6;
            const int lc = lane & 15, lk = lane >> 4;
            const int col = 4 * k4l + lk;
            const int k4 = (PKC / 4) * kc + k4l;
            o4[(size_t)(k4 * WT_GROUPS + g) * 64 + lane] =
                f4{(float)tile[lc * PK_S + col], (float)tile[(16 + lc) * PK_S + col], (float)tile[(32 + lc) * PK_S + col],
                   (float)tile[(48 + lc) * PK_S + col]};
        }
    }
}

void launch_pack_w(hipStream_t s, const double* W, int N, int NP, void* Wf, int dtype, int task, double scale) {
    const int nbt = NP / WT;
    const size_t lds = (size_t)PKR * PK_S * sizeof(double);
    static PerDeviceOnce once;
    once.run([&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_pack_w<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_pack_w<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    const dim3 grid(nbt * (WT / PKC), nbt * (WT / PKR));
    const size_t off = (size_t)task * wf_elems(NP);
    if (dtype == DT_F32) hipLaunchKernelGGL(k_pack_w<float>, grid, dim3(256), lds, s, W, N, NP, static_cast<float*>(Wf) + off, scale);
    else hipLaunchKernelGGL(k_pack_w<double>, grid, dim3(256), lds, s, W, N, NP, static_cast<double*>(Wf) + off, scale);
}

// dst[r][dst_col0 + c] = src[r][src_col0 + c] * scale for c < ncol: moves alpha columns from the fp64 fit workspace into
// the model blob's A4 image (element type of the model; the SVGP model puts task t's single column at column t,
// times the task's outputscale).
template <typename TO>
__global__ __launch_bounds__(256) void k_store4(const double* __restrict__ src4, int rows, TO* __restrict__ dst4, int src_col0,
                                                int dst_col0, int ncol, double scale) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    for (int c = 0; c < ncol; ++c) dst4[(size_t)r * 4 + dst_col0 + c] = (TO)(src4[(size_t)r * 4 + src_col0 + c] * scale);
}

void launch_store4(hipStream_t s, const double* src4, int rows, void* dst4, int dtype, int src_col0, int dst_col0, int ncol, double scale) {
    const dim3 grid((rows + 255) / 256);
    if (dtype == DT_F32) hipLaunchKernelGGL(k_store4<float>, grid, dim3(256), 0, s, src4, rows, static_cast<float*>(dst4), src_col0, dst_col0, ncol, scale);
    else hipLaunchKernelGGL(k_store4<double>, grid, dim3(256), 0, s, src4, rows, static_cast<double*>(dst4), src_col0, dst_col0, ncol, scale);
}

// =====================================================================================
// Gradient of the log-marginal likelihood (sklearn/_gpr.py:625-648):
//   grad_p = 0.5 * sum_ij ( sum_o alpha_io alpha_jo - O * Kinv_ij ) * dK_ij/dtheta_p ,  theta = log params.
// launch_kinv: Kinv = W^T W (lower block triangle, MFMA GEMM) into `Kout`.
// k_lml_terms: one workgroup per 64x64 lower tile recomputes the RBF part of K from the scaled sources
// and accumulates  S[0] = sum inner*Krbf (d/dlog c),  S[1+d] = sum inner*Krbf*(xs_id-xs_jd)^2 (d/dlog l_d),
// and sum_i inner_ii (d/dlog noise, times noise on the host); off-diagonal elements count twice.
// Per-workgroup partials are summed in a fixed order by k_sum_partials (deterministic).
// =====================================================================================
void launch_kinv(hipStream_t s, const double* W, int NP, double* Kout) {
    GemmArgs g{};
    g.A = W; g.lda = NP; g.B = W; g.ldb = NP; g.C = Kout; g.ldc = NP;
    g.M = g.M_last = NP; g.N = NP; g.K = g.K_last = NP; g.nbatch = 1;
    g.alpha = 1.0; g.beta = 0.0; g.lower_only = 1; g.k_from_ij = 1;
    launch_gemm<false, true>(s, g);
}

// Partial sums per workgroup: [0] d/dlog c, [1 + d] d/dlog l_d, [1 + DW] the noise term; DW = 3 / 8 / 16 as in k_gram.
template <int DW>
__global__ __launch_bounds__(256) void k_lml_terms(const double* __restrict__ Xs, const double* __restrict__ A4, int npass,
                                                   const double* __restrict__ Kinv, int N, int NP, int O, int ktype, double c,
                                                   double* __restrict__ partial /* [blocks][LML_PARTIAL_STRIDE] */) {
    constexpr int XS = DW == 3 ? 4 : DW;
    constexpr int NS = DW + 2;
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int t = threadIdx.x;
    double S[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) S[e] = 0.0;
    if (bj <= bi) {
        const int r = t >> 2, cs = (t & 3) * 16;
        const int i = bi * 64 + r;
        if (i < N) {
            double xi[DW];
#pragma unroll
            for (int d = 0; d < DW; ++d) xi[d] = Xs[(size_t)i * XS + d];
            for (int u = 0; u < 16; ++u) {
                const int j = bj * 64 + cs + u;
                if (j > i || j >= N) continue;
                double aa = 0.0;
                for (int ps = 0; ps < npass; ++ps) {
                    const d4 ai = *reinterpret_cast<const d4*>(A4 + ((size_t)ps * NP + i) * 4);
                    const d4 aj = *reinterpret_cast<const d4*>(A4 + ((size_t)ps * NP + j) * 4);
                    aa += ai[0] * aj[0] + ai[1] * aj[1] + ai[2] * aj[2] + ai[3] * aj[3];
                }
                const double inner = aa - (double)O * Kinv[(size_t)i * NP + j];
                const double wgt = (i == j) ? 1.0 : 2.0;
                double df[DW], r2 = 0.0;
#pragma unroll
                for (int d = 0; d < DW; ++d) { df[d] = xi[d] - Xs[(size_t)j * XS + d]; r2 += df[d] * df[d]; }
                const double kr = kernel_libm(ktype, c, r2);
                // dK/dlog l_d = gk * (xs_id - xs_jd)^2  (kernels.py: RBF 1568-1580, Matern 1747-1778)
                double gk;
                if (ktype == KT_RBF) gk = kr;
                else {
                    const double r = sqrt(r2);
                    if (ktype == KT_MATERN12) gk = (r > 0.0) ? kr / r : 0.0;
                    else if (ktype == KT_MATERN32) gk = 3.0 * c * exp(-1.7320508075688772 * r);
                    else { const double t = 2.23606797749979 * r; gk = (5.0 / 3.0) * c * (t + 1.0) * exp(-t); }
                }
                const double wi = wgt * inner;
                S[0] += wi * kr;
                const double v = wi * gk;
#pragma unroll
                for (int d = 0; d < DW; ++d) S[1 + d] += v * df[d] * df[d];
                if (i == j) S[1 + DW] += inner;
            }
        }
    }
    __shared__ double red[256][NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) red[t][e] = S[e];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o)
#pragma unroll
            for (int e = 0; e < NS; ++e) red[t][e] += red[t + o][e];
        __syncthreads();
    }
    if (t < NS) partial[((size_t)bi * gridDim.x + bj) * LML_PARTIAL_STRIDE + t] = red[0][t];
}

// out[0] = d/dlog c, out[1 + d] (d < MAX_D) = d/dlog l_d, out[1 + MAX_D] = the noise term; per-workgroup partials added
// in a fixed order
template <int DW>
__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
    constexpr int NS = DW + 2;
    __shared__ double red[256][NS];
    double S[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) S[e] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int e = 0; e < NS; ++e) S[e] += partial[(size_t)b * LML_PARTIAL_STRIDE + e];
#pragma unroll
    for (int e = 0; e < NS; ++e) red[threadIdx.x][e] = S[e];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int e = 0; e < NS; ++e) red[threadIdx.x][e] += red[threadIdx.x + o][e];
        __syncthreads();
    }
    if (threadIdx.x < LML_TERMS) {
        const int e = threadIdx.x;
        double v = 0.0;
        if (e == 0) v = red[0][0];
        else if (e == LML_TERMS - 1) v = red[0][1 + DW];
        else if (e - 1 < DW) v = red[0][e];
        out[e] = v;
    }
}

// partial: (NP/64)^2 * LML_PARTIAL_STRIDE doubles of scratch; out: LML_TERMS doubles
void launch_lml_terms(hipStream_t s, const double* Xs, int D, const double* A4, int npass, const double* Kinv, int N, int NP,
                      int O, int ktype, double c, double* partial, double* out) {
    const int nb = NP / 64;
    if (D <= 3) {
        hipLaunchKernelGGL(k_lml_terms<3>, dim3(nb, nb), dim3(256), 0, s, Xs, A4, npass, Kinv, N, NP, O, ktype, c, partial);
        hipLaunchKernelGGL(k_sum_partials<3>, dim3(1), dim3(256), 0, s, partial, nb * nb, out);
    } else if (D <= WIDE_D) {
        hipLaunchKernelGGL(k_lml_terms<WIDE_D>, dim3(nb, nb), dim3(256), 0, s, Xs, A4, npass, Kinv, N, NP, O, ktype, c, partial);
        hipLaunchKernelGGL(k_sum_partials<WIDE_D>, dim3(1), dim3(256), 0, s, partial, nb * nb, out);
    } else {
        hipLaunchKernelGGL(k_lml_terms<MAX_D>, dim3(nb, nb), dim3(256), 0, s, Xs, A4, npass, Kinv, N, NP, O, ktype, c, partial);
        hipLaunchKernelGGL(k_sum_partials<MAX_D>, dim3(1), dim3(256), 0, s, partial, nb * nb, out);
    }
}

// =====================================================================================
// Posterior covariance  cov = k(Xq,Xq) + noise*I - V^T V,  V = W K*^T  (sklearn/_gpr.py:458-468,
// where V = L \ K*^T).  Small-M path (sampling, return_cov): K*^T and V are materialised (NP x Mp).
// =====================================================================================
__global__ __launch_bounds__(256) void k_cross_t(const double* __restrict__ Xs, const double* __restrict__ Xq, int N, int NP,
                                                 int64_t M, int Mp, int D, int XS, int ktype, double c, InvLs il,
                                                 double* __restrict__ KsT /* [NP][Mp] */) {
    const int t = threadIdx.x;
    const int n = blockIdx.y * 64 + (t >> 2);
    const int m0 = blockIdx.x * 64 + (t & 3) * 16;
    double x[MAX_D];
#pragma unroll
    for (int d = 0; d < MAX_D; ++d) x[d] = (d < D) ? Xs[(size_t)n * XS + d] : 0.0;
    for (int u = 0; u < 16; ++u) {
        const int m = m0 + u;
        double v = 0.0;
        if (n < N && m < M) {
            double r2 = 0.0;
#pragma unroll
            for (int d = 0; d < MAX_D; ++d)
                if (d < D) { const double df = x[d] - Xq[(size_t)m * D + d] * il.v[d]; r2 += df * df; }
            v = kernel_libm(ktype, c, r2);
        }
        KsT[(size_t)n * Mp + m] = v;
    }
}

__global__ __launch_bounds__(256) void k_cov_finish(const double* __restrict__ Xq, int64_t M, int Mp, int D, int ktype, double c, double noise,
                                                    InvLs il, const double* __restrict__ VtV, double* __restrict__ cov /* [M][M] */) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * M) return;
    const int64_t i = e / M, j = e % M;
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < MAX_D; ++d)
        if (d < D) { const double df = (Xq[i * D + d] - Xq[j * D + d]) * il.v[d]; d2 += df * df; }
    double v = (i == j) ? (c + noise) : kernel_libm(ktype, c, d2);      // k(0) = 1 exactly (kernels.py:1562)
    cov[e] = v - VtV[(size_t)i * Mp + j];
}

// scratch: KsT (NP*Mp), V (NP*Mp), VtV (Mp*Mp) doubles
void launch_cov(hipStream_t s, const KernelParams& p, const double* Xs, const double* W, const double* Xq_dev, int64_t M,
                int Mp, double* KsT, double* V, double* VtV, double* cov_dev) {
    const int NP = p.NP;
    InvLs il;
    for (int d = 0; d < MAX_D; ++d) il.v[d] = p.inv_ls[d];
    hipLaunchKernelGGL(k_cross_t, dim3(Mp / 64, NP / 64), dim3(256), 0, s, Xs, Xq_dev, p.N, NP, M, Mp, p.D, xs_stride(p.D), p.ktype, p.c, il, KsT);
    GemmArgs a{};
    a.A = W; a.lda = NP; a.B = KsT; a.ldb = Mp; a.C = V; a.ldc = Mp;
    a.M = a.M_last = NP; a.N = Mp; a.K = a.K_last = NP; a.nbatch = 1;
    a.alpha = 1.0; a.beta = 0.0; a.a_lower = 1;
    launch_gemm<false>(s, a);
    GemmArgs b{};
    b.A = V; b.lda = Mp; b.B = V; b.ldb = Mp; b.C = VtV; b.ldc = Mp;
    b.M = b.M_last = Mp; b.N = Mp; b.K = b.K_last = NP; b.nbatch = 1;
    b.alpha = 1.0; b.beta = 0.0;
    launch_gemm<false, true>(s, b);
    const int64_t tot = M * M;
    hipLaunchKernelGGL(k_cov_finish, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Xq_dev, M, Mp, p.D, p.ktype, p.c, p.noise,
                       il, VtV, cov_dev);
}

// sum(log(diag(L))) over the first N rows (LML, sklearn/_gpr.py:603).  One workgroup.
__global__ __launch_bounds__(256) void k_logdet(const double* __restrict__ K, int N, int NP, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < N; i += 256) s += log(K[(size_t)i * NP + i]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

void launch_logdet(hipStream_t s, const double* K, int N, int NP, double* out) {
    hipLaunchKernelGGL(k_logdet, dim3(1), dim3(256), 0, s, K, N, NP, out);
}

}  // namespace gpt
