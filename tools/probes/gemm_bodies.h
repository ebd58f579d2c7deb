// Alternative bodies of the fp64 MFMA GEMM tile, measured against the shipped one-buffer body (gpt_fit.hip: gemm_tile) in
// round 3 and NOT shipped: none of them was faster on the rank-256 trailing update of the Cholesky
// (profiles/r03_fit_gemm_study.txt).  Compiled into the probes only (-DGPT_GEMM_VARIANTS; selected at run time with
// GPT_GEMM_BODY = 1 two LDS buffers + register staging, 2 LDS-DMA ring, 3 two chunks of loads in flight), never into
// libgpt_hip.so.  Included from inside namespace gpt, after gemm_decode / lds_ld / GemmArgs.
#pragma once

// =====================================================================================
// The one-buffer tile body with TWO chunks of operand loads in flight (two register sets, chunk c + 2 requested as soon as
// chunk c has been staged).  Why: the trailing updates are bound by bytes in flight per CU, not by LDS, L2 hit rate or
// matrix-pipe issue (profiles/r03_fit_gemm_pmc.txt: a workgroup's loads take 3-4 us to come back once every CU is
// streaming, a 32-deep chunk of MFMAs lasts 0.85 us; three bodies with 64, 96 and 109 KB of loads in flight per CU ran at
// 34, 44 and 47 TFLOP/s).  A second chunk in flight doubles that without touching the LDS budget.
// =====================================================================================
// NCH > 0: the chunk count is a compile-time constant and the chunk loop is straight-line code, so hipcc counts its vmcnt waits
// exactly (with a loop it merges the two register sets' pending loads at the loop header and waits for BOTH sets before
// staging the older one, which halves the second chunk's time in flight).  NCH = 0: any chunk count, loop form.
template <bool BT, bool AT, int TS, int NCH>
__device__ __forceinline__ void gemm_tile_deep(const GemmArgs& g, int TM, int TN, int G, int fold_tm, const int vid, double* smem) {
    constexpr int WS = TS / 2, RT = WS / 16, GB_S = TS + 16, NP_ = TS / 16;
    double* As = smem;
    double* Bs = smem + (AT ? 32 * GB_S : TS * GA_S);
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
    if (g.k_from_ij) kbeg = i0 > j0 ? i0 : j0;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int lc = lane & 15, lk = lane >> 4;
    const bool active = (i0 + WS * wr < M) && (j0 + WS * wc < N) && !(g.lower_only && i0 == j0 && wc > wr);
    double* const ctile = C + (size_t)(i0 + WS * wr + lk) * g.ldc + j0 + WS * wc + lc;
    const bool use_c = active && g.beta != 0.0;
    const double cscale = g.beta != 0.0 ? g.beta / g.alpha : 0.0;
    const int rk_r = t >> 4, rk_k = (t & 15) * 2;
    constexpr int KN_L = TS / 2, KN_R = 256 / KN_L;
    const int kn_k = t / KN_L, kn_n = (t % KN_L) * 2;
    const int offA = AT ? kn_k * (int)g.lda + kn_n : rk_r * (int)g.lda + rk_k;
    const int offB = BT ? rk_r * (int)g.ldb + rk_k : kn_k * (int)g.ldb + kn_n;
    const bool fullA = TS == 64 || (i0 + TS) <= M, fullB = TS == 64 || (j0 + TS) <= N;
    d2 pa0[NP_], pb0[NP_], pa1[NP_], pb1[NP_];
    auto fetch = [&](d2 (&pa)[NP_], d2 (&pb)[NP_], int kc) {
#pragma unroll
        for (int u = 0; u < NP_; ++u) {
            if (!AT) {
                const double* base = A + (size_t)(i0 + 16 * u) * g.lda + kc;
                pa[u] = (TS == 64 || fullA || 16 * u < TS / 2) ? *reinterpret_cast<const d2*>(base + offA) : d2{0, 0};
            } else {
                const double* base = A + (size_t)(kc + KN_R * u) * g.lda + i0;
                pa[u] = (TS == 64 || fullA || kn_n < TS / 2) ? *reinterpret_cast<const d2*>(base + offA) : d2{0, 0};
            }
            if (BT) {
                const double* base = B + (size_t)(j0 + 16 * u) * g.ldb + kc;
                pb[u] = (TS == 64 || fullB || 16 * u < TS / 2) ? *reinterpret_cast<const d2*>(base + offB) : d2{0, 0};
            } else {
                const double* base = B + (size_t)(kc + KN_R * u) * g.ldb + j0;
                pb[u] = (TS == 64 || fullB || kn_n < TS / 2) ? *reinterpret_cast<const d2*>(base + offB) : d2{0, 0};
            }
        }
    };
    auto stage = [&](const d2 (&pa)[NP_], const d2 (&pb)[NP_]) {
#pragma unroll
        for (int u = 0; u < NP_; ++u) {
            if (!AT) *reinterpret_cast<d2*>(&As[(rk_r + 16 * u) * GA_S + rk_k]) = pa[u];
            else     *reinterpret_cast<d2*>(&As[(kn_k + KN_R * u) * GB_S + kn_n]) = pa[u];
            if (BT)  *reinterpret_cast<d2*>(&Bs[(rk_r + 16 * u) * GA_S + rk_k]) = pb[u];
            else     *reinterpret_cast<d2*>(&Bs[(kn_k + KN_R * u) * GB_S + kn_n]) = pb[u];
        }
    };
    d4 acc[RT][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < RT; ++c) acc[r][c] = d4{0, 0, 0, 0};
    auto mfmas = [&]() {
        if (!active) return;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            double a[RT], bb[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r)
                a[r] = lds_ld(AT ? &As[(4 * s_ + lk) * GB_S + WS * wr + 16 * r + lc] : &As[(WS * wr + 16 * r + lc) * GA_S + 4 * s_ + lk]);
#pragma unroll
            for (int q = 0; q < RT; ++q)
                bb[q] = lds_ld(BT ? &Bs[(WS * wc + 16 * q + lc) * GA_S + 4 * s_ + lk] : &Bs[(4 * s_ + lk) * GB_S + WS * wc + 16 * q + lc]);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < RT; ++q)
                    acc[r][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], bb[q], acc[r][q], 0, 0, 0);
        }
    };
    const int nch = NCH > 0 ? NCH : (kend - kbeg + 31) / 32;
    // the two first chunks, then C (its latency under theirs); C is scaled in front of the first MFMA
    if (nch > 0) fetch(pa0, pb0, kbeg);
    if (nch > 1) fetch(pa1, pb1, kbeg + 32);
    if (use_c) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < RT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][c][e] = ctile[(size_t)(16 * r + 4 * e) * g.ldc + 16 * c];
    }
    auto scale_c = [&]() {
        if (!use_c) return;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < RT; ++q) acc[r][q] *= cscale;
    };
    if constexpr (NCH > 0) {
        unroll_ints(std::make_integer_sequence<int, (NCH + 1) / 2>{}, [&](auto hc) {
            constexpr int c = 2 * decltype(hc)::value;
            stage(pa0, pb0);
            __syncthreads();
            if constexpr (c + 2 < NCH) fetch(pa0, pb0, kbeg + 32 * (c + 2));
            if constexpr (c == 0) scale_c();
            mfmas();
            __syncthreads();
            if constexpr (c + 1 < NCH) {
                stage(pa1, pb1);
                __syncthreads();
                if constexpr (c + 3 < NCH) fetch(pa1, pb1, kbeg + 32 * (c + 3));
                mfmas();
                __syncthreads();
            }
        });
    } else {
        bool first = true;
        for (int c = 0; c < nch; c += 2) {
            stage(pa0, pb0);
            __syncthreads();
            if (c + 2 < nch) fetch(pa0, pb0, kbeg + 32 * (c + 2));
            if (first) scale_c();
            first = false;
            mfmas();
            __syncthreads();
            if (c + 1 >= nch) break;
            stage(pa1, pb1);
            __syncthreads();
            if (c + 3 < nch) fetch(pa1, pb1, kbeg + 32 * (c + 3));
            mfmas();
            __syncthreads();
        }
    }
    if (!active) return;
    if (nch <= 0 && use_c) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < RT; ++q) acc[r][q] *= cscale;
    }
    const double alpha = g.alpha;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < RT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) ctile[(size_t)(16 * r + 4 * e) * g.ldc + 16 * c] = alpha * acc[r][c][e];
}

template <bool BT, bool AT, int TS, int NCH>
__global__ __launch_bounds__(256, 2) void k_gemm_deep(GemmArgs g, int TM, int TN, int G, int fold_tm) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gemm_tile_deep<BT, AT, TS, NCH>(g, TM, TN, G, fold_tm, blockIdx.x, smem);
}

// =====================================================================================
// The same GEMM tile, software-pipelined: TWO LDS buffers and ONE barrier per 32-deep chunk.  While the MFMAs of chunk c
// run from buffer c & 1, the registers holding chunk c + 1 (requested one chunk earlier) are written to the other
// buffer and the global loads of chunk c + 2 are issued, all in the shadow of the 64-cycle MFMAs; MFMA operands are read
// from LDS one k-step ahead of their use; C (beta != 0) is requested during the last chunk and combined in the epilogue,
// so neither its latency nor the first chunk's sits in front of the first MFMA.
// Why (tools/probes/gemm_tile_trace.hip, shader-clock stamps of a lone tile, K = 256): the one-buffer body above spends
// 930 cycles staging + 1100-1500 cycles barrier / load issue + 2600 cycles of MFMAs per chunk, strictly one after the
// other, and waits for C before it even requests the first chunk: 43 k cycles (18 us) per tile where the MFMAs need 16 k.
// With several workgroups per CU those phases overlap between workgroups, with few tiles (the late trailing updates,
// the thin updates on the chain, everything at N <= 2500) they do not.
// =====================================================================================
template <bool BT, bool AT, int TS>
__device__ __forceinline__ void gemm_tile_pipe(const GemmArgs& g, int TM, int TN, int G, int fold_tm, const int vid, double* smem) {
    constexpr int WS = TS / 2, RT = WS / 16, GB_S = TS + 16, NP_ = TS / 16;
    constexpr int A_DBL = AT ? 32 * GB_S : TS * GA_S, B_DBL = BT ? TS * GA_S : 32 * GB_S, BUF = A_DBL + B_DBL;
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
    if (g.k_from_ij) kbeg = i0 > j0 ? i0 : j0;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int lc = lane & 15, lk = lane >> 4;
    const bool active = (i0 + WS * wr < M) && (j0 + WS * wc < N) && !(g.lower_only && i0 == j0 && wc > wr);
    double* const ctile = C + (size_t)(i0 + WS * wr + lk) * g.ldc + j0 + WS * wc + lc;     // this lane's first element
    const bool use_c = active && g.beta != 0.0;
    const double cscale = g.beta != 0.0 ? g.beta / g.alpha : 0.0;

    const int rk_r = t >> 4, rk_k = (t & 15) * 2;
    constexpr int KN_L = TS / 2, KN_R = 256 / KN_L;
    const int kn_k = t / KN_L, kn_n = (t % KN_L) * 2;
    const int offA = AT ? kn_k * (int)g.lda + kn_n : rk_r * (int)g.lda + rk_k;
    const int offB = BT ? rk_r * (int)g.ldb + rk_k : kn_k * (int)g.ldb + kn_n;
    // rows / columns of the tile beyond M / N (dims are multiples of 64: only the second half of a 128-tile can be)
    // (a 64-tile is always whole: no conditions around its loads — a condition, even a uniform one, makes hipcc branch
    // around every single load)
    const bool fullA = TS == 64 || (i0 + TS) <= M, fullB = TS == 64 || (j0 + TS) <= N;
    d2 pa[NP_], pb[NP_];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int u = 0; u < NP_; ++u) {
            if (!AT) {
                const double* base = A + (size_t)(i0 + 16 * u) * g.lda + kc;
                pa[u] = (TS == 64 || fullA || 16 * u < TS / 2) ? *reinterpret_cast<const d2*>(base + offA) : d2{0, 0};
            } else {
                const double* base = A + (size_t)(kc + KN_R * u) * g.lda + i0;
                pa[u] = (TS == 64 || fullA || kn_n < TS / 2) ? *reinterpret_cast<const d2*>(base + offA) : d2{0, 0};
            }
            if (BT) {
                const double* base = B + (size_t)(j0 + 16 * u) * g.ldb + kc;
                pb[u] = (TS == 64 || fullB || 16 * u < TS / 2) ? *reinterpret_cast<const d2*>(base + offB) : d2{0, 0};
            } else {
                const double* base = B + (size_t)(kc + KN_R * u) * g.ldb + j0;
                pb[u] = (TS == 64 || fullB || kn_n < TS / 2) ? *reinterpret_cast<const d2*>(base + offB) : d2{0, 0};
            }
        }
    };
    auto stage = [&](double* As, double* Bs) {
#pragma unroll
        for (int u = 0; u < NP_; ++u) {
            if (!AT) *reinterpret_cast<d2*>(&As[(rk_r + 16 * u) * GA_S + rk_k]) = pa[u];
            else     *reinterpret_cast<d2*>(&As[(kn_k + KN_R * u) * GB_S + kn_n]) = pa[u];
            if (BT)  *reinterpret_cast<d2*>(&Bs[(rk_r + 16 * u) * GA_S + rk_k]) = pb[u];
            else     *reinterpret_cast<d2*>(&Bs[(kn_k + KN_R * u) * GB_S + kn_n]) = pb[u];
        }
    };
    d4 acc[RT][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < RT; ++c) acc[r][c] = d4{0, 0, 0, 0};
    // beta != 0: the accumulators start from (beta / alpha) C.  C is requested right behind the first chunk, so the two
    // latencies overlap, and is waited for only in front of the first MFMA (vmcnt counts in issue order: staging the first
    // chunk does not wait for C, nor does scaling C wait for the second chunk's loads)
    auto load_c = [&]() {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < RT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][c][e] = ctile[(size_t)(16 * r + 4 * e) * g.ldc + 16 * c];
    };
    auto scale_c = [&]() {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < RT; ++c) acc[r][c] *= cscale;
    };
    const int nch = (kend - kbeg + 31) / 32;
    if (nch <= 0) {
        if (use_c) { load_c(); scale_c(); }
    } else {
        GPT_GT(0);
        fetch(kbeg);
        __builtin_amdgcn_sched_barrier(0);
        if (use_c) load_c();
        __builtin_amdgcn_sched_barrier(0);
        stage(smem, smem + A_DBL);
        if (nch > 1) fetch(kbeg + 32);
        __builtin_amdgcn_sched_barrier(0);
        GPT_GT(1);
        __syncthreads();
        if (use_c) scale_c();
        GPT_GT(2);
        for (int c = 0; c < nch; ++c) {
            const double* As = smem + (c & 1) * BUF;
            const double* Bs = As + A_DBL;
            double* Asn = smem + ((c + 1) & 1) * BUF;
            double* Bsn = Asn + A_DBL;
            const bool more1 = c + 1 < nch, more2 = c + 2 < nch;
            if (active) {
                auto ldfrag = [&](const int s_, double (&a)[RT], double (&bb)[RT]) {
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        a[r] = lds_ld(AT ? &As[(4 * s_ + lk) * GB_S + WS * wr + 16 * r + lc]
                                         : &As[(WS * wr + 16 * r + lc) * GA_S + 4 * s_ + lk]);
#pragma unroll
                    for (int q = 0; q < RT; ++q)
                        bb[q] = lds_ld(BT ? &Bs[(WS * wc + 16 * q + lc) * GA_S + 4 * s_ + lk]
                                          : &Bs[(4 * s_ + lk) * GB_S + WS * wc + 16 * q + lc]);
                };
                double fa[2][RT], fb[2][RT];
                ldfrag(0, fa[0], fb[0]);
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) {
                    if (s_ + 1 < 8) ldfrag(s_ + 1, fa[(s_ + 1) & 1], fb[(s_ + 1) & 1]);
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int q = 0; q < RT; ++q)
                            acc[r][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s_ & 1][r], fb[s_ & 1][q], acc[r][q], 0, 0, 0);
                    if (s_ == 1) GPT_GT(3 + 3 * c);
                    if (s_ == 1) {
                        // chunk c + 1: registers -> the other buffer (free since the barrier that closed chunk c - 1); then the
                        // registers are free for chunk c + 2.  Program order pinned: the stores and loads sit behind MFMAs that
                        // are already issued, and the loads have six k-steps plus a barrier before their data is needed.
                        __builtin_amdgcn_sched_barrier(0);
                        if (more1) stage(Asn, Bsn);
                        if (more2) fetch(kbeg + 32 * (c + 2));
                        __builtin_amdgcn_sched_barrier(0);
                        GPT_GT(4 + 3 * c);
                    }
                }
            } else {
                if (more1) stage(Asn, Bsn);
                if (more2) fetch(kbeg + 32 * (c + 2));
            }
            GPT_GT(5 + 3 * c);
            __syncthreads();
        }
    }
    GPT_GT(60);
    if (!active) return;
    const double alpha = g.alpha;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < RT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) ctile[(size_t)(16 * r + 4 * e) * g.ldc + 16 * c] = alpha * acc[r][c][e];
    GPT_GT(61);
}

// (a 128-tile's two buffers take 139-143 KiB of LDS: one workgroup per CU, so it may as well have the registers of one)
template <bool BT, bool AT, int TS>
__global__ __launch_bounds__(256, TS == 64 ? 2 : 1) void k_gemm_pipe(GemmArgs g, int TM, int TN, int G, int fold_tm) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gemm_tile_pipe<BT, AT, TS>(g, TM, TN, G, fold_tm, blockIdx.x, smem);
}

// =====================================================================================
// The GEMM tile with the operands moved global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write pass, the copy engine fills LDS while the waves issue MFMAs) through a ring of NBUF buffers of KC-deep chunks:
// chunk c + NBUF - 1 is requested at the top of iteration c, a counted s_waitcnt vmcnt leaves the younger chunks in flight
// across the ONE raw s_barrier per chunk (guide section 5, "Pipelining across barriers": __syncthreads() would drain
// them), MFMA operands are read from LDS one k-step ahead of their use.
// An LDS-DMA instruction writes 64 lanes x 16 B = 1 KiB contiguously, so the images cannot be padded against bank
// conflicts; they are XOR-swizzled instead, through the per-lane SOURCE address (guide rule 21):
//   [row][k] images (A, and B when BT), KC doubles per row, 16-byte slot sl of row r holds source slot sl ^ f(r),
//       f(r) = r & 15 (KC = 32: 256-B rows) or (r >> 1) & 7 (KC = 16: 128-B rows): the 16 rows a ds_read_b64 half-wave
//       touches land on distinct bank groups;
//   [k][n] images (B when !BT, A when AT), TS doubles per k row, slot sl of row k holds source slot sl ^ ((k & 1) << 3):
//       the two k rows of a half-wave land on the two halves of the bank window.
// =====================================================================================
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 2 || N == 4 || N == 8 || N == 12 || N == 16 || N == 24, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

template <bool BT, bool AT, int TS, int KC, int NBUF>
__device__ __forceinline__ void gemm_tile_dma(const GemmArgs& g, int TM, int TN, int G, int fold_tm, const int vid, double* smem) {
    constexpr int WS = TS / 2, RT = WS / 16;
    constexpr int OP_DBL = TS * KC, BUF = 2 * OP_DBL;            // doubles per operand image / per buffer
    constexpr int KB_OP = OP_DBL * 8 / 1024;                      // 1-KiB DMA pieces per operand and chunk
    constexpr int PW = KB_OP / 4;                                 // ... per wave
    constexpr int GL = 2 * PW;                                    // DMA instructions per wave and chunk
    constexpr int KS = KC / 4;                                    // MFMA k-steps per chunk
    static_assert(KB_OP % 4 == 0 && (KC == 16 || KC == 32) && NBUF >= 2 && NBUF <= 4, "geometry");
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
    if (g.k_from_ij) kbeg = i0 > j0 ? i0 : j0;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int lc = lane & 15, lk = lane >> 4;
    const bool active = (i0 + WS * wr < M) && (j0 + WS * wc < N) && !(g.lower_only && i0 == j0 && wc > wr);
    double* const ctile = C + (size_t)(i0 + WS * wr + lk) * g.ldc + j0 + WS * wc + lc;
    const bool use_c = active && g.beta != 0.0;
    const double cscale = g.beta != 0.0 ? g.beta / g.alpha : 0.0;

    // ---- DMA sources.  Piece p (0 .. KB_OP) of an operand image is LDS bytes [1024 p, 1024 p + 1024); wave w moves pieces
    // w, w + 4, ...  Rows of the tile beyond M / N (second half of a 128-tile on a 64-granular edge) are clamped to the
    // last valid row: their products belong to inactive wave quadrants.
    // [row][k]: SPR = KC / 2 slots per row, RPP = 1024 / (8 KC) rows per piece
    constexpr int RK_SPR = KC / 2, RK_RPP = 128 / KC;
    constexpr int KN_SPR = TS / 2, KN_RPP = 128 / TS > 0 ? 128 / TS : 1;      // [k][n]: TS = 64: 2 k rows per piece, 128: 1
    auto rk_f = [](int r) { return KC == 32 ? (r & 15) : ((r >> 1) & 7); };
    // per-lane element offset (doubles) of this lane's 16 bytes inside piece p's source, p = w + 4 u
    long offA[PW], offB[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        const int p = w + 4 * u;
        if (!AT) {
            int r = p * RK_RPP + lane / RK_SPR;
            const int sl = (lane % RK_SPR) ^ rk_f(r);
            if (i0 + r >= M) r = M - 1 - i0;
            offA[u] = (long)(i0 + r) * g.lda + 2 * sl;
        } else {
            const int k = (TS == 64 ? 2 * p + lane / KN_SPR : p);
            int sl = (lane % KN_SPR) ^ ((k & 1) << 3);
            if (i0 + 2 * sl >= M) sl = (M - 2 - i0) / 2;
            offA[u] = (long)k * g.lda + i0 + 2 * sl;
        }
        if (BT) {
            int r = p * RK_RPP + lane / RK_SPR;
            const int sl = (lane % RK_SPR) ^ rk_f(r);
            if (j0 + r >= N) r = N - 1 - j0;
            offB[u] = (long)(j0 + r) * g.ldb + 2 * sl;
        } else {
            const int k = (TS == 64 ? 2 * p + lane / KN_SPR : p);
            int sl = (lane % KN_SPR) ^ ((k & 1) << 3);
            if (j0 + 2 * sl >= N) sl = (N - 2 - j0) / 2;
            offB[u] = (long)k * g.ldb + j0 + 2 * sl;
        }
    }
    auto issue = [&](const int c) {          // chunk c -> buffer c % NBUF
        const int kc = kbeg + c * KC;
        double* const buf = smem + (c % NBUF) * BUF;
        const double* const srcA = AT ? A + (size_t)kc * g.lda : A + kc;
        const double* const srcB = BT ? B + kc : B + (size_t)kc * g.ldb;
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA + offA[u]), (lptr_t)(buf + (w + 4 * u) * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB + offB[u]), (lptr_t)(buf + OP_DBL + (w + 4 * u) * 128), 16, 0, 0);
        }
    };
    // ---- fragment addresses (doubles, inside a buffer)
    int fa_base, fb_base, fa_g = 0, fb_g = 0;     // [row][k]: base = row * KC + (lk & 1), g = (lk >> 1) ^ f(row): slot of k-step s is (2 s) ^ g
    int fb_q[RT], fa_q[RT];                       // [k][n]: per column tile (n >> 1 ^ f(k)) * 2 + (n & 1); base = lk * TS
    if (!AT) { const int r = WS * wr + lc; fa_base = r * KC + (lk & 1); fa_g = (lk >> 1) ^ rk_f(r); }
    else {
        fa_base = lk * TS;
#pragma unroll
        for (int q = 0; q < RT; ++q) { const int n = WS * wr + 16 * q + lc; fa_q[q] = (((n >> 1) ^ ((lk & 1) << 3)) << 1) + (n & 1); }
    }
    if (BT) { const int r = WS * wc + lc; fb_base = OP_DBL + r * KC + (lk & 1); fb_g = (lk >> 1) ^ rk_f(r); }
    else {
        fb_base = OP_DBL + lk * TS;
#pragma unroll
        for (int q = 0; q < RT; ++q) { const int n = WS * wc + 16 * q + lc; fb_q[q] = (((n >> 1) ^ ((lk & 1) << 3)) << 1) + (n & 1); }
    }
    auto ldfrag = [&](const double* buf, const int s_, double (&a)[RT], double (&bb)[RT]) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
            a[r] = lds_ld(AT ? &buf[fa_base + 4 * s_ * TS + fa_q[r]] : &buf[fa_base + 16 * r * KC + (((2 * s_) ^ fa_g) << 1)]);
#pragma unroll
        for (int q = 0; q < RT; ++q)
            bb[q] = lds_ld(BT ? &buf[fb_base + 16 * q * KC + (((2 * s_) ^ fb_g) << 1)] : &buf[fb_base + 4 * s_ * TS + fb_q[q]]);
    };

    d4 acc[RT][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < RT; ++c) acc[r][c] = d4{0, 0, 0, 0};
    const int nch = (kend - kbeg) / KC;
    // prologue: the first NBUF - 1 chunks in flight, then C (ordinary loads: their first use drains every counter, the
    // DMAs of the prologue included — once per tile)
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < nch) issue(c);
    if (use_c) {
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
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < nch; ++c) {
        const bool more = c + NBUF - 1 < nch;
        if (more) issue(c + NBUF - 1);             // into the buffer chunk c - 1 was read from: every wave is past that barrier
        const double* buf = smem + (c % NBUF) * BUF;
        if (active) {
            double fa[2][RT], fb[2][RT];
            ldfrag(buf, 0, fa[0], fb[0]);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                // Program order pinned: the LDS reads of k-step s + 1 are issued after the first MFMAs of k-step s, i.e. with
                // at least half a k-step of matrix work still queued behind them.  (Left alone hipcc sinks them behind ALL the
                // MFMAs of step s, and it waits for LDS with lgkmcnt(0), never a counted wait, in this kernel: reads issued
                // just before a wait would stall the matrix pipe for their whole latency.)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (r == 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (s_ + 1 < KS) ldfrag(buf, s_ + 1, fa[(s_ + 1) & 1], fb[(s_ + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int q = 0; q < RT; ++q)
                        acc[r][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s_ & 1][r], fb[s_ & 1][q], acc[r][q], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // chunk c + 1 must have landed (this wave's share; the barrier makes it everybody's) — the younger chunks stay in flight
        if (more) wait_vmcnt<GL * (NBUF - 2)>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (!active) return;
    const double alpha = g.alpha;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < RT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) ctile[(size_t)(16 * r + 4 * e) * g.ldc + 16 * c] = alpha * acc[r][c][e];
}

template <bool BT, bool AT, int TS, int KC, int NBUF>
__global__ __launch_bounds__(256, (2 * TS * KC * 8 * NBUF <= 80 * 1024 && TS == 64) ? 2 : 1) void k_gemm_dma(GemmArgs g, int TM, int TN, int G, int fold_tm) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gemm_tile_dma<BT, AT, TS, KC, NBUF>(g, TM, TN, G, fold_tm, blockIdx.x, smem);
}


static int gemm_dma_geom() {
    static const int v = [] { const char* e = getenv("GPT_GEMM_DMA_GEOM"); return e ? atoi(e) : 1; }();
    return v;
}

template <bool BT, bool AT, int TS, int KC, int NBUF>
static void launch_gemm_dma(hipStream_t s, const GemmArgs& g, const GemmGrid& q) {
    constexpr size_t lds = (size_t)2 * TS * KC * NBUF * sizeof(double);
    static PerDeviceOnce once;
    once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dma<BT, AT, TS, KC, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((k_gemm_dma<BT, AT, TS, KC, NBUF>), dim3(q.nvid), dim3(256), lds, s, g, q.TM, q.TN, q.G, q.fold_tm);
}

template <bool BT, bool AT, int TS>
static void launch_gemm_variant(hipStream_t s, const GemmArgs& g, const GemmGrid& q, size_t lds) {
    const int body = gemm_body();
    if (body == 2) {
        if (TS == 128) launch_gemm_dma<BT, AT, 128, 16, 3>(s, g, q);
        else if (gemm_dma_geom() == 1) launch_gemm_dma<BT, AT, 64, 16, 4>(s, g, q);
        else launch_gemm_dma<BT, AT, 64, 32, 3>(s, g, q);
        return;
    }
    static PerDeviceOnce once;
    once.run([&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_deep<BT, AT, TS, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (BT && !AT && TS == 64) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_deep<BT, AT, TS, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_deep<BT, AT, TS, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_deep<BT, AT, TS, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_pipe<BT, AT, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds));
    });
    if (body == 3) {
        const bool fixed_k = !g.a_lower && !g.b_lower && !g.k_from_ij && g.nbatch == 1;
        const int nch = fixed_k ? g.K / 32 : 0;
        if constexpr (BT && !AT && TS == 64) {
            if (nch == 4) { hipLaunchKernelGGL((k_gemm_deep<BT, AT, TS, 4>), dim3(q.nvid), dim3(256), lds, s, g, q.TM, q.TN, q.G, q.fold_tm); return; }
            if (nch == 8) { hipLaunchKernelGGL((k_gemm_deep<BT, AT, TS, 8>), dim3(q.nvid), dim3(256), lds, s, g, q.TM, q.TN, q.G, q.fold_tm); return; }
            if (nch == 16) { hipLaunchKernelGGL((k_gemm_deep<BT, AT, TS, 16>), dim3(q.nvid), dim3(256), lds, s, g, q.TM, q.TN, q.G, q.fold_tm); return; }
        }
        hipLaunchKernelGGL((k_gemm_deep<BT, AT, TS, 0>), dim3(q.nvid), dim3(256), lds, s, g, q.TM, q.TN, q.G, q.fold_tm);
        return;
    }
    hipLaunchKernelGGL((k_gemm_pipe<BT, AT, TS>), dim3(q.nvid), dim3(256), 2 * lds, s, g, q.TM, q.TN, q.G, q.fold_tm);
}
