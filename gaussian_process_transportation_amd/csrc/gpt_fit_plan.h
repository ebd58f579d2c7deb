// Plan of the recursive factor + inverse (host-only C++, no HIP: gpt_fit.hip executes it, the CPU sanitizer build's
// stand-in replays its memory regions, tests/test_host_cpu.py checks them through gpt_debug_fit_plan).
//
// L = chol(A) in place and W = L^-1 for the diagonal block [off, off + n) of an NP x NP matrix:
//   n <= leaf :  the right-looking blocked factorisation (k_potrf_step chain + rank-128/256 updates confined to the block),
//                then the block's triangular inverse by recursive doubling                      (LEAF_FACTOR, LEAF_INVERSE)
//   otherwise :  split n = n1 + n2;   (L11, W11) = rec(A11)
//                L21 = A21 W11^T                       one GEMM, K = n1 (triangular)           (L21 -> bounce buffer, COPY_L21)
//                A22 -= L21 L21^T                      one SYRK, K = n1                        (SYRK)
//                T21 = L21 W11                         side branch: only needed by W21         (FORK, T21 on the side stream)
//                (L22, W22) = rec(A22)
//                W21 = -W22 T21                        one GEMM, K = n2 (triangular)           (JOIN, W21)
// so that three quarters of the 2 N^3 / 3 flops run in products with K = N/2 and nine tenths with K >= 512 — where the fp64 tile
// GEMM reaches its deep-K rate — instead of in rank-128/256 updates (sklearn/_gpr.py:346-364's cholesky + the explicit inverse of
// models/gaussian_process.py:42-43 are one LAPACK call each on the CPU; this is their blocked restatement for the GPU).
// Every scratch region is taken from ONE arena by a stack allocator HERE, sized by the same walk that assigns the
// offsets: factor_scratch_doubles(NP) = fit_plan(NP).arena — the layout cannot disagree with the size (round 3's fault was
// a region sized for n/2 used with a split off the half).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace gpt {

enum FitOpKind {
    FOP_LEAF_FACTOR = 0,   // block columns [off, off + n1) of the diagonal block, rows < off + n1
    FOP_LEAF_INVERSE = 1,  // W[off.., off..] (n1 x n1) from the 64-blocks' inverses; scratch region r0
    FOP_L21 = 2,           // r0 (n2 x n1, compact) = A21 W11^T
    FOP_COPY_L21 = 3,      // K21 = r0
    FOP_SYRK = 4,          // A22 -= r0 r0^T (lower)
    FOP_FORK = 5,          // side stream waits for everything the main stream has issued (event pair `depth`)
    FOP_T21 = 6,           // r1 (n2 x n1, compact) = K21 W11; on the side stream when side != 0
    FOP_JOIN = 7,          // main stream waits for the side stream
    FOP_W21 = 8,           // W21 = -W22 r1
    FOP_FACTORED = 9,      // L is complete here (the caller's event)
};

struct FitOp {
    int kind, side, depth;
    int off, n1, n2;           // block [off, off + n1 + n2): first part n1, second part n2 (leaf ops: n2 = 0)
    size_t r0, r0_size;        // arena region (doubles)
    size_t r1, r1_size;
};

struct FitPlan {
    int NP = 0, leaf = 0, align = 256, fork_min = 0;
    bool recursive = false;
    std::vector<FitOp> ops;
    size_t arena = 0;          // doubles
    int max_depth = 0;
};

// extent (doubles) of the scratch the recursive-doubling inverse of an n x n block touches (gpt_fit.hip trinv_levels:
// level sz writes nbp - 1 blocks of sz x sz and the last pair's m_last x sz)
inline size_t trinv_extent(int n, int nb64 = 64) {
    size_t ext = 0;
    for (long sz = nb64; sz < n; sz *= 2) {
        const int npairs = (int)((n + 2 * sz - 1) / (2 * sz));
        const long r0_last = (long)(npairs - 1) * 2 * sz;
        long m_last = n - r0_last - sz;
        int nbp = npairs;
        if (m_last <= 0) { nbp = npairs - 1; m_last = sz; }
        if (nbp <= 0) continue;
        if (m_last > sz) m_last = sz;
        const size_t e = (size_t)(nbp - 1) * sz * sz + (size_t)m_last * sz;
        if (e > ext) ext = e;
    }
    return ext;
}

inline int fit_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// leaf / rec_min / fork_min < 0: from the environment (GPT_FIT_LEAF, GPT_FIT_REC_MIN, GPT_FIT_FORK_MIN) or the defaults
inline FitPlan fit_plan(int NP, int leaf = -1, int rec_min = -1, int fork_min = -1) {
    FitPlan pl;
    pl.NP = NP;
    pl.leaf = leaf > 0 ? leaf : fit_env_int("GPT_FIT_LEAF", 1024);
    if (pl.leaf < 256) pl.leaf = 256;
    const int rmin = rec_min >= 0 ? rec_min : fit_env_int("GPT_FIT_REC_MIN", 4096);
    pl.fork_min = fork_min >= 0 ? fork_min : fit_env_int("GPT_FIT_FORK_MIN", 4096);
    pl.recursive = fit_env_int("GPT_FIT_RECURSIVE", 0) != 0 && NP >= rmin && NP > pl.leaf;
    size_t top = 0;
    auto push = [&](size_t n) { const size_t o = top; top += (n + 511) / 512 * 512; if (top > pl.arena) pl.arena = top; return o; };
    struct Rec {
        FitPlan& pl; size_t& top; decltype(push)& push_;
        void leaf_ops(int off, int n, int depth) {
            pl.ops.push_back(FitOp{FOP_LEAF_FACTOR, 0, depth, off, n, 0, 0, 0, 0, 0});
            const size_t mark = top;
            const size_t ext = trinv_extent(n);
            const size_t r = push_(ext);
            pl.ops.push_back(FitOp{FOP_LEAF_INVERSE, 0, depth, off, n, 0, r, ext, 0, 0});
            top = mark;
        }
        void run(int off, int n, int depth) {
            if (depth > pl.max_depth) pl.max_depth = depth;
            if (!pl.recursive || n <= pl.leaf) { leaf_ops(off, n, depth); return; }
            int n1 = (n / 2 + pl.align - 1) / pl.align * pl.align;
            if (n1 >= n) n1 = n - pl.align;
            const int n2 = n - n1;
            run(off, n1, depth + 1);
            const size_t mark = top;
            const size_t sz = (size_t)n2 * n1;
            const size_t t = push_(sz);         // T21: lives until W21
            const size_t p = push_(sz);         // bounce buffer of L21: dead after the SYRK
            const int side = n >= pl.fork_min ? 1 : 0;
            pl.ops.push_back(FitOp{FOP_L21, 0, depth, off, n1, n2, p, sz, 0, 0});
            pl.ops.push_back(FitOp{FOP_COPY_L21, 0, depth, off, n1, n2, p, sz, 0, 0});
            if (side) pl.ops.push_back(FitOp{FOP_FORK, 1, depth, off, n1, n2, 0, 0, 0, 0});
            pl.ops.push_back(FitOp{FOP_T21, side, depth, off, n1, n2, 0, 0, t, sz});
            pl.ops.push_back(FitOp{FOP_SYRK, 0, depth, off, n1, n2, p, sz, 0, 0});
            top = t + (sz + 511) / 512 * 512;   // the bounce buffer is free again
            run(off + n1, n2, depth + 1);
            if (side) pl.ops.push_back(FitOp{FOP_JOIN, 1, depth, off, n1, n2, 0, 0, 0, 0});
            pl.ops.push_back(FitOp{FOP_W21, 0, depth, off, n1, n2, 0, 0, t, sz});
            top = mark;
        }
    } rec{pl, top, push};
    rec.run(0, NP, 0);
    // L is complete after the last leaf has been factored
    for (size_t i = pl.ops.size(); i-- > 0;)
        if (pl.ops[i].kind == FOP_LEAF_FACTOR) {
            pl.ops.insert(pl.ops.begin() + (long)i + 1, FitOp{FOP_FACTORED, 0, 0, 0, 0, 0, 0, 0, 0, 0});
            break;
        }
    pl.arena += 4096;
    return pl;
}

// Scratch of launch_factor_inverse in doubles: the arena of the plan this size runs with, and never less than what the
// non-recursive forms use (T of the whole-matrix inverse NP^2/4; the overlapped form's half-size inverses NP^2/16 more).
inline size_t factor_scratch_doubles_of(int NP) {
    const size_t legacy = (size_t)NP * NP / 4 + (size_t)NP * NP / 16 + 4096;
    const size_t planned = fit_plan(NP).arena;
    return planned > legacy ? planned : legacy;
}

}  // namespace gpt
