// Plan of the factor + inverse for large models (host-only C++, no HIP: gpt_fit.hip executes it, the CPU sanitizer build's
// stand-in replays its memory regions, tests/test_host_cpu.py checks regions AND ordering through gpt_debug_fit_plan).
//
// L = chol(A) in place (sklearn/_gpr.py:346-364's cholesky) and W = L^-1 (the factor of models/gaussian_process.py:42-43's
// explicit K^-1) are one LAPACK call each on the CPU.  Here, for NP >= rec_min: a LEFT-LOOKING blocked form with panels of
// `panel` columns (1024) and look-ahead, laid out so that
//   * every O(N^3) flop sits in a product with K >= panel — where the fp64 tile GEMM reaches its deep-K rate (60 TFLOP/s against
//     37-54 in the rank-128/256 trailing updates of the right-looking form: profiles/r03_fit_gemm_study.txt, r04_fit_summary.txt);
//   * the launch chain of the factorisation — 128 k_potrf_step launches at N = 8192, 2.4 ms that nothing can shorten — has the bulk
//     of the update flops and the WHOLE inverse running beside it instead of behind it.
// Panel p = columns [o, o + b), r = NP - o - b rows below it:
//   UPD_BULK(p)  A[o:, o:o+b] -= L[o:, 0:o'] L[o:o+b, 0:o']^T     o' = start of panel p-1: the panels final since TRSM(p-2)   side
//   UPD_LAST(p)  A[o:, o:o+b] -= L[o:, o':o] L[o:o+b, o':o]^T     panel p-1 alone (K = b), the only update on the chain      main
//   LEAF(p)      right-looking factorisation of the b x b diagonal block (k_potrf_step chain) + its inverse W_pp               chain
//   TRSM(p)      L[o+b:, o:o+b] = A[o+b:, o:o+b] W_pp^T            (bounce buffer + copy back)                                 main
//   T(p)         T = L[o:o+b, 0:o] W[0:o, 0:o]                     needs row p of L (TRSM(p-1)) and W rows < p                 side
//   WFIN(p)      W[o:o+b, 0:o] = -W_pp T                           needs LEAF(p)                                               side
// Three streams: `main` (all CUs), `side` (7/8 of the CUs: look-ahead updates and the inverse), `chain` (the other 1/8: the
// leaves, so that a k_potrf_step never queues behind a side GEMM's grid — measured: 19 -> 68-83 us per step when it does).
// Cross-stream order is by events; every op lists the events it waits for and the one it records.  The checker in
// tests/test_host_cpu.py rebuilds the happens-before relation from (stream order + events) and asserts it covers every pair
// of ops that touch overlapping memory, so a missing wait is a CPU test failure, not a GPU race.
// Every scratch region comes from ONE arena sized by the same walk that assigns the offsets (factor_scratch_doubles).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace gpt {

enum FitOpKind {
    FOP_LEAF_FACTOR = 0,   // block columns [off, off + n1) of the diagonal block, rows < off + n1
    FOP_LEAF_INVERSE = 1,  // W[off.., off..] (n1 x n1) from the 64-blocks' inverses; scratch region r0
    FOP_UPDATE = 2,        // A[off:, off:off+n1] -= L[off:, k0:k0+kw] L[off:off+n1, k0:k0+kw]^T   (rows off .. NP)
    FOP_TRSM = 3,          // r0 (n2 x n1, compact) = A[off+n1:, off:off+n1] W_pp^T
    FOP_COPY_L21 = 4,      // L[off+n1:, off:off+n1] = r0
    FOP_T = 5,             // r1 (n1 x off, compact) = L[off:off+n1, 0:off] W[0:off, 0:off]
    FOP_WFIN = 6,          // W[off:off+n1, 0:off] = -W_pp r1
    FOP_FACTORED = 7,      // L is complete here (the caller's event)
};
enum { FS_MAIN = 0, FS_SIDE = 1, FS_CHAIN = 2 };
constexpr int FIT_MAX_EVENTS = 256;

struct FitOp {
    int kind, stream;
    int off, n1, n2;           // panel [off, off + n1), n2 rows below it
    int k0, kw;                // FOP_UPDATE: source columns [k0, k0 + kw)
    size_t r0, r0_size;        // arena regions (doubles)
    size_t r1, r1_size;
    int wait[3];               // event ids this op's stream waits for before it (-1: none)
    int record;                // event id recorded in this op's stream after it (-1: none)
};

struct FitPlan {
    int NP = 0, panel = 0;
    bool blocked = false;      // false: the whole matrix is one leaf (the right-looking form + recursive-doubling inverse)
    std::vector<FitOp> ops;
    size_t arena = 0;          // doubles
    int n_events = 0;
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

// panel / rec_min < 0: from the environment (GPT_FIT_PANEL, GPT_FIT_REC_MIN) or the defaults.  side: 0 = everything in the
// main stream (no look-ahead: the serial form of the same algebra), 1 = three streams; < 0: GPT_FIT_OVERLAP or 1.
inline FitPlan fit_plan(int NP, int panel = -1, int rec_min = -1, int side = -1) {
    FitPlan pl;
    pl.NP = NP;
    pl.panel = panel > 0 ? panel : fit_env_int("GPT_FIT_PANEL", 1024);
    pl.panel = (pl.panel + 255) / 256 * 256;
    const int rmin = rec_min >= 0 ? rec_min : fit_env_int("GPT_FIT_REC_MIN", 4096);
    const bool streams = (side >= 0 ? side : fit_env_int("GPT_FIT_OVERLAP", 1)) != 0;
    pl.blocked = fit_env_int("GPT_FIT_BLOCKED", 1) != 0 && NP >= rmin && NP > pl.panel;
    auto op = [](int kind, int stream, int off, int n1, int n2) {
        FitOp o{};
        o.kind = kind; o.stream = stream; o.off = off; o.n1 = n1; o.n2 = n2;
        o.wait[0] = o.wait[1] = o.wait[2] = -1; o.record = -1;
        return o;
    };
    auto al = [](size_t n) { return (n + 511) / 512 * 512; };
    if (!pl.blocked) {
        pl.ops.push_back(op(FOP_LEAF_FACTOR, FS_MAIN, 0, NP, 0));
        pl.ops.push_back(op(FOP_FACTORED, FS_MAIN, 0, 0, 0));
        FitOp inv = op(FOP_LEAF_INVERSE, FS_MAIN, 0, NP, 0);
        inv.r0 = 0; inv.r0_size = trinv_extent(NP);
        pl.ops.push_back(inv);
        pl.arena = al(inv.r0_size) + 4096;
        return pl;
    }
    // panels: P equal widths, multiples of 256
    const int P = (NP + pl.panel - 1) / pl.panel;
    std::vector<int> o(P + 1, 0);
    {
        const int wdt = ((NP + P - 1) / P + 255) / 256 * 256;
        for (int p = 1; p <= P; ++p) o[p] = (o[p - 1] + wdt < NP) ? o[p - 1] + wdt : NP;
        o[P] = NP;
    }
    // arena: [leaf inverse scratch | TRSM bounce | T]   (each used by one stream at a time; see the op list)
    size_t leaf_scr = 0, bounce = 0, tbuf = 0;
    for (int p = 0; p < P; ++p) {
        const int b = o[p + 1] - o[p], r = NP - o[p + 1];
        if (trinv_extent(b) > leaf_scr) leaf_scr = trinv_extent(b);
        if ((size_t)r * b > bounce) bounce = (size_t)r * b;
        if ((size_t)b * o[p] > tbuf) tbuf = (size_t)b * o[p];
    }
    const size_t off_leaf = 0, off_bounce = al(leaf_scr), off_t = off_bounce + al(bounce);
    pl.arena = off_t + al(tbuf) + 4096;
    int nev = 0;
    auto new_event = [&]() { return nev++; };
    const int S_SIDE = streams ? FS_SIDE : FS_MAIN, S_CHAIN = streams ? FS_CHAIN : FS_MAIN;
    std::vector<int> ev_trsm(P, -1), ev_leaf(P, -1), ev_bulk(P + 1, -1), ev_wfin(P, -1);
    int ev_t_prev = -1;                      // the T buffer: T(p) must not overwrite what WFIN(p-1) still reads (same stream: ordered)
    for (int p = 0; p < P; ++p) {
        const int off = o[p], b = o[p + 1] - o[p], r = NP - o[p + 1];
        // ---- main: the update by panel p-1 (after the look-ahead part by the earlier panels has landed)
        if (p >= 1) {
            FitOp u = op(FOP_UPDATE, FS_MAIN, off, b, r);
            u.k0 = o[p - 1]; u.kw = o[p] - o[p - 1];
            u.wait[0] = ev_bulk[p];                                  // (-1 for p == 1: no earlier panels)
            pl.ops.push_back(u);
        }
        // ---- chain: the leaf
        {
            FitOp lf = op(FOP_LEAF_FACTOR, p == 0 ? FS_MAIN : S_CHAIN, off, b, 0);
            FitOp li = op(FOP_LEAF_INVERSE, lf.stream, off, b, 0);
            li.r0 = off_leaf; li.r0_size = trinv_extent(b);
            if (lf.stream != FS_MAIN) {                              // hand-over main -> chain -> main
                const int e_in = new_event();
                pl.ops.back().record = e_in;                         // the update just issued in main
                lf.wait[0] = e_in;
            }
            pl.ops.push_back(lf);
            if (p == P - 1) {
                FitOp f = op(FOP_FACTORED, lf.stream, 0, 0, 0);
                pl.ops.push_back(f);
            }
            ev_leaf[p] = new_event();
            li.record = ev_leaf[p];
            pl.ops.push_back(li);
        }
        // ---- main: the rows below the panel
        if (r > 0) {
            FitOp t = op(FOP_TRSM, FS_MAIN, off, b, r);
            t.r0 = off_bounce; t.r0_size = (size_t)r * b;
            t.wait[0] = ev_leaf[p];
            pl.ops.push_back(t);
            FitOp c = op(FOP_COPY_L21, FS_MAIN, off, b, r);
            c.r0 = off_bounce; c.r0_size = (size_t)r * b;
            ev_trsm[p] = new_event();
            c.record = ev_trsm[p];
            pl.ops.push_back(c);
        }
        // ---- side: the inverse's block row p (its small last product), then the look-ahead update of panel p+1 by the panels
        //      < p, then the big product of block row p+1 — in that order: the update is what the chain will wait for
        if (p >= 1) {
            FitOp w = op(FOP_WFIN, S_SIDE, off, b, 0);
            w.r1 = off_t; w.r1_size = (size_t)b * off;
            w.wait[0] = ev_leaf[p];                                  // W_pp; T(p) precedes it in the same stream
            if (p == P - 1) { w.stream = FS_MAIN; w.wait[1] = ev_t_prev; }       // the tail of the whole thing: in the caller's stream
            pl.ops.push_back(w);
        }
        if (p + 1 < P && p >= 1) {
            const int q = p + 1;                                     // panel q gets the panels < p now (final since TRSM(p-1))
            FitOp u = op(FOP_UPDATE, S_SIDE, o[q], o[q + 1] - o[q], NP - o[q + 1]);
            u.k0 = 0; u.kw = o[p];
            u.wait[0] = ev_trsm[p - 1];
            ev_bulk[q] = new_event();
            u.record = ev_bulk[q];
            pl.ops.push_back(u);
        }
        if (p + 1 < P) {
            const int q = p + 1;                                     // T(q) = L[q, 0:o_q] W[0:o_q, 0:o_q]: row q of L is final after TRSM(p)
            FitOp t = op(FOP_T, S_SIDE, o[q], o[q + 1] - o[q], 0);
            t.r1 = off_t; t.r1_size = (size_t)(o[q + 1] - o[q]) * o[q];
            t.wait[0] = ev_trsm[p];                                  // (W rows < q: WFIN(<= p) precede in the same stream; WFIN(p) above)
            ev_t_prev = new_event();
            t.record = ev_t_prev;
            pl.ops.push_back(t);
        }
    }
    pl.n_events = nev;
    return pl;
}

// Scratch of launch_factor_inverse in doubles: the arena of the plan this size runs with, and never less than what the
// one-leaf form of any size uses (T of the whole-matrix inverse, NP^2/4).
inline size_t factor_scratch_doubles_of(int NP) {
    const size_t legacy = (size_t)NP * NP / 4 + (size_t)NP * NP / 16 + 4096;
    const size_t planned = fit_plan(NP).arena;
    return planned > legacy ? planned : legacy;
}

}  // namespace gpt
