// Plan of the factor + inverse (host-only C++, no HIP: gpt_fit.hip executes it, the CPU sanitizer build's stand-in replays its
// memory regions, tests/test_host_cpu.py checks regions AND cross-stream ordering through gpt_debug_fit_plan).
//
// L = chol(A) in place (sklearn/_gpr.py:346-364's cholesky) and W = L^-1 (the factor of models/gaussian_process.py:42-43's
// explicit K^-1) are one LAPACK call each on the CPU.  On the GPU they are a list of operations over up to three streams:
// `main` (the caller's, all CUs), `side` (the first side_eighths/8 of the CUs) and `chain` (the other CUs) — CU-masked, because a
// launch chain of k_potrf_step kernels and a bulk GEMM only run side by side without delaying each other when they cannot land
// on the same CUs (tools/probes/cumask_probe.hip; a step beside an unmasked GEMM: 19 -> 68-83 us, profiles/r04_fit_summary.txt).
// Every operation names the events it waits for and the event it records; every scratch region comes from ONE arena sized by the
// walk that assigns the offsets.  The checker in tests/test_host_cpu.py rebuilds the happens-before relation from (stream order
// + events) and asserts it covers every pair of operations that touch overlapping memory, and that every region lies inside
// the arena: a layout or ordering mistake is a CPU test failure, not a GPU fault (round 3 had one: a scratch region sized for
// a half split used with another split).
//
// Three forms:
//   0  one leaf: the right-looking blocked factorisation of the whole matrix, then its inverse by recursive doubling; one stream.
//   1  split (the shipped form for 4096 < NP <= 12288): right-looking over the first half's columns h; then the rest of the
//      factorisation — bound by its launch chain, most CUs idle — on `chain` while W11 = L11^-1 and T21 = L21 W11 (5/8 of the
//      inverse's flops, needing only the first half's columns) run on `side`; afterwards W22 and W21 = -W22 T21 on the whole chip.
//   2  left-looking panels with look-ahead (opt-in, GPT_FIT_FORM=2; measured slower than form 1 at every size tried:
//      profiles/r04_fit_summary.txt): panel p = columns [o, o + b)
//        UPDATE(p, look-ahead)  A[o:, o:o+b] -= L[o:, 0:o'] L[o:o+b, 0:o']^T    o' = start of panel p-1        side
//        UPDATE(p, last)        ... -= L[o:, o':o] L[o:o+b, o':o]^T              panel p-1 alone (K = b)        main
//        POTRF, FINISH, TRINV   the b x b diagonal block and its inverse W_pp                                  chain
//        TRSM + COPY            L[o+b:, o:o+b] = A[o+b:, o:o+b] W_pp^T  (bounce buffer)                        main
//        T, WFIN                T = L[o:o+b, 0:o] W[0:o, 0:o];  W[o:o+b, 0:o] = -W_pp T                        side
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace gpt {

enum FitOpKind {
    FOP_POTRF = 0,         // right-looking blocked factorisation of block columns [off, off + n1), trailing updates over rows < row_end
    FOP_FINISH = 1,        // diagonal 64-blocks of [off, off + n1): parked L_kk from W into K, inv(L_kk) into W
    FOP_TRINV = 2,         // W[off.., off..] (n1 x n1) = L^-1 of that diagonal block by recursive doubling; scratch region r0
    FOP_UPDATE = 3,        // A[off:, off:off+n1] -= L[off:, k0:k0+kw] L[off:off+n1, k0:k0+kw]^T   (rows off .. NP)
    FOP_TRSM = 4,          // r0 (n2 x n1, compact) = A[off+n1:, off:off+n1] W_pp^T
    FOP_COPY_L21 = 5,      // L[off+n1:, off:off+n1] = r0
    FOP_T = 6,             // r1 (n1 x off, compact) = L[off:off+n1, 0:off] W[0:off, 0:off]
    FOP_WFIN = 7,          // W[off:off+n1, 0:off] = -W[off:off+n1, off:off+n1] r1
    FOP_FACTORED = 8,      // L is complete here (the caller's event)
};
enum { FS_MAIN = 0, FS_SIDE = 1, FS_CHAIN = 2 };
constexpr int FIT_MAX_EVENTS = 256;
constexpr int FIT_NB = 64;              // = NB of gpt_common.h: the Cholesky's diagonal block

struct FitOp {
    int kind, stream;
    int off, n1, n2;           // block / panel [off, off + n1); n2: rows below it (TRSM, COPY)
    int k0, kw;                // FOP_UPDATE: source columns [k0, k0 + kw)
    int row_end, grp;          // FOP_POTRF: rows taking part; panels per trailing update (0: by size)
    size_t r0, r0_size;        // arena regions (doubles)
    size_t r1, r1_size;
    int wait[3];               // event ids this op's stream waits for before it (-1: none)
    int record;                // event id recorded in this op's stream after it (-1: none)
};

struct FitPlan {
    int NP = 0, form = 0, panel = 0, side_eighths = 7;
    std::vector<FitOp> ops;
    size_t arena = 0;          // doubles
    int n_events = 0;
    bool multi_stream() const { for (const FitOp& o : ops) if (o.stream != FS_MAIN) return true; return false; }
};

// extent (doubles) of the scratch the recursive-doubling inverse of an n x n block touches (gpt_fit.hip trinv_levels:
// level sz writes nbp - 1 blocks of sz x sz and the last pair's m_last x sz)
inline size_t trinv_extent(int n, int nb64 = FIT_NB) {
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

// blocking of the right-looking factorisation (gpt_fit.hip potrf_groups): panels of `ob` 64-blocks, `grp` panels per trailing update
inline int potrf_outer_blocks() {
    static const int ob = [] { const int v = fit_env_int("GPT_POTRF_OB", 128) / FIT_NB; return (v >= 1 && v <= 8) ? v : 2; }();
    return ob;
}
inline int potrf_group(int NP) {
    const int v = fit_env_int("GPT_POTRF_GROUP", 0);
    return (v >= 1 && v <= 8) ? v : (NP >= 4096 ? 2 : 1);
}

// form < 0: GPT_FIT_FORM or by size; panel < 0: GPT_FIT_PANEL or 1024 (form 2); streams: 0 = everything in the main stream (the
// serial order of the same operations), 1 = CU-masked streams, < 0: GPT_FIT_OVERLAP or 1
inline FitPlan fit_plan(int NP, int form = -1, int panel = -1, int streams = -1) {
    FitPlan pl;
    pl.NP = NP;
    const bool want_streams = (streams >= 0 ? streams : fit_env_int("GPT_FIT_OVERLAP", 1)) != 0;
    if (form < 0) form = fit_env_int("GPT_FIT_FORM", -1);
    const int nb = NP / FIT_NB;
    const int gw = potrf_group(NP) * potrf_outer_blocks();
    const int hb = nb / 2 / gw * gw;                           // form 1: split block column, aligned to the trailing-update groups
    // by size: the split form where its second half is chain-bound (measured: N = 8192 10.4 -> 9.7 ms; N = 6000 and 12000 unchanged,
    // N = 16384 slower: its second half is GEMM-bound, profiles/r02_fit_overlap.log)
    if (form < 0) form = (NP > 4096 && NP <= 12288 && want_streams) ? 1 : 0;
    if (form == 1 && !(hb >= gw && hb < nb)) form = 0;
    pl.panel = ((panel > 0 ? panel : fit_env_int("GPT_FIT_PANEL", 1024)) + 255) / 256 * 256;
    if (form == 2 && NP <= pl.panel) form = 0;
    if (form < 0 || form > 2) form = 0;
    pl.form = form;
    auto op = [](int kind, int stream, int off, int n1, int n2) {
        FitOp o{};
        o.kind = kind; o.stream = stream; o.off = off; o.n1 = n1; o.n2 = n2;
        o.wait[0] = o.wait[1] = o.wait[2] = -1; o.record = -1;
        return o;
    };
    auto al = [](size_t n) { return (n + 511) / 512 * 512; };
    int nev = 0;
    auto new_event = [&]() { return nev++; };
    const int S_SIDE = want_streams ? FS_SIDE : FS_MAIN, S_CHAIN = want_streams ? FS_CHAIN : FS_MAIN;

    if (form == 0) {
        FitOp f = op(FOP_POTRF, FS_MAIN, 0, NP, 0); f.row_end = NP;
        pl.ops.push_back(f);
        pl.ops.push_back(op(FOP_FINISH, FS_MAIN, 0, NP, 0));
        pl.ops.push_back(op(FOP_FACTORED, FS_MAIN, 0, 0, 0));
        FitOp inv = op(FOP_TRINV, FS_MAIN, 0, NP, 0);
        inv.r0 = 0; inv.r0_size = trinv_extent(NP);
        pl.ops.push_back(inv);
        pl.arena = al(inv.r0_size) + 4096;
        return pl;
    }

    if (form == 1) {
        // the chain (rest of the factorisation) gets 3/8 of the CUs above NP = 6144, 4/8 up to it (fit at N = 5000 / 6000: 3.76-3.89 /
        // 5.14-5.29 ms with 96 CUs, 3.67 / 5.02 with 128; N = 7000 and 8192: 96 is best of 64 .. 160; sessions r3cus, r3cus2)
        pl.side_eighths = fit_env_int("GPT_FIT_SIDE_EIGHTHS", NP <= 6144 ? 4 : 5);
        const int h = hb * FIT_NB, r = NP - h;
        const size_t t21 = (size_t)r * h;
        const size_t inner = trinv_extent(h) > trinv_extent(r) ? trinv_extent(h) : trinv_extent(r);
        const size_t off_t = 0, off_inner = al(t21);
        pl.arena = off_inner + al(inner) + 4096;
        const int e_fork = new_event(), e_chain = new_event(), e_side = new_event();
        FitOp f1 = op(FOP_POTRF, FS_MAIN, 0, h, 0); f1.row_end = NP; f1.record = e_fork;       // columns [0, h) final, A22 carries their update
        pl.ops.push_back(f1);
        // chain: the rest of the factorisation
        FitOp f2 = op(FOP_POTRF, S_CHAIN, h, r, 0); f2.row_end = NP; f2.wait[0] = e_fork;
        pl.ops.push_back(f2);
        pl.ops.push_back(op(FOP_FINISH, S_CHAIN, h, r, 0));
        FitOp fac = op(FOP_FACTORED, S_CHAIN, 0, 0, 0); fac.record = e_chain;
        pl.ops.push_back(fac);
        // side: W11 = L11^-1, T21 = L21 W11
        FitOp fin1 = op(FOP_FINISH, S_SIDE, 0, h, 0); fin1.wait[0] = e_fork;
        pl.ops.push_back(fin1);
        FitOp inv1 = op(FOP_TRINV, S_SIDE, 0, h, 0); inv1.r0 = off_inner; inv1.r0_size = trinv_extent(h);
        pl.ops.push_back(inv1);
        FitOp t = op(FOP_T, S_SIDE, h, r, 0); t.r1 = off_t; t.r1_size = t21; t.record = e_side;
        pl.ops.push_back(t);
        // whole chip again: W22 = L22^-1, W21 = -W22 T21
        FitOp inv2 = op(FOP_TRINV, FS_MAIN, h, r, 0); inv2.r0 = off_inner; inv2.r0_size = trinv_extent(r);
        inv2.wait[0] = e_chain; inv2.wait[1] = e_side;
        pl.ops.push_back(inv2);
        FitOp w = op(FOP_WFIN, FS_MAIN, h, r, 0); w.r1 = off_t; w.r1_size = t21;
        pl.ops.push_back(w);
        pl.n_events = nev;
        return pl;
    }

    // ---- form 2: left-looking panels with look-ahead
    pl.side_eighths = fit_env_int("GPT_FIT_SIDE_EIGHTHS", 7);
    const int leaf_grp = fit_env_int("GPT_FIT_LEAF_GROUP", 1);
    const int P = (NP + pl.panel - 1) / pl.panel;
    std::vector<int> o(P + 1, 0);
    {
        const int wdt = ((NP + P - 1) / P + 255) / 256 * 256;
        for (int p = 1; p <= P; ++p) o[p] = (o[p - 1] + wdt < NP) ? o[p - 1] + wdt : NP;
        o[P] = NP;
    }
    // arena: [leaf inverse scratch | TRSM bounce | T]   (each used by one stream at a time: the checker verifies it)
    size_t leaf_scr = 0, bounce = 0, tbuf = 0;
    for (int p = 0; p < P; ++p) {
        const int b = o[p + 1] - o[p], r = NP - o[p + 1];
        if (trinv_extent(b) > leaf_scr) leaf_scr = trinv_extent(b);
        if ((size_t)r * b > bounce) bounce = (size_t)r * b;
        if ((size_t)b * o[p] > tbuf) tbuf = (size_t)b * o[p];
    }
    const size_t off_leaf = 0, off_bounce = al(leaf_scr), off_t = off_bounce + al(bounce);
    pl.arena = off_t + al(tbuf) + 4096;
    std::vector<int> ev_trsm(P, -1), ev_leaf(P, -1), ev_bulk(P + 1, -1);
    int ev_t_prev = -1;
    for (int p = 0; p < P; ++p) {
        const int off = o[p], b = o[p + 1] - o[p], r = NP - o[p + 1];
        // ---- main: the update by panel p-1 (after the look-ahead part by the earlier panels has landed)
        if (p >= 1) {
            FitOp u = op(FOP_UPDATE, FS_MAIN, off, b, r);
            u.k0 = o[p - 1]; u.kw = o[p] - o[p - 1];
            u.wait[0] = ev_bulk[p];                                  // (-1 for p == 1: no earlier panels)
            pl.ops.push_back(u);
        }
        // ---- chain: the diagonal block and its inverse
        {
            const int st = p == 0 ? FS_MAIN : S_CHAIN;
            FitOp lf = op(FOP_POTRF, st, off, b, 0); lf.row_end = off + b; lf.grp = leaf_grp;
            if (p >= 1) {                                            // hand-over main -> chain
                const int e_in = new_event();
                pl.ops.back().record = e_in;                         // the update just issued in main
                lf.wait[0] = e_in;
            }
            pl.ops.push_back(lf);
            pl.ops.push_back(op(FOP_FINISH, st, off, b, 0));
            if (p == P - 1) pl.ops.push_back(op(FOP_FACTORED, st, 0, 0, 0));
            FitOp li = op(FOP_TRINV, st, off, b, 0);
            li.r0 = off_leaf; li.r0_size = trinv_extent(b);
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
        // ---- side: block row p of the inverse (its small last product), then the look-ahead update of panel p+1 by the panels
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
            t.wait[0] = ev_trsm[p];                                  // (W rows < q: the WFINs precede in the same stream)
            ev_t_prev = new_event();
            t.record = ev_t_prev;
            pl.ops.push_back(t);
        }
    }
    pl.n_events = nev;
    return pl;
}

// Scratch of launch_factor_inverse in doubles: the largest arena of the forms a handle may run at this size (the one the
// environment selects and the one-leaf form it falls back to without masked streams).
inline size_t factor_scratch_doubles_of(int NP) {
    size_t need = fit_plan(NP).arena;
    const size_t one_leaf = fit_plan(NP, 0).arena;
    if (one_leaf > need) need = one_leaf;
    return need;
}

}  // namespace gpt
