// Work decomposition of the variance kernel (host-only C++, no HIP: also compiled into the CPU sanitizer build
// and exercised by tests/test_host_cpu.py through gpt_debug_var_plan).
//
// The product V = A B has A = the stack of `ntask` block-lower-triangular inverse factors (nbi x nbi tiles of
// 512 x 512 each; ntask = 1 for the exact GP, T for the SVGP exact-conversion model) and B = the generated kernel
// columns, 64 per column block.  A *sweep* is (column block cb, task, i-block ib, k range [k_lo, k_hi)) counted in QUARTER
// tiles (VAR_KQ per tile: 32 k4-steps, one LDS chunk of the kernel), k_hi <= 4 (ib + 1); a whole sweep has k_lo = 0,
// k_hi = 4 (ib + 1) and its 512 x 64 result is folded straight into the per-column sums of squares.  The diagonal tile
// [4 ib, 4 ib + 4) is divided only where a workgroup's share of the list is small against it (cut_diag below).
//
//   * Rounds: while at least P (= workgroups = CUs) column blocks are left, workgroup p takes block r P + p whole, all
//     tasks, longest sweep first — every workgroup then walks the same tiles of A at the same time, which is what
//     keeps the A stream in L2.  Rounds need no list: the kernel derives them from (round, blockIdx).
//   * Tail (and every launch with fewer than P blocks — the reference's own batch sizes, M = 400 .. 10^4): the
//     remaining sweeps are laid end to end, costed, and cut into P ranges of equal cost AT QUARTER-TILE GRANULARITY (tile
//     granularity until round 4: with 3.2 tiles of work per workgroup — N = 2500, M = 4096 — shares of 3 and 4 tiles are 25 %
//     apart, and the 42 tiles of configs[1]'s 14-block tail kept 42 of 256 workgroups busy).  A sweep
//     that is cut leaves partial products: each part writes its 512 x 64 partial V to `vslab`, and k_var_combine adds
//     the parts in order, squares and reduces.  The explicit list of items per workgroup built here is what the
//     kernel executes.  The list is block-major: a workgroup keeps its generated B image for the following sweeps of
//     the same block, and workgroups p and p + 16 k (same XCD) work on the same tiles of A at the same time.  (The
//     sweep-major order — all column blocks of one (task, ib) next to each other — is kept as a diagnostic: it lost at
//     every size, 4.1 -> 5.9 ms at N = 8192, M = 4096: neighbouring workgroups sit on different XCDs and every item
//     has to generate its B fragments again; profiles/r02_small_m.log.)
//   * Per-column partial sums go to unique slots of `slab`, a (block, task)'s slots are consecutive and
//     k_var_finalize adds them in order: deterministic, no atomics.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace gpt {

constexpr int VAR_COLS = 64;           // columns per column block
constexpr int VAR_ROWS = 512;          // rows of an i-block (= WT of gpt_common.h)
constexpr int VAR_SLOT = 2 * VAR_COLS;             // elements per slab slot: ssq[64], crs[64]
constexpr int VAR_VSLOT = VAR_ROWS * VAR_COLS;     // elements per vslab slot: one 512 x 64 partial product
constexpr int VAR_DIAG_COST = 72;      // k4-steps a diagonal tile costs a SIMD (waves g and 7-g: 16 (g+1) + 16 (8-g), halved)
constexpr int VAR_TILE_COST = 128;     // k4-steps of a full tile
constexpr int VAR_KQ = 4;              // item k ranges count quarter tiles
constexpr int VAR_Q_COST = VAR_TILE_COST / VAR_KQ;     // 32 k4-steps = one LDS chunk of k_var (both element types)
constexpr int VAR_SWEEP_OVERHEAD = 6;  // fixed cost of a sweep in k4-steps (first fill, accumulator fold)
constexpr int VAR_SPLIT_SLOTS = 8;     // slab slots of a cut sweep: k_var_combine reduces each of the 8 row groups in a workgroup of its own
constexpr int VAR_DIAG_SHARE = 4 * VAR_TILE_COST;      // the diagonal tile may be cut when a workgroup's share of the list is below this

struct VarItem {
    int cb, task, ib;     // column block, task, i-block
    int k_lo, k_hi;       // quarter tiles [k_lo, k_hi), k_hi <= 4 (ib + 1)
    int flags;            // VI_*
    int slot;             // >= 0: after this item the running column sums go to slab slot `slot`
    int vslot;            // >= 0: partial product, stored to vslab slot `vslot` instead of being folded
};
enum { VI_GEN = 1,        // B fragments of [k_lo, k_hi) are generated (and copied to the scratch image), not reloaded
       VI_FIRST = 2,      // first item of this workgroup for this column block: the scratch image changes owner
       VI_ZERO = 4 };     // the running column sums start from zero

struct VarSplit { int v_begin, v_end, slot; };   // combine vslab slots [v_begin, v_end) into slab slots [slot, slot + VAR_SPLIT_SLOTS)

// Passed by value to the kernels.
struct VarPlanDev {
    int64_t ncb;          // column blocks
    int64_t nfull;        // blocks handled whole in rounds (a multiple of P)
    int P, nbi, ntask;
    int tiles_per_task;   // nbi (nbi + 1) / 2
    int n_splits;
    const int* item_begin;      // [P + 1] into items
    const VarItem* items;
    const int* fin;             // [(ncb - nfull) * ntask][2]: slab slot range of a tail (block, task)
    const VarSplit* splits;
    // the part of the plan ONE launch of k_var executes (set per launch by launch_var: the rounds are dealt to several
    // launches so that the workgroups start each stretch of rounds together; the item list goes with the last one)
    int64_t rnd_begin = 0, rnd_end = 0;
    int with_tail = 1;
};

struct VarPlanHost {
    VarPlanDev d{};
    std::vector<int> item_begin;
    std::vector<VarItem> items;
    std::vector<int> fin;
    std::vector<VarSplit> splits;
    int64_t n_slots = 0;        // slab slots
    int64_t n_vslots = 0;       // vslab slots (partial products)
    int order = 0;              // 0 block-major, 1 sweep-major
    int cut_diag = 0;           // the list was cut with the diagonal tiles divisible (small shares)
    int cohorts = 0, cohort_s = 0, cohort_f = 0;   // the tail was split into cohorts: long sweeps ib >= s (+ the last f tiles of s-1) whole
};

// Quarters [qa, qb) of a diagonal tile, in the cost unit of the full tiles (k4-steps of a SIMD that holds two waves).  Row group g
// has 16 (g + 1) k4-steps in the tile and shares its SIMD with group 7 - g: whole, every SIMD has 144 wave-steps (72); a
// part has what the two groups have inside [32 qa, 32 qb), and a wave that is alone on its SIMD runs at 0.6 of the
// pair's time per step, not 0.5 (1 250 against 2 x 1 024 clocks: profiles/r04_small_n.txt).
inline int var_diag_cost(int qa, int qb) {
    if (qa == 0 && qb == VAR_KQ) return VAR_DIAG_COST;
    int worst = 0;
    for (int g = 0; g < 4; ++g) {
        auto steps = [&](int gg) { const int hi = std::min(16 * (gg + 1), VAR_Q_COST * qb); return std::max(hi - VAR_Q_COST * qa, 0); };
        const int a = steps(g), b = steps(7 - g);
        worst = std::max(worst, std::max((a + b) / 2, (6 * std::max(a, b) + 9) / 10));
    }
    return worst;
}
// k_lo, k_hi in quarter tiles
inline int var_sweep_cost(int ib, int k_lo, int k_hi) {
    const int diag_lo = VAR_KQ * ib;
    const int full_end = k_hi < diag_lo ? k_hi : diag_lo;
    int c = VAR_SWEEP_OVERHEAD;
    if (full_end > k_lo) c += VAR_Q_COST * (full_end - k_lo);
    if (k_hi > diag_lo) c += var_diag_cost((k_lo > diag_lo ? k_lo : diag_lo) - diag_lo, k_hi - diag_lo);
    return c;
}
inline int var_whole_sweep_cost(int ib) { return var_sweep_cost(ib, 0, VAR_KQ * (ib + 1)); }

// ncols = columns of B (queries x columns per query).  order: 0 (or -1) block-major, 1 sweep-major (diagnostic).
inline VarPlanHost build_var_plan(int64_t ncols, int nbi, int ntask, int P, int order = -1) {
    VarPlanHost h;
    VarPlanDev& d = h.d;
    d.ncb = (ncols + VAR_COLS - 1) / VAR_COLS;
    d.P = P; d.nbi = nbi; d.ntask = ntask;
    d.tiles_per_task = nbi * (nbi + 1) / 2;
    d.nfull = d.ncb / P * P;
    const int ncb_t = (int)(d.ncb - d.nfull);
    h.order = order > 0 ? 1 : 0;
    h.item_begin.assign(P + 1, 0);
    h.n_slots = d.nfull * ntask;
    if (ncb_t == 0) return h;

    // ---- the tail's sweeps in execution order
    // k1 >= 0: only the quarter tiles [0, k1) of this sweep are laid out here; [k1, 4 (ib + 1)) went to workgroup post_wg (its
    // item post_idx) as the last partial product of the sweep
    struct Sweep { int cbt, task, ib; int k1 = -1, post_wg = -1, post_idx = -1; };
    std::vector<Sweep> sweeps;
    sweeps.reserve((size_t)ncb_t * ntask * nbi);
    if (h.order == 0) {
        for (int c = 0; c < ncb_t; ++c)
            for (int t = 0; t < ntask; ++t)
                for (int ib = nbi - 1; ib >= 0; --ib) sweeps.push_back(Sweep{c, t, ib});
    } else {
        for (int t = 0; t < ntask; ++t)
            for (int ib = nbi - 1; ib >= 0; --ib)
                for (int c = 0; c < ncb_t; ++c) sweeps.push_back(Sweep{c, t, ib});
    }

    // ---- cut into P ranges of equal cost at tile granularity
    struct Contribution { bool split; int ref; };                        // ref: item index (flush) or split index
    std::vector<std::vector<Contribution>> contrib;
    std::vector<std::vector<VarItem>> wg;
    constexpr int TOL = VAR_Q_COST / 2;
    // diagnostic: GPT_VAR_CUT_TILES=1 cuts at whole tiles (the granularity of rounds 2 and 3) for A/B runs
    const int gran = [] { const char* e = getenv("GPT_VAR_CUT_TILES"); return (e && atoi(e) != 0) ? VAR_KQ : 1; }();
    // diagnostic: GPT_VAR_CUT_DIAG=0 / 1 never / always allows cuts inside a diagonal tile
    const int cut_diag_env = [] { const char* e = getenv("GPT_VAR_CUT_DIAG"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    // lays `sw` end to end over the workgroups [p_begin, p_begin + Pn) in Pn ranges of equal cost; U = total cost to share
    auto cut_range = [&](const std::vector<Sweep>& sw, const int p_begin, const int Pn, const int64_t U) -> int64_t {
        // A diagonal tile is 72 units whole and 32 + 32 + 20 + 20 in quarters: cut only where a whole one is a large part of
        // a workgroup's share — the 14-block tail of configs[1] (N = 1024: 42 tiles for 256 workgroups) took as long as its
        // diagonal tiles, 0.085 ms of a 0.98 ms launch; the lists of the large models (shares of thousands of units) stay as they were.
        const bool cut_diag = cut_diag_env >= 0 ? cut_diag_env != 0 : U / Pn < VAR_DIAG_SHARE;
        h.cut_diag = cut_diag ? 1 : 0;
        int p = 0;
        int64_t cum = 0;
        auto boundary = [&](int q) { return U / Pn * (q + 1) + (U % Pn) * (q + 1) / Pn; };
        std::vector<std::pair<int, int>> parts;                          // (workgroup, index in wg[workgroup]) of the current sweep
        for (const Sweep& s : sw) {
            parts.clear();
            int k = 0;
            const int kend = s.k1 >= 0 ? s.k1 : VAR_KQ * (s.ib + 1);
            const int diag_lo = VAR_KQ * s.ib;
            while (k < kend) {
                const int rem = var_sweep_cost(s.ib, k, kend);
                const int64_t room = boundary(p) - cum;
                int take = kend - k;
                if (p < Pn - 1 && rem > room + TOL) {
                    // quarter tiles that fit into what is left of this workgroup's share (full tiles come first, the diagonal
                    // tile last — whole, unless the shares are small against it: cut_diag); something is left for the next workgroup
                    const int max_take = (cut_diag ? kend - 1 : (kend > diag_lo ? diag_lo : kend - 1)) - k;
                    int nt = 0;
                    while (nt < max_take && var_sweep_cost(s.ib, k, k + nt + 1) <= room + TOL) ++nt;
                    if (!cut_diag || k + nt <= diag_lo) nt = nt / gran * gran;
                    if (nt < 1) { ++p; continue; }                       // nothing fits: close this workgroup
                    take = nt;
                }
                VarItem it{(int)d.nfull + s.cbt, s.task, s.ib, k, k + take, 0, -1, -1};
                parts.emplace_back(p_begin + p, (int)wg[p_begin + p].size());
                wg[p_begin + p].push_back(it);
                cum += var_sweep_cost(s.ib, k, k + take);
                k += take;
                if (k < kend) ++p;                                       // the rest of this sweep belongs to the next one
            }
            if (s.post_wg >= 0) parts.emplace_back(s.post_wg, s.post_idx);
            if (parts.size() > 1) {                                      // cut: every part is a partial product
                VarSplit sp{(int)h.n_vslots, (int)(h.n_vslots + (int64_t)parts.size()), -1};
                for (auto& pr : parts) wg[pr.first][pr.second].vslot = (int)h.n_vslots++;
                contrib[(size_t)s.cbt * ntask + s.task].push_back({true, (int)h.splits.size()});
                h.splits.push_back(sp);
            }
        }
        return cum;
    };
    auto reset = [&]() {
        contrib.assign((size_t)ncb_t * ntask, {});
        wg.assign(P, {});
        h.splits.clear();
        h.n_vslots = 0;
    };
    // Every part of a cut sweep pays the fixed sweep overhead again, so the total to share is only known once the cuts
    // are — and a diagonal tile costs more in parts than whole: the cut is repeated with the total the previous pass produced
    // until a pass stays within the total it was given (the shares are then upper bounds; a pass that exceeds its total leaves
    // the excess to the last workgroup).
    auto cut_settled = [&](std::vector<Sweep>& sw, const int p_begin, const int Pn, auto&& before_each) {
        int64_t U = 0;
        for (const Sweep& x : sw) U += var_sweep_cost(x.ib, 0, x.k1 >= 0 ? x.k1 : VAR_KQ * (x.ib + 1));
        for (int rep = 0; rep < 8; ++rep) {
            reset();
            before_each();                       // (may set the k0 / pre_* fields of sw: the U of the first pass is then an over-estimate, corrected by the repetition)
            const int64_t U2 = cut_range(sw, p_begin, Pn, U);
            if (U2 <= U && (rep > 0 || U2 == U)) break;
            U = U2;
        }
    };

    // ---- Cohorts (block-major order, at least half as many blocks as workgroups, i.e. no block can have two workgroups of
    // its own): workgroup b < ncb_t takes the long sweeps ib = nbi-1 .. s of block b WHOLE — those workgroups start together
    // and do the same work, so they walk the tiles of A in step and an XCD's 32 of them share one copy of that stream in
    // L2, as in the rounds — and the other P - ncb_t workgroups share the short sweeps ib < s of all blocks, cut at tile
    // granularity as below.  s balances the two cohorts.  Why: laid end to end, every workgroup is at a different tile
    // of A at any time and nothing of that stream is shared: at N = 8192, M = 10^4 (157 blocks on 256 workgroups) the launch
    // pulled 157 x 285 MB = 45 GB through L2 in 10.6 ms and ran at the HBM rate, 80 % of its MFMA floor
    // (profiles/r02_small_m.log); the short sweeps only touch the first s (s + 1) / 2 tiles of every task (110 MB at s = 10).
    // Used when ONE boundary s balances the cohorts to 4 % (many i-blocks: N >= 4096 or so); with few i-blocks the list cut
    // tile by tile below balances better and the factor is small enough to stay in cache anyway.
    bool cohorts = h.order == 0 && 2 * ncb_t >= P && ncb_t < P && nbi >= 2;
    const int nB = P - ncb_t;
    // The long cohort also takes the LAST best_f tiles (the diagonal one included) of sweep best_s - 1, as a partial product:
    // what balances the two cohorts to a tile.  The short cohort runs its part [0, best_s - best_f) of that sweep after
    // sweep best_s - 2, whose generating pass has produced every kernel column the part needs.
    int best_s = 1, best_f = 0;
    if (cohorts) {
        int64_t best = -1, all = 0;
        for (int ib = 0; ib < nbi; ++ib) all += var_whole_sweep_cost(ib);
        for (int s = 1; s < nbi; ++s)
            for (int f = 0; f < s; ++f) {
                if (f > 0 && s < 2) continue;                         // the short cohort's part needs a sweep in front of it
                int64_t a = f ? var_sweep_cost(s - 1, VAR_KQ * (s - f), VAR_KQ * s) : 0, b = var_sweep_cost(s - 1, 0, VAR_KQ * (s - f));
                for (int ib = 0; ib < nbi; ++ib)
                    if (ib >= s) a += var_whole_sweep_cost(ib);
                    else if (ib < s - 1) b += var_whole_sweep_cost(ib);
                a *= ntask; b *= ntask;
                const int64_t span = std::max<int64_t>(a, (b * ncb_t + nB - 1) / nB);
                if (best < 0 || span < best) { best = span; best_s = s; best_f = f; }
            }
        const int64_t ideal = (all * ntask * ncb_t + P - 1) / P;
        cohorts = 100 * best <= 104 * ideal;
    }
    h.cohorts = cohorts ? 1 : 0; h.cohort_s = cohorts ? best_s : 0; h.cohort_f = cohorts ? best_f : 0;
    if (cohorts) {
        std::vector<Sweep> shortsw;
        shortsw.reserve((size_t)ncb_t * ntask * best_s);
        std::vector<size_t> cut_at((size_t)ncb_t * ntask, 0);             // index in shortsw of the (c, t, best_s - 1) entry
        for (int c = 0; c < ncb_t; ++c)
            for (int t = 0; t < ntask; ++t) {
                // order: best_s - 2 (generates tiles [0, best_s - 1)), then the part of best_s - 1, then the rest, longest first
                if (best_s >= 2 && best_f > 0) shortsw.push_back(Sweep{c, t, best_s - 2});
                cut_at[(size_t)c * ntask + t] = shortsw.size();
                shortsw.push_back(Sweep{c, t, best_s - 1});
                for (int ib = best_s - 2 - ((best_s >= 2 && best_f > 0) ? 1 : 0); ib >= 0; --ib) shortsw.push_back(Sweep{c, t, ib});
            }
        cut_settled(shortsw, ncb_t, nB, [&]() {
            for (int c = 0; c < ncb_t; ++c)
                for (int t = 0; t < ntask; ++t) {
                    for (int ib = nbi - 1; ib >= best_s; --ib)
                        wg[c].push_back(VarItem{(int)d.nfull + c, t, ib, 0, VAR_KQ * (ib + 1), 0, -1, -1});
                    if (best_f > 0) {
                        Sweep& sp = shortsw[cut_at[(size_t)c * ntask + t]];
                        sp.k1 = VAR_KQ * (best_s - best_f); sp.post_wg = c; sp.post_idx = (int)wg[c].size();
                        wg[c].push_back(VarItem{(int)d.nfull + c, t, best_s - 1, VAR_KQ * (best_s - best_f), VAR_KQ * best_s, 0, -1, -1});
                    }
                }
        });
    } else {
        cut_settled(sweeps, 0, P, []() {});
    }

    // ---- flags and flush groups per workgroup; items concatenated in workgroup order
    for (int q = 0; q < P; ++q) {
        h.item_begin[q] = (int)h.items.size();
        int cur_cb = -1, g_lo = 0, g_hi = 0;                              // generated k-tile extent of the current block
        int grp_cb = -1, grp_task = -1, grp_last = -1;                    // open flush group (whole sweeps of one (block, task))
        auto close_group = [&]() {
            if (grp_last >= 0) contrib[(size_t)(grp_cb - d.nfull) * ntask + grp_task].push_back({false, grp_last});
            grp_last = -1;
        };
        for (VarItem it : wg[q]) {
            if (it.cb != cur_cb) { it.flags |= VI_FIRST; cur_cb = it.cb; g_lo = g_hi = 0; }
            if (it.k_lo < g_lo || it.k_hi > g_hi) {
                it.flags |= VI_GEN;
                if (g_hi > g_lo && it.k_lo <= g_hi && it.k_hi >= g_lo) {   // touches what is there: extend
                    g_lo = it.k_lo < g_lo ? it.k_lo : g_lo;
                    g_hi = it.k_hi > g_hi ? it.k_hi : g_hi;
                } else { g_lo = it.k_lo; g_hi = it.k_hi; }
            }
            if (it.vslot < 0) {
                if (grp_last < 0 || grp_cb != it.cb || grp_task != it.task) {
                    close_group();
                    it.flags |= VI_ZERO;
                    grp_cb = it.cb; grp_task = it.task;
                }
                grp_last = (int)h.items.size();
            }
            h.items.push_back(it);
        }
        close_group();
    }
    h.item_begin[P] = (int)h.items.size();

    // ---- slab slots: the contributions of a (block, task) get consecutive slots
    h.fin.assign((size_t)ncb_t * ntask * 2, 0);
    for (size_t ct = 0; ct < contrib.size(); ++ct) {
        h.fin[2 * ct] = (int)h.n_slots;
        for (const Contribution& c : contrib[ct]) {
            if (c.split) { h.splits[c.ref].slot = (int)h.n_slots; h.n_slots += VAR_SPLIT_SLOTS; }
            else h.items[c.ref].slot = (int)h.n_slots++;
        }
        h.fin[2 * ct + 1] = (int)h.n_slots;
    }
    d.n_splits = (int)h.splits.size();
    return h;
}

}  // namespace gpt
