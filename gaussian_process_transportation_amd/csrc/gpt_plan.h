// Work decomposition of the variance kernel (host-only C++, no HIP: also compiled into the CPU sanitizer build
// and exercised by tests/test_host_cpu.py through gpt_debug_var_plan).
//
// The product V = A B has A = the stack of `ntask` block-lower-triangular inverse factors (nbi x nbi tiles of
// 512 x 512 each; ntask = 1 for the exact GP, T for the SVGP exact-conversion model) and B = the generated kernel
// columns, 64 per column block.  A *sweep* is (column block cb, task, i-block ib, k tiles [k_lo, k_hi)) with
// k_hi <= ib + 1; a whole sweep has k_lo = 0, k_hi = ib + 1 and its 512 x 64 result is folded straight into the
// per-column sums of squares.
//
//   * Rounds: while at least P (= workgroups = CUs) column blocks are left, workgroup p takes block r P + p whole, all
//     tasks, longest sweep first — every workgroup then walks the same tiles of A at the same time, which is what
//     keeps the A stream in L2.  Rounds need no list: the kernel derives them from (round, blockIdx).
//   * Tail (and every launch with fewer than P blocks — the reference's own batch sizes, M = 400 .. 10^4): the
//     remaining sweeps are laid end to end, costed, and cut into P ranges of equal cost AT TILE GRANULARITY.  A sweep
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
#include <vector>

namespace gpt {

constexpr int VAR_COLS = 64;           // columns per column block
constexpr int VAR_ROWS = 512;          // rows of an i-block (= WT of gpt_common.h)
constexpr int VAR_SLOT = 2 * VAR_COLS;             // elements per slab slot: ssq[64], crs[64]
constexpr int VAR_VSLOT = VAR_ROWS * VAR_COLS;     // elements per vslab slot: one 512 x 64 partial product
constexpr int VAR_DIAG_COST = 72;      // k4-steps a diagonal tile costs a SIMD (waves g and 7-g: 16 (g+1) + 16 (8-g), halved)
constexpr int VAR_TILE_COST = 128;     // k4-steps of a full tile
constexpr int VAR_SWEEP_OVERHEAD = 6;  // fixed cost of a sweep in k4-steps (first fill, accumulator fold)

struct VarItem {
    int cb, task, ib;     // column block, task, i-block
    int k_lo, k_hi;       // k tiles [k_lo, k_hi), k_hi <= ib + 1
    int flags;            // VI_*
    int slot;             // >= 0: after this item the running column sums go to slab slot `slot`
    int vslot;            // >= 0: partial product, stored to vslab slot `vslot` instead of being folded
};
enum { VI_GEN = 1,        // B fragments of [k_lo, k_hi) are generated (and copied to the scratch image), not reloaded
       VI_FIRST = 2,      // first item of this workgroup for this column block: the scratch image changes owner
       VI_ZERO = 4 };     // the running column sums start from zero

struct VarSplit { int v_begin, v_end, slot; };   // combine vslab slots [v_begin, v_end) into slab slot `slot`

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
};

inline int var_sweep_cost(int ib, int k_lo, int k_hi) {
    const int full_end = k_hi < ib ? k_hi : ib;
    int c = VAR_SWEEP_OVERHEAD;
    if (full_end > k_lo) c += VAR_TILE_COST * (full_end - k_lo);
    if (k_hi == ib + 1) c += VAR_DIAG_COST;
    return c;
}

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
    struct Sweep { int cbt, task, ib; };
    std::vector<Sweep> sweeps;
    sweeps.reserve((size_t)ncb_t * ntask * nbi);
    if (h.order == 0) {
        for (int c = 0; c < ncb_t; ++c)
            for (int t = 0; t < ntask; ++t)
                for (int ib = nbi - 1; ib >= 0; --ib) sweeps.push_back({c, t, ib});
    } else {
        for (int t = 0; t < ntask; ++t)
            for (int ib = nbi - 1; ib >= 0; --ib)
                for (int c = 0; c < ncb_t; ++c) sweeps.push_back({c, t, ib});
    }
    int64_t U0 = 0;
    for (const Sweep& s : sweeps) U0 += var_sweep_cost(s.ib, 0, s.ib + 1);

    // ---- cut into P ranges of equal cost at tile granularity
    struct Contribution { bool split; int ref; };                        // ref: item index (flush) or split index
    std::vector<std::vector<Contribution>> contrib;
    std::vector<std::vector<VarItem>> wg;
    constexpr int TOL = VAR_TILE_COST / 2;
    // Every part of a cut sweep pays the fixed sweep overhead again, so the total to share is only known once the cuts
    // are: the cut is repeated with the total the previous pass produced (settles after one repetition).
    auto cut = [&](const int64_t U) -> int64_t {
        contrib.assign((size_t)ncb_t * ntask, {});
        wg.assign(P, {});
        h.splits.clear();
        h.n_vslots = 0;
        int p = 0;
        int64_t cum = 0;
        auto boundary = [&](int q) { return U / P * (q + 1) + (U % P) * (q + 1) / P; };
        std::vector<std::pair<int, int>> parts;                          // (workgroup, index in wg[workgroup]) of the current sweep
        for (const Sweep& s : sweeps) {
            parts.clear();
            int k = 0;
            const int kend = s.ib + 1;
            while (k < kend) {
                const int rem = var_sweep_cost(s.ib, k, kend);
                const int64_t room = boundary(p) - cum;
                int take = kend - k;
                if (p < P - 1 && rem > room + TOL) {
                    // tiles that fit into what is left of this workgroup's share (full tiles come first, the diagonal last)
                    int64_t nt = (room - VAR_SWEEP_OVERHEAD + TOL) / VAR_TILE_COST;
                    if (nt > kend - k - 1) nt = kend - k - 1;            // leave something for the next workgroup
                    if (nt < 1) { ++p; continue; }                       // nothing fits: close this workgroup
                    take = (int)nt;
                }
                VarItem it{(int)d.nfull + s.cbt, s.task, s.ib, k, k + take, 0, -1, -1};
                parts.emplace_back(p, (int)wg[p].size());
                wg[p].push_back(it);
                cum += var_sweep_cost(s.ib, k, k + take);
                k += take;
                if (k < kend) ++p;                                       // the rest of this sweep belongs to the next one
            }
            if (parts.size() > 1) {                                      // cut: every part is a partial product
                VarSplit sp{(int)h.n_vslots, (int)(h.n_vslots + (int64_t)parts.size()), -1};
                for (auto& pr : parts) wg[pr.first][pr.second].vslot = (int)h.n_vslots++;
                contrib[(size_t)s.cbt * ntask + s.task].push_back({true, (int)h.splits.size()});
                h.splits.push_back(sp);
            }
        }
        return cum;
    };
    int64_t U = cut(U0);
    for (int rep = 0; rep < 3; ++rep) {
        const int64_t U2 = cut(U);
        if (U2 == U) break;
        U = U2;
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
            if (c.split) h.splits[c.ref].slot = (int)h.n_slots++;
            else h.items[c.ref].slot = (int)h.n_slots++;
        }
        h.fin[2 * ct + 1] = (int)h.n_slots;
    }
    d.n_splits = (int)h.splits.size();
    return h;
}

}  // namespace gpt
