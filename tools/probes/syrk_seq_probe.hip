// The trailing-update GEMMs of the blocked Cholesky, without the panel steps: the same launch sequence potrf_groups
// issues (thin update of a panel's own columns by the group's earlier panels, bulk update of everything behind the
// group), timed per launch and in total, for a choice of grouping and tile size (env GPT_POTRF_GROUP,
// GPT_GEMM_TS64_BELOW, GPT_SYRK_*).  Flops are counted on the block lower triangle (what the kernels compute).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/syrk_seq_probe.hip -o tools/probes/syrk_seq_probe
// usage: syrk_seq_probe [NP=8192] [reps=5] [verbose=0]
#include "../../gaussian_process_transportation_amd/csrc/gpt_fit.hip"
#include <cstdio>
#include <vector>
using namespace gpt;

struct Call { int row0, ncols, kcol0, kw; bool bulk; };

int main(int argc, char** argv) {
    const int NP = argc > 1 ? atoi(argv[1]) : 8192;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int verbose = argc > 3 ? atoi(argv[3]) : 0;
    double* K;
    const size_t bytes = (size_t)NP * NP * sizeof(double);
    hipMalloc(&K, bytes);
    {   // small random-ish values so the data is not all zero (clock behaviour) and the repeated updates stay finite
        std::vector<double> h((size_t)NP * NP);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0 * 1e-3; }
        hipMemcpy(K, h.data(), bytes, hipMemcpyHostToDevice);
    }
    const int nb = NP / NB, ob = potrf_outer_blocks(), grp = potrf_group(NP), gw = grp * ob;
    if (argc > 5) {      // single update, repeated: syrk_seq_probe NP reps verbose rem K   (PMC passes)
        const int rem = atoi(argv[4]), kw = atoi(argv[5]);
        hipStream_t s1; hipStreamCreate(&s1);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        syrk_update(s1, K, NP, NP - rem, rem, NP - rem - kw, kw);
        hipStreamSynchronize(s1);
        hipEventRecord(a, s1);
        for (int r = 0; r < reps; ++r) syrk_update(s1, K, NP, NP - rem, rem, NP - rem - kw, kw);
        hipEventRecord(b, s1); hipStreamSynchronize(s1);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("single update rem=%d K=%d: %.1f us, %.1f TF\n", rem, kw, ms * 1e3 / reps, (double)rem * (rem + 64) * kw / (ms * 1e-3 / reps) / 1e12);
        return 0;
    }
    std::vector<Call> calls;
    for (int g0 = 0; g0 < nb; g0 += gw) {
        const int gend = g0 + gw < nb ? g0 + gw : nb;
        for (int p0 = g0; p0 < gend; p0 += ob) {
            const int pend = p0 + ob < gend ? p0 + ob : gend;
            if (p0 > g0) calls.push_back({p0 * NB, (pend - p0) * NB, g0 * NB, (p0 - g0) * NB, false});
        }
        if (NP - gend * NB > 0) calls.push_back({gend * NB, NP - gend * NB, g0 * NB, (gend - g0) * NB, true});
    }
    hipStream_t s; hipStreamCreate(&s);
    auto issue = [&](const Call& c) { syrk_update(s, K, NP, c.row0, c.ncols, c.kcol0, c.kw); };
    std::vector<hipEvent_t> ev(calls.size() + 1);
    for (auto& e : ev) hipEventCreate(&e);
    for (const Call& c : calls) issue(c);            // warm-up
    hipStreamSynchronize(s);
    std::vector<double> us(calls.size(), 1e30);
    double best_total = 1e30;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(ev[0], s);
        for (size_t i = 0; i < calls.size(); ++i) { issue(calls[i]); hipEventRecord(ev[i + 1], s); }
        hipStreamSynchronize(s);
        float ms;
        for (size_t i = 0; i < calls.size(); ++i) { hipEventElapsedTime(&ms, ev[i], ev[i + 1]); if (ms * 1e3 < us[i]) us[i] = ms * 1e3; }
        hipEventElapsedTime(&ms, ev[0], ev[calls.size()]);
        if (ms < best_total) best_total = ms;
    }
    // back-to-back without events in between
    hipEventRecord(ev[0], s);
    for (int r = 0; r < reps; ++r) for (const Call& c : calls) issue(c);
    hipEventRecord(ev[1], s);
    hipStreamSynchronize(s);
    float ms_b2b; hipEventElapsedTime(&ms_b2b, ev[0], ev[1]); ms_b2b /= reps;
    double flop_bulk = 0, flop_thin = 0, us_bulk = 0, us_thin = 0;
    for (size_t i = 0; i < calls.size(); ++i) {
        const Call& c = calls[i];
        const double rem = NP - c.row0;
        // lower block triangle at 64-granularity: ncols x ncols triangle (incl. diagonal blocks) + rectangle below
        const double nc = c.ncols < rem ? c.ncols : rem;
        const double elems = nc * (nc + 64) / 2 + (rem - nc) * nc;
        const double fl = 2.0 * elems * c.kw;
        (c.bulk ? flop_bulk : flop_thin) += fl;
        (c.bulk ? us_bulk : us_thin) += us[i];
        if (verbose) printf("%s row0=%5d ncols=%5d K=%4d: %7.1f us  %5.1f TF\n", c.bulk ? "bulk" : "thin", c.row0, c.ncols, c.kw, us[i], fl / us[i] / 1e6);
    }
    printf("NP=%d grp=%d ob=%d ts64_below=%d: %zu launches, total %.3f ms (events between) / %.3f ms (back to back); bulk %.3f ms %.1f TF; thin %.3f ms %.1f TF; all %.1f TF\n",
           NP, grp, ob, gemm_tile_threshold(), calls.size(), best_total, ms_b2b, us_bulk / 1e3, flop_bulk / us_bulk / 1e6, us_thin / 1e3,
           us_thin > 0 ? flop_thin / us_thin / 1e6 : 0.0, (flop_bulk + flop_thin) / (ms_b2b * 1e-3) / 1e12);
    return 0;
}
