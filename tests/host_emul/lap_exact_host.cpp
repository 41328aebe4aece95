// tests/host_emul/lap_exact_host.cpp -- TEST INFRASTRUCTURE.  Compiles the product's exact-tie code
// (bot-sort-onnx-tensorrt_amd/csrc/lap_exact.h, the very source the device kernel k_lap_exact instantiates) for the CPU with an
// executor that runs the threads of a phase one after the other -- in ascending or in descending thread order, so that a
// hidden dependence between threads of one phase shows up as a difference.  tests/test_lap_exact_host.py drives it against
// oracle/lapjv_oracle.c.  Nothing in the product loads this library.
#include "../../bot-sort-onnx-tensorrt_amd/csrc/lap_exact.h"
#include "host_ex.h"
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <utility>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#endif

namespace {

struct DenseSrc {
    static constexpr bool patched = false;
    const double *c; int ld;
    double real(int i, int j) const { return c[(size_t)i * ld + j]; }
};
struct PatchedSrc {           // the tracker's situation: `base` = the matrix without the fused appearance term, edges override it
    static constexpr bool patched = true;
    const double *c; int ld;
    double real(int i, int j) const { return c[(size_t)i * ld + j]; }
};

}  // namespace

// cost[N, M] (ld) -> feasible edges (<= L); x[N], y[M]: in = some optimal visible assignment, out = the exact-tie step's.
// base != NULL: run with the patched source (non-edges read from base, edges from the list built out of cost).
// info[0] = certificate said "tied", [1] = error flags, [2] = replay ran, [3] = edges.  shuffle: edge list order seed.
extern "C" int lapx_host_exact(int N, int M, const double *cost, int ld, double L, int32_t *x, int32_t *y, int nth, int order,
                               int force_replay, const double *base, unsigned shuffle, int32_t *info) {
    std::vector<int32_t> er, ec; std::vector<double> ev;
    std::vector<int32_t> orow, ocol; std::vector<double> oval;       // overrides: entries above L that differ from the base matrix
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j) {
            const double c = cost[(size_t)i * ld + j];
            if (c <= L) { er.push_back(i); ec.push_back(j); ev.push_back(c); }
            else if (base && c != base[(size_t)i * M + j]) { orow.push_back(i); ocol.push_back(j); oval.push_back(c); }
        }
    // the device appends edges in no particular order: permute them (deterministically), and sprinkle vacant slots
    unsigned s = shuffle * 2654435761u + 12345u;
    for (size_t k = er.size(); shuffle && k > 1; --k) {
        s = s * 1664525u + 1013904223u;
        const size_t q = (s >> 8) % k;
        std::swap(er[k - 1], er[q]); std::swap(ec[k - 1], ec[q]); std::swap(ev[k - 1], ev[q]);
    }
    if (shuffle & 1) { er.push_back(-1); ec.push_back(0); ev.push_back(0.0); }
    const long E = (long)er.size();
    // the device keeps the overrides at the TOP END of the same arrays (filled downwards): same shape here, with a hole between
    const long O = (long)orow.size(), hole = 3;
    for (long k = 0; k < hole; ++k) { er.push_back(-7); ec.push_back(-7); ev.push_back(-7.0); }
    for (long k = O - 1; k >= 0; --k) { er.push_back(orow[k]); ec.push_back(ocol[k]); ev.push_back(oval[k]); }
    lapx::Work W;
    memset(&W, 0, sizeof(W));
    // scratch carved WITH GAPS behind every array: sentinel pattern (checked after the call) + ASan poison when built with it
    const int GAP = 6;
    std::vector<double> dbl(lapx::work_doubles(N, M, E + O, nth, GAP), -777.25);
    std::vector<int32_t> ints(lapx::work_ints(N, M, E + O, nth, GAP), (int32_t)0x5e471e1);
    if (!lapx::work_carve(W, N, M, E + O, nth, dbl.data(), dbl.size(), ints.data(), ints.size(), GAP)) return -1;
    std::vector<std::pair<void *, size_t>> gaps;
    {
        const size_t n = (size_t)N + M;
        double *dp[] = {W.ce_cost, W.p0, W.p1, W.cm, W.v, W.d, W.w, W.shd, W.rv1, W.rv2};
        const size_t dn[] = {(size_t)(E + O), (size_t)M, (size_t)M, (size_t)N, n, n, n, (size_t)lapx::D_SIZE, (size_t)nth, (size_t)nth};
        for (int k = 0; k < 10; ++k) gaps.push_back({dp[k] + dn[k], GAP * sizeof(double)});
        int32_t *ip[] = {W.rp, W.rcur, W.ce_col, W.alive, W.has, W.xx, W.yy, W.cols, W.pred, W.freer, W.tmpa, W.tmpb, W.flag, W.runs, W.uniq, W.sh, W.rj1, W.rj2, W.ri};
        const size_t in[] = {(size_t)N + 2, (size_t)N + 1, (size_t)(E + O), (size_t)N + 1, (size_t)N + 1, n, n, n, n, n, n, n, n, n, n, (size_t)lapx::S_SIZE, (size_t)nth, (size_t)nth, (size_t)nth};
        for (int k = 0; k < 19; ++k) gaps.push_back({ip[k] + in[k], GAP * sizeof(int32_t)});
    }
#if defined(__SANITIZE_ADDRESS__)
    for (auto &g : gaps) __asan_poison_memory_region(g.first, g.second);
#endif
    W.N = N; W.M = M; W.L = L; W.e_row = er.data(); W.e_col = ec.data(); W.e_cost = ev.data(); W.E = (int)E;
    W.O = (int)O; W.o_base = E + hole;
    W.x = x; W.y = y; W.max_steps = 50000000L;
    HostEx ex{nth, order};
    lapx::build_csr(ex, W);
    lapx::tie_check(ex, W);
    info[0] = W.sh[lapx::S_TIED]; info[3] = (int)(E + (O << 16));
    info[2] = 0;
    if (info[0] || force_replay) {
        info[2] = 1;
        if (base) { PatchedSrc S{base, M}; lapx::jv_replay(ex, W, S); }
        else { DenseSrc S{cost, ld}; lapx::jv_replay(ex, W, S); }
    }
    info[1] = W.sh[lapx::S_ERR];
#if defined(__SANITIZE_ADDRESS__)
    for (auto &g : gaps) __asan_unpoison_memory_region(g.first, g.second);
#endif
    for (auto &g : gaps) {           // nothing may have written behind the end of a scratch array
        const unsigned char *b = (const unsigned char *)g.first;
        const bool is_dbl = g.second == GAP * sizeof(double);
        for (int k = 0; k < GAP; ++k)
            if (is_dbl ? ((const double *)b)[k] != -777.25 : ((const int32_t *)b)[k] != (int32_t)0x5e471e1) return -2;
    }
    return 0;
}
