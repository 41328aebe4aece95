// tests/host_emul/lap_exact_host.cpp -- TEST INFRASTRUCTURE.  Compiles the product's exact-tie code
// (bot-sort-onnx-tensorrt_amd/csrc/lap_exact.h, the very source the device kernel k_lap_exact instantiates) for the CPU with an
// executor that runs the threads of a phase one after the other -- in ascending or in descending thread order, so that a
// hidden dependence between threads of one phase shows up as a difference.  tests/test_lap_exact_host.py drives it against
// oracle/lapjv_oracle.c.  Nothing in the product loads this library.
#include "../../bot-sort-onnx-tensorrt_amd/csrc/lap_exact.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

struct HostEx {
    int n, order;
    int nth() const { return n; }
    template <class F> void phase(F f) {
        if (order == 0) for (int t = 0; t < n; ++t) f(t);
        else for (int t = n - 1; t >= 0; --t) f(t);
    }
    int atomic_add(int32_t *p, int v) { const int o = *p; *p = o + v; return o; }
    int atomic_max(int32_t *p, int v) { const int o = *p; if (v > o) *p = v; return o; }
};

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
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j)
            if (cost[(size_t)i * ld + j] <= L) { er.push_back(i); ec.push_back(j); ev.push_back(cost[(size_t)i * ld + j]); }
    // the device appends edges in no particular order: permute them (deterministically), and sprinkle vacant slots
    unsigned s = shuffle * 2654435761u + 12345u;
    for (size_t k = er.size(); shuffle && k > 1; --k) {
        s = s * 1664525u + 1013904223u;
        const size_t q = (s >> 8) % k;
        std::swap(er[k - 1], er[q]); std::swap(ec[k - 1], ec[q]); std::swap(ev[k - 1], ev[q]);
    }
    if (shuffle & 1) { er.push_back(-1); ec.push_back(0); ev.push_back(0.0); }
    const long E = (long)er.size();
    lapx::Work W;
    memset(&W, 0, sizeof(W));
    std::vector<double> dbl(lapx::work_doubles(N, M, E, nth));
    std::vector<int32_t> ints(lapx::work_ints(N, M, E, nth));
    if (!lapx::work_carve(W, N, M, E, nth, dbl.data(), dbl.size(), ints.data(), ints.size())) return -1;
    W.N = N; W.M = M; W.L = L; W.e_row = er.data(); W.e_col = ec.data(); W.e_cost = ev.data(); W.E = (int)E;
    W.x = x; W.y = y; W.max_steps = 50000000L;
    HostEx ex{nth, order};
    lapx::build_csr(ex, W);
    lapx::tie_check(ex, W);
    info[0] = W.sh[lapx::S_TIED]; info[3] = (int)E;
    info[2] = 0;
    if (info[0] || force_replay) {
        info[2] = 1;
        if (base) { PatchedSrc S{base, M}; lapx::jv_replay(ex, W, S); }
        else { DenseSrc S{cost, ld}; lapx::jv_replay(ex, W, S); }
    }
    info[1] = W.sh[lapx::S_ERR];
    return 0;
}
