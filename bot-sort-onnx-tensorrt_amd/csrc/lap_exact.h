// lap_exact.h -- exact-tie mode of the assignment stage (bts_tracker_config.tie_mode = 1, bts_lapjv_exact).
//
// The fast solver (lap.hip) returns AN optimum of  lap.lapjv(cost, extend_cost=True, cost_limit=L)  (reference call site
// D:1682-1693); it equals the reference's result whenever the optimum is unique.  On exact cost ties the reference's result
// is whatever the dense Jonker-Volgenant code of lap 0.4.0 does on the (N+M) x (N+M) extended matrix -- an artefact of its
// scan orders that no sparse solver reproduces by construction.  This file closes that gap in two steps, both run by ONE
// workgroup behind the fast solver:
//
//   1. tie_check: a CERTIFICATE that the visible assignment (x, y) the fast solver produced is the only one within
//      TIE_TOL of the optimum.  Optimal duals are rebuilt from the assignment itself (difference constraints on the column
//      prices, Bellman-Ford sweeps in Jacobi form), the near-tight edges form a digraph on {matched rows} + Z (Z = "the
//      padding": every unmatched row / free column / dummy of the extended problem), and another optimum exists iff that
//      digraph has a cycle (alternating cycle, or alternating path between two padding ends = a cycle through Z).
//      No cycle  =>  every exact solver, the dense JV included, returns this (x, y): done.
//   2. jv_replay (only when the certificate fails): the dense JV itself -- column reduction, reduction transfer, two sweeps
//      of augmenting row reduction, shortest augmenting paths with lap's ready / scan / todo column lists -- executed phase
//      by phase on the IMPLICIT extended matrix (real block from the cost source, padding L/2 and 0), with every scan
//      order, strictness and swap of lap 0.4.0's lapjv.cpp reproduced (the tests compare it with the CPU checker's restatement of
//      that code).
//      Data-parallel inside a scan, sequential across scans like the original; its column-list bookkeeping (the order of
//      `cols[]` decides ties) is done in closed form per RUN of consecutive list insertions (a run of r insertions rotates
//      the examined todo segment by r -- see apply_runs).
//
// The code is written against an EXECUTOR: `ex.phase(f)` runs f(tid) for every thread of the workgroup between two
// barriers; code outside phases is uniform (it only reads what earlier phases left in memory).  The device executor is
// __syncthreads; tests/host_emul compiles this very file for the CPU with an executor that runs the threads of a phase one
// after the other (in either order), and compares (x, y) with the CPU checker's dense JV bit for bit on tie families
// (tests/test_lap_exact_host.py).  No thread keeps private state across phases, so the emulation is faithful.
#pragma once
#include <stdint.h>
#include <float.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LAPX_HD __host__ __device__ __forceinline__
#else
#define LAPX_HD inline
#endif

namespace lapx {

constexpr double TIE_TOL = 1e-9;         // slack below which a non-matching edge counts as tight (costs live in [0, 1])
constexpr double BF_EPS = 1e-12;         // smallest price improvement a Bellman-Ford sweep follows
constexpr double BIGD = DBL_MAX;         // lap's LARGE
constexpr int NONE = 0x7fffffff;

// shared scalars (Work::sh) -----------------------------------------------------------------------------------------------
enum { S_TIED = 0, S_CHANGED, S_ROUND, S_NFREE, S_CUR, S_NNEW, S_LO, S_HI, S_FINAL, S_NREADY, S_NRUN, S_ERR, S_STEPS,
       S_CNT, S_A, S_B, S_C, S_D, S_SIZE = 32 };
enum { D_MIND = 0, D_H, D_A, D_SIZE = 8 };
enum { ERR_STEPS = 1, ERR_BF = 2 };

struct Work {
    int N, M;                 // real rows / columns
    double L;                 // cost_limit
    // feasible edges (cost <= L), unsorted, at [0, E); entries with row < 0 are vacant.  OVERRIDES at [o_base, o_base + O) of the
    // same arrays: entries of the cost matrix ABOVE L that are not what the cost source says (face features: a pair whose face
    // passes the appearance gate keeps its body distance even above the match threshold, D:1543-1554).  They never take
    // part in the certificate (nothing above L is a feasible edge); the replay reads them like edges.
    const int32_t *e_row, *e_col;
    const double *e_cost;
    int E, O;
    long o_base;
    int32_t *x, *y;           // visible assignment, in: the fast solver's, out: the reference's
    // scratch, sizes in []; n = N + M
    int32_t *rp;              // [N + 2] row CSR of the edges
    int32_t *rcur;            // [N + 1]
    int32_t *ce_col;          // [E]
    double *ce_cost;          // [E]
    double *p0, *p1;          // [M] column prices (tie_check)
    double *cm;               // [N] cost of a row's matched edge
    int32_t *alive, *has;     // [N + 1]
    double *v, *d, *w;        // [n]
    int32_t *xx, *yy, *cols, *pred, *freer, *tmpa, *tmpb, *flag, *runs, *uniq;   // [n] each
    int32_t *sh;              // [S_SIZE]
    double *shd;              // [D_SIZE]
    double *rv1, *rv2;        // [nth] reduction scratch
    int32_t *rj1, *rj2, *ri;  // [nth]
    long max_steps;           // watchdog: phases a call may run before it gives up (sets S_ERR)
};

// elements of scratch for a problem of the given capacities (host side sizing).  `gap`: unused elements left behind every
// array -- 0 in the product; the CPU emulation harness carves with gaps, poisons them (AddressSanitizer) and checks a
// sentinel pattern in them after every call, so that an index running off the end of one scratch array cannot hide in its
// neighbour.
constexpr int N_DBL_ARRAYS = 10, N_INT_ARRAYS = 19;
LAPX_HD size_t work_doubles(int N, int M, long E, int nth, int gap = 0) {
    return (size_t)E + 2 * (size_t)M + (size_t)N + 3 * (size_t)(N + M) + D_SIZE + 2 * (size_t)nth + (size_t)gap * N_DBL_ARRAYS;
}
LAPX_HD size_t work_ints(int N, int M, long E, int nth, int gap = 0) {
    return (size_t)(N + 2) + (size_t)(N + 1) + (size_t)E + 2 * (size_t)(N + 1) + 10 * (size_t)(N + M) + S_SIZE + 3 * (size_t)nth + (size_t)gap * N_INT_ARRAYS;
}

// carve a Work out of two blocks; returns false if either block is too small
LAPX_HD bool work_carve(Work &W, int N, int M, long E, int nth, double *dbl, size_t ndbl, int32_t *ints, size_t nints, int gap = 0) {
    if (ndbl < work_doubles(N, M, E, nth, gap) || nints < work_ints(N, M, E, nth, gap)) return false;
    const size_t n = (size_t)N + M;
    int nd = 0, ni = 0;
    auto D = [&](double *&field, size_t count) { field = dbl; dbl += count + gap; ++nd; };
    auto I = [&](int32_t *&field, size_t count) { field = ints; ints += count + gap; ++ni; };
    D(W.ce_cost, E); D(W.p0, M); D(W.p1, M); D(W.cm, N); D(W.v, n); D(W.d, n); D(W.w, n); D(W.shd, D_SIZE); D(W.rv1, nth); D(W.rv2, nth);
    I(W.rp, N + 2); I(W.rcur, N + 1); I(W.ce_col, E); I(W.alive, N + 1); I(W.has, N + 1);
    I(W.xx, n); I(W.yy, n); I(W.cols, n); I(W.pred, n); I(W.freer, n); I(W.tmpa, n); I(W.tmpb, n); I(W.flag, n); I(W.runs, n); I(W.uniq, n);
    I(W.sh, S_SIZE); I(W.rj1, nth); I(W.rj2, nth); I(W.ri, nth);
    return nd == N_DBL_ARRAYS && ni == N_INT_ARRAYS;
}

// ---- small helpers -------------------------------------------------------------------------------------------------------
LAPX_HD bool lexless(double a, int ja, double b, int jb) { return a < b || (a == b && ja < jb); }
struct Top2 { double v1; int j1; double v2; int j2; };
LAPX_HD void top2_init(Top2 &t) { t.v1 = BIGD; t.j1 = NONE; t.v2 = BIGD; t.j2 = NONE; }
// (value, index) pairs in ANY order: the two lexicographically smallest.  Equals lap's sequential scan
//   if (c < v2) { if (c >= v1) { v2 = c; j2 = j; } else { v2 = v1; j2 = j1; v1 = c; j1 = j; } }       over j ascending
// (first index of the minimum; first index, other than j1, of the smallest remaining value).
LAPX_HD void top2_push(Top2 &t, double c, int j) {
    if (lexless(c, j, t.v1, t.j1)) { t.v2 = t.v1; t.j2 = t.j1; t.v1 = c; t.j1 = j; }
    else if (lexless(c, j, t.v2, t.j2)) { t.v2 = c; t.j2 = j; }
}

// contiguous share of [0, len) of thread t
LAPX_HD void chunk(int len, int nth, int t, int &lo, int &hi) {
    const int per = (len + nth - 1) / nth;
    lo = t * per < len ? t * per : len;
    hi = lo + per < len ? lo + per : len;
}

// the watchdog: every phase counts; past max_steps the call gives up loudly (S_ERR) instead of hanging the queue
template <class Ex> LAPX_HD bool alive_steps(Ex &ex, Work &W, long &steps) { return ++steps <= W.max_steps; }

// Ordered compaction of the k in [0, len) with pred(k) into out[], count to sh[S_CNT]: three phases.
template <class Ex, class Pred> LAPX_HD int compact(Ex &ex, Work &W, int len, int32_t *out, Pred pred) {
    const int nth = ex.nth();
    ex.phase([&](int t) { int lo, hi, c = 0; chunk(len, nth, t, lo, hi); for (int k = lo; k < hi; ++k) c += pred(k) ? 1 : 0; W.ri[t] = c; });
    ex.phase([&](int t) { if (t == 0) { int run = 0; for (int q = 0; q < nth; ++q) { const int c = W.ri[q]; W.ri[q] = run; run += c; } W.sh[S_CNT] = run; } });
    ex.phase([&](int t) { int lo, hi; chunk(len, nth, t, lo, hi); int pos = W.ri[t]; for (int k = lo; k < hi; ++k) if (pred(k)) out[pos++] = k; });
    return W.sh[S_CNT];
}

// ---- the row CSR of the feasible edges (both steps use it) ---------------------------------------------------------------
template <class Ex> LAPX_HD void build_csr(Ex &ex, Work &W) {
    const int nth = ex.nth(), N = W.N, E = W.E, EO = W.E + W.O;
    auto at = [&](int k) -> long { return k < E ? (long)k : W.o_base + (k - E); };      // list position k -> array index
    ex.phase([&](int t) { for (int i = t; i <= N + 1; i += nth) W.rp[i] = 0; });
    ex.phase([&](int t) { for (int k = t; k < EO; k += nth) { const int i = W.e_row[at(k)]; if (i >= 0) ex.atomic_add(&W.rp[i + 2], 1); } });
    // rp[i + 2] holds deg(i): an exclusive scan leaves rp[i + 1] = start of row i (two-level: a chunk of rows per thread --
    // the certificate runs every frame in tie_mode 1, so no phase walks all rows on one thread)
    ex.phase([&](int t) { int lo, hi, c = 0; chunk(N, nth, t, lo, hi); for (int i = lo; i < hi; ++i) c += W.rp[i + 2]; W.ri[t] = c; });
    ex.phase([&](int t) { if (t == 0) { int run = 0; for (int q = 0; q < nth; ++q) { const int c = W.ri[q]; W.ri[q] = run; run += c; } W.sh[S_CNT] = run; } });
    // starts go to rcur first (they double as the fill cursors): rp still holds the degrees other threads are reading
    ex.phase([&](int t) { int lo, hi; chunk(N, nth, t, lo, hi); int run = W.ri[t]; for (int i = lo; i < hi; ++i) { W.rcur[i] = run; run += W.rp[i + 2]; } });
    ex.phase([&](int t) { for (int i = t; i < N; i += nth) W.rp[i + 1] = W.rcur[i]; if (t == 0) { W.rp[0] = 0; W.rp[N + 1] = W.sh[S_CNT]; } });
    ex.phase([&](int t) {
        for (int k = t; k < EO; k += nth) {
            const long e = at(k);
            const int i = W.e_row[e];
            if (i < 0) continue;
            const int pos = ex.atomic_add(&W.rcur[i], 1);
            W.ce_col[pos] = W.e_col[e]; W.ce_cost[pos] = W.e_cost[e];
        }
    });
    // row i's entries (edges and overrides): [rp[i + 1], rp[i + 2])   (order inside a row is arbitrary: nothing below depends on it)
}
LAPX_HD int row_lo(const Work &W, int i) { return W.rp[i + 1]; }
LAPX_HD int row_hi(const Work &W, int i) { return W.rp[i + 2]; }

// ======================================================================================================= 1. the certificate
// Formulation (lap.hip's): minimise sum(matched cost) + L * (#unmatched rows); columns may stay free at no cost.
// Duals: u_i + v_j <= c_ij, u_i <= L, v_j <= 0; tight on matched edges, u = L on unmatched rows, v = 0 on free columns.
// With prices p_j = -v_j >= 0:  u_i = c_i,x(i) + p_x(i)  and for a matched row i and each of its other edges (i, j):
//     p_x(i) <= (c_ij - c_i,x(i)) + p_j ,   p_x(i) <= L - c_i,x(i)          (difference constraints; p = 0 on free columns)
// Their MAXIMAL solution (shortest distances from the zero node) is a valid optimal dual whenever (x, y) is optimal.
// Returns through sh[S_TIED]: 0 = (x, y) is the unique optimum (to TIE_TOL), 1 = another one may exist (or the sweeps did
// not settle: conservative).
template <class Ex> LAPX_HD void tie_check(Ex &ex, Work &W) {
    const int nth = ex.nth(), N = W.N, M = W.M;
    const double L = W.L;
    long steps = 0;
    ex.phase([&](int t) {
        if (t == 0) { W.sh[S_TIED] = 0; W.sh[S_ERR] = 0; }
        for (int j = t; j < M; j += nth) W.p0[j] = W.y[j] >= 0 ? BIGD : 0.0;
        for (int i = t; i < N; i += nth) W.cm[i] = 0.0;
    });
    // matched edges: their cost, and the price cap L - c of their column
    ex.phase([&](int t) {
        for (int i = t; i < N; i += nth) {
            const int j = W.x[i];
            if (j < 0) continue;
            for (int e = row_lo(W, i); e < row_hi(W, i); ++e)
                if (W.ce_col[e] == j) { W.cm[i] = W.ce_cost[e]; W.p0[j] = L - W.ce_cost[e]; }
        }
    });
    // an assignment that uses a pair which is not a feasible edge cannot be certified (never produced by the fast solver)
    ex.phase([&](int t) { for (int j = t; j < M; j += nth) if (W.y[j] >= 0 && W.p0[j] == BIGD) W.sh[S_TIED] = 1; });
    if (W.sh[S_TIED]) return;
    // Bellman-Ford sweeps, Jacobi form (p0 -> p1 -> p0 ...): row i pulls the cap of ITS column from its other edges
    double *pa = W.p0, *pb = W.p1;
    for (int round = 0;; ++round) {
        if (round > N + M + 1 || !alive_steps(ex, W, steps)) { ex.phase([&](int t) { if (t == 0) { W.sh[S_TIED] = 1; W.sh[S_ERR] |= ERR_BF; } }); return; }
        ex.phase([&](int t) { if (t == 0) W.sh[S_CHANGED] = 0; for (int j = t; j < M; j += nth) pb[j] = pa[j]; });
        ex.phase([&](int t) {
            for (int i = t; i < N; i += nth) {
                const int js = W.x[i];
                if (js < 0) continue;
                double best = pa[js];
                for (int e = row_lo(W, i); e < row_hi(W, i); ++e) {
                    const int j = W.ce_col[e];
                    if (j == js || W.ce_cost[e] > L) continue;         // (above L: an override, not a feasible edge)
                    const double cand = (W.ce_cost[e] - W.cm[i]) + pa[j];
                    if (cand < best) best = cand;
                }
                // (an improvement below BF_EPS is rounding noise around a zero-weight cycle -- exactly the tie this check
                //  is looking for; following it would never settle.  The prices stay accurate to ~BF_EPS * path length,
                //  three orders of magnitude below TIE_TOL)
                if (best < pa[js] - BF_EPS) { pb[js] = best; W.sh[S_CHANGED] = 1; }
            }
        });
        double *sw = pa; pa = pb; pb = sw;
        if (!W.sh[S_CHANGED]) break;
    }
    const double *p = pa;
    // near-tight arcs on {matched rows} + Z (node N).  alive[] / has[]: Kahn peeling by out-degree -- a node with no arc
    // into the alive set cannot lie on (or lead to) a cycle and dies; whatever survives contains a cycle.
    ex.phase([&](int t) { for (int i = t; i <= N; i += nth) W.alive[i] = i == N ? 1 : (W.x[i] >= 0 ? 1 : 0); });
    for (int round = 1;; ++round) {
        if (round > N + 2 || !alive_steps(ex, W, steps)) { ex.phase([&](int t) { if (t == 0) { W.sh[S_TIED] = 1; W.sh[S_ERR] |= ERR_BF; } }); return; }
        ex.phase([&](int t) { if (t == 0) W.sh[S_CHANGED] = 0; for (int i = t; i <= N; i += nth) W.has[i] = 0; });
        ex.phase([&](int t) {
            for (int i = t; i < N; i += nth) {
                const int js = W.x[i];
                const bool matched = js >= 0;
                const int src = matched ? i : N;
                if (!W.alive[src]) continue;
                const double ui = matched ? W.cm[i] + p[js] : L;
                bool out = false;
                for (int e = row_lo(W, i); e < row_hi(W, i); ++e) {
                    const int j = W.ce_col[e];
                    if (j == js || W.ce_cost[e] > L) continue;
                    if ((W.ce_cost[e] + p[j]) - ui > TIE_TOL) continue;           // not tight
                    const int owner = W.y[j];
                    const int dst = owner >= 0 ? owner : N;
                    if (src == N && dst == N) { W.sh[S_TIED] = 1; continue; }     // unmatched row + free column at cost ~L
                    if (W.alive[dst]) out = true;
                }
                if (matched) {
                    if (ui >= L - TIE_TOL && W.alive[N]) out = true;              // i could go unmatched:            i -> Z
                    if (p[js] <= TIE_TOL && W.alive[i]) W.has[N] = 1;             // its column could become free:    Z -> i
                }
                if (out) W.has[src] = 1;
            }
        });
        if (W.sh[S_TIED]) return;
        ex.phase([&](int t) { for (int i = t; i <= N; i += nth) if (W.alive[i] && !W.has[i]) { W.alive[i] = 0; W.sh[S_CHANGED] = 1; } });
        if (!W.sh[S_CHANGED]) break;
    }
    // anything left alive lies on or leads to a cycle.  (Z alone cannot survive: it needs an arc into an alive row.)
    ex.phase([&](int t) { for (int i = t; i <= N; i += nth) if (W.alive[i] && i != N) W.sh[S_TIED] = 1; });
}

// ============================================================================================================ 2. the replay
// Cost sources: Src::real(i, j) = the cost-matrix entry of a real pair BEFORE the feasible edges are patched in;
// Src::patched = whether the edge list overrides entries (the tracker's box source: non-edges are the IoU cost, edges carry
// min(IoU cost, appearance cost), D:1539-1554 / D:1590-1602) or the source already is the whole matrix (bts_lapjv_exact).
template <class Src> LAPX_HD double real_cost(const Work &W, const Src &S, int i, int j) {
    if (Src::patched)
        for (int e = row_lo(W, i); e < row_hi(W, i); ++e)
            if (W.ce_col[e] == j) return W.ce_cost[e];
    return S.real(i, j);
}
// entry (i, j) of the extended matrix: lap's  cost_c_extended[:] = cost_limit / 2;  [n_rows:, n_cols:] = 0;  [:n_rows, :n_cols] = cost
template <class Src> LAPX_HD double ext(const Work &W, const Src &S, int i, int j) {
    if (i < W.N) return j < W.M ? real_cost(W, S, i, j) : W.L / 2.0;
    return j < W.M ? W.L / 2.0 : 0.0;
}

// reduce the per-thread Top2 partials in rv1/rj1/rv2/rj2 to one (left in slot 0): two phases
template <class Ex> LAPX_HD void reduce_top2(Ex &ex, Work &W) {
    const int nth = ex.nth();
    const int g = 32;
    ex.phase([&](int t) {
        if (t * g >= nth || t >= g) return;
        Top2 a; top2_init(a);
        for (int q = t * g; q < nth && q < (t + 1) * g; ++q) {
            if (W.rj1[q] != NONE) top2_push(a, W.rv1[q], W.rj1[q]);
            if (W.rj2[q] != NONE) top2_push(a, W.rv2[q], W.rj2[q]);
        }
        W.rv1[t * g] = a.v1; W.rj1[t * g] = a.j1; W.rv2[t * g] = a.v2; W.rj2[t * g] = a.j2;
    });
    ex.phase([&](int t) {
        if (t != 0) return;
        Top2 a; top2_init(a);
        for (int q = 0; q < nth; q += g) {
            if (W.rj1[q] != NONE) top2_push(a, W.rv1[q], W.rj1[q]);
            if (W.rj2[q] != NONE) top2_push(a, W.rv2[q], W.rj2[q]);
        }
        W.rv1[0] = a.v1; W.rj1[0] = a.j1; W.rv2[0] = a.v2; W.rj2[0] = a.j2;
    });
}
// (nth > 32 * 32 = 1024 threads would need a third level)

// Move the FLAGGED positions of cols[a .. b) onto the scan list [lo, hi) exactly as lap's loops do one element at a time:
//     for k in a .. b-1:  if strict(k): hi = lo;   if flagged(k): { j = cols[k]; cols[k] = cols[hi]; cols[hi++] = j; }
// (flag[k] = 1 flagged, 2 flagged + "strict": the scan list restarts at lo first -- _find_dense's  d[j] < mind).
// Invariant of the sequential code: [lo, hi) list, [hi, k) the examined unflagged elements, a QUEUE: a flagged element
// swaps with the queue's FRONT, which thereby moves to its back -- a run of r consecutive flagged positions rotates the
// queue left by r and puts the r elements in front of it.  One run = two phases, whatever its length.
// Precondition: hi <= a (true at both call sites: a = hi for scans, a = lo + 1 = hi for finds).
template <class Ex> LAPX_HD bool apply_runs(Ex &ex, Work &W, int lo, int hi_in, int a, int b, long &steps) {
    const int nth = ex.nth();
    // run starts: flagged positions whose predecessor is not flagged, or that are strict themselves
    const int nrun = compact(ex, W, b - a, W.runs, [&](int q) { const int k = a + q; return W.flag[k] && (k == a || !W.flag[k - 1] || W.flag[k] == 2); });
    steps += 3;
    int hi = hi_in;
    for (int rix = 0; rix < nrun; ++rix) {
        if (!alive_steps(ex, W, steps)) return false;
        const int k = a + W.runs[rix];
        const int kend = rix + 1 < nrun ? a + W.runs[rix + 1] : b;      // the run ends at the first unflagged position before kend
        if (W.flag[k] == 2) hi = lo;
        // run length: flagged positions from k on (uniform: every thread reads the same flags)
        int r = 1;
        while (k + r < kend && W.flag[k + r] == 1) ++r;
        const int q = k - hi;                                         // queue length
        if (q > 0) {
            ex.phase([&](int t) {
                for (int s = t; s < q; s += nth) W.tmpa[s] = W.cols[hi + s];
                for (int s = t; s < r; s += nth) W.tmpb[s] = W.cols[k + s];
            });
            ex.phase([&](int t) {
                for (int s = t; s < r; s += nth) W.cols[hi + s] = W.tmpb[s];
                for (int s = t; s < q; s += nth) W.cols[hi + r + s] = W.tmpa[(int)(((long)s + r) % q)];
            });
        }
        hi += r;
    }
    ex.phase([&](int t) { if (t == 0) W.sh[S_HI] = hi; });
    return true;
}

// _find_dense: the todo columns cols[lo .. n) with minimal d are moved to cols[lo .. hi); returns hi (sh[S_HI]) and
// leaves mind in shd[D_MIND].  flag[k] for k > lo: d[cols[k]] <= min(d[cols[lo .. k))) (2 if <).
template <class Ex> LAPX_HD bool find_dense(Ex &ex, Work &W, int lo, long &steps) {
    const int nth = ex.nth(), n = W.N + W.M, len = n - lo;
    ex.phase([&](int t) { int a, b; chunk(len, nth, t, a, b); double mn = BIGD; for (int q = a; q < b; ++q) { const double wv = W.d[W.cols[lo + q]]; W.w[lo + q] = wv; if (wv < mn) mn = wv; } W.rv1[t] = mn; });
    ex.phase([&](int t) { if (t == 0) { double run = BIGD; for (int q = 0; q < nth; ++q) { const double c = W.rv1[q]; W.rv1[q] = run; if (c < run) run = c; } W.shd[D_MIND] = run; } });
    ex.phase([&](int t) {
        int a, b; chunk(len, nth, t, a, b);
        double run = W.rv1[t];
        for (int q = a; q < b; ++q) {
            const double wv = W.w[lo + q];
            W.flag[lo + q] = q == 0 ? 0 : (wv < run ? 2 : wv == run ? 1 : 0);
            if (wv < run) run = wv;
        }
    });
    steps += 3;
    return apply_runs(ex, W, lo, lo + 1, lo + 1, n, steps);
}

// The dense JV of lap 0.4.0 (lapjv.cpp: _ccrrt_dense, _carr_dense x 2, _ca_dense) on the implicit extended matrix.
// Out: W.x / W.y rewritten with the visible part of its assignment.  sh[S_ERR] != 0: gave up (watchdog).
template <class Ex, class Src> LAPX_HD void jv_replay(Ex &ex, Work &W, const Src &S) {
    const int nth = ex.nth(), N = W.N, M = W.M, n = N + M;
    long steps = 0;
    if (n == 0) return;
    auto fail = [&]() { ex.phase([&](int t) { if (t == 0) W.sh[S_ERR] |= ERR_STEPS; }); };
    // ---- column reduction (_ccrrt_dense): v[j] = min_i E[i][j], y[j] = FIRST minimal row
    ex.phase([&](int t) {
        for (int i = t; i < n; i += nth) { W.xx[i] = -1; W.uniq[i] = 0; }
        for (int j = t; j < n; j += nth) {
            double mn = BIGD; int arg = 0;
            for (int i = 0; i < n; ++i) { const double c = ext(W, S, i, j); if (c < mn) { mn = c; arg = i; } }
            W.v[j] = mn; W.yy[j] = arg;
        }
    });
    // columns high -> low: a row keeps the HIGHEST column that chose it; the others lose their row; "unique" = chosen once
    ex.phase([&](int t) { for (int j = t; j < n; j += nth) { ex.atomic_max(&W.xx[W.yy[j]], j); ex.atomic_add(&W.uniq[W.yy[j]], 1); } });
    ex.phase([&](int t) { for (int j = t; j < n; j += nth) if (W.xx[W.yy[j]] != j) W.yy[j] = -1; });
    // free rows ascending
    const int nfree0 = compact(ex, W, n, W.freer, [&](int i) { return W.xx[i] < 0; });
    // reduction transfer, rows ascending: v[x[i]] -= min_{j != x[i]} (E[i][j] - v[j])    (sequential: each step changes v)
    const int nuniq = compact(ex, W, n, W.tmpa, [&](int i) { return W.xx[i] >= 0 && W.uniq[i] == 1; });
    steps += 9;
    for (int q = 0; q < nuniq; ++q) {
        if (!alive_steps(ex, W, steps)) { fail(); return; }
        const int i = W.tmpa[q], j1 = W.xx[i];
        ex.phase([&](int t) { double mn = BIGD; for (int j = t; j < n; j += nth) if (j != j1) { const double c = ext(W, S, i, j) - W.v[j]; if (c < mn) mn = c; } W.rv1[t] = mn; });
        ex.phase([&](int t) { if (t == 0) { double mn = BIGD; for (int s = 0; s < nth; ++s) if (W.rv1[s] < mn) mn = W.rv1[s]; W.v[j1] -= mn; } });
        steps += 1;
    }
    // ---- augmenting row reduction (_carr_dense), two sweeps
    int nfree = nfree0;
    for (int sweep = 0; sweep < 2 && nfree > 0; ++sweep) {
        ex.phase([&](int t) { if (t == 0) { W.sh[S_CUR] = 0; W.sh[S_NNEW] = 0; W.sh[S_A] = 0; /* rr_cnt */ } });
        for (;;) {
            const int cur = W.sh[S_CUR];
            if (cur >= nfree) break;
            if (!alive_steps(ex, W, steps)) { fail(); return; }
            const int fi = W.freer[cur];
            ex.phase([&](int t) {
                Top2 a; top2_init(a);
                for (int j = t; j < n; j += nth) top2_push(a, ext(W, S, fi, j) - W.v[j], j);
                W.rv1[t] = a.v1; W.rj1[t] = a.j1; W.rv2[t] = a.v2; W.rj2[t] = a.j2;
            });
            reduce_top2(ex, W);
            ex.phase([&](int t) {
                if (t != 0) return;
                int cur1 = W.sh[S_CUR] + 1, nnew = W.sh[S_NNEW];
                const long rr = (long)W.sh[S_A] + 1;                   // (n^2 < 2^31 is required by the caller: rr fits)
                int j1 = W.rj1[0], j2 = W.rj2[0] == NONE ? -1 : W.rj2[0];
                const double r1 = W.rv1[0], r2 = W.rj2[0] == NONE ? BIGD : W.rv2[0];
                int i0 = W.yy[j1];
                const double v1new = W.v[j1] - (r2 - r1);
                const bool lowers = v1new < W.v[j1];
                if (rr < (long)cur1 * (long)n) {
                    if (lowers) W.v[j1] = v1new;
                    else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = W.yy[j2]; }
                    if (i0 >= 0) { if (lowers) W.freer[--cur1] = i0; else W.freer[nnew++] = i0; }
                } else if (i0 >= 0) W.freer[nnew++] = i0;
                W.xx[fi] = j1; W.yy[j1] = fi;
                W.sh[S_CUR] = cur1; W.sh[S_NNEW] = nnew; W.sh[S_A] = (int)rr;
            });
            steps += 3;
        }
        nfree = W.sh[S_NNEW];
    }
    // ---- augmentation (_ca_dense): one shortest augmenting path per remaining free row
    for (int f = 0; f < nfree; ++f) {
        const int start = W.freer[f];
        ex.phase([&](int t) {
            for (int j = t; j < n; j += nth) { W.cols[j] = j; W.pred[j] = start; W.d[j] = ext(W, S, start, j) - W.v[j]; }
            if (t == 0) { W.sh[S_LO] = 0; W.sh[S_HI] = 0; W.sh[S_NREADY] = 0; W.sh[S_FINAL] = -1; }
        });
        for (;;) {                                                     // find_path_dense
            if (!alive_steps(ex, W, steps)) { fail(); return; }
            int lo = W.sh[S_LO], hi = W.sh[S_HI];
            if (W.sh[S_FINAL] >= 0) break;
            if (lo == hi) {
                ex.phase([&](int t) { if (t == 0) W.sh[S_NREADY] = lo; });
                if (!find_dense(ex, W, lo, steps)) { fail(); return; }
                hi = W.sh[S_HI];
                // a free column among the minimal ones ends the search: the LAST of them in list order (the loop does not break)
                ex.phase([&](int t) { int best = -1; for (int k = lo + t; k < hi; k += nth) if (W.yy[W.cols[k]] < 0) best = k; W.ri[t] = best; });
                ex.phase([&](int t) { if (t == 0) { int best = -1; for (int s = 0; s < nth; ++s) if (W.ri[s] > best) best = W.ri[s]; if (best >= 0) W.sh[S_FINAL] = W.cols[best]; } });
                steps += 3;
                if (W.sh[S_FINAL] >= 0) break;
            }
            // _scan_dense, one scan-list column per turn: j = cols[lo++], relax every todo column through its row
            const int js = W.cols[lo], is = W.yy[js];
            const double mind = W.d[js];
            const double hh = (ext(W, S, is, js) - W.v[js]) - mind;
            ex.phase([&](int t) {
                int first = NONE;
                for (int k = hi + t; k < n; k += nth) {
                    const int j = W.cols[k];
                    const double cred = (ext(W, S, is, j) - W.v[j]) - hh;
                    int fl = 0;
                    if (cred < W.d[j]) { fl = cred == mind ? (W.yy[j] < 0 ? 3 : 1) : 4; W.w[k] = cred; }
                    W.flag[k] = fl;                                    // 1 tight -> scan list, 3 tight + free column, 4 plain relaxation
                    if (fl == 3 && k < first) first = k;
                }
                W.ri[t] = first;
            });
            ex.phase([&](int t) { if (t == 0) { int first = NONE; for (int s = 0; s < nth; ++s) if (W.ri[s] < first) first = W.ri[s]; W.sh[S_B] = first; } });
            const int kstop = W.sh[S_B];                              // the loop returns at this position (after relaxing it)
            const int kend = kstop == NONE ? n : kstop + 1;
            ex.phase([&](int t) {
                for (int k = hi + t; k < kend; k += nth) {
                    const int fl = W.flag[k];
                    if (fl) { const int j = W.cols[k]; W.d[j] = W.w[k]; W.pred[j] = is; }
                    if (k >= (kstop == NONE ? n : kstop) || fl != 1) W.flag[k] = 0;     // only tight, owned columns before the stop join the list
                }
                if (t == 0) { W.sh[S_LO] = lo + 1; if (kstop != NONE) W.sh[S_FINAL] = W.cols[kstop]; }
            });
            steps += 3;
            if (!apply_runs(ex, W, lo + 1, hi, hi, kstop == NONE ? n : kstop, steps)) { fail(); return; }
        }
        // dual update of the ready columns, then flip the path
        {
            const int nready = W.sh[S_NREADY], endj = W.sh[S_FINAL];
            const double mind = W.shd[D_MIND];      // = d of every scan-list column (lap reads d[cols[lo]] with the lo of the last find / scan entry)
            ex.phase([&](int t) { for (int k = t; k < nready; k += nth) { const int j = W.cols[k]; W.v[j] += W.d[j] - mind; } });
            ex.phase([&](int t) {
                if (t != 0) return;
                int j = endj, i = -1, guard = 0;
                while (i != start && guard++ <= n) { i = W.pred[j]; W.yy[j] = i; const int tj = W.xx[i]; W.xx[i] = j; j = tj; }
                if (i != start) W.sh[S_ERR] |= ERR_STEPS;
            });
            steps += 2;
            if (W.sh[S_ERR]) return;
        }
    }
    // ---- visible part (lap: x[x >= n_cols] = -1, y[y >= n_rows] = -1)
    ex.phase([&](int t) {
        for (int i = t; i < N; i += nth) W.x[i] = W.xx[i] >= M ? -1 : W.xx[i];
        for (int j = t; j < M; j += nth) W.y[j] = W.yy[j] >= N ? -1 : W.yy[j];
    });
}

// the whole exact-tie step: certificate, replay if it fails.  sh[S_TIED] tells which happened, sh[S_ERR] != 0 = watchdog.
template <class Ex, class Src> LAPX_HD void exact_step(Ex &ex, Work &W, const Src &S, bool force_replay = false) {
    build_csr(ex, W);
    tie_check(ex, W);
    if (W.sh[S_TIED] || force_replay) jv_replay(ex, W, S);
}

}  // namespace lapx
