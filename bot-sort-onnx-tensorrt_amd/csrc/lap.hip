// lap.hip -- exact linear assignment with the semantics of
//     lap.lapjv(cost, extend_cost=True, cost_limit=L)          (reference call site D:1682-1693)
// i.e. minimise  sum(matched cost) + (L/2) * (#unmatched rows + #unmatched columns).
//
// The reference reaches a dense Jonker-Volgenant solver (third-party lap==0.4.0) on an
// (n+m) x (n+m) padded matrix.  On the GPU that matrix is never formed.  Instead:
//   1. only entries with cost <= L can appear in an optimum (matching such a pair must beat two
//      dummies at L/2 each), so the problem is reduced to its *feasible edges* (sparsify);
//   2. the objective equals  const + sum(matched cost) + L * (#unmatched rows): every row gets a
//      private zero-reduced "exit" of cost L and columns may stay free (dual v_j = 0 while free);
//   3. the feasible graph splits into connected components (union-find, root = smallest row),
//      which are independent sub-problems: one wavefront solves one component with the
//      Jonker-Volgenant / Hungarian shortest-augmenting-path step (Dijkstra over reduced costs,
//      row duals u, column duals v), lanes working across the component's columns/edges and
//      __shfl butterflies picking the (distance, column) lexicographic minimum.
// The result is the optimal assignment; when the optimum is unique it is what lapjv returns.
// Tie-breaking (unpinned in the reference): rows in ascending order, smallest
// column on equal distance, a real column preferred over staying unmatched on equality.
// Everything is deterministic regardless of the order edges were appended in.
#include "bts_internal.h"
#include "block_scan.h"
#include "lap_exact.h"
#include "lap_src.h"

namespace {

enum { H_EDGES = 0, H_COMPS = 1, H_ERR = 2, H_HARD = 3, H_NA = 4, H_MA = 5, H_NH = 6, H_OVR = 7, H_MODE = 8,
       H_READY = 32 };   // the flag the helper workgroups poll sits in its own 128-byte line   // hdr: edges, components, overflow flag,
                                                                              // components for k_solve, active rows / columns

// Every word of the work area that is the target of an atomic (or is read back after atomics) is
// accessed ONLY with agent-scope atomics: a plain store that is still on its way to L2 can land
// after atomics issued behind a later barrier and wipe them out (observed: lost increments).
#define RLX(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

typedef bts_lap_problem LapArgs;

// Phase timeline of a launch (make TRACE=1, scripts/dbg_laptrace.py): 100 MHz timestamps in hdr[9 + k].
#ifdef BTS_LAP_TRACE
#define TR(w, k) do { if (threadIdx.x == 0) RST(&(w).hdr[9 + (k)], (int)__builtin_amdgcn_s_memrealtime()); } while (0)
#define TRH(w, k) do { if (blockIdx.y == 1 && threadIdx.x == 0) RST(&(w).hdr[9 + (k)], (int)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define TR(w, k) do { } while (0)
#define TRH(w, k) do { } while (0)
#endif



// Step counters of the component solvers: emulator build only (tests/hip_emu; scripts/lap_steps_cpu.py reads them) -- the
// device build compiles LSTAT to nothing.
#if defined(BTS_HOST_EMU)
static long long g_lapstat[24];
extern "C" void bts_emu_lap_stats(long long *out, int reset) {
    for (int k = 0; k < 24; ++k) { if (out) out[k] = g_lapstat[k]; if (reset) g_lapstat[k] = 0; }
}
#define LSTAT(k, n) do { if (lane == 0) g_lapstat[k] += (n); } while (0)
#define LSTAT_MAX(k, n) do { if (lane == 0 && g_lapstat[k] < (n)) g_lapstat[k] = (n); } while (0)
#define LSTAT_DECL(v) long long v = 0
#define PSTAT(k) do { if (threadIdx.x == 0) g_lapstat[k] += 1; } while (0)
#define LSTAT_INC(v) (++v)
#else
#define LSTAT_DECL(v)
#define PSTAT(k) do { } while (0)
#define LSTAT_INC(v)
#define LSTAT(k, n) do { } while (0)
#define LSTAT_MAX(k, n) do { } while (0)
#endif
enum { LS_PACKED = 0, LS_TALL = 1, LS_GLOBAL = 2, LS_AUG = 3, LS_STEPS = 4, LS_RELAX = 5, LS_TALL_AUG = 6, LS_TALL_STEPS = 7, LS_TALL_RELAX = 8,
       LS_MAX_COMP_STEPS = 9, LS_PATH = 10, LS_TALL_PRELUDE = 11, LS_GLOBAL_ROWS = 12, LS_GLOBAL_COLS = 13, LS_GLOBAL_STEPS = 14, LS_BIG = 15,
       LS_PREP_A = 16, LS_PREP_B = 17, LS_PREP_LARGE = 18, LS_PREP_AGAIN = 19 };

// Lanes of one wavefront hand values to each other through global memory (lane 0 publishes a
// decision, every lane reads it back): drain this wave's outstanding stores/loads before going on.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#if !defined(BTS_HOST_EMU)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_wave_barrier();
}

// order LDS traffic between lanes of one wavefront (DS operations of a wave execute in order; this
// only stops the compiler from moving accesses across the hand-off)
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- 1. sparsify a dense matrix: one wavefront per row, ordered ballot append -------------------
__global__ __launch_bounds__(256) void k_edges_dense(bts_lap_work w, const double *__restrict__ cost, int ld,
                                                     const int32_t *__restrict__ dims, int n_max, int m_max,
                                                     double limit) {
    const int n = dims ? dims[0] : n_max, m = dims ? dims[1] : m_max;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < n; i += nwave) {
        const double *row = cost + (long)i * ld;
        for (int j0 = 0; j0 < m; j0 += 64) {
            const int j = j0 + lane;
            const double c = j < m ? row[j] : 0.0;
            const bool ok = j < m && c <= limit;
            const unsigned long long mask = __ballot(ok);
            if (mask == 0) continue;
            const int cnt = __popcll(mask);
            int base = 0;
            if (lane == 0) base = atomicAdd(&w.hdr[H_EDGES], cnt);
            base = __shfl(base, 0);
            if ((long)base + cnt > w.e_cap) { if (lane == 0) w.hdr[H_ERR] = 1; continue; }
            if (ok) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1));
                w.e_row[pos] = i; w.e_col[pos] = j; w.e_cost[pos] = c;
            }
        }
    }
}

// ---- 2. single-workgroup preparation: CSR by row, components, grouping --------------------------
__device__ int find_root(int32_t *parent, int x) {
    int p = RLX(&parent[x]);
    while (p != x) {
        const int gp = RLX(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
        x = p; p = gp;
    }
    return x;
}

// same, for an array that no atomic ever touches
__device__ int block_exscan_plain(int32_t *arr, int len, int32_t *tmp) {
    const int nth = blockDim.x, tid = threadIdx.x;
    const int per = (len + nth - 1) / nth;
    const int lo = min(len, tid * per), hi = min(len, lo + per);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += arr[k];
    int total;
    int run = bscan::exscan(s, tmp, &total);
    for (int k = lo; k < hi; ++k) { const int v = arr[k]; arr[k] = run; run += v; }
    __syncthreads();
    return total;
}

// same for an LDS array (hand-offs through LDS only: no wait for global stores at the barriers).  Short arrays
// (the usual case: a few hundred components / rows) are scanned by ONE wavefront -- lane chunks, a shuffle scan --
// behind a single workgroup barrier instead of the four of the general form.
__device__ int block_exscan_lds(int32_t *arr, int len, int32_t *tmp) {
    const int nth = blockDim.x, tid = threadIdx.x;
    if (len <= 64 * 32) {
        if (tid < 64) {
            const int per = (len + 63) / 64;
            const int lo = min(len, tid * per), hi = min(len, lo + per);
            int sum = 0;
            for (int k = lo; k < hi; ++k) sum += arr[k];
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (tid >= off) incl += t; }
            int run = incl - sum;
            for (int k = lo; k < hi; ++k) { const int v = arr[k]; arr[k] = run; run += v; }
            if (tid == 63) tmp[0] = incl;
        }
        bscan::lds_barrier();
        return tmp[0];
    }
    const int per = (len + nth - 1) / nth;
    const int lo = min(len, tid * per), hi = min(len, lo + per);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += arr[k];
    int total;
    int run = bscan::exscan(s, tmp, &total);
    for (int k = lo; k < hi; ++k) { const int v = arr[k]; arr[k] = run; run += v; }
    bscan::lds_barrier();
    return total;
}

// read-only walk to the root: used once every union is done, so that what it returns is final
// (a compressing find running next to it could still move parent[] entries to non-root ancestors)
__device__ int find_root_ro(const int32_t *parent, int x) {
    int p = RLX(&parent[x]);
    while (p != x) { x = p; p = RLX(&parent[x]); }
    return x;
}

__device__ void unite(int32_t *parent, int a, int b) {
    int ra = find_root(parent, a), rb = find_root(parent, b);
    while (ra != rb) {
        if (ra < rb) { const int t = ra; ra = rb; rb = t; }      // hook the larger root under the smaller
        const int old = atomicCAS(&parent[ra], ra, rb);
        if (old == ra) return;
        ra = find_root(parent, old);
        rb = find_root(parent, rb);
    }
}

// exclusive scan of arr[0..len) in place (values via RLX), total returned to every thread.
__device__ int block_exscan(int32_t *arr, int len, int32_t *tmp) {
    const int nth = blockDim.x, tid = threadIdx.x;
    const int per = (len + nth - 1) / nth;
    const int lo = min(len, tid * per), hi = min(len, lo + per);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += RLX(&arr[k]);
    int total;
    int run = bscan::exscan(s, tmp, &total);
    for (int k = lo; k < hi; ++k) { const int v = RLX(&arr[k]); RST(&arr[k], run); run += v; }
    __syncthreads();
    return total;
}

// Work is proportional to the number of feasible edges E, not to n + m, and the heavy part only to
// the edges that are NOT isolated pairs:
//   * an edge whose row and column have degree 1 is a whole component by itself: it is matched on
//     the spot (any feasible edge beats two dummies, cost <= limit; ties prefer the match);
//   * the remaining "hard" edges get dense row/column slots, a union-find pass (components) and
//     are exported as PACKED COMPONENTS: descriptor {first row, rows, first col, cols}, per-row
//     {row id, edge range}, column ids, and edges {column lane, cost}.  One wavefront solves one
//     component from three coalesced reads.
// Workgroup 0 of the launch prepares; it publishes the packed problem on hdr[H_READY] and the helper
// workgroups of the same launch (spinning on that one word) share the components with it.
// Two forms of the preparation:
//   SMALL (<= SM_EH_MAX hard edges, two layouts; components up to what solve_big_cols takes): slot space, in LDS;
//   LARGE: node space, in the global work area (any size; oversize components go to the general
//          global-memory solver).
// Invariant kept across calls: deg[] / cdeg[] are zero outside this kernel.
// Memory discipline: words that are targets of atomics (deg, cdeg, hdr, parent, crow_ptr, ccol_ptr,
// cursor, ccur_*) are only ever touched with agent-scope atomics (RLX / RST / atomicAdd / CAS); the
// other arrays are written once and read after a workgroup barrier / the publish with plain accesses.
constexpr int FIRST = 0x40000000;    // "already has a slot" flag inside deg / cdeg
constexpr int SM_EH = 4096;          // hard-edge budget of the LDS form
constexpr int SM_C = SM_EH / 2;      // components: a hard edge shares its row or its column with another hard edge, so a
                                     // component has >= 2 edges (and, being connected, at most edges + 1 rows and columns:
                                     // row slots + column slots <= SM_EH + SM_C)
constexpr int SM_EDGES = 512;        // edge budget of the register/LDS solver, per wavefront
constexpr int BIG_R = 2048, BIG_C = 512, BIG_E = 6144;      // what ONE wavefront solves with the workgroup's whole LDS (solve_big_cols)
constexpr int TALL_S = 4, TALL_R = 254;
// Two FORMS of the solver kernel, chosen by the host per launch (k_lap<EC, CROWD>; bts_lap_enqueue):
//   CROWD = true  (default): components with more rows than columns in column form, components beyond a wavefront's share of
//                 LDS by one wavefront with the workgroup's whole LDS (solve_tall_cols / solve_big_cols), both LDS layouts of
//                 the preparation;
//   CROWD = false (BTS_LAP_FORM=row): the round-2 solvers -- row forms only (solve_packed / solve_packed_tall), the LDS
//                 preparation admits nothing beyond a wavefront's share of LDS and has one layout; everything else takes the
//                 global-memory path.  Kept as a selectable form (the column forms had never run on hardware), and
//                 about a third smaller in code.
// What the LDS preparation admits: rows / columns / edges of a component (the packed descriptor keeps the row count in 8 bits,
// 255 = "255 or more"; solve_packed takes 64 rows, solve_packed_tall TALL_S row slots per lane).
template <bool CROWD> struct Admit {
    static constexpr int R = CROWD ? BIG_R : 254, C = CROWD ? BIG_C : 64, E = CROWD ? BIG_E : 512;
};

// LDS image of the hard sub-problem (slot space).  The HEAD (edge -> row / column, slot -> id) is the same for every layout;
// the TAIL comes in two sizes (round 3): A = up to SM_EH edges and SM_EH / 2 components (24 B + 8 B per edge: plain scenes,
// many tiny components), B = up to SM_EH_MAX edges and SM_EH_MAX / 4 components (dense crowds: 5 000+ edges in ~230
// components) -- which one is known once the non-trivial edges are counted.  Row slots are numbered from the bottom of a
// node array and column slots from its top (rows + columns <= edges + components).
constexpr int SM_EH_MAX = 6144;      // hard-edge budget of the LDS form, layout B
struct SmallHead {
    uint16_t hr[SM_EH_MAX], hc[SM_EH_MAX];        // hard edge -> row, column (global ids first, then slots)
    int32_t NA, MA, C, maxr, EH, pad_[3];
};
template <int EHC, int CC> struct SmallTail {
    static constexpr int kEdges = EHC, kComps = CC, kNodes = EHC + CC;
    // ---- everything below is dead while the node table is alive: the table borrows it ----
    int32_t parent[kNodes];                       // union-find: row slots [0,NA), column slots [NA, NA+MA); a final root: -1 - component
    int32_t rptr[EHC + 1];                        // row slot -> edges
    int32_t crp[CC + 1], ccp[CC + 1];             // component -> first row / column of its group
    int32_t ecnt[CC], cur_c_[CC];                 // component -> edges; column cursor
    int32_t ebase_[CC + 1], ecur_[CC];            // component -> first edge / edge cursor
    uint16_t anode[kNodes];                       // slot -> global row (from the bottom) / column (from the top)
    uint16_t cnode[kNodes];                       // row slot / column slot -> component; column part later lane_of_col
    // arrays with disjoint lifetimes share storage
    __device__ int32_t *cursor() { return parent; }            // [NA] (after the roots are known)
    __device__ int32_t *cur_r() { return parent + EHC; }       // [C]
    __device__ uint16_t &arow(int s) { return anode[s]; }
    __device__ uint16_t &acol(int s) { return anode[kNodes - 1 - s]; }
    __device__ uint16_t &comp(int s) { return cnode[s]; }
    __device__ uint16_t &ccomp(int s) { return cnode[kNodes - 1 - s]; }
};
typedef SmallTail<SM_EH, SM_C> SmallTailA;
typedef SmallTail<SM_EH_MAX, SM_EH_MAX / 4> SmallTailB;
constexpr size_t SMALL_TAIL_BYTES = sizeof(SmallTailA) > sizeof(SmallTailB) ? sizeof(SmallTailA) : sizeof(SmallTailB);
constexpr size_t SMALL_PIPE_BYTES = sizeof(SmallHead) + SMALL_TAIL_BYTES;
static_assert(sizeof(SmallHead) % 16 == 0, "the tail starts aligned");

__device__ int find_root_lds(int32_t *parent, int x) {
    int p = parent[x];
    while (p != x) { x = p; p = parent[x]; }
    return x;
}

// SMALL preparation, entirely in LDS (the global work area's deg/cdeg are not touched).  The node
// table (degree, then slot, of every row and column) borrows the storage of the tail, which only
// comes alive after it is dead.  Returns 0 when the problem is fully prepared/solved as far as this
// workgroup goes, -1 when the LARGE form has to take over (nothing global was changed except x/y of
// isolated pairs, which the LARGE form rewrites identically), -2 when it has to start over.
constexpr int TABLE_CAP = (int)(SMALL_TAIL_BYTES / sizeof(int32_t)) < 65535 ? (int)(SMALL_TAIL_BYTES / sizeof(int32_t)) : 65535;  // node-table entries (global ids are kept in 16 bits)

// inclusive prefix sum across the 64 lanes: DPP row shifts, then the row broadcasts (lane 15 -> next row, lane 31 -> rows 2, 3)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// ONE wavefront: out_a[k] = out_b[k] = sum of src[0..k) for k < len, out_a[len] = total (src may be out_a).
// Returns whether any src[k] exceeds `limit` (lane-local; the caller raises a flag).
__device__ __forceinline__ bool wave_exscan_lds(const int32_t *src, int len, int32_t *out_a, int32_t *out_b, int limit, int lane) {
    const int per = (len + 63) >> 6;
    const int lo = min(len, lane * per), hi = min(len, lo + per);
    int sum = 0;
    bool over = false;
    for (int k = lo; k < hi; ++k) { const int v = src[k]; sum += v; over |= v > limit; }
    const int incl = wave_incl_scan(sum);
    int run = incl - sum;
    for (int k = lo; k < hi; ++k) { const int v = src[k]; out_a[k] = run; out_b[k] = run; run += v; }
    if (lane == 63) out_a[len] = incl;
    return over;
}

// ---- SELF preparation (bts_lap_work::x_slices > 0, BTS_LAP_SELF=1 when a tracker is created; NOT yet measured) -----------------
// The classic launch has workgroup 0 prepare the packed components and hand them to the helper workgroups through memory:
// an agent-scope release, a flag the helpers poll, an acquire, and a cold re-read of the packed arrays on other compute units
// -- a fixed cost of the launch, and the one hand-written protocol BETWEEN workgroups on the step's critical path.  In the SELF
// form every workgroup of the launch runs the (LDS) preparation itself, at the same time, exports the packed components into
// its OWN slice of the work area and solves its share of them: nothing is exchanged between workgroups.  Which components a
// workgroup's wavefronts take must not depend on the order its atomics happened to run in, so a component is named by what
// every workgroup agrees on -- the smallest global row id it contains: the components are ranked by that id (bitmap + prefix
// popcount in LDS), rank r goes to wavefront r mod (wavefronts of the launch), exactly the classic round-robin.  Only for
// problems the tracker issues (`fresh`: x / y preset) and only while the LDS preparation takes the problem; whenever that one
// gives up (every workgroup reaches the same verdict from the same edge list) the launch continues in the classic form.
struct Packed {                      // the packed-component arrays of one slice
    int4 *rowinfo, *cdesc;
    int32_t *colid, *csr_lcol, *order;
    double *csr_cost;
};
__device__ __forceinline__ Packed packed_of(const bts_lap_work &w, int xw) {
    Packed p;
    if (xw <= 0) { p.rowinfo = w.rowinfo; p.cdesc = w.cdesc; p.colid = w.colid; p.csr_lcol = w.csr_lcol; p.csr_cost = w.csr_cost; p.order = w.x_order; return p; }
    const size_t k = (size_t)(xw - 1);
    p.rowinfo = w.x_rowinfo + k * (size_t)w.xr_cap; p.cdesc = w.x_cdesc + k * (size_t)w.xd_cap; p.colid = w.x_colid + k * (size_t)w.xc_cap;
    p.csr_lcol = w.x_lcol + k * (size_t)w.xe_cap; p.csr_cost = w.x_cost + k * (size_t)w.xe_cap; p.order = w.x_order + (k + 1) * (size_t)w.xd_cap;
    return p;
}

// second half of the SMALL preparation, for one tail layout: slots -> components -> packed export
template <int EC, bool CROWD, bool SELF, class Tail>
__device__ __forceinline__ int small_tail(const LapArgs &A, SmallHead &H, Tail &L, int32_t *tab, int EH, int n, bool cached,
                                          const int (&hs)[EC], const double (&hcost)[EC], int xw) {
    const bts_lap_work &w = A.w;
    const Packed pk = packed_of(w, xw);
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int h = tid; h < EH; h += nth) {                 // dense slots; the table entry becomes FIRST | slot
        const int i = H.hr[h], j = H.hc[h];
        if (!(atomicOr(&tab[i], FIRST) & FIRST)) atomicExch(&tab[i], FIRST | atomicAdd(&H.NA, 1));
        if (!(atomicOr(&tab[n + j], FIRST) & FIRST)) atomicExch(&tab[n + j], FIRST | atomicAdd(&H.MA, 1));
    }
    bscan::lds_barrier();
    const int NA = H.NA, MA = H.MA;
    // (row, col) -> slots in place; the slot -> id arrays live in the tail, i.e. inside the table: their values wait in registers
    // of the edge's thread until the table is dead
    int gi[(SM_EH_MAX + 1023) / 1024], gj[(SM_EH_MAX + 1023) / 1024];
#pragma unroll
    for (int k = 0; k < (SM_EH_MAX + 1023) / 1024; ++k) {
        const int h = tid + k * 1024;
        gi[k] = gj[k] = -1;
        if (h < EH) {
            gi[k] = H.hr[h]; gj[k] = H.hc[h];
            H.hr[h] = (uint16_t)(tab[gi[k]] & ~FIRST); H.hc[h] = (uint16_t)(tab[n + gj[k]] & ~FIRST);
        }
    }
    bscan::lds_barrier();                                      // the table is dead from here on
    TR(w, 3);
    if (NA + MA > Tail::kNodes) return -2;                     // (more components than this layout holds: checked before anything is indexed by slot)
#pragma unroll
    for (int k = 0; k < (SM_EH_MAX + 1023) / 1024; ++k)        // (every edge of a row writes the same id: benign)
        if (gi[k] >= 0) { L.arow(H.hr[tid + k * 1024]) = (uint16_t)gi[k]; L.acol(H.hc[tid + k * 1024]) = (uint16_t)gj[k]; }
    for (int s = tid; s < NA; s += nth) { L.parent[s] = s; L.rptr[s] = 0; }
    for (int s = tid; s < MA; s += nth) L.parent[NA + s] = NA + s;
    bscan::lds_barrier();
    for (int h = tid; h < EH; h += nth) { unite(L.parent, H.hr[h], NA + H.hc[h]); atomicAdd(&L.rptr[H.hr[h]], 1); }
    bscan::lds_barrier();
    for (int s = tid; s < NA; s += nth) L.comp(s) = (uint16_t)find_root_lds(L.parent, s);       // (read-only walks; a root is always a row slot)
    for (int s = tid; s < MA; s += nth) L.ccomp(s) = (uint16_t)find_root_lds(L.parent, NA + s);
    bscan::lds_barrier();
    for (int s = tid; s < NA; s += nth)                   // the roots hand out component ids: kept in their (now dead) parent cell
        if (L.comp(s) == s) {
            const int c = atomicAdd(&H.C, 1);
            L.parent[s] = -1 - c;
            if (c < Tail::kComps) { L.crp[c] = 0; L.ccp[c] = 0; L.ecnt[c] = 0; }
        }
    bscan::lds_barrier();
    const int C = H.C;
    TR(w, 4);
    if (C > Tail::kComps) return -2;
    for (int s = tid; s < NA; s += nth) { const int c = -1 - L.parent[L.comp(s)]; L.comp(s) = (uint16_t)c; atomicAdd(&L.crp[c], 1); atomicAdd(&L.ecnt[c], L.rptr[s]); }
    for (int s = tid; s < MA; s += nth) { const int c = -1 - L.parent[L.ccomp(s)]; L.ccomp(s) = (uint16_t)c; atomicAdd(&L.ccp[c], 1); }
    bscan::lds_barrier();
    // Three wavefronts scan the component tables side by side -- rows, columns, edges: group offsets plus the cursors
    // the grouping pass bumps -- and check them against the solver's limits (a plain flag: no contended atomics).
    // (cursor / cur_r alias parent[]: every read of the component ids kept there happened before the barrier above)
    int32_t *cursor = L.cursor(), *cur_r = L.cur_r(), *cur_c = L.cur_c_, *ebase = L.ebase_, *ecur = L.ecur_;
    if (tid < 192) {
        const int wv = tid >> 6, ln = tid & 63;
        const bool over = wv == 0 ? wave_exscan_lds(L.crp, C, L.crp, cur_r, Admit<CROWD>::R, ln)     // (what the LDS solvers take)
                        : wv == 1 ? wave_exscan_lds(L.ccp, C, L.ccp, cur_c, Admit<CROWD>::C, ln)
                                  : wave_exscan_lds(L.ecnt, C, ebase, ecur, Admit<CROWD>::E, ln);
        if (over) H.maxr = 65;
    }
    bscan::lds_barrier();
    const bool fits = H.maxr <= 64;
    TR(w, 5);
    if (!fits) return -2;                                 // (the table is gone by now: the other form starts over)
    // rows / columns grouped by component; a row takes its edge range from its component's block, so that the
    // edges of one component are contiguous (rows and ranges in arbitrary order inside: the solver sorts rows by id)
    for (int s = tid; s < NA; s += nth) {
        const int c = L.comp(s), deg = L.rptr[s];
        const int pos = atomicAdd(&cur_r[c], 1), e0 = atomicAdd(&ecur[c], deg);
        cursor[s] = e0;
        pk.rowinfo[pos] = make_int4(L.arow(s), e0, e0 + deg, 0);
    }
    for (int s = tid; s < MA; s += nth) {
        const int c = L.ccomp(s), pos = atomicAdd(&cur_c[c], 1);
        L.ccomp(s) = (uint16_t)(pos - L.ccp[c]);          // from here on: the column's lane inside its component
        pk.colid[pos] = L.acol(s);
    }
    for (int c = tid; c <= C; c += nth) {                 // (entry C: the end of the last component's blocks)
        const int r = c < C ? L.crp[c + 1] - L.crp[c] : 0, cn = c < C ? L.ccp[c + 1] - L.ccp[c] : 0, ne = c < C ? L.ecnt[c] : 0;
        pk.cdesc[c] = make_int4(L.crp[c], L.ccp[c], (r > 255 ? 255 : r) | ((cn > 255 ? 255 : cn) << 8) | ((ne > 0x7fff ? 0x7fff : ne) << 16), ebase[c]);
    }
    TR(w, 6);
    if (!cached) __syncthreads();                         // (w.hard[] is read back below: its stores must have landed)
    else bscan::lds_barrier();
    if (cached) {
#pragma unroll
        for (int k = 0; k < EC; ++k)
            if (hs[k] >= 0) {
                const int pos = atomicAdd(&cursor[H.hr[hs[k]]], 1);
                pk.csr_lcol[pos] = L.ccomp(H.hc[hs[k]]);
                pk.csr_cost[pos] = hcost[k];
            }
    } else {
        for (int h = tid; h < EH; h += nth) {
            const int pos = atomicAdd(&cursor[H.hr[h]], 1);
            pk.csr_lcol[pos] = L.ccomp(H.hc[h]);
            pk.csr_cost[pos] = w.e_cost[RLX(&w.hard[h])];
        }
    }
    if (tid == 0) { RST(&w.hdr[H_COMPS], C); RST(&w.hdr[H_HARD], C); RST(&w.hdr[H_MODE], xw >= 0 ? 3 : 1); RST(&w.hdr[H_NH], EH); }      // (mode 3: SELF form)
    if (SELF && xw >= 0) {               // (compiled into the SELF instantiations only: k_lap<EC, CROWD, true>)
        // SELF form: rank -> component, by the smallest global row id of each component (see Packed).  Everything of the tail is
        // dead except comp() / arow(): cmin and the bitmap live in parent[], the word prefix in rptr[] (+ a throw-away copy in crp[])
        static_assert(Tail::kNodes >= 2 * 2048 && Tail::kEdges >= 2048, "the ranking borrows parent[] (2 x 2048 words) and rptr[] (2049)");
        int32_t *cmin = L.parent, *bm = L.parent + 2048, *pre = L.rptr;
        const int nwords = (n + 31) >> 5;                 // (n + m <= TABLE_CAP <= 65535: at most 2048 words)
        bscan::lds_barrier();                             // the export above is done with cursor[] (= parent[])
        for (int c = tid; c < C; c += nth) cmin[c] = 0x7fffffff;
        for (int k = tid; k < nwords; k += nth) bm[k] = 0;
        bscan::lds_barrier();
        for (int s = tid; s < NA; s += nth) atomicMin(&cmin[L.comp(s)], (int)L.arow(s));
        bscan::lds_barrier();
        for (int c = tid; c < C; c += nth) atomicOr(&bm[cmin[c] >> 5], (int)(1u << (cmin[c] & 31)));
        bscan::lds_barrier();
        for (int k = tid; k < nwords; k += nth) pre[k] = __popc((unsigned)bm[k]);
        bscan::lds_barrier();
        if (tid < 64) wave_exscan_lds(pre, nwords, pre, pre, 0x7fffffff, tid);
        bscan::lds_barrier();
        for (int c = tid; c < C; c += nth) {
            const int id = cmin[c], r = pre[id >> 5] + __popc((unsigned)bm[id >> 5] & ((1u << (id & 31)) - 1u));
            pk.order[r] = c;
        }
    }
    TR(w, 7);
    return 0;
}

// xw: -1 = classic (this is workgroup 0, the only one that prepares); >= 0 = SELF form, this workgroup's slice (0 = workgroup 0)
template <int EC, bool CROWD, bool SELF> __device__ __forceinline__ int small_pipeline(const LapArgs &A, char *lds, int E, int n, int m, int32_t *tmp, int xw) {
    constexpr int EH_MAX = CROWD ? SM_EH_MAX : SM_EH;      // (the row form keeps the one layout of round 2)
    const bts_lap_work &w = A.w;
    const int tid = threadIdx.x, nth = blockDim.x;
    SmallHead &H = *reinterpret_cast<SmallHead *>(lds);
    int32_t *tab = reinterpret_cast<int32_t *>(lds + sizeof(SmallHead));      // [n + m]: rows, then columns
    if (tid == 0) { H.NA = 0; H.MA = 0; H.C = 0; H.maxr = 0; H.EH = 0; }
    // The edge list is read from global memory ONCE: every pass that walks it uses the same edge -> thread mapping
    // and keeps (row, col) in registers (up to EC per thread; longer lists are re-read).  A thread also remembers
    // the queue position of each of its hard edges and fetches their costs while the component phases run: the
    // final export is done in this mapping too, so the costs never pass through LDS.
    int er[EC], ec[EC], hs[EC];
    double hcost[EC];
    const bool cached = E <= EC * nth;
#pragma unroll
    for (int k = 0; k < EC; ++k) {
        const int e = tid + k * nth;
        er[k] = (cached && e < E) ? w.e_row[e] : -1;
        ec[k] = (cached && e < E) ? w.e_col[e] : 0;
        hs[k] = -1;
    }
    for (int k = tid; k < n + m; k += nth) tab[k] = 0;
    bscan::lds_barrier();
    if (cached) {
#pragma unroll
        for (int k = 0; k < EC; ++k) if (er[k] >= 0) { atomicAdd(&tab[er[k]], 1); atomicAdd(&tab[n + ec[k]], 1); }      // (row -1: vacant slot)
    } else {
        for (int e = tid; e < E; e += nth) { const int i = w.e_row[e]; if (i >= 0) { atomicAdd(&tab[i], 1); atomicAdd(&tab[n + w.e_col[e]], 1); } }
    }
    bscan::lds_barrier();
    auto classify = [&](int e, int i, int j) -> int {     // isolated pairs are final; the rest is queued
        if (i < 0) return -1;
        if (tab[i] == 1 && tab[n + j] == 1) { A.x[i] = j; A.y[j] = i; return -1; }
        const int h = atomicAdd(&H.EH, 1);
        if (h < EH_MAX) { H.hr[h] = (uint16_t)i; H.hc[h] = (uint16_t)j; }
        if (xw <= 0) w.hard[h] = e;                       // (the queue of the global-memory form, should it take over: workgroup 0's)
        return h;
    };
    if (cached) {
#pragma unroll
        for (int k = 0; k < EC; ++k) hs[k] = classify(tid + k * nth, er[k], ec[k]);
    } else {
        for (int e = tid; e < E; e += nth) classify(e, w.e_row[e], w.e_col[e]);
    }
    bscan::lds_barrier();
    const int EH = H.EH;
    TR(w, 2);
    if (EH == 0) return 0;
    if (EH > EH_MAX) {
        // Too many non-trivial edges for the LDS form: hand over to the global-memory form WITHOUT redoing the degree
        // count and the isolated pairs there (three passes of global atomics): its queue w.hard[] is filled, and the
        // degrees of the rows / columns it will touch are copied out of the LDS table.
        if (xw > 0) return -1;                            // (SELF form, a helper: workgroup 0 carries on alone, classic hand-over)
        auto publish = [&](int i, int j) {
            if (i < 0 || (tab[i] == 1 && tab[n + j] == 1)) return;
            RST(&w.deg[i], tab[i]); RST(&w.cdeg[j], tab[n + j]);
        };
        if (cached) {
#pragma unroll
            for (int k = 0; k < EC; ++k) publish(er[k], ec[k]);
        } else {
            for (int e = tid; e < E; e += nth) publish(w.e_row[e], w.e_col[e]);
        }
        if (tid == 0) RST(&w.hdr[H_NH], EH);
        return -1;
    }
#pragma unroll
    for (int k = 0; k < EC; ++k) hcost[k] = hs[k] >= 0 ? w.e_cost[tid + k * nth] : 0.0;      // consumed by the export at the very end
    PSTAT(EH <= SM_EH ? LS_PREP_A : LS_PREP_B);
    if (!CROWD || EH <= SM_EH) return small_tail<EC, CROWD, SELF>(A, H, *reinterpret_cast<SmallTailA *>(lds + sizeof(SmallHead)), tab, EH, n, cached, hs, hcost, xw);
    if constexpr (CROWD) return small_tail<EC, CROWD, SELF>(A, H, *reinterpret_cast<SmallTailB *>(lds + sizeof(SmallHead)), tab, EH, n, cached, hs, hcost, xw);
    return -2;
}

// one counter bump per wavefront instead of one contended global atomic per lane: returns this lane's slot (pred lanes)
__device__ __forceinline__ int wave_alloc(int32_t *counter, bool pred) {
    const unsigned long long mk = __ballot(pred);
    if (!mk) return 0;
    const int lane = threadIdx.x & 63, first = __ffsll((long long)mk) - 1;
    int base = 0;
    if (lane == first) base = atomicAdd(counter, __popcll(mk));
    base = __shfl(base, first);
    return base + __popcll(mk & ((1ull << lane) - 1ull));
}

// LARGE preparation: global work area, node space (phases as in the LDS form)
__device__ __forceinline__ void large_pipeline(const LapArgs &A, int EH, int32_t *tmp, int32_t *lds_nodes, int lds_cap) {
    const bts_lap_work &w = A.w;
    const int n = A.dims ? A.dims[0] : A.n_max, m = A.dims ? A.dims[1] : A.m_max;
    const int tid = threadIdx.x, nth = blockDim.x;
    // the union-find forest is the pointer-chasing part: it lives in LDS whenever the node space fits
    int32_t *parent = n + m <= lds_cap ? lds_nodes : w.parent;
    // 3. dense slots for the rows / columns of hard edges, union-find cells
    for (int h0 = 0; h0 < EH; h0 += nth) {                 // (uniform trip count: wave_alloc is convergent)
        const int h = h0 + tid;
        const bool on = h < EH;
        const int e = on ? w.hard[h] : 0, i = on ? w.e_row[e] : 0, j = on ? w.e_col[e] : 0;
        const bool newr = on && !(atomicOr(&w.deg[i], FIRST) & FIRST);
        const int sr = wave_alloc(&w.hdr[H_NA], newr);
        if (newr) { w.arow[sr] = i; w.slot_of_row[i] = sr; RST(&parent[i], i); }
        const bool newc = on && !(atomicOr(&w.cdeg[j], FIRST) & FIRST);
        const int sc = wave_alloc(&w.hdr[H_MA], newc);
        if (newc) { w.acol[sc] = j; RST(&parent[n + j], n + j); w.v[j] = 0.0; }
    }
    __syncthreads();
    // 4. connected components of the hard graph
    for (int h = tid; h < EH; h += nth) { const int e = w.hard[h]; unite(parent, w.e_row[e], n + w.e_col[e]); }
    __syncthreads();
    // 5. final roots (read-only walks: every union is done); the roots hand out component ids
    const int NA = RLX(&w.hdr[H_NA]), MA = RLX(&w.hdr[H_MA]);
    for (int s0 = 0; s0 < NA; s0 += nth) {
        const int s = s0 + tid;
        const bool on = s < NA;
        const int i = on ? w.arow[s] : 0, r = on ? find_root_ro(parent, i) : -1;
        if (on) { w.rootv[i] = r; w.rptr[s] = RLX(&w.deg[i]) & ~FIRST; }
        const bool root = on && r == i;                   // roots are rows: rows have the smaller node ids
        const int c = wave_alloc(&w.hdr[H_COMPS], root);
        if (root) { w.comp_of[i] = c; RST(&w.crow_ptr[c], 0); RST(&w.ccol_ptr[c], 0); }
    }
    for (int s = tid; s < MA; s += nth) { const int j = w.acol[s]; w.rootv[n + j] = find_root_ro(parent, n + j); }
    __syncthreads();
    // 6. component sizes
    const int C = RLX(&w.hdr[H_COMPS]);
    for (int s = tid; s < NA; s += nth) atomicAdd(&w.crow_ptr[w.comp_of[w.rootv[w.arow[s]]]], 1);
    for (int s = tid; s < MA; s += nth) atomicAdd(&w.ccol_ptr[w.comp_of[w.rootv[n + w.acol[s]]]], 1);
    __syncthreads();
    // 7. pointers
    block_exscan(w.crow_ptr, C, tmp);
    block_exscan(w.ccol_ptr, C, tmp);
    if (tid == 0) { RST(&w.crow_ptr[C], NA); RST(&w.ccol_ptr[C], MA); }
    __syncthreads();
    for (int c = tid; c < C; c += nth) { RST(&w.ccur_r[c], RLX(&w.crow_ptr[c])); RST(&w.ccur_c[c], RLX(&w.ccol_ptr[c])); }
    __syncthreads();
    // 8. rows / columns grouped by component; edge ranges in grouped row order (contiguous per component)
    for (int s = tid; s < NA; s += nth) {
        const int i = w.arow[s];
        const int pos = atomicAdd(&w.ccur_r[w.comp_of[w.rootv[i]]], 1);
        w.crows[pos] = i; w.rpos[s] = pos; w.gstart[pos] = w.rptr[s];
    }
    for (int s = tid; s < MA; s += nth) {
        const int j = w.acol[s];
        const int c = w.comp_of[w.rootv[n + j]], pos = atomicAdd(&w.ccur_c[c], 1);
        w.ccols[pos] = j;
        w.colid[pos] = j;
        w.col_local[j] = pos - RLX(&w.ccol_ptr[c]);
        RST(&w.cdeg[j], 0);
    }
    __syncthreads();
    block_exscan_plain(w.gstart, NA, tmp);
    if (tid == 0) w.gstart[NA] = EH;
    __syncthreads();
    for (int s = tid; s < NA; s += nth) {
        const int i = w.arow[s], pos = w.rpos[s], e0 = w.gstart[pos], e1 = e0 + w.rptr[s];
        RST(&w.cursor[s], e0);
        w.rlo[s] = e0; w.rhi[s] = e1;
        w.rowinfo[pos] = make_int4(i, e0, e1, 0);
        RST(&w.deg[i], 0);
    }
    for (int c = tid; c < C; c += nth) {
        const int r0 = RLX(&w.crow_ptr[c]), c0 = RLX(&w.ccol_ptr[c]);
        const int r = RLX(&w.crow_ptr[c + 1]) - r0, cn = RLX(&w.ccol_ptr[c + 1]) - c0;
        const int ne = w.gstart[r0 + r] - w.gstart[r0];
        w.cdesc[c] = make_int4(r0, c0, (r > 255 ? 255 : r) | ((cn > 255 ? 255 : cn) << 8) | ((ne > 0x7fff ? 0x7fff : ne) << 16), w.gstart[r0]);
    }
    if (tid == 0) w.cdesc[C] = make_int4(NA, MA, 0, EH);  // (the end of the last component's blocks)
    __syncthreads();
    for (int h = tid; h < EH; h += nth) {
        const int e = w.hard[h];
        const int pos = atomicAdd(&w.cursor[w.slot_of_row[w.e_row[e]]], 1);
        w.csr_col[pos] = w.e_col[e];
        w.csr_lcol[pos] = w.col_local[w.e_col[e]];
        w.csr_cost[pos] = w.e_cost[e];
    }
    if (tid == 0) { RST(&w.hdr[H_HARD], C); RST(&w.hdr[H_MODE], 2); }
}

// Degrees + isolated pairs; returns the number of hard edges (uniform).
__device__ __forceinline__ int isolated_pairs(const LapArgs &A, int E) {
    const bts_lap_work &w = A.w;
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int e = tid; e < E; e += nth) { const int i = w.e_row[e]; if (i >= 0) { atomicAdd(&w.deg[i], 1); atomicAdd(&w.cdeg[w.e_col[e]], 1); } }
    __syncthreads();
    for (int e0 = 0; e0 < E; e0 += nth) {
        const int e = e0 + tid;
        const int i = e < E ? w.e_row[e] : -1, j = e < E ? w.e_col[e] : 0;
        const bool iso = i >= 0 && RLX(&w.deg[i]) == 1 && RLX(&w.cdeg[j]) == 1;
        if (iso) { A.x[i] = j; A.y[j] = i; }
        const bool hardq = i >= 0 && !iso;
        const int hq = wave_alloc(&w.hdr[H_NH], hardq);
        if (hardq) w.hard[hq] = e;
    }
    __syncthreads();
    // (degrees of the isolated pairs go back to zero only now: the test above must see 1 for both ends)
    for (int e = tid; e < E; e += nth) {
        const int i = w.e_row[e], j = w.e_col[e];
        if (i < 0) continue;
        if (RLX(&w.deg[i]) == 1 && RLX(&w.cdeg[j]) == 1) { RST(&w.deg[i], 0); RST(&w.cdeg[j], 0); }
    }
    return RLX(&w.hdr[H_NH]);
}

// Lanes of the general solver exchange state through memory that the same wave wrote earlier in the
// kernel: such accesses go straight to L2 (relaxed agent-scope atomics).
#define LDG(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define STG(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// ---- cross-lane helpers for the register solver --------------------------------------------------------
// ds_bpermute (what __shfl compiles to) costs ~100+ cycles of latency per use; the search below is a
// chain of such exchanges, so: broadcasts from a wave-uniform lane go through v_readlane (scalar
// path), and the (distance, column) minimum runs on DPP row operations over the 16 lanes that a
// component of <= 16 columns occupies (wider components fall back to the xor butterfly).
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
__device__ __forceinline__ double bcast_d(double v, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), s);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), s);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)dpp_i<CTRL>((int)(b & 0xffffffffll)), hi = (unsigned)dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// minimum over the wavefront's first cn lanes (the others pass +inf), wave-uniform.  Row operations inside the
// 16-lane DPP rows, the xor butterfly (ds_bpermute) only for components wider than a row.
__device__ __forceinline__ double wave_min_d(double d, int cn) {
    d = fmin(d, dpp_d<0xB1>(d));      // quad_perm [1,0,3,2]  : lane ^ 1
    d = fmin(d, dpp_d<0x4E>(d));      // quad_perm [2,3,0,1]  : lane ^ 2
    d = fmin(d, dpp_d<0x141>(d));     // row_half_mirror       : i <-> 7 - i  (pairs the two quads)
    d = fmin(d, dpp_d<0x140>(d));     // row_mirror            : i <-> 15 - i (pairs the two halves)
    if (cn > 16) d = fmin(d, __shfl_xor(d, 16));
    if (cn > 32) d = fmin(d, __shfl_xor(d, 32));
    return bcast_d(d, 0);
}

// ---- 3. one wavefront per packed component: shortest augmenting paths ---------------------------------
constexpr int DN_R = 8, DN_C = 32;   // components up to DN_R x DN_C also get a dense cost block
struct SmallLds {                    // per-wavefront adjacency copy
    double ecost[SM_EDGES];          // (dense form: [0, DN_R * DN_C) the block, the edge list behind it)
    int32_t ecol[SM_EDGES];          // column lane
};
static_assert(2 * DN_R * DN_C <= SM_EDGES, "dense block + its edge list share ecost[]");

// Components with <= 64 rows, <= 64 columns and <= SM_EDGES edges: lane k owns local row k (u, its
// column) AND local column k (v, owner, distance, predecessor, scanned); the adjacency sits in LDS;
// every step of the Dijkstra search is a handful of DPP / v_readlane operations -- no global memory
// inside the search.  Small components (most of them: a few rows) keep a dense rows x columns cost
// block, so that relaxing from a row is ONE LDS read per lane instead of a walk over the row's edge list.
// Returns false if the component does not fit.
__device__ __forceinline__ bool solve_packed(const LapArgs &A, int c, int lane, SmallLds &L, int xw = 0) {
    const bts_lap_work &w = A.w;
    const Packed pk = packed_of(w, xw);
    const double limit = A.limit, INF = __builtin_huge_val();
    TRH(w, 13);
    const int4 d = pk.cdesc[c];
#ifdef BTS_LAP_TRACE
    if (d.x < 0) return false;
#endif
    TRH(w, 14);
    const int rs = d.x, cs = d.y, r = d.z & 0xff, cn = (d.z >> 8) & 0xff, ne = d.z >> 16, eb = d.w;
    if (r > 64 || cn > 64 || ne > SM_EDGES) return false;
    const bool dense = r <= DN_R && cn <= DN_C && ne <= DN_R * DN_C;
    LSTAT(LS_PACKED, 1);
    double *ecs = dense ? L.ecost + DN_R * DN_C : L.ecost;
    const int4 ri = lane < r ? pk.rowinfo[rs + lane] : make_int4(0x7fffffff, 0, 0, 0);
    const int my_col = lane < cn ? pk.colid[cs + lane] : 0x7fffffff;
    for (int e = lane; e < ne; e += 64) { ecs[e] = pk.csr_cost[eb + e]; L.ecol[e] = pk.csr_lcol[eb + e]; }   // one contiguous block
    if (dense)
        for (int k = lane; k < DN_R * DN_C; k += 64) L.ecost[k] = INF;
    // rows in ascending id order (the grouping appended them in arbitrary order)
    int my_row = 0x7fffffff, e_lo = 0, e_hi = 0;
    {
        int rank = 0;
        for (int q = 0; q < r; ++q) rank += bcast_i(ri.x, q) < ri.x ? 1 : 0;
        for (int q = 0; q < r; ++q) {
            const int rq = bcast_i(rank, q), v0 = bcast_i(ri.x, q), v1 = bcast_i(ri.y, q), v2 = bcast_i(ri.z, q);
            if (lane == rq) { my_row = v0; e_lo = v1 - eb; e_hi = v2 - eb; }
        }
    }
    lds_sync();
    double u = limit;
    for (int e = e_lo; e < e_hi; ++e) u = fmin(u, ecs[e]);
    if (dense) {
        for (int e = e_lo; e < e_hi; ++e) L.ecost[lane * DN_C + L.ecol[e]] = ecs[e];
        lds_sync();
    }
#ifdef BTS_LAP_TRACE
    if (u < -1e300) return false;
#endif
    TRH(w, 15);
    int xr = -1, yc = -1;       // row -> local column, column -> local row
    double v = 0.0;
    // relax(cur, ucur, dcur, dist, pred, scanned): offers every unscanned column the path through row `cur`
    auto run = [&](auto relax) {
        for (int i0 = 0; i0 < r; ++i0) {
            double dist = INF;
            int pred = -1;
            bool scanned = false;
            double exit_d = limit - bcast_d(u, i0);
            int exit_row = i0, cur = i0, end_col = -1;
            double dcur = 0.0, mu = 0.0;
            LSTAT(LS_AUG, 1);
            for (;;) {
                LSTAT(LS_STEPS, 1);
                relax(cur, bcast_d(u, cur), dcur, dist, pred, scanned);
                const bool cand = lane < cn && !scanned;
                const double bd = wave_min_d(cand ? dist : INF, cn);
                if (!(bd < INF) || exit_d < bd) { mu = exit_d; break; }
                // the column at that distance; several: the smallest column id
                const unsigned long long tie = __ballot(cand && dist == bd);
                int bl = __ffsll((long long)tie) - 1;
                if (tie & (tie - 1)) {
                    int bc = 0x7fffffff;
                    for (unsigned long long t = tie; t; t &= t - 1) {
                        const int l = __ffsll((long long)t) - 1, cc = bcast_i(my_col, l);
                        if (cc < bc) { bc = cc; bl = l; }
                    }
                }
                const int owner = bcast_i(yc, bl);
                if (owner < 0) { mu = bd; end_col = bl; break; }
                if (lane == bl) scanned = true;
                cur = owner; dcur = bd;
                const double e2 = bd + (limit - bcast_d(u, owner));
                if (e2 < exit_d) { exit_d = e2; exit_row = owner; }
            }
            // duals: every scanned column hands its delta to the row that owns it
            const double delta = scanned ? mu - dist : 0.0;
            if (scanned) v -= delta;
            const double dfrom = __shfl(delta, xr >= 0 ? xr : 0);
            if (xr >= 0) u += dfrom;
            if (lane == i0) u += mu;
            // augment
            int j = end_col;
            if (j < 0) {
                if (exit_row == i0) continue;
                j = bcast_i(xr, exit_row);
                if (lane == exit_row) xr = -1;
            }
            for (;;) {
                const int i = bcast_i(pred, j);
                const int jn = bcast_i(xr, i);
                if (lane == i) xr = j;
                if (lane == j) yc = i;
                if (i == i0) break;
                j = jn;
            }
        }
    };
    if (dense) {
        run([&](int cur, double ucur, double dcur, double &dist, int &pred, bool scanned) {
            const double cst = L.ecost[__builtin_amdgcn_readfirstlane(cur) * DN_C + (lane & (DN_C - 1))];
            const double nd = dcur + ((cst - ucur) - v);       // (absent pair: +inf, never below dist)
            if (lane < DN_C && !scanned && nd < dist) { dist = nd; pred = cur; }
        });
    } else {
        run([&](int cur, double ucur, double dcur, double &dist, int &pred, bool scanned) {
            const int e0 = bcast_i(e_lo, cur), e1 = bcast_i(e_hi, cur);
            LSTAT(LS_RELAX, e1 - e0);
            for (int e = e0; e < e1; ++e) {
                if (L.ecol[e] == lane && !scanned) {
                    const double nd = dcur + ((ecs[e] - ucur) - v);
                    if (nd < dist) { dist = nd; pred = cur; }
                }
            }
        });
    }
    const int xcol = __shfl(my_col, xr >= 0 ? xr : 0), yrow = __shfl(my_row, yc >= 0 ? yc : 0);
#ifdef BTS_LAP_TRACE
    if (xcol < -12345) return false;
#endif
    TRH(w, 16);
    if (lane < r) A.x[my_row] = xr >= 0 ? xcol : -1;
    if (lane < cn) A.y[my_col] = yc >= 0 ? yrow : -1;
    lds_sync();                 // the next component of this wavefront reuses L
    return true;
}

// Tall components (dense crowds: a detection column overlapped by many track rows): up to TALL_R rows, <= 64 columns,
// <= SM_EDGES edges.  Same search as solve_packed with the row state spread over TALL_S slots per lane (local row
// i lives in lane i & 63, slot i >> 6); the column state stays one per lane.  Slots are selected with wave-uniform
// indices through unrolled selects (a runtime-indexed register array would go to scratch).
template <class T> __device__ __forceinline__ T sel4(const T (&a)[TALL_S], int s) { return s == 0 ? a[0] : s == 1 ? a[1] : s == 2 ? a[2] : a[3]; }
__device__ __forceinline__ bool solve_packed_tall(const LapArgs &A, int c, int lane, SmallLds &L, int xw = 0) {
    const bts_lap_work &w = A.w;
    const Packed pk = packed_of(w, xw);
    const double limit = A.limit, INF = __builtin_huge_val();
    const int4 d = pk.cdesc[c];
    const int rs = d.x, cs = d.y, r = d.z & 0xff, cn = (d.z >> 8) & 0xff, ne = d.z >> 16, eb = d.w;
    if (r >= TALL_R + 1 || cn > 64 || ne > SM_EDGES) return false;
    LSTAT(LS_TALL, 1);
    LSTAT(LS_TALL_PRELUDE, 2 * r);
    LSTAT_DECL(steps_here);
    int4 ri[TALL_S];
#pragma unroll
    for (int s = 0; s < TALL_S; ++s) ri[s] = lane + 64 * s < r ? pk.rowinfo[rs + lane + 64 * s] : make_int4(0x7fffffff, 0, 0, 0);
    const int my_col = lane < cn ? pk.colid[cs + lane] : 0x7fffffff;
    for (int e = lane; e < ne; e += 64) { L.ecost[e] = pk.csr_cost[eb + e]; L.ecol[e] = pk.csr_lcol[eb + e]; }
    // rows in ascending id order: rank = number of rows with a smaller id, then every row moves to (rank & 63, rank >> 6)
    int my_row[TALL_S], e_lo[TALL_S], e_hi[TALL_S], rank[TALL_S];
#pragma unroll
    for (int s = 0; s < TALL_S; ++s) { my_row[s] = 0x7fffffff; e_lo[s] = 0; e_hi[s] = 0; rank[s] = 0; }
    for (int q = 0; q < r; ++q) {
        const int idq = bcast_i(q < 64 ? ri[0].x : q < 128 ? ri[1].x : q < 192 ? ri[2].x : ri[3].x, q & 63);
#pragma unroll
        for (int s = 0; s < TALL_S; ++s) rank[s] += idq < ri[s].x ? 1 : 0;
    }
    for (int q = 0; q < r; ++q) {
        const int qs = q >> 6, ql = q & 63;
        const int4 rq = make_int4(bcast_i(qs == 0 ? ri[0].x : qs == 1 ? ri[1].x : qs == 2 ? ri[2].x : ri[3].x, ql),
                                  bcast_i(qs == 0 ? ri[0].y : qs == 1 ? ri[1].y : qs == 2 ? ri[2].y : ri[3].y, ql),
                                  bcast_i(qs == 0 ? ri[0].z : qs == 1 ? ri[1].z : qs == 2 ? ri[2].z : ri[3].z, ql), 0);
        const int rk = bcast_i(sel4(rank, qs), ql);
#pragma unroll
        for (int s = 0; s < TALL_S; ++s)
            if (lane == (rk & 63) && s == (rk >> 6)) { my_row[s] = rq.x; e_lo[s] = rq.y - eb; e_hi[s] = rq.z - eb; }
    }
    lds_sync();
    double u[TALL_S];
    int xr[TALL_S];
#pragma unroll
    for (int s = 0; s < TALL_S; ++s) {
        u[s] = limit; xr[s] = -1;
        for (int e = e_lo[s]; e < e_hi[s]; ++e) u[s] = fmin(u[s], L.ecost[e]);
    }
    int yc = -1;                 // column -> local row
    double v = 0.0;
    for (int i0 = 0; i0 < r; ++i0) {
        double dist = INF;
        int pred = -1;
        bool scanned = false;
        double exit_d = limit - bcast_d(sel4(u, i0 >> 6), i0 & 63);
        int exit_row = i0, cur = i0, end_col = -1;
        double dcur = 0.0, mu = 0.0;
        LSTAT(LS_TALL_AUG, 1);
        for (;;) {
            const int cs_ = cur >> 6, cl = cur & 63;
            const double ucur = bcast_d(sel4(u, cs_), cl);
            const int e0 = bcast_i(sel4(e_lo, cs_), cl), e1 = bcast_i(sel4(e_hi, cs_), cl);
            LSTAT(LS_TALL_STEPS, 1);
            LSTAT(LS_TALL_RELAX, e1 - e0);
            LSTAT_INC(steps_here);
            for (int e = e0; e < e1; ++e) {
                if (L.ecol[e] == lane && !scanned) {
                    const double nd = dcur + ((L.ecost[e] - ucur) - v);
                    if (nd < dist) { dist = nd; pred = cur; }
                }
            }
            const bool cand = lane < cn && !scanned;
            const double bd = wave_min_d(cand ? dist : INF, cn);
            if (!(bd < INF) || exit_d < bd) { mu = exit_d; break; }
            const unsigned long long tie = __ballot(cand && dist == bd);
            int bl = __ffsll((long long)tie) - 1;
            if (tie & (tie - 1)) {
                int bc = 0x7fffffff;
                for (unsigned long long t = tie; t; t &= t - 1) {
                    const int l = __ffsll((long long)t) - 1, cc = bcast_i(my_col, l);
                    if (cc < bc) { bc = cc; bl = l; }
                }
            }
            const int owner = bcast_i(yc, bl);
            if (owner < 0) { mu = bd; end_col = bl; break; }
            if (lane == bl) scanned = true;
            cur = owner; dcur = bd;
            const double e2 = bd + (limit - bcast_d(sel4(u, owner >> 6), owner & 63));
            if (e2 < exit_d) { exit_d = e2; exit_row = owner; }
        }
        // duals: every scanned column hands its delta to the row that owns it
        const double delta = scanned ? mu - dist : 0.0;
        if (scanned) v -= delta;
#pragma unroll
        for (int s = 0; s < TALL_S; ++s) {
            const double dfrom = __shfl(delta, xr[s] >= 0 ? xr[s] : 0);
            if (xr[s] >= 0) u[s] += dfrom;
            if (lane == (i0 & 63) && s == (i0 >> 6)) u[s] += mu;
        }
        // augment
        int j = end_col;
        if (j < 0) {
            if (exit_row == i0) continue;
            j = bcast_i(sel4(xr, exit_row >> 6), exit_row & 63);
#pragma unroll
            for (int s = 0; s < TALL_S; ++s) if (lane == (exit_row & 63) && s == (exit_row >> 6)) xr[s] = -1;
        }
        for (;;) {
            const int i = bcast_i(pred, j);
            const int jn = bcast_i(sel4(xr, i >> 6), i & 63);
#pragma unroll
            for (int s = 0; s < TALL_S; ++s) if (lane == (i & 63) && s == (i >> 6)) xr[s] = j;
            if (lane == j) yc = i;
            if (i == i0) break;
            j = jn;
        }
    }
#pragma unroll
    for (int s = 0; s < TALL_S; ++s) {
        const int xcol = __shfl(my_col, xr[s] >= 0 ? xr[s] : 0);
        if (lane + 64 * s < r) A.x[my_row[s]] = xr[s] >= 0 ? xcol : -1;
    }
    {
        const int ys = yc >= 0 ? yc >> 6 : 0, yl = yc >= 0 ? yc & 63 : 0;
        int yrow = 0;
#pragma unroll
        for (int s = 0; s < TALL_S; ++s) { const int t = __shfl(my_row[s], yl); if (ys == s) yrow = t; }
        if (lane < cn) A.y[my_col] = yc >= 0 ? yrow : -1;
    }
    LSTAT_MAX(LS_MAX_COMP_STEPS, steps_here);
    lds_sync();                 // the next component of this wavefront reuses L
    return true;
}

// Tall components, COLUMN form (round 3; the row form above is the BTS_LAP_FORM=row instantiation, k_lap<EC, false>).  A tall component has
// more rows than columns (r > 64 >= cn; dense crowds: 100-250 track rows around 20-60 detections), and most of its rows
// end unmatched.  The row form pays one shortest-path search per ROW; this one starts with every row on its private exit
// (u = limit, tight) and runs one search per COLUMN over the rows -- cn searches instead of r, and short ones, because a
// search ends at the first row that is still on its exit.  Same objective, same optimum (unique unless costs tie).
// State: rows in LDS, indexed by the row's rank in ascending id order (u, dist, pred, partner, scanned); columns one per
// lane in registers (v, partner).  The adjacency is re-sorted by column (CSC) so that relaxing from a column is one
// parallel pass of the lanes over that column's edges.  Tie-breaking: columns in ascending id order, the smallest row on
// equal distance, a match preferred over leaving the column free on equality.
struct TallLds {
    double ecost[SM_EDGES];                      // by column
    double u[256], dist[256];                    // (dist doubles as the id table of the ranking pass, u as the column counters)
    uint8_t erow[SM_EDGES];
    uint8_t pred[256], xr[256], scanned[256];    // pred: column lane; xr: column lane or TALL_NONE
    uint16_t cptr[66];
};
constexpr int TALL_NONE = 255;
constexpr int COLFORM_MIN_ROWS = 16;      // components with more rows than this AND more rows than columns take the column form
union WaveLds { SmallLds s; TallLds t; };        // one stride for both layouts: wavefronts of a workgroup use either

__device__ __forceinline__ bool solve_tall_cols(const LapArgs &A, int c, int lane, TallLds &L, int xw = 0) {
    const bts_lap_work &w = A.w;
    const Packed pk = packed_of(w, xw);
    const double limit = A.limit, INF = __builtin_huge_val();
    const int4 d = pk.cdesc[c];
    const int rs = d.x, cs = d.y, r = d.z & 0xff, cn = (d.z >> 8) & 0xff, ne = d.z >> 16;
    if (r >= TALL_R + 1 || cn > 64 || ne > SM_EDGES) return false;
    LSTAT(LS_TALL, 1);
    LSTAT_DECL(steps_here);
    int4 ri[TALL_S];
    int rk[TALL_S];
    int32_t *ids = reinterpret_cast<int32_t *>(L.dist), *cnt = reinterpret_cast<int32_t *>(L.u);
#pragma unroll
    for (int s = 0; s < TALL_S; ++s) {
        const int k = lane + 64 * s;
        ri[s] = k < r ? pk.rowinfo[rs + k] : make_int4(0x7fffffff, 0, 0, 0);
        rk[s] = 0;
        if (k < r) ids[k] = ri[s].x;
    }
    cnt[lane] = 0;
    const int my_col = lane < cn ? pk.colid[cs + lane] : 0x7fffffff;
    int crank = 0;
    for (int q = 0; q < cn; ++q) crank += bcast_i(my_col, q) < my_col ? 1 : 0;
    lds_sync();
    for (int q = 0; q < r; ++q) {                 // rank = rows with a smaller id (LDS broadcast reads)
        const int idq = ids[q];
#pragma unroll
        for (int s = 0; s < TALL_S; ++s) rk[s] += idq < ri[s].x ? 1 : 0;
    }
    // adjacency by column: count, scan, fill (the order of a column's edges does not matter: one edge per row)
#pragma unroll
    for (int s = 0; s < TALL_S; ++s)
        for (int e = ri[s].y; e < ri[s].z; ++e) atomicAdd(&cnt[pk.csr_lcol[e]], 1);
    lds_sync();
    {
        const int deg = cnt[lane], incl = wave_incl_scan(deg);
        lds_sync();
        cnt[lane] = incl - deg;
        L.cptr[lane] = (uint16_t)(incl - deg);
        if (lane == 63) L.cptr[64] = (uint16_t)incl;
    }
    lds_sync();
    double vmin = INF;                            // (kept per edge below, reduced per column afterwards)
#pragma unroll
    for (int s = 0; s < TALL_S; ++s)
        for (int e = ri[s].y; e < ri[s].z; ++e) {
            const int pos = atomicAdd(&cnt[pk.csr_lcol[e]], 1);
            L.ecost[pos] = pk.csr_cost[e];
            L.erow[pos] = (uint8_t)rk[s];
        }
    (void)vmin;
    lds_sync();
    // duals: every row on its exit (u = limit), v = column minimum - limit (<= 0: every edge costs <= limit)
#pragma unroll
    for (int s = 0; s < TALL_S; ++s) { const int k = lane + 64 * s; L.u[k] = limit; L.xr[k] = TALL_NONE; }
    double v = 0.0;
    {
        double mn = limit;
        for (int k = L.cptr[lane]; k < L.cptr[lane + 1]; ++k) mn = fmin(mn, L.ecost[k]);
        v = mn - limit;
    }
    int yc = -1;                                  // column -> row rank
    lds_sync();
    for (int t = 0; t < cn; ++t) {
        const int j0 = __ffsll((long long)__ballot(lane < cn && crank == t)) - 1;
#pragma unroll
        for (int s = 0; s < TALL_S; ++s) { const int k = lane + 64 * s; L.dist[k] = INF; L.scanned[k] = 0; }
        lds_sync();
        double exit_d = -bcast_d(v, j0), dcur = 0.0, mu = 0.0;
        int exit_col = j0, cur = j0, end_row = -1;
        LSTAT(LS_TALL_AUG, 1);
        for (;;) {
            const double vcur = bcast_d(v, cur);
            const int k1 = L.cptr[cur + 1];
            LSTAT(LS_TALL_STEPS, 1);
            LSTAT(LS_TALL_RELAX, (k1 - L.cptr[cur] + 63) / 64);
            LSTAT_INC(steps_here);
            for (int k = L.cptr[cur] + lane; k < k1; k += 64) {
                const int i = L.erow[k];
                if (!L.scanned[i]) {
                    const double nd = dcur + ((L.ecost[k] - L.u[i]) - vcur);
                    if (nd < L.dist[i]) { L.dist[i] = nd; L.pred[i] = (uint8_t)cur; }
                }
            }
            lds_sync();
            double bl = INF;
            int bi_l = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < TALL_S; ++s) {
                const int k = lane + 64 * s;
                const double dd = (k < r && !L.scanned[k]) ? L.dist[k] : INF;
                if (dd < bl) { bl = dd; bi_l = k; }
            }
            const double bd = wave_min_d(bl, 64);
            if (!(bd < INF) || exit_d < bd) { mu = exit_d; break; }
            const unsigned long long tie = __ballot(bl == bd);
            int bi = bcast_i(bi_l, __ffsll((long long)tie) - 1);
            if (tie & (tie - 1))
                for (unsigned long long q = tie & (tie - 1); q; q &= q - 1) { const int o = bcast_i(bi_l, __ffsll((long long)q) - 1); bi = o < bi ? o : bi; }
            const int own = L.xr[bi];
            if (own == TALL_NONE) { mu = bd; end_row = bi; break; }
            if (lane == 0) L.scanned[bi] = 1;
            cur = own; dcur = bd;
            const double e2 = bd - bcast_d(v, own);
            if (e2 < exit_d) { exit_d = e2; exit_col = own; }
            lds_sync();
        }
        // duals: every scanned row hands its delta to the column that owns it
        if (lane < cn && yc >= 0 && L.scanned[yc]) v += mu - L.dist[yc];
        if (lane == j0) v += mu;
#pragma unroll
        for (int s = 0; s < TALL_S; ++s) { const int k = lane + 64 * s; if (L.scanned[k]) L.u[k] -= mu - L.dist[k]; }
        // augment
        int i = end_row;
        if (i < 0) {
            if (exit_col == j0) { lds_sync(); continue; }
            i = bcast_i(yc, exit_col);
            if (lane == exit_col) yc = -1;
        }
        for (;;) {
            const int j = L.pred[i];
            const int prev = bcast_i(yc, j);
            LSTAT(LS_PATH, 1);
            if (lane == 0) L.xr[i] = (uint8_t)j;
            if (lane == j) yc = i;
            if (j == j0) break;
            i = prev;
        }
        lds_sync();
    }
#pragma unroll
    for (int s = 0; s < TALL_S; ++s) {
        const int mine = lane + 64 * s < r ? L.xr[rk[s]] : TALL_NONE;
        const int xcol = __shfl(my_col, mine != TALL_NONE ? mine : 0);
        if (lane + 64 * s < r) {
            A.x[ri[s].x] = mine != TALL_NONE ? xcol : -1;
            if (mine != TALL_NONE) A.y[xcol] = ri[s].x;
        }
    }
    if (lane < cn && yc < 0) A.y[my_col] = -1;
    LSTAT_MAX(LS_MAX_COMP_STEPS, steps_here);
    lds_sync();                 // the next component of this wavefront reuses L
    return true;
}

// Components beyond a wavefront's share of LDS (more than TALL_R rows, 64 columns or SM_EDGES edges: the core of a dense
// crowd now and then -- 261 rows x 76 columns in the crowd scene) used to go to the global-memory solver below, where every
// step of a search is a chain of L2 round trips (~2 ms for that component).  They are queued instead, and once every
// wavefront of the workgroup is done with its own components, wavefront 0 solves them with the WHOLE dynamic LDS of the
// workgroup: the column form of solve_tall_cols with the column state in LDS as well (up to BIG_R x BIG_C, BIG_E edges).
constexpr int BIG_NONE = 0xffff, BIG_QUEUE = 96;      // (a SMALL preparation holds <= SM_EH_MAX edges, a queued component >= 65 of them -- more than
                                                       //  64 columns, > 254 rows or > 512 edges --: at most 94 can exist, the queue cannot overflow)
static_assert(BIG_QUEUE > SM_EH_MAX / 65, "the big-component queue holds every component a SMALL preparation can admit");
struct BigLds {
    double ecost[BIG_E];                         // by column
    double u[BIG_R], dist[BIG_R];                // (dist doubles as the id table of the ranking pass)
    double v[BIG_C];                             // (doubles as the column counters while the adjacency is re-sorted)
    int32_t colid[BIG_C];
    uint16_t erow[BIG_E];
    uint16_t pred[BIG_R], xr[BIG_R], rank[BIG_R];
    uint16_t yc[BIG_C], cptr[BIG_C + 2], corder[BIG_C];
    uint8_t scanned[BIG_R];
};

__device__ __forceinline__ bool fits_big(const bts_lap_work &w, int c, int xw = 0) {
    const Packed pk = packed_of(w, xw);
    const int4 d = pk.cdesc[c], e = pk.cdesc[c + 1];
    return e.x - d.x <= BIG_R && e.y - d.y <= BIG_C && e.w - d.w <= BIG_E;
}

__device__ __forceinline__ void solve_big_cols(const LapArgs &A, int c, int lane, BigLds &L, int xw = 0) {
    const bts_lap_work &w = A.w;
    const Packed pk = packed_of(w, xw);
    const double limit = A.limit, INF = __builtin_huge_val();
    const int4 d = pk.cdesc[c], dn = pk.cdesc[c + 1];
    const int rs = d.x, cs = d.y, r = dn.x - rs, cn = dn.y - cs;
    LSTAT(LS_BIG, 1);
    LSTAT_MAX(LS_GLOBAL_ROWS, r);
    LSTAT_MAX(LS_GLOBAL_COLS, cn);
    int32_t *ids = reinterpret_cast<int32_t *>(L.dist), *cnt = reinterpret_cast<int32_t *>(L.v);
    for (int k = lane; k < r; k += 64) ids[k] = pk.rowinfo[rs + k].x;
    for (int j = lane; j < cn; j += 64) { L.colid[j] = pk.colid[cs + j]; cnt[j] = 0; }
    lds_sync();
    for (int k = lane; k < r; k += 64) {          // rows / columns in ascending id order
        const int id = ids[k];
        int rk = 0;
        for (int q = 0; q < r; ++q) rk += ids[q] < id ? 1 : 0;
        L.rank[k] = (uint16_t)rk;
    }
    for (int j = lane; j < cn; j += 64) {
        const int id = L.colid[j];
        int rk = 0;
        for (int q = 0; q < cn; ++q) rk += L.colid[q] < id ? 1 : 0;
        L.corder[rk] = (uint16_t)j;
    }
    for (int k = lane; k < r; k += 64) {
        const int4 ri = pk.rowinfo[rs + k];
        for (int e = ri.y; e < ri.z; ++e) atomicAdd(&cnt[pk.csr_lcol[e]], 1);
    }
    lds_sync();
    {
        const int per = (cn + 63) >> 6, lo = min(cn, lane * per), hi = min(cn, lo + per);
        int sum = 0;
        for (int k = lo; k < hi; ++k) sum += cnt[k];
        int run = wave_incl_scan(sum) - sum;
        lds_sync();
        for (int k = lo; k < hi; ++k) { const int t = cnt[k]; cnt[k] = run; L.cptr[k] = (uint16_t)run; run += t; }
        if (hi == cn) L.cptr[cn] = (uint16_t)run;
    }
    lds_sync();
    for (int k = lane; k < r; k += 64) {
        const int4 ri = pk.rowinfo[rs + k];
        const uint16_t rk = L.rank[k];
        for (int e = ri.y; e < ri.z; ++e) {
            const int pos = atomicAdd(&cnt[pk.csr_lcol[e]], 1);
            L.ecost[pos] = pk.csr_cost[e];
            L.erow[pos] = rk;
        }
    }
    lds_sync();
    for (int i = lane; i < r; i += 64) { L.u[i] = limit; L.xr[i] = (uint16_t)BIG_NONE; }
    for (int j = lane; j < cn; j += 64) {
        double mn = limit;
        for (int k = L.cptr[j]; k < L.cptr[j + 1]; ++k) mn = fmin(mn, L.ecost[k]);
        L.v[j] = mn - limit;                      // (cnt[] is dead: lds_sync above)
        L.yc[j] = (uint16_t)BIG_NONE;
    }
    lds_sync();
    for (int t = 0; t < cn; ++t) {
        const int j0 = L.corder[t];
        for (int i = lane; i < r; i += 64) { L.dist[i] = INF; L.scanned[i] = 0; }
        lds_sync();
        double exit_d = -L.v[j0], dcur = 0.0, mu = 0.0;
        int exit_col = j0, cur = j0, end_row = -1;
        LSTAT(LS_TALL_AUG, 1);
        for (;;) {
            const double vcur = L.v[cur];
            const int k1 = L.cptr[cur + 1];
            LSTAT(LS_GLOBAL_STEPS, 1);
            for (int k = L.cptr[cur] + lane; k < k1; k += 64) {
                const int i = L.erow[k];
                if (!L.scanned[i]) {
                    const double nd = dcur + ((L.ecost[k] - L.u[i]) - vcur);
                    if (nd < L.dist[i]) { L.dist[i] = nd; L.pred[i] = (uint16_t)cur; }
                }
            }
            lds_sync();
            double bl = INF;
            int bi_l = 0x7fffffff;
            for (int i = lane; i < r; i += 64) {
                const double dd = L.scanned[i] ? INF : L.dist[i];
                if (dd < bl) { bl = dd; bi_l = i; }
            }
            const double bd = wave_min_d(bl, 64);
            if (!(bd < INF) || exit_d < bd) { mu = exit_d; break; }
            const unsigned long long tie = __ballot(bl == bd);
            int bi = bcast_i(bi_l, __ffsll((long long)tie) - 1);
            if (tie & (tie - 1))
                for (unsigned long long q = tie & (tie - 1); q; q &= q - 1) { const int o = bcast_i(bi_l, __ffsll((long long)q) - 1); bi = o < bi ? o : bi; }
            const int own = L.xr[bi];
            if (own == BIG_NONE) { mu = bd; end_row = bi; break; }
            cur = own; dcur = bd;
            const double e2 = bd - L.v[own];
            if (e2 < exit_d) { exit_d = e2; exit_col = own; }
            lds_sync();
            if (lane == 0) L.scanned[bi] = 1;
            lds_sync();
        }
        // duals (every lane has left the search: nothing below is read by another lane before the next lds_sync)
        for (int j = lane; j < cn; j += 64) {
            const int y = L.yc[j];
            if (y != BIG_NONE && L.scanned[y]) L.v[j] += mu - L.dist[y];
        }
        for (int i = lane; i < r; i += 64) if (L.scanned[i]) L.u[i] -= mu - L.dist[i];
        lds_sync();
        if (lane == 0) {
            L.v[j0] += mu;
            int i = end_row;
            bool go = true;
            if (i < 0) {
                if (exit_col == j0) go = false;
                else { i = L.yc[exit_col]; L.yc[exit_col] = (uint16_t)BIG_NONE; }
            }
            while (go) {
                const int j = L.pred[i], prev = L.yc[j];
                L.xr[i] = (uint16_t)j;
                L.yc[j] = (uint16_t)i;
                if (j == j0) break;
                i = prev;
            }
        }
        lds_sync();
    }
    for (int k = lane; k < r; k += 64) {
        const int gid = pk.rowinfo[rs + k].x, mcol = L.xr[L.rank[k]];
        A.x[gid] = mcol != BIG_NONE ? L.colid[mcol] : -1;
        if (mcol != BIG_NONE) A.y[L.colid[mcol]] = gid;
    }
    for (int j = lane; j < cn; j += 64) if (L.yc[j] == BIG_NONE) A.y[L.colid[j]] = -1;
    lds_sync();
}

// General path (any component size): state in global memory.
__device__ __forceinline__ void solve_component(const LapArgs &A, int c, int lane) {
    const bts_lap_work &w = A.w;
    const double limit = A.limit;
    const double INF = __builtin_huge_val();
    int32_t *x = A.x, *y = A.y;
    LSTAT(LS_GLOBAL, 1);
    const int rs = LDG(&w.crow_ptr[c]), r = LDG(&w.crow_ptr[c + 1]) - rs;
    const int cs = LDG(&w.ccol_ptr[c]), cn = LDG(&w.ccol_ptr[c + 1]) - cs;
    LSTAT_MAX(LS_GLOBAL_ROWS, r);
    LSTAT_MAX(LS_GLOBAL_COLS, cn);

    // rows in ascending order (the grouping pass appended them in arbitrary order)
    for (int k = lane; k < r; k += 64) {
        const int mine = LDG(&w.crows[rs + k]);
        int rank = 0;
        for (int q = 0; q < r; ++q) rank += LDG(&w.crows[rs + q]) < mine;
        STG(&w.crows_s[rs + rank], mine);
    }
    wave_sync();
    // row duals start at the row minimum (<= limit): all reduced costs >= 0, exits >= 0
    for (int k = lane; k < r; k += 64) {
        const int i = LDG(&w.crows_s[rs + k]);
        double mn = limit;
        const int sl = LDG(&w.slot_of_row[i]);
        for (int e = LDG(&w.rlo[sl]); e < LDG(&w.rhi[sl]); ++e) mn = fmin(mn, LDG(&w.csr_cost[e]));
        STG(&w.u[i], mn);
    }
    wave_sync();

    for (int t = 0; t < r; ++t) {
        const int i0 = LDG(&w.crows_s[rs + t]);
        for (int k = lane; k < cn; k += 64) { const int j = LDG(&w.ccols[cs + k]); STG(&w.dist[j], INF); STG(&w.scanned[j], 0); }
        wave_sync();
        double exit_d = limit - LDG(&w.u[i0]);   // leave i0 unmatched
        int exit_row = i0;
        int cur = i0;
        double dcur = 0.0, mu = 0.0;
        int end_col = -1;                  // >= 0: path ends at this free column
        for (;;) {
            LSTAT(LS_GLOBAL_STEPS, 1);
            const double ucur = LDG(&w.u[cur]);
            const int scur = LDG(&w.slot_of_row[cur]);
            const int e1 = LDG(&w.rhi[scur]);
            for (int e = LDG(&w.rlo[scur]) + lane; e < e1; e += 64) {
                const int j = LDG(&w.csr_col[e]);
                if (!LDG(&w.scanned[j])) {
                    const double nd = dcur + ((LDG(&w.csr_cost[e]) - ucur) - LDG(&w.v[j]));
                    if (nd < LDG(&w.dist[j])) { STG(&w.dist[j], nd); STG(&w.pred[j], cur); }
                }
            }
            wave_sync();
            double bd = INF;
            int bj = 0x7fffffff;
            for (int k = lane; k < cn; k += 64) {
                const int j = LDG(&w.ccols[cs + k]);
                if (!LDG(&w.scanned[j])) {
                    const double dj = LDG(&w.dist[j]);
                    if (dj < bd || (dj == bd && j < bj)) { bd = dj; bj = j; }
                }
            }
#pragma unroll
            for (int off = 32; off; off >>= 1) {
                const double od = __shfl_xor(bd, off);
                const int oj = __shfl_xor(bj, off);
                if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
            }
            if (!(bd < INF) || exit_d < bd) { mu = exit_d; break; }
            const int owner = LDG(&y[bj]);
            if (owner < 0) { mu = bd; end_col = bj; break; }
            if (lane == 0) STG(&w.scanned[bj], 1);
            wave_sync();
            cur = owner; dcur = bd;
            const double e2 = bd + (limit - LDG(&w.u[owner]));
            if (e2 < exit_d) { exit_d = e2; exit_row = owner; }
        }
        // dual update (before the matching changes: y[] still names the tree rows)
        for (int k = lane; k < cn; k += 64) {
            const int j = LDG(&w.ccols[cs + k]);
            if (LDG(&w.scanned[j])) {
                const double delta = mu - LDG(&w.dist[j]);
                const int own = LDG(&y[j]);
                STG(&w.u[own], LDG(&w.u[own]) + delta);
                STG(&w.v[j], LDG(&w.v[j]) - delta);
            }
        }
        wave_sync();
        if (lane == 0) STG(&w.u[i0], LDG(&w.u[i0]) + mu);
        wave_sync();
        // augment
        int j = end_col;
        if (j < 0) {
            if (exit_row == i0) continue;     // i0 stays unmatched
            j = LDG(&x[exit_row]);
            wave_sync();
            if (lane == 0) STG(&x[exit_row], -1);
        }
        for (;;) {
            const int i = LDG(&w.pred[j]);
            const int jn = LDG(&x[i]);
            wave_sync();
            if (lane == 0) { STG(&x[i], j); STG(&y[j], i); }
            if (i == i0) break;
            j = jn;
        }
        wave_sync();
    }
}

// ---- the one launch -----------------------------------------------------------------------------------
// grid = (problems, 1 + helpers); 1024 threads.  Workgroup y = 0 of a problem prepares; the helper
// workgroups wait for hdr[H_READY] (1 + number of packed components) and share the components.
constexpr int LAP_SOLO_EDGES = 64;   // edge lists up to this length are solved by the preparing workgroup alone
constexpr int LAP_HELPERS = 15, LAP_HELPERS_BIG = 63;    // helper workgroups per problem (16 wavefronts each): small / big problems
constexpr unsigned LAP_LDS_BYTES = SMALL_PIPE_BYTES > 16 * sizeof(WaveLds) ? (unsigned)SMALL_PIPE_BYTES : 16 * (unsigned)sizeof(WaveLds);
static_assert(sizeof(BigLds) <= LAP_LDS_BYTES, "solve_big_cols: its image must fit the kernel's dynamic LDS");
static_assert(LAP_LDS_BYTES + 4096 + 512 <= 160 * 1024, "k_lap: dynamic + static LDS exceed a CU's 160 KiB");

struct LapPair { LapArgs p[2]; };     // (indexed by blockIdx.x straight out of the kernel-argument segment)

template <int EC, bool CROWD = true, bool SELF = false> __global__ __launch_bounds__(1024) void k_lap(LapPair P) {
#if defined(BTS_HOST_EMU)
    char *const lds_raw = emu::dynamic_lds();
#else
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
#endif
    __shared__ int32_t tmp[1024];
    __shared__ int s_c, s_nbig;
    __shared__ int32_t s_big[BIG_QUEUE];
    if (threadIdx.x == 0) s_nbig = 0;             // (a workgroup barrier sits on every path to the component loop)
    const LapArgs &A = P.p[blockIdx.x];
    const bts_lap_work &w = A.w;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wid = tid >> 6;
    int C = 0;
    bool solo = false;
    // SELF form (see Packed): every workgroup of the launch prepares for itself -- a tracker's problem (x / y preset), slices for
    // every helper, an edge list short enough to stay in registers across the passes (the export then needs no queue in memory)
    int E0 = 0;
    bool self = SELF && A.fresh && gridDim.y > 1 && w.x_slices >= (int)gridDim.y - 1;      // (SELF = false: the classic launch only)
    if (self) {
        E0 = RLX(&w.hdr[H_EDGES]);
        if ((long)E0 > w.e_cap) E0 = (int)w.e_cap;
        const int n0 = A.dims ? A.dims[0] : A.n_max, m0 = A.dims ? A.dims[1] : A.m_max;
        self = E0 > LAP_SOLO_EDGES && E0 <= EC * nth && n0 + m0 <= TABLE_CAP;
        if (!self && blockIdx.y > 0 && E0 <= LAP_SOLO_EDGES) return;      // (workgroup 0 solves a short list alone)
    }
    int xw = self ? (int)blockIdx.y : -1;
    const bool prepares = blockIdx.y == 0 || self;        // workgroup 0 always; in the SELF form every workgroup (ONE call site of
    bool wait_for_0 = !prepares;                          //  the preparation: it is the bulk of this kernel's code)
    if (prepares) {
        const int n = A.dims ? A.dims[0] : A.n_max, m = A.dims ? A.dims[1] : A.m_max;
        TR(w, 0);
        int E = RLX(&w.hdr[H_EDGES]);
        TR(w, 1);
        if ((long)E > w.e_cap) E = (int)w.e_cap;
        // A short edge list means few components: this workgroup's 16 wavefronts solve them straight away
        // (no release / poll / re-read of the packed problem through memory); the helpers are sent home.
        solo = E <= LAP_SOLO_EDGES;
        if (blockIdx.y == 0) {
            if (solo && tid == 0) RST(&w.hdr[H_READY], 1);
            if (!A.fresh) {
                for (int i = tid; i < n; i += nth) { RST(&w.deg[i], 0); A.x[i] = -1; }
                for (int j = tid; j < m; j += nth) { RST(&w.cdeg[j], 0); A.y[j] = -1; }
                __syncthreads();
            }
        }
        int state = E > 0 ? -2 : 0;
        if (E > 0 && n + m <= TABLE_CAP) state = small_pipeline<EC, CROWD, SELF>(A, lds_raw, E, n, m, tmp, xw);
        if (self && state < 0) { self = false; xw = -1; }      // (every workgroup reaches the same verdict: classic hand-over)
        if (blockIdx.y == 0) {
            if (state < 0) {                              // LARGE form: global degree arrays
                __syncthreads();
                const int EH = state == -1 && n + m <= TABLE_CAP ? RLX(&w.hdr[H_NH]) : isolated_pairs(A, E);
                PSTAT(LS_PREP_LARGE);
                if (state == -2) PSTAT(LS_PREP_AGAIN);
                if (EH > 0) large_pipeline(A, EH, tmp, reinterpret_cast<int32_t *>(lds_raw), (int)(LAP_LDS_BYTES / sizeof(int32_t)));
            }
            __syncthreads();
            C = RLX(&w.hdr[H_HARD]);
            __syncthreads();
            if (!solo && !self && tid == 0) {       // publish (stores -> barrier -> one lane: agent release, drain, relaxed flag)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#if !defined(BTS_HOST_EMU)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                RST(&w.hdr[H_READY], 1 + C);
            }
            TR(w, 8);
        } else {
            // a helper of the SELF form: its own preparation, into its own slice
            __syncthreads();
            if (state == 0) C = RLX(&w.hdr[H_HARD]);      // (this workgroup's own store -- every workgroup stores the same count)
            else wait_for_0 = true;                       // the LDS preparation gave up (so did workgroup 0's)
            __syncthreads();
        }
    }
    if (wait_for_0) {
        TRH(w, 9);
        if (tid == 0) {
            int v = 0;
            for (long spins = 0; (v = RLX(&w.hdr[H_READY])) == 0 && spins < (1l << 22); ++spins) __builtin_amdgcn_s_sleep(8);
            if (v == 0) { RST(&w.hdr[H_ERR], 2); v = 1; }      // give up loudly rather than hang the queue
            s_c = v - 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        C = s_c;
        TRH(w, 10);
    }
    if (C == 0) return;
    WaveLds *lds = reinterpret_cast<WaveLds *>(lds_raw);
    const int gw = blockIdx.y * (nth >> 6) + wid, nw = (solo ? 1 : gridDim.y) * (nth >> 6);
    const int32_t *order = xw >= 0 ? packed_of(w, xw).order : nullptr;      // SELF form: rank -> this workgroup's component id
    for (int c0 = gw; c0 < C; c0 += nw) {
        const int c = order ? order[c0] : c0;
        if constexpr (!CROWD) {
            if (solve_packed(A, c, lane, lds[wid].s, xw) || solve_packed_tall(A, c, lane, lds[wid].s, xw)) continue;
        } else {
            // one search per row (solve_packed) or per column (solve_tall_cols), whichever side is shorter; small components
            // stay with the row form and its dense cost block
            const int dz = packed_of(w, xw).cdesc[c].z, r_ = dz & 0xff, cn_ = (dz >> 8) & 0xff;
            if (r_ > COLFORM_MIN_ROWS && r_ > cn_ ? solve_tall_cols(A, c, lane, lds[wid].t, xw) : solve_packed(A, c, lane, lds[wid].s, xw)) continue;
            if (solve_tall_cols(A, c, lane, lds[wid].t, xw)) continue;       // (more than 64 rows but not more rows than columns)
            // beyond a wavefront's share of LDS: queued for the end of the loop (the SMALL preparation admits nothing that
            // solve_big_cols cannot take, and never more such components than the queue holds; the LARGE one also feeds the
            // global-memory solver)
            if (fits_big(w, c, xw)) {
                int slot = 0;
                if (lane == 0) slot = atomicAdd(&s_nbig, 1);
                slot = __shfl(slot, 0);
                if (slot < BIG_QUEUE) { if (lane == 0) s_big[slot] = c; continue; }
            }
            if (RLX(&w.hdr[H_MODE]) != 2) { if (lane == 0) RST(&w.hdr[H_ERR], 8); continue; }      // (cannot happen: loud, not wrong)
        }
        solve_component(A, c, lane);
    }
    if constexpr (CROWD) {
        __syncthreads();                              // every wavefront is done with its share of the dynamic LDS
        if (wid == 0) {
            const int nb = s_nbig < BIG_QUEUE ? s_nbig : BIG_QUEUE;
            for (int k = 0; k < nb; ++k) solve_big_cols(A, s_big[k], lane, *reinterpret_cast<BigLds *>(lds_raw), xw);
        }
    }
#ifdef BTS_LAP_TRACE
    __syncthreads();
    if (blockIdx.y <= 1) TR(w, 11 + blockIdx.y);
#endif
}

// ---- exact-tie mode: one workgroup per problem behind k_lap (lap_exact.h holds the algorithm; the same source runs on the
//      CPU under tests/host_emul) ----------------------------------------------------------------------------------------
struct DevEx {                          // the executor: a phase = every thread between two workgroup barriers
    __device__ __forceinline__ int nth() const { return (int)blockDim.x; }
    template <class F> __device__ __forceinline__ void phase(F f) { __syncthreads(); f((int)threadIdx.x); __syncthreads(); }
    __device__ __forceinline__ int atomic_add(int32_t *p, int v) { return atomicAdd(p, v); }
    __device__ __forceinline__ int atomic_max(int32_t *p, int v) { return atomicMax(p, v); }
};
struct ExactPair { LapArgs p[2]; bts_lap_exact x[2]; };

__global__ __launch_bounds__(1024) void k_lap_exact(ExactPair P) {
    const LapArgs &A = P.p[blockIdx.x];
    const bts_lap_exact &X = P.x[blockIdx.x];
    const bts_lap_work &w = A.w;
    const int n = A.dims ? A.dims[0] : A.n_max, m = A.dims ? A.dims[1] : A.m_max;
    int E = RLX(&w.hdr[H_EDGES]);
    const int O = RLX(&w.hdr[H_OVR]);                                     // overrides above the limit, filled from the top end downwards
    if (RLX(&w.hdr[H_ERR]) || (long)E > w.e_cap) return;                  // (an overflowed edge list is already an error of the frame)
    if ((long)O > w.o_cap) { if (threadIdx.x == 0) { RST(&w.hdr[BTS_LAP_H_XERR], 16); RST(&w.hdr[H_ERR], 4); } return; }      // (flagged where they were written, too)
    lapx::Work W;
    W.N = n; W.M = m; W.L = A.limit; W.e_row = w.e_row; W.e_col = w.e_col; W.e_cost = w.e_cost; W.E = E;
    W.O = O; W.o_base = w.e_cap + w.o_cap - O;
    W.x = A.x; W.y = A.y;
    W.max_steps = 8000000L;                                               // ~40x what a 10 000 x 2 000 replay needs
    if (!lapx::work_carve(W, n, m, (long)E + O, (int)blockDim.x, X.dbl, X.ndbl, X.ints, X.nints)) {
        if (threadIdx.x == 0) { RST(&w.hdr[BTS_LAP_H_XERR], 8); RST(&w.hdr[H_ERR], 4); }
        return;
    }
    DevEx ex;
    lapx::build_csr(ex, W);
    lapx::tie_check(ex, W);
    const int tied = W.sh[lapx::S_TIED];
    const bool replay = tied || X.force_replay;
    if (replay) {
        if (X.cost) { const DenseSrc S{X.cost, X.ld}; lapx::jv_replay(ex, W, S); }
        else { const BoxSrc S{X.abox, X.af32, X.bbox, X.bidx}; lapx::jv_replay(ex, W, S); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int err = W.sh[lapx::S_ERR];
        RST(&w.hdr[BTS_LAP_H_TIED], tied); RST(&w.hdr[BTS_LAP_H_REPLAYED], replay ? 1 : 0); RST(&w.hdr[BTS_LAP_H_XERR], err);
        if (err & lapx::ERR_STEPS) RST(&w.hdr[H_ERR], 4);                 // the replay gave up: the frame's assignment is not the reference's
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------- host
#ifdef BTS_HOST_EMU
// (CPU emulation build only: red zones behind every array of a work area, like the track table's -- tracker.hip, take())
thread_local std::vector<std::pair<size_t, size_t>> *bts_emu_lap_zones = nullptr;
static size_t carve(size_t &off, size_t bytes) {
    const size_t o = off;
    off = bts_align(off + bytes) + 256;
    if (bts_emu_lap_zones) bts_emu_lap_zones->push_back({o + bytes, off - (o + bytes)});
    return o;
}
#else
static size_t carve(size_t &off, size_t bytes) { const size_t o = off; off = bts_align(off + bytes); return o; }
#endif

static void layout(bts_lap_work *w, char *base, size_t &off, int n, int m, int64_t e, int64_t ovr, int slices) {
#define SLOT(field, type, count) { size_t o = carve(off, sizeof(type) * (size_t)(count)); w->field = base ? (type *)(base + o) : nullptr; }
    SLOT(hdr, int32_t, BTS_LAP_HDR_INTS);
    SLOT(e_row, int32_t, e + ovr); SLOT(e_col, int32_t, e + ovr); SLOT(e_cost, double, e + ovr);      // (+ the overrides' own region)
    SLOT(deg, int32_t, n); SLOT(cdeg, int32_t, m); SLOT(rptr, int32_t, n + 1); SLOT(cursor, int32_t, n);
    SLOT(csr_col, int32_t, e); SLOT(csr_cost, double, e);
    SLOT(parent, int32_t, n + m); SLOT(rootv, int32_t, n + m); SLOT(comp_of, int32_t, n);
    SLOT(crow_ptr, int32_t, n + 1); SLOT(ccol_ptr, int32_t, n + 1);
    SLOT(ccur_r, int32_t, n + 1); SLOT(ccur_c, int32_t, n + 1);
    SLOT(crows, int32_t, n); SLOT(crows_s, int32_t, n); SLOT(ccols, int32_t, m); SLOT(comp_list, int32_t, n);
    SLOT(slot_of_row, int32_t, n); SLOT(slot_of_col, int32_t, m); SLOT(col_local, int32_t, m); SLOT(colid, int32_t, m);
    SLOT(cdesc, int4, n + 1); SLOT(rowinfo, int4, n); SLOT(rpos, int32_t, n); SLOT(gstart, int32_t, n + 1); SLOT(rlo, int32_t, n); SLOT(rhi, int32_t, n); SLOT(csr_lcol, int32_t, e); SLOT(arow, int32_t, n); SLOT(acol, int32_t, m); SLOT(hard, int32_t, e);
    SLOT(u, double, n); SLOT(v, double, m); SLOT(dist, double, m);
    SLOT(pred, int32_t, m); SLOT(scanned, int32_t, m);
    // SELF form (bts_lap_work::x_slices): the packed-component arrays once more per helper workgroup, sized by what the LDS
    // preparation can export (SM_EH_MAX edges, that many + SM_EH / 2 nodes, SM_EH / 2 components), plus the rank -> component
    // table of every slice incl. workgroup 0's
    w->x_slices = slices;
    if (slices > 0) {
        const int nodes = SM_EH_MAX + SM_C;
        w->xr_cap = n < nodes ? n : nodes; w->xc_cap = m < nodes ? m : nodes;
        w->xd_cap = (n < SM_C ? n : SM_C) + 1; w->xe_cap = (int32_t)(e < SM_EH_MAX ? e : SM_EH_MAX);
        SLOT(x_rowinfo, int4, (size_t)slices * w->xr_cap); SLOT(x_cdesc, int4, (size_t)slices * w->xd_cap);
        SLOT(x_colid, int32_t, (size_t)slices * w->xc_cap); SLOT(x_lcol, int32_t, (size_t)slices * w->xe_cap);
        SLOT(x_cost, double, (size_t)slices * w->xe_cap); SLOT(x_order, int32_t, (size_t)(slices + 1) * w->xd_cap);
    }
#undef SLOT
}

size_t bts_lap_work_bytes(int n_cap, int m_cap, int64_t e_cap, int64_t o_cap, int self_slices) {
    bts_lap_work w; size_t off = 0;
    layout(&w, nullptr, off, n_cap < 1 ? 1 : n_cap, m_cap < 1 ? 1 : m_cap, e_cap < 1 ? 1 : e_cap, o_cap < 0 ? 0 : o_cap, self_slices < 0 ? 0 : self_slices);
    return off;
}

void bts_lap_work_carve(bts_lap_work *w, void *block, int n_cap, int m_cap, int64_t e_cap, int64_t o_cap, int self_slices) {
    size_t off = 0;
    n_cap = n_cap < 1 ? 1 : n_cap; m_cap = m_cap < 1 ? 1 : m_cap; e_cap = e_cap < 1 ? 1 : e_cap; o_cap = o_cap < 0 ? 0 : o_cap;
    layout(w, (char *)block, off, n_cap, m_cap, e_cap, o_cap, self_slices < 0 ? 0 : self_slices);
    w->n_cap = n_cap; w->m_cap = m_cap; w->e_cap = e_cap; w->o_cap = o_cap; w->bytes = off;
}

int bts_lap_enqueue_reset(const bts_lap_work *w, hipStream_t stream) {
    BTS_HIP(hipMemsetAsync(w->hdr, 0, BTS_LAP_HDR_INTS * sizeof(int32_t), stream));
    return BTS_OK;
}

int bts_lap_init() {      // the solver kernel needs more than the default 64 KiB of dynamic LDS
    static bool attr_set = false;
    if (!attr_set) {
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<8, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        BTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lap<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LAP_LDS_BYTES));
        attr_set = true;
    }
    return BTS_OK;
}

int bts_lap_helpers_for(int n_max, int helpers) {
    if (helpers >= 0 && helpers <= 255) return helpers;
    return n_max >= 2048 ? LAP_HELPERS_BIG : LAP_HELPERS;
}

int bts_lap_form_from_env() { const char *e = bts_form_get("BTS_LAP_FORM"); return e && !strcmp(e, "row") ? 1 : 0; }

int bts_lap_enqueue(const bts_lap_problem *probs, int count, hipStream_t stream, hipEvent_t *ev, int form, int helpers_want) {
    int rc0 = bts_lap_init();
    if (rc0) return rc0;
    const bts_lap_problem &A = probs[0], &B = probs[count > 1 ? 1 : 0];
    LapPair P;
    P.p[0] = A; P.p[1] = B;
    if (ev) (void)hipEventRecord(ev[0], stream);
    // a component is one wavefront's job: big problems (hundreds of components) get more helper workgroups
    int helpers = (A.n_max >= 2048 || B.n_max >= 2048) ? LAP_HELPERS_BIG : LAP_HELPERS;
    {   // BTS_LAP_HELPERS: sweep knob (host side only: the grid's second dimension); a tracker passes what it read at create,
        // the standalone entry points read the process environment per call (no latching: the same key has the same lifetime
        // on both paths)
        int want = helpers_want;
        if (want < 0) { const char *e = getenv("BTS_LAP_HELPERS"); want = e ? atoi(e) : -1; }
        if (want >= 0 && want <= 255) helpers = want;
    }
    // edges a thread keeps in registers across the preparation passes: 4 (lists up to 4096 edges are read once), or 8 for
    // problems wide enough to produce longer lists (the instantiation costs the short ones a few percent)
    // the solver's FORM (see Admit<>): column forms unless the caller (a tracker: BTS_LAP_FORM when it was created) or, for the
    // standalone entry points, the process environment asks for the round-2 row forms.  Both are exact -- on tie-free costs they
    // return the same indices
    const bool row_form = (form < 0 ? bts_lap_form_from_env() : form) == 1;
    const bool wide = A.m_max > 3072 || B.m_max > 3072;
    const dim3 grid(count, 1 + helpers);
    // (the SELF form -- work areas carved with slices, BTS_LAP_SELF -- has its own instantiations: the classic kernels do not
    //  carry its ranking pass)
    const bool selfk = A.w.x_slices > 0 || B.w.x_slices > 0;
    if (selfk) {
        if (row_form) {
            if (wide) hipLaunchKernelGGL((k_lap<8, false, true>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
            else hipLaunchKernelGGL((k_lap<4, false, true>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
        } else if (wide) hipLaunchKernelGGL((k_lap<8, true, true>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
        else hipLaunchKernelGGL((k_lap<4, true, true>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
    } else if (row_form) {
        if (wide) hipLaunchKernelGGL((k_lap<8, false>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
        else hipLaunchKernelGGL((k_lap<4, false>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
    } else if (wide) hipLaunchKernelGGL((k_lap<8, true>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
    else hipLaunchKernelGGL((k_lap<4, true>), grid, dim3(1024), LAP_LDS_BYTES, stream, P);
    BTS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[1], stream);
    return BTS_OK;
}

void bts_lap_exact_sizes(int n_cap, int m_cap, int64_t e_cap, size_t *ndbl, size_t *nints) {
    n_cap = n_cap < 1 ? 1 : n_cap; m_cap = m_cap < 1 ? 1 : m_cap; e_cap = e_cap < 1 ? 1 : e_cap;
    *ndbl = lapx::work_doubles(n_cap, m_cap, (long)e_cap, 1024);
    *nints = lapx::work_ints(n_cap, m_cap, (long)e_cap, 1024);
}

int bts_lap_exact_enqueue(const bts_lap_problem *probs, const bts_lap_exact *xs, int count, hipStream_t stream) {
    ExactPair P;
    P.p[0] = probs[0]; P.p[1] = probs[count > 1 ? 1 : 0];
    P.x[0] = xs[0]; P.x[1] = xs[count > 1 ? 1 : 0];
    hipLaunchKernelGGL(k_lap_exact, dim3(count), dim3(1024), 0, stream, P);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_lap_enqueue_dense(const bts_lap_work *w, const double *cost, int ld, const int32_t *dims,
                          int n_max, int m_max, double cost_limit, int32_t *x, int32_t *y,
                          hipStream_t stream) {
    int rc = bts_lap_enqueue_reset(w, stream);
    if (rc) return rc;
    if (n_max > 0 && m_max > 0) {
        int blocks = bts_cdiv(n_max, 4);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_edges_dense, dim3(blocks), dim3(256), 0, stream, *w, cost, ld, dims, n_max, m_max,
                           cost_limit);
        BTS_LAUNCH_CHECK();
    }
    bts_lap_problem P{*w, dims, n_max, m_max, cost_limit, x, y, 0};
    return bts_lap_enqueue(&P, 1, stream);
}

extern "C" int bts_lapjv(bts_context *ctx, const double *cost, int n, int m, int ld, double cost_limit,
                         int32_t *x, int32_t *y, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0, "ctx, n, m");
    BTS_ON_DEVICE(ctx);
    BTS_REQUIRE(cost_limit == cost_limit && cost_limit < 1e300, "cost_limit must be finite");
    if (n == 0 && m == 0) return BTS_OK;
    BTS_REQUIRE((n == 0 || x) && (m == 0 || y), "null x / y");
    BTS_REQUIRE(n == 0 || m == 0 || (cost && ld >= m), "null cost / ld < m");
    const int64_t e_cap = (int64_t)n * m;
    BTS_REQUIRE(e_cap < (1ll << 31), "n*m must fit int32");
    const size_t bytes = bts_lap_work_bytes(n, m, e_cap);
    // the arena is reused across calls on one context (bts_arena::ensure synchronises the device before it grows);
    // calls that share a context must be issued on ONE stream (botsort_hip.h, conventions)
    int rc = ctx->scratch.ensure(bytes);
    if (rc) return rc;
    bts_lap_work w;
    bts_lap_work_carve(&w, ctx->scratch.base, n, m, e_cap);
    return bts_lap_enqueue_dense(&w, cost, ld, nullptr, n, m, cost_limit, x, y, (hipStream_t)stream);
}

// lap.lapjv(cost, extend_cost=True, cost_limit=L) with the reference's behaviour on exact cost ties as well: the fast
// solver, then the exact-tie step (lap_exact.h).  info3_host (nullable; synchronises when given): {certificate said "tied",
// the dense JV was replayed, error flags}.
extern "C" int bts_lapjv_exact(bts_context *ctx, const double *cost, int n, int m, int ld, double cost_limit,
                               int32_t *x, int32_t *y, int force_replay, int32_t *info3_host, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0, "ctx, n, m");
    BTS_ON_DEVICE(ctx);
    BTS_REQUIRE(cost_limit == cost_limit && cost_limit < 1e300 && cost_limit >= 0.0, "cost_limit must be finite and >= 0");
    if (info3_host) { info3_host[0] = info3_host[1] = info3_host[2] = 0; }
    if (n == 0 && m == 0) return BTS_OK;
    BTS_REQUIRE((n == 0 || x) && (m == 0 || y), "null x / y");
    BTS_REQUIRE(n == 0 || m == 0 || (cost && ld >= m), "null cost / ld < m");
    const int64_t e_cap = (int64_t)n * m;
    BTS_REQUIRE(e_cap < (1ll << 31) && (int64_t)(n + m) * (n + m) < (1ll << 31), "n*m and (n+m)^2 must fit int32");
    const size_t lap_bytes = bts_align(bts_lap_work_bytes(n, m, e_cap));
    size_t ndbl = 0, nints = 0;
    bts_lap_exact_sizes(n, m, e_cap, &ndbl, &nints);
    const size_t dbl_bytes = bts_align(ndbl * sizeof(double));
    int rc = ctx->scratch.ensure(lap_bytes + dbl_bytes + nints * sizeof(int32_t));
    if (rc) return rc;
    bts_lap_work w;
    bts_lap_work_carve(&w, ctx->scratch.base, n, m, e_cap);
    rc = bts_lap_enqueue_dense(&w, cost, ld, nullptr, n, m, cost_limit, x, y, (hipStream_t)stream);
    if (rc) return rc;
    bts_lap_problem P{w, nullptr, n, m, cost_limit, x, y, 0};
    bts_lap_exact X;
    X.dbl = (double *)((char *)ctx->scratch.base + lap_bytes); X.ndbl = ndbl;
    X.ints = (int32_t *)((char *)ctx->scratch.base + lap_bytes + dbl_bytes); X.nints = nints;
    X.cost = cost; X.ld = ld; X.force_replay = force_replay ? 1 : 0;
    rc = bts_lap_exact_enqueue(&P, &X, 1, (hipStream_t)stream);
    if (rc) return rc;
    if (info3_host) {
        int32_t h[BTS_LAP_HDR_INTS];
        BTS_HIP(hipMemcpyAsync(h, w.hdr, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
        BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
        info3_host[0] = h[BTS_LAP_H_TIED]; info3_host[1] = h[BTS_LAP_H_REPLAYED]; info3_host[2] = h[BTS_LAP_H_XERR] | (h[2] ? 0x100 : 0);
    }
    return BTS_OK;
}
