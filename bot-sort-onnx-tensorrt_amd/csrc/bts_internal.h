// bts_internal.h -- shared host-side plumbing of libbotsort_hip.so (gfx950 only).
//
// Error handling: every C-ABI entry point is wrapped so that no C++ exception leaves the library;
// failures set a thread-local message (bts_last_error) and return a negative bts_status.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/botsort_hip.h"

#define BTS_WAVE 64

void bts_set_error(const char *fmt, ...);

#define BTS_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            bts_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                          __LINE__);                                                         \
            return BTS_EHIP;                                                                 \
        }                                                                                    \
    } while (0)

#define BTS_REQUIRE(cond, msg)                                      \
    do {                                                            \
        if (!(cond)) {                                              \
            bts_set_error("invalid argument: %s (%s)", msg, #cond); \
            return BTS_EINVAL;                                      \
        }                                                           \
    } while (0)

#define BTS_LAUNCH_CHECK()                                                                   \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) {                                                              \
            bts_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),         \
                          __FILE__, __LINE__);                                               \
            return BTS_EHIP;                                                                 \
        }                                                                                    \
    } while (0)

// Execution forms (which of several result-identical kernel forms / launch structures a tracker uses): every switch is a
// KEY[=VALUE] pair looked up first in the string given to bts_tracker_create_forms (while a bts_forms_scope is alive on this
// thread), then in the process environment.  "KEY=" (empty value) in the string turns a form off whatever the environment says.
// The returned pointer is valid until the next bts_form_get on this thread.
const char *bts_form_get(const char *key);
bool bts_form_flag(const char *key);       // boolean forms: "", "0", "false", "off", "no" are OFF
struct bts_forms_scope {
    const char *prev;
    explicit bts_forms_scope(const char *forms);
    ~bts_forms_scope();
    bts_forms_scope(const bts_forms_scope &) = delete;
    bts_forms_scope &operator=(const bts_forms_scope &) = delete;
};

static inline int bts_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t bts_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// A growable device scratch arena owned by a context (standalone API calls) -- never touched
// inside a tracker step (the tracker preallocates everything at create time so that a step is
// pure kernel enqueues and can be captured in a hipGraph).
struct bts_arena {
    char *base = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
};

// Every entry point runs on its context's device and leaves the caller's current device as it found it
// (a rank with LOCAL_RANK != 0 may call in with another device current).
struct bts_device_guard {
    int prev = -1, want;
    explicit bts_device_guard(int dev) : want(dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~bts_device_guard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    bts_device_guard(const bts_device_guard &) = delete;
    bts_device_guard &operator=(const bts_device_guard &) = delete;
};
#define BTS_ON_DEVICE(ctx) bts_device_guard dev_guard_((ctx)->device)

struct bts_context {
    int device = 0;
    int cos_kernel = BTS_COS_MFMA_F32;
    int n_cu = 256;
    bts_arena scratch;
};

// ---- assignment solver (lap.hip) -------------------------------------------------------------
// Device-side work area of the exact sparse linear-assignment solver.  All sizes are capacities;
// the live sizes are read by the kernels from `dims` (device memory) so that a tracker step never
// needs the host to know them.
constexpr int BTS_LAP_HDR_INTS = 64;   // counters (first 128-byte line) + the ready flag (second line)

struct bts_lap_work {
    int32_t n_cap = 0, m_cap = 0;
    int64_t e_cap = 0;           // feasible edges the list holds: [0, e_cap) of e_row / e_col / e_cost
    int64_t o_cap = 0;           // room BEHIND them for the exact-tie step's overrides, filled from the top end of the arrays
                                 // downwards: [e_cap + o_cap - O, e_cap + o_cap) -- the two regions never meet
    int32_t *hdr = nullptr;      // [BTS_LAP_HDR_INTS] counters, see lap.hip
    int32_t *e_row = nullptr;    // [e_cap] unsorted feasible edges (row, col, cost)
    int32_t *e_col = nullptr;
    double *e_cost = nullptr;
    int32_t *deg = nullptr;      // [n_cap] feasible edges per row
    int32_t *cdeg = nullptr;     // [m_cap] feasible edges per column
    int32_t *rptr = nullptr;     // [n_cap+1] CSR row pointers, indexed by slot_of_row
    int32_t *cursor = nullptr;   // [n_cap] fill cursors / component cursors
    int32_t *csr_col = nullptr;  // [e_cap]
    double *csr_cost = nullptr;  // [e_cap]
    int32_t *parent = nullptr;   // [n_cap+m_cap] union-find forest (when it does not fit LDS)
    int32_t *rootv = nullptr;    // [n_cap+m_cap] root of every node, written after all unions are done
    int32_t *comp_of = nullptr;  // [n_cap] component id of a root row
    int32_t *crow_ptr = nullptr; // [n_cap+1]
    int32_t *ccol_ptr = nullptr; // [n_cap+1]
    int32_t *ccur_r = nullptr;   // [n_cap+1] grouping cursors
    int32_t *ccur_c = nullptr;   // [n_cap+1]
    int32_t *crows = nullptr;    // [n_cap] rows grouped by component
    int32_t *crows_s = nullptr;  // [n_cap] ... sorted ascending inside each component
    int32_t *ccols = nullptr;    // [m_cap] columns grouped by component
    int32_t *comp_list = nullptr; // [n_cap] components that need the augmenting-path solver
    int32_t *slot_of_row = nullptr; // [n_cap] dense index of an active row (valid while deg > 0)
    int32_t *slot_of_col = nullptr; // [m_cap]
    int32_t *col_local = nullptr; // [m_cap] lane of a column inside its component
    int32_t *colid = nullptr;     // [m_cap] packed components: column ids, grouped
    int4 *cdesc = nullptr;        // [n_cap] packed components: {first row, rows, first column, columns}
    int4 *rowinfo = nullptr;      // [n_cap] packed components: {row id, first edge, end edge, -}
    int32_t *rpos = nullptr, *gstart = nullptr, *rlo = nullptr, *rhi = nullptr;   // [n_cap(+1)] grouped positions / edge ranges
    int32_t *csr_lcol = nullptr;  // [e_cap] packed components: column lane of each CSR edge
    int32_t *arow = nullptr;      // [n_cap] active rows (have a feasible edge), in order of first sight
    int32_t *acol = nullptr;      // [m_cap] active columns
    int32_t *hard = nullptr;      // [e_cap] indices of edges that are not isolated pairs
    double *u = nullptr;         // [n_cap] row duals
    double *v = nullptr;         // [m_cap] column duals
    double *dist = nullptr;      // [m_cap] tentative distances of a search
    int32_t *pred = nullptr;     // [m_cap] predecessor row of a column in the search tree
    int32_t *scanned = nullptr;  // [m_cap]
    // SELF form of the launch (lap.hip, struct Packed): x_slices > 0 = every helper workgroup has its own copy of the packed-
    // component arrays (slice k >= 1 at x_* + (k - 1) * x?_cap; slice 0 = the arrays above) and x_order holds every slice's
    // rank -> component table (slice k at x_order + k * xd_cap)
    int32_t x_slices = 0, xr_cap = 0, xc_cap = 0, xd_cap = 0, xe_cap = 0;
    int4 *x_rowinfo = nullptr, *x_cdesc = nullptr;
    int32_t *x_colid = nullptr, *x_lcol = nullptr, *x_order = nullptr;
    double *x_cost = nullptr;
    size_t bytes = 0;
};

size_t bts_lap_work_bytes(int n_cap, int m_cap, int64_t e_cap, int64_t o_cap = 0, int self_slices = 0);
// carve `w` out of a device block of bts_lap_work_bytes() bytes
void bts_lap_work_carve(bts_lap_work *w, void *block, int n_cap, int m_cap, int64_t e_cap, int64_t o_cap = 0, int self_slices = 0);
// helper workgroups bts_lap_enqueue launches per problem (`helpers` < 0: automatic, from the problem's row capacity)
int bts_lap_helpers_for(int n_max, int helpers);

// Enqueue: dense cost[n x m] (leading dimension ld, live sizes in dims[0], dims[1] on device,
// bounded by n_max/m_max known to the host) -> x[n], y[m].
int bts_lap_enqueue_dense(const bts_lap_work *w, const double *cost, int ld, const int32_t *dims,
                          int n_max, int m_max, double cost_limit, int32_t *x, int32_t *y,
                          hipStream_t stream);
// Enqueue the solver on an edge list already appended to w->e_* (hdr[0] = edge count).
// One assignment problem on an edge list already appended to w.e_* (w.hdr[0] = edge count).
// `fresh`: x/y already hold -1 and deg/cdeg are zero (the tracker keeps both invariants; the
// standalone entry point passes 0 and the kernel establishes them itself, O(n+m)).
struct bts_lap_problem {
    bts_lap_work w;
    const int32_t *dims;   // device: live n, m (nullable -> n_max, m_max)
    int n_max, m_max;
    double limit;
    int32_t *x, *y;
    int fresh;
};
int bts_lap_init();
// Enqueue 1 or 2 independent problems (they share one launch).
// `ev` (nullable): four pre-created events recorded around k_prepare and k_solve (measurement hook).
// `form`: 0 = column forms (default), 1 = the round-2 row forms (k_lap<EC, false>), -1 = the process default (BTS_LAP_FORM=row | col).
// form: 0 column forms, 1 row forms, < 0 the process default (BTS_LAP_FORM); helpers: helper workgroups per problem, < 0 automatic
int bts_lap_enqueue(const bts_lap_problem *probs, int count, hipStream_t stream, hipEvent_t *ev = nullptr, int form = -1, int helpers = -1);
#ifdef BTS_HOST_EMU
#include <utility>
#include <vector>
extern thread_local std::vector<std::pair<size_t, size_t>> *bts_emu_lap_zones;      // red zones of the work area being carved (lap.hip)
#endif
int bts_lap_form_from_env();      // (BTS_LAP_FORM through bts_form_get)
// reset hdr before edges are appended
int bts_lap_enqueue_reset(const bts_lap_work *w, hipStream_t stream);

// ---- exact-tie mode (lap_exact.h, kernel k_lap_exact in lap.hip) -----------------------------------
// Runs behind bts_lap_enqueue on the same problems: certificate that the fast solver's (x, y) is the only optimum, else the
// dense JV of lap 0.4.0 replayed on the implicit extended matrix.  One descriptor per problem: scratch + cost source.
struct bts_lap_exact {
    double *dbl = nullptr; size_t ndbl = 0;        // scratch (bts_lap_exact_sizes)
    int32_t *ints = nullptr; size_t nints = 0;
    // cost source of a real (row, column) pair that is NOT a feasible edge: a dense matrix (standalone entry point) ...
    const double *cost = nullptr; int ld = 0;
    // ... or the boxes of the stage (the tracker: non-edges are IoU costs; the edge list overrides the fused entries)
    const double *abox = nullptr; const uint8_t *af32 = nullptr; const double *bbox = nullptr; const int32_t *bidx = nullptr;
    int force_replay = 0;                          // tests: replay even when the certificate holds
};
void bts_lap_exact_sizes(int n_cap, int m_cap, int64_t e_cap, size_t *ndbl, size_t *nints);
int bts_lap_exact_enqueue(const bts_lap_problem *probs, const bts_lap_exact *xs, int count, hipStream_t stream);
// lap hdr words the exact step reports in (bts_tracker_debug_lap): certificate verdict, replay ran, watchdog / sweep flags
constexpr int BTS_LAP_H_TIED = 40, BTS_LAP_H_REPLAYED = 41, BTS_LAP_H_XERR = 42;

// ---- cosine similarity (cosine.hip) ------------------------------------------------------------
// sim[i*ld+j] = max(0, <A[a_idx[i]], B[b_idx[j]]>) ; a_idx/b_idx nullable (= identity); the live
// sizes come from dims (device, nullable -> use n_max/m_max).
int bts_cos_enqueue(int which, const float *a, const int32_t *a_idx, int lda, const float *b,
                    const int32_t *b_idx, int ldb, int d, const int32_t *dims, int n_max,
                    int m_max, float *sim, int ld, hipStream_t stream);
