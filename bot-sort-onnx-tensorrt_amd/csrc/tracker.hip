// tracker.hip -- the resident BoT-SORT tracker: one BoTSORT.update (tracking half, reference
// demo_bottrack_onnx_tflite.py:1414-1639) as a chain of kernel enqueues with NO host round trip.
//
// State lives in HBM in a slot-indexed structure-of-arrays track table (DESIGN.md "data layout"):
//   mean[T][8], cov[T][64]      float64 rows: 8 lanes own one track, a wavefront reads 8 contiguous 576-B rows
//   body_curr/smooth[T][D]      float32 ReID rows (the matcher reads `curr`, quirk Q1)
//   state/ids/frame counters[T] int32
//   tracked[], lost[]           ordered slot lists = the reference's tracked_stracks / lost_stracks
// All per-frame sizes (pool rows N, high/low detections, ...) are computed on the device and read
// by later kernels from `hdr`; grids are sized for capacity and surplus workgroups exit at once.
// That is what makes a step capturable in a hipGraph and lets frames be enqueued back to back.
#include "bts_internal.h"
#include "block_scan.h"
#include "cost_math.h"
#include "feat_math.h"
#include "iou_math.h"
#include "kf_math.h"

namespace {

enum {   // hdr[] fields
    H_FRAME = 0, H_NTRACKED, H_NLOST, H_NREMOVED, H_LASTID, H_NPOOL, H_NHI, H_NLO, H_NUNC, H_NR2, H_NLEFT,
    H_ERR, H_NCONF, H_ALLF32, H_NFREE, H_NDETS, H_NKEEP, H_NB, H_NRF, H_NL0, H_NL1, H_NBORN,
    H_DIMS1 = 24, H_DIMS2 = 26, H_DIMS3 = 28,      // ([30], [31]: frame and count of the first capacity error)
    // BTS_LISTS_AHEAD: the NEXT frame's pool / unconfirmed sizes, produced by tk_finalize, moved into place by the next tk_front_dets
    H_NPOOL_N = 32, H_NCONF_N, H_NUNC_N, H_ALLF32_N, H_SIZE = 40
};
enum { ERR_TRACK_CAPACITY = 1, ERR_LAP_EDGES = 2, ERR_CHUNK_WAIT = 4 };
// list jobs that are split over `Dev::nch` consecutive workgroups (bscan::Chunk): their slices of Dev::agg
enum { J_POOL = 0, J_UNC, J_R2, J_KEEP, J_RF, J_L0, J_L1, J_FT, J_FL, J_PB, J_COUNT };

// Per-workgroup durations of the list jobs (make TRACE=1, bts_debug_jobtrace): 100 MHz ticks at g_jobtrace[kernel * 8 + job].
__device__ int g_jobtrace[64];
#ifdef BTS_LAP_TRACE
#define TJ_BEGIN() const unsigned long long tj_t0 = __builtin_amdgcn_s_memrealtime()
#define TJ_END(kern, job) do { __syncthreads(); if (threadIdx.x == 0) g_jobtrace[(kern) * 8 + (job)] = (int)(__builtin_amdgcn_s_memrealtime() - tj_t0); } while (0)
#else
#define TJ_BEGIN() do { } while (0)
#define TJ_END(kern, job) do { } while (0)
#endif

// The frame's inputs: eager steps carry them in the kernel arguments; under hipGraph replay the
// kernel arguments are frozen at capture, so they come from one small device-resident descriptor
// refreshed by a 40-byte async copy per step.
struct InDesc {
    const int32_t *boxes;
    const float *scores;
    float *feats, *face_feats;
    int m;
    unsigned epoch;            // step counter (never 0): tags the chunk counts of the split list jobs
};

struct Dev {
    int T, Mx, D, Df;
    // thresholds (float64 like the python attributes)
    double high, low, new_thr, match1, match2, match3, prox, dedupe_thr;
    float app, ema_a, ema_b;
    int max_time_lost;
    // track table
    double *mean, *cov;
    uint8_t *f32, *in_removed, *in_use;
    unsigned long long *agg;   // [J_COUNT][CHUNK_MAX] chunk counts of the split list jobs
    int nch;                   // workgroups per split list job
    int32_t *state, *is_activated, *track_id, *frame_id, *start_frame, *tracklet_len, *det_index;
    float *score;
    float *body_curr, *body_smooth, *face_curr, *face_smooth;
    uint8_t *has_body, *has_face;
    int32_t *tracked, *lost, *tracked2, *lost2, *free_list;
    int32_t *hdr;
    // per-frame work
    int32_t *pool_slot, *unc_slot, *hi_idx, *lo_idx, *r2_row, *r2_slot, *left_hi, *left_det, *born_slot;
    int32_t *r2_of_slot, *unc_of_slot;                 // BTS_MERGE_EARLY: slot -> index in the stage-2 row list / the unconfirmed list (validated by a read-back)
    int renorm_fused;                                  // BTS_RENORM_FUSED=1: a stage-1 matched detection row is renormalised by the workgroup that absorbs it (tk_mid2)
    int merge_early;                                   // BTS_MERGE_EARLY=1: the list jobs of tk_merge ride in the launch of tk_apply23
    double *pool_tlbr, *unc_tlbr, *det_tlbr, *r2_tlbr;
    uint8_t *unc_f32, *ones;
    float *det_xywh;
    float *sim, *fsim;
    double *cost1, *cost2, *cost3;
    int32_t *x1, *y1, *x2, *y2, *x3, *y3;
    // dedupe
    uint8_t *drop_t, *drop_l;
    int32_t *lap_hdr;          // 3 x 64 ints: per-stage counters of the assignment solver (BTS_LAP_HDR_INTS each)
    int record;                // keep dense cost matrices (tests / inspection)
    int motion_gate;           // optional Mahalanobis gate on the stage-1 edges (tk_gate_edges)
    double gate_thr;           // chi2inv95[4], D:139-149
    int filter;                // appearance term by int8-MFMA pre-filter + exact float32 refine
    int8_t *body_h, *face_h, *det_h, *dface_h;        // int8 shadows of body_curr/face_curr and of the frame's rows (feat::to_i8_row)
    float *body_m, *face_m, *det_m, *dface_m;         // their records [rows][feat::I8_META]: scale, residual, prefix norms
    int32_t *approx, *approx_f;                        // [T][Mx] split-K accumulation of the pre-filter, body / face (self-cleaning; exact integers)
    uint32_t *cbits;                                   // [T][cb_ld] candidate bitmap of the one-pass pre-filter (self-cleaning)
    int cb_ld;                                         // 32-bit words per bitmap row
    // BTS_FILTER_AHEAD=1 at create (opt-in throughput mode, round 6; see enqueue_step): the bitmap of a frame is [T + Mx][cb_ld]
    // -- rows [0, T) by table SLOT (main pass), rows [T, T + Mx) by the detection index of the PREVIOUS frame (det x det pass) --
    // and pool row i finds its bitmap row in pool_cbrow[i] (written with the pool list by tk_finalize); nullptr = bitmap row i
    int32_t fahead;
    int32_t *pool_cbrow;
    // BTS_SMOOTH_DEFERRED=1 at create (with BTS_FILTER_AHEAD): a step writes a rewritten track's `curr` row only and marks the slot
    // (1: update with a previous feature, 2: update without, 3: newborn); tk_smooth, on a third stream behind the step, brings the
    // `smooth` row up to date from `curr` -- nothing in a step reads `smooth` (the matcher reads `curr`, Q1)
    int32_t smooth_def;
    uint8_t *ema_pending;
    // BTS_EDGES_SORTED together with BTS_FILTER_AHEAD: the candidate LISTS of the two side passes sit one behind the other (fa_mt main
    // tiles, then fa_dt det x det tiles, four slots each; the det x det region starts at slot fa_dd_slot0) and name rows by table slot /
    // previous-frame detection; the candidate wavefronts translate them to pool rows through pool_of_slot / pool_of_pdet (written
    // with pool_cbrow by tk_finalize, validated by reading pool_cbrow back: a stale entry is never taken for a membership)
    int32_t fa_mt, fa_dt, fa_dd_slot0, fa_nprev;
    int32_t *pool_of_slot, *pool_of_pdet;
    float gate_sim;                                    // the similarity the appearance term must reach: 1 - appearance_thresh
    int sf32;                                          // kalman_dtype BTS_F32: mean / cov hold float32 elements (same rows, half the bytes)
    int c16;                                           // cost_dtype BTS_F16: association costs rounded to binary16 before thresholds and the solver
    // shadow layout (feat_math.h, i8_elem): row stride / chunk stride in bytes of the track-side and detection-side shadow arrays,
    // body [0] and face [1] (row-major: D, 128; BTS_SHADOW_PLANES: 128, rows * 128)
    unsigned sh_rt[2], sh_ct[2], sh_rd[2], sh_cd[2];
    int sh_planes;
    int f_guess;                                       // BTS_FILTER_GUESS=1: a workgroup's first tile is already predicted to stop at the first checkpoint
    int chk_first;                                     // first prefix checkpoint of the pre-filter, in sixteenths of the row (feat::i8_checkpoints)
    int32_t use_grid;                                   // BTS_EDGES_GRID=1 at create: stage-1 edges by neighbour search (experimental; wins only for very large frames)
    int32_t *grid_hdr, *cell_start, *cell_items;       // uniform grid over the frame's high-score detections (prep_dets -> tk_edges_grid)
    // BTS_EDGES_SORTED=1 at create (opt-in stage-1 form, tk_edges_sorted): the pre-filter's candidates also leave it as a LIST --
    // one slot of CAND_W {pool row, high-score column} pairs per (tile, wavefront) quadrant, count in cand_cnt (-1: more than a
    // slot holds, the quadrant's bitmap words are swept instead); consumed by the candidate wavefronts of tk_edges_sorted,
    // cleared (with the bitmap bits) by tk_mid1
    int32_t sorted;                                    // the form is on for this tracker
    int32_t ahead;                                     // BTS_LISTS_AHEAD=1 at create: tk_finalize also builds the next frame's pool / unconfirmed / free lists
    int32_t *cand_cnt;                                 // [f_slots]
    int2 *cand;                                        // [f_slots][CAND_W]
    int32_t f_ntiles, f_mc;                            // this step's pre-filter tiling (tiles, detection bound): what a slot index decodes with
    int32_t *fstat;                                    // [8] pre-filter counters of the frame: K stages run, tiles, prefix exits, candidates
    const InDesc *in;          // inputs of the current step: descriptor in memory (hipGraph replay) ...
    InDesc iv;                 // ... or by value in the kernel arguments (eager launches, in == nullptr)
};
#define IN(d, f) ((d).in ? (d).in->f : (d).iv.f)

__device__ __forceinline__ bscan::Chunk chunk_of(const Dev &d, int job, int ch) {
    return bscan::Chunk{d.agg + job * bscan::CHUNK_MAX, IN(d, epoch), ch, d.nch, d.hdr + H_ERR, ERR_CHUNK_WAIT};
}
// The list kernels exist in two instantiations: CH = false is the single-workgroup form of every job (tables up to
// ~2048 tracks: the chunk machinery only costs there -- code size on a cold instruction cache), CH = true the split form.
template <bool CH> __device__ __forceinline__ int nchunks(const Dev &d) { return CH ? d.nch : 1; }
template <bool CH, class Load, class Pred, class Emit>
__device__ __forceinline__ int job_compact(const bscan::Chunk &cx, int L, int32_t *tmp, Load load, Pred pred, Emit emit, bool flag = false,
                                           bool want_all = false, int *grand_total = nullptr, bool *any_flag = nullptr) {
    if constexpr (CH) {
        return bscan::compact_chunk(cx, L, tmp, load, pred, emit, flag, want_all, grand_total, any_flag);
    } else {
        const int n = bscan::compact_ws(L, tmp, load, pred, emit);
        if (grand_total) *grand_total = n;
        if (any_flag) *any_flag = flag;
        return n;
    }
}
template <bool CH> __device__ __forceinline__ void job_range(const bscan::Chunk &cx, int L, int &lo, int &hi) {
    if constexpr (CH) bscan::chunk_range(cx, L, lo, hi);
    else { lo = 0; hi = L; }
}

// ---- single-workgroup helpers: bscan::exscan / bscan::compact (block_scan.h) ------------------------
template <class Load, class Pred, class Emit>
__device__ __forceinline__ int block_compact(int L, int32_t *tmp, Load load, Pred pred, Emit emit) { return bscan::compact_ws(L, tmp, load, pred, emit); }
__device__ __forceinline__ int block_exscan_i(int v, int32_t *tmp, int *total) { return bscan::exscan(v, tmp, total); }

// The Kalman state of a slot: float64 rows (default, the reference's dtype), or -- kalman_dtype BTS_F32 -- float32 elements in
// the same arrays; either way the arithmetic works on float64 registers (the float32 form widens at the load and rounds at
// the store).  d.sf32 is uniform over the launch: one scalar branch per access group.
__device__ __forceinline__ double ld_mean(const Dev &d, long slot, int c) {
    return d.sf32 ? (double)reinterpret_cast<const float *>(d.mean)[slot * 8 + c] : d.mean[slot * 8 + c];
}
__device__ __forceinline__ void st_mean(const Dev &d, long slot, int c, double v) {
    if (d.sf32) reinterpret_cast<float *>(d.mean)[slot * 8 + c] = (float)v; else d.mean[slot * 8 + c] = v;
}
__device__ __forceinline__ double ld_cov(const Dev &d, long slot, int c) {
    return d.sf32 ? (double)reinterpret_cast<const float *>(d.cov)[slot * 64 + c] : d.cov[slot * 64 + c];
}
__device__ __forceinline__ void st_cov(const Dev &d, long slot, int c, double v) {
    if (d.sf32) reinterpret_cast<float *>(d.cov)[slot * 64 + c] = (float)v; else d.cov[slot * 64 + c] = v;
}
__device__ __forceinline__ void load_state(const Dev &d, int slot, double m[8], double P[64]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = ld_mean(d, slot, c);
#pragma unroll
    for (int c = 0; c < 64; ++c) P[c] = ld_cov(d, slot, c);
}
__device__ __forceinline__ void store_state(const Dev &d, int slot, const double m[8], const double P[64]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) st_mean(d, slot, c, m[c]);
#pragma unroll
    for (int c = 0; c < 64; ++c) st_cov(d, slot, c, P[c]);
}
// 8-lane form: lane i of the group holds element i of the mean and row i of the covariance
__device__ __forceinline__ void load_row(const Dev &d, int slot, int i, double &mi, double Pr[8]) {
    if (d.sf32) {
        mi = (double)reinterpret_cast<const float *>(d.mean)[(long)slot * 8 + i];
        const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(d.cov) + (long)slot * 64 + i * 8);
        const float4 a = p[0], b = p[1];
        Pr[0] = a.x; Pr[1] = a.y; Pr[2] = a.z; Pr[3] = a.w; Pr[4] = b.x; Pr[5] = b.y; Pr[6] = b.z; Pr[7] = b.w;
        return;
    }
    mi = d.mean[(long)slot * 8 + i];
    const double2 *p = reinterpret_cast<const double2 *>(d.cov + (long)slot * 64 + i * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const double2 v = p[j]; Pr[2 * j] = v.x; Pr[2 * j + 1] = v.y; }
}
__device__ __forceinline__ void store_row(const Dev &d, int slot, int i, double mi, const double Pr[8]) {
    if (d.sf32) {
        reinterpret_cast<float *>(d.mean)[(long)slot * 8 + i] = (float)mi;
        float4 *p = reinterpret_cast<float4 *>(reinterpret_cast<float *>(d.cov) + (long)slot * 64 + i * 8);
        p[0] = make_float4((float)Pr[0], (float)Pr[1], (float)Pr[2], (float)Pr[3]);
        p[1] = make_float4((float)Pr[4], (float)Pr[5], (float)Pr[6], (float)Pr[7]);
        return;
    }
    d.mean[(long)slot * 8 + i] = mi;
    double2 *p = reinterpret_cast<double2 *>(d.cov + (long)slot * 64 + i * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = make_double2(Pr[2 * j], Pr[2 * j + 1]);
}
__device__ __forceinline__ void slot_tlbr(const Dev &d, int slot, double out[4]) {
    double m[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) m[c] = ld_mean(d, slot, c);
    kf::mean_to_tlbr(m, d.f32[slot] != 0, out);
}

// ---- K1: frame prologue: four independent single-workgroup jobs + the detection-row conversion ---------
// D:1414-1423 (split tracked into confirmed / unconfirmed, pool = confirmed ++ lost) and the
// detection records of D:1488-1532 (high: score > 0.40, low: 0.1 <= score <= 0.40; float32 boxes).
// Each job owns its hdr words; the list order is semantics (it defines the row indices).
template <bool CH> __device__ __forceinline__ void prep_pool(const Dev &d, int ch) {
    __shared__ int32_t tmp[1024];
    int32_t *h = d.hdr;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int nt = h[H_NTRACKED], nl = h[H_NLOST];
    const bscan::Chunk cx = chunk_of(d, J_POOL, ch);
    int lo, hi;
    job_range<CH>(cx, nl, lo, hi);
    // "every pool row still holds float32 state" (np.asarray of the batch is then float32, Q10) can only be true on
    // the frame after the one that activated its births at once (frame 1, D:565-566): later births enter the pool
    // through an update, which makes their state float64.  Each chunk looks at its own rows; the verdict travels
    // with the chunk counts.
    const bool chk = h[H_FRAME] == 1;
    int f64 = 0;
    if (chk) {
        int a, b;
        job_range<CH>(cx, nt, a, b);
        for (int k = a + tid; k < b; k += nth) { const int s = d.tracked[k]; if ((d.is_activated[s] != 0) & !d.f32[s]) f64 = 1; }
        for (int k = lo + tid; k < hi; k += nth) if (!d.f32[d.lost[k]]) f64 = 1;
        f64 = __syncthreads_or(f64);
    }
    int nconf = 0;
    bool any64 = false;
    job_compact<CH>(cx, nt, tmp, [&](int k) { return d.tracked[k]; }, [&](int k, int s) { return d.is_activated[s] != 0; },
                         [&](int p, int k, int s) { d.pool_slot[p] = s; }, f64 != 0, true, &nconf, &any64);
    for (int k = lo + tid; k < hi; k += nth) d.pool_slot[nconf + k] = d.lost[k];
    if (ch == 0 && tid == 0) {           // (every chunk has read H_FRAME by now: all of them have published)
        const int N = nconf + nl;
        h[H_FRAME] += 1;
        h[H_NPOOL] = N; h[H_NCONF] = nconf;
        h[H_ALLF32] = (N > 0 && chk && !any64) ? 1 : 0;
        h[H_DIMS1] = N;
        h[H_NR2] = 0; h[H_DIMS2] = 0;
    }
}

template <bool CH> __device__ __forceinline__ void prep_unconfirmed(const Dev &d, int ch) {
    __shared__ int32_t tmp[1024];
    int32_t *h = d.hdr;
    const int nt = h[H_NTRACKED];
    const int nunc = job_compact<CH>(chunk_of(d, J_UNC, ch), nt, tmp, [&](int k) { return d.tracked[k]; },
                                          [&](int k, int s) { return d.is_activated[s] == 0; }, [&](int p, int k, int s) { d.unc_slot[p] = s; if (d.merge_early) d.unc_of_slot[s] = p; });
    if (ch == nchunks<CH>(d) - 1 && threadIdx.x == 0) { h[H_NUNC] = nunc; h[H_DIMS3] = nunc; }
}

// ---- uniform grid over the high-score detections, for the stage-1 neighbour search (tk_edges_grid) ---------------------
// A pair (track, detection) can only get an IoU edge when the boxes overlap, i.e. when the detection's CENTRE lies inside
// the track's box widened by half the detection's size.  The detections are binned by centre into square cells at least
// as large as the largest detection of the frame, so a track looks at the few cells its widened box touches (3 x 3 for
// boxes of similar size) instead of at every detection.  Coordinates are DOUBLED integers (2 cx = x1 + x2: no halves).
// Built by the workgroup that has just compacted the high-score list: reductions, counts (LDS atomics), scan, scatter.
constexpr int GRID_MAX = 128, GRID_CELLS = GRID_MAX * GRID_MAX;
enum { G_OX = 0, G_OY, G_CS, G_GX, G_GY, G_MAXW, G_MAXH, G_NCELL };
__device__ __forceinline__ void build_grid(const Dev &d, int nhi, int32_t *tmp) {
    __shared__ int gcnt[GRID_CELLS];
    __shared__ int gred[8];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int4 *boxes = reinterpret_cast<const int4 *>(IN(d, boxes));
    if (tid < 8) gred[tid] = tid < 2 ? 0x7fffffff : tid < 4 ? (int)0x80000000 : 0;
    __syncthreads();                                   // (also: the high-score list written above is visible to the whole workgroup)
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = (int)0x80000000, mxy = (int)0x80000000, mw = 0, mh = 0;
    for (int p = tid; p < nhi; p += nth) {
        const int4 b = boxes[d.hi_idx[p]];
        const int cx2 = b.x + b.z, cy2 = b.y + b.w;
        mnx = min(mnx, cx2); mny = min(mny, cy2); mxx = max(mxx, cx2); mxy = max(mxy, cy2);
        mw = max(mw, b.z - b.x); mh = max(mh, b.w - b.y);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, off)); mny = min(mny, __shfl_xor(mny, off));
        mxx = max(mxx, __shfl_xor(mxx, off)); mxy = max(mxy, __shfl_xor(mxy, off));
        mw = max(mw, __shfl_xor(mw, off)); mh = max(mh, __shfl_xor(mh, off));
    }
    if ((tid & 63) == 0) {
        atomicMin(&gred[0], mnx); atomicMin(&gred[1], mny); atomicMax(&gred[2], mxx); atomicMax(&gred[3], mxy);
        atomicMax(&gred[4], mw); atomicMax(&gred[5], mh);
    }
    __syncthreads();
    const int ox = gred[0], oy = gred[1], maxw = max(gred[4], 0), maxh = max(gred[5], 0);
    const long spanx = nhi ? (long)gred[2] - ox : 0, spany = nhi ? (long)gred[3] - oy : 0;
    long cs = 2L * max(max(maxw, maxh), 1);            // cell side (doubled coordinates) >= the largest detection
    cs = max(cs, max(spanx, spany) / GRID_MAX + 1);    // ... and no more than GRID_MAX cells per axis
    const int gx = (int)(spanx / cs) + 1, gy = (int)(spany / cs) + 1, ncell = gx * gy;
    for (int c = tid; c < ncell; c += nth) gcnt[c] = 0;
    __syncthreads();
    auto cell_of = [&](int p) {
        const int4 b = boxes[d.hi_idx[p]];
        return (int)(((long)(b.y + b.w) - oy) / cs) * gx + (int)(((long)(b.x + b.z) - ox) / cs);
    };
    for (int p = tid; p < nhi; p += nth) atomicAdd(&gcnt[cell_of(p)], 1);
    __syncthreads();
    {   // exclusive scan of the counts: every thread owns a contiguous run of cells
        const int per = (ncell + nth - 1) / nth, lo = min(ncell, tid * per), hi = min(ncell, lo + per);
        int sum = 0;
        for (int c = lo; c < hi; ++c) sum += gcnt[c];
        int total;
        int run = bscan::exscan(sum, tmp, &total);
        for (int c = lo; c < hi; ++c) { const int v = gcnt[c]; gcnt[c] = run; d.cell_start[c] = run; run += v; }
        if (tid == 0) d.cell_start[ncell] = nhi;
    }
    __syncthreads();
    for (int p = tid; p < nhi; p += nth) d.cell_items[atomicAdd(&gcnt[cell_of(p)], 1)] = p;
    if (tid == 0) {
        int32_t *g = d.grid_hdr;
        g[G_OX] = ox; g[G_OY] = oy; g[G_CS] = (int)min(cs, 0x7fffffffL); g[G_GX] = gx; g[G_GY] = gy;
        g[G_MAXW] = maxw; g[G_MAXH] = maxh; g[G_NCELL] = ncell;
    }
    __syncthreads();
}

// GRID: also bin the high-score detections for tk_edges_grid (its 64 KiB of LDS counters would otherwise sit in EVERY
// tk_front workgroup and leave one workgroup per CU instead of two)
template <bool GRID> __device__ __forceinline__ void prep_dets(const Dev &d) {
    __shared__ int32_t tmp[1024];
    int32_t *h = d.hdr;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int m = IN(d, m);
    const int32_t *boxes = IN(d, boxes);
    const float *scores = IN(d, scores);
    for (int j = tid; j < m; j += nth) {
        const int4 b = reinterpret_cast<const int4 *>(boxes)[j];
        const float tx = (float)b.x, ty = (float)b.y, w = (float)(b.z - b.x), hh = (float)(b.w - b.y);   // _tlwh, D:465
        d.det_tlbr[j * 4 + 0] = (double)tx; d.det_tlbr[j * 4 + 1] = (double)ty;
        d.det_tlbr[j * 4 + 2] = (double)(w + tx); d.det_tlbr[j * 4 + 3] = (double)(hh + ty);         // D:635-642
        d.det_xywh[j * 4 + 0] = tx + w / 2; d.det_xywh[j * 4 + 1] = ty + hh / 2;                     // D:663-670
        d.det_xywh[j * 4 + 2] = w; d.det_xywh[j * 4 + 3] = hh;
    }
    // high- and low-score lists in one pass over the scores
    const int2 nn = bscan::compact2_ws(m, tmp, [&](int j) { return __float_as_int(scores[j]); },
                                       [&](int j, int v) { return (double)__int_as_float(v) > d.high; },
                                       [&](int j, int v) { const double s = (double)__int_as_float(v); return (s <= d.high) & (s >= d.low); },
                                       [&](int p, int j, int) { d.hi_idx[p] = j; }, [&](int p, int j, int) { d.lo_idx[p] = j; });
    const int nhi = nn.x, nlo = nn.y;
    if constexpr (GRID) { if (d.filter && (d.use_grid || d.sorted)) build_grid(d, nhi, tmp); }
    if (tid < 3 * BTS_LAP_HDR_INTS) d.lap_hdr[tid] = 0;
    if (tid < 16 && d.filter) d.fstat[tid] = tid == 8 || tid == 12 ? 0x7fffffff : 0;
    if (tid == 0) {
        h[H_NHI] = nhi; h[H_NLO] = nlo; h[H_NDETS] = m;
        h[H_DIMS1 + 1] = nhi; h[H_DIMS2 + 1] = nlo;
        h[H_NLEFT] = 0; h[H_NBORN] = 0; h[H_DIMS3 + 1] = 0;
    }
}

// slot recycling: a slot is free iff it is on neither list (its track was removed, dropped as a
// duplicate or filtered by quirk Q6).  Rebuilt here, off the critical path: births happen in tk_merge.
constexpr int FREE_BITS_WORDS = 8192;       // LDS bitmap of the slots in use: tables up to 262144 tracks
__device__ __forceinline__ void prep_free(const Dev &d) {
    __shared__ int32_t tmp[1024];
    __shared__ uint32_t used[FREE_BITS_WORDS];
    int32_t *h = d.hdr;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int nt = h[H_NTRACKED], nl = h[H_NLOST];
    const bool lds = d.T <= 32 * FREE_BITS_WORDS;
    if (lds) {                               // marks go to LDS: the lists are read coalesced, nothing is scattered to memory
        for (int k = tid; k < (d.T + 31) / 32; k += nth) used[k] = 0u;
        __syncthreads();
        bscan::for_batched<4>(nt, [&](int k) { return d.tracked[k]; }, [&](int k, int s) { atomicOr(&used[s >> 5], 1u << (s & 31)); });
        bscan::for_batched<8>(nl, [&](int k) { return d.lost[k]; }, [&](int k, int s) { atomicOr(&used[s >> 5], 1u << (s & 31)); });
        __syncthreads();
    } else {
        for (int s = tid; s < d.T; s += nth) d.in_use[s] = 0;
        __syncthreads();
        bscan::for_batched<4>(nt, [&](int k) { return d.tracked[k]; }, [&](int k, int s) { d.in_use[s] = 1; });
        bscan::for_batched<8>(nl, [&](int k) { return d.lost[k]; }, [&](int k, int s) { d.in_use[s] = 1; });
        __syncthreads();
    }
    const int nfree = block_compact(d.T, tmp, [&](int s) { return lds ? (int)((used[s >> 5] >> (s & 31)) & 1u) : (int)d.in_use[s]; },
                                    [&](int s, int v) { return v == 0; }, [&](int p, int s, int) { d.free_list[p] = s; });
    if (tid == 0) h[H_NFREE] = nfree;
}

// ---- K2: STrack.multi_predict over the pool (D:524-536 + D:265-302) ---------------------------------
// int8 shadows + records of this frame's (raw) detection rows for the appearance pre-filter
__device__ __forceinline__ void cvt_body(const Dev &d, long blk) {        // four rows per 1024-thread workgroup
    const long j = blk * 4 + feat::row_group();
    const bool on = j < IN(d, m);
    feat::to_i8_row(IN(d, feats) + j * d.D, d.det_h + j * d.sh_rd[0], d.det_m + j * feat::I8_META, d.D, d.gate_sim, on, d.chk_first, d.sh_cd[0]);
    if (d.Df) feat::to_i8_row(IN(d, face_feats) + j * d.Df, d.dface_h + j * d.sh_rd[1], d.dface_m + j * feat::I8_META, d.Df, d.gate_sim, on, d.chk_first, d.sh_cd[1]);
}
// A track's latest feature IS a detection row (curr = the matched / born detection's renormalised row, D:492-502): its
// shadow is that detection's, copied, and so is its record -- which tk_mid1 has by then turned into the record of the
// UNIT row x / |x| (feat::i8_track_side: a detection row need not arrive unit-norm; the record's slack covers the 1e-7
// between x / |x| in exact arithmetic and the float32 row).  Called by the 256 threads that handle the row.
// (src_cs / dst_cs: chunk strides of the two shadow arrays, feat::i8_elem)
__device__ __forceinline__ void copy_shadow(const int8_t *src_q, const float *src_m, int8_t *dst_q, float *dst_m, int dim, unsigned src_cs, unsigned dst_cs) {
    const int t = feat::row_tid();
    if (dim % 16 == 0) {
        for (int k = t; k < dim / 16; k += feat::ROW_G)          // (sixteen elements never straddle a chunk)
            *reinterpret_cast<uint4 *>(dst_q + feat::i8_elem(16 * k, dst_cs)) = *reinterpret_cast<const uint4 *>(src_q + feat::i8_elem(16 * k, src_cs));
    } else {
        for (int k = t; k < dim; k += feat::ROW_G) dst_q[feat::i8_elem(k, dst_cs)] = src_q[feat::i8_elem(k, src_cs)];
    }
    if (t < feat::I8_META) dst_m[t] = src_m[t];
}

// NEXT = true (BTS_PREDICT_FRONT, with BTS_LISTS_AHEAD): called from the frame's FIRST launch, where the pool's sizes still sit in
// the *_N header words the previous tk_finalize left (workgroup 0 of that launch moves them into place at the same time)
template <bool NEXT = false> __device__ __forceinline__ void predict_body(const Dev &d, int bid) {
    const int g = bid * (int)blockDim.x + threadIdx.x;
    // the three assignment vectors start out "unmatched" (the solver only writes matches)
    if (g < d.T) { d.x1[g] = -1; d.x2[g] = -1; d.x3[g] = -1; }
    if (g < IN(d, m)) { d.y1[g] = -1; d.y2[g] = -1; d.y3[g] = -1; }
    const int r = g >> 3, i = g & 7;                     // 8 lanes per pool row
    const int s_raw = d.pool_slot[min(r, d.T - 1)];      // (travels with the extent)
    if (r >= d.hdr[NEXT ? H_NPOOL_N : H_NPOOL]) return;
    const int s = s_raw;
    double mi, Pr[8];
    load_row(d, s, i, mi, Pr);
    if (i >= 6 && d.state[s] != BTS_TRACKED) mi = 0.0;   // D:529-532
    kf8::predict(mi, Pr, i, d.hdr[NEXT ? H_ALLF32_N : H_ALLF32] != 0);       // a batch is float32 only if every row is (np.asarray)
    store_row(d, s, i, mi, Pr);
    double b[4];
    kf8::tlbr(mi, false, b);
    if (i < 4) d.pool_tlbr[r * 4 + i] = b[i];
    if (i == 0) d.f32[s] = 0;
}

// one launch: the first workgroups build the frame's lists (two jobs of nch chunks each, the detection lists, the free slots),
// the others convert the detection rows
__host__ __device__ constexpr int front_jobs(int nch) { return 2 * nch + 2; }
// CVT = false (opt-in, BTS_CVT_OVERLAP=1 at create; NOT yet measured): the detection-row conversion is not part of this launch --
// it runs as tk_cvt on a side stream, where frame t+1's rows (which depend on nothing the tracker computes) are converted
// under frame t's latency-bound kernels (lever (c)).  The default instantiations are unchanged code.
template <bool CH, bool GRID = false, bool CVT = true> __global__ __launch_bounds__(1024) void tk_front(Dev d) {
    TJ_BEGIN();
    const int b = blockIdx.x, nch = nchunks<CH>(d);
    if (b < nch) { prep_pool<CH>(d, b); if (b == 0) TJ_END(0, 0); }
    else if (b < 2 * nch) { prep_unconfirmed<CH>(d, b - nch); if (b == 2 * nch - 1) TJ_END(0, 1); }
    else if (b == 2 * nch) { prep_dets<GRID>(d); TJ_END(0, 2); }
    else if (b == 2 * nch + 1) { prep_free(d); TJ_END(0, 3); }
    else if constexpr (CVT) { cvt_body(d, (long)b - front_jobs(nch)); if (b == front_jobs(nch)) TJ_END(0, 4); }
}
__global__ __launch_bounds__(1024) void tk_cvt(Dev d) { cvt_body(d, (long)blockIdx.x); }
// BTS_FILTER_AHEAD: the side stream's opening launch for a frame.  The row workgroups convert the frame's detection rows like
// tk_cvt and ALSO leave each row's record in its track-side form in `det_mts` (what feat::i8_track_side makes of it when the row is
// renormalised and absorbed by a track, with the norm taken in renorm_row's own summation order: bit-identical to what the
// absorbing track's record will be) -- the row records of the NEXT frame's det x det pass.  The last workgroup builds the frame's
// high-score list for the side passes (the predicate and order of prep_dets, which builds the step's own copy later) and the two
// little headers the unchanged pre-filter kernel reads its extents from.
// The first `clear_blocks` workgroups clear what the two passes behind this launch write with atomics: the words of the frame's mc
// columns (`row_words` per row) in bitmap rows [0, T + n_prev) -- the row stride is sized for max_dets, far above a frame's count in
// some callers -- and, for the sorted edges, the counts of the buffer's two candidate lists.  (Three memsets on the side stream
// before: three launches and three host calls per step.)
__global__ __launch_bounds__(1024) void tk_cvt_ahead(Dev d, float *det_mts, int32_t *hi_b, int32_t *hdr_main, int32_t *hdr_dd, int32_t *fstat_b, int n_prev,
                                                     int clear_blocks, int row_words, int cand_slots_n) {
    if ((int)blockIdx.x < clear_blocks) {
        const long nth = (long)clear_blocks * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
        const long words = (long)(d.T + n_prev) * row_words;
        for (long k = t0; k < words; k += nth) d.cbits[(k / row_words) * d.cb_ld + k % row_words] = 0u;
        for (long k = t0; k < cand_slots_n; k += nth) d.cand_cnt[k] = 0;
        return;
    }
    const int b = (int)blockIdx.x - clear_blocks;
    if (blockIdx.x + 1 == gridDim.x) {
        __shared__ int32_t tmp[1024];
        const float *scores = IN(d, scores);
        const int nhi = bscan::compact_ws(IN(d, m), tmp, [&](int j) { return __float_as_int(scores[j]); },
                                          [&](int j, int v) { return (double)__int_as_float(v) > d.high; }, [&](int p, int j, int) { hi_b[p] = j; });
        if (threadIdx.x < 16) fstat_b[threadIdx.x] = threadIdx.x == 8 || threadIdx.x == 12 ? 0x7fffffff : 0;
        if (threadIdx.x == 0) { hdr_main[H_NPOOL] = d.T; hdr_main[H_NHI] = nhi; hdr_dd[H_NPOOL] = n_prev; hdr_dd[H_NHI] = nhi; }
        return;
    }
    const long j = (long)b * 4 + feat::row_group();
    const bool on = j < IN(d, m);
    cvt_body(d, (long)b);
    const float nrm = feat::row_norm(IN(d, feats) + j * d.D, d.D, on);      // (every thread of the workgroup: barriers inside)
    if (on && feat::row_tid() == 0) {       // (the thread that wrote the row's record in to_i8_row)
        float rec[feat::I8_META];
        for (int k = 0; k < feat::I8_META; ++k) rec[k] = d.det_m[j * feat::I8_META + k];
        feat::i8_track_side(rec, nrm, d.gate_sim);
        for (int k = 0; k < feat::I8_META; ++k) det_mts[j * feat::I8_META + k] = rec[k];
    }
}
// BTS_LISTS_AHEAD: the frame opens with the detection side alone -- workgroup 0 builds the detection lists, moves the sizes the
// previous tk_finalize left in the *_N header words into place (what prep_pool / prep_unconfirmed wrote at this point) and
// clears the duplicate-removal flags; the others convert the detection rows (unless tk_cvt does, CVT = false)
// pred_blocks > 0 (BTS_PREDICT_FRONT=1 at create; NOT yet measured): the Kalman predict of the pool -- whose list exists since the
// previous frame's tk_finalize -- runs in the LAST pred_blocks workgroups of this launch instead of behind the pre-filter's
// persistent workgroups, where it is the tail of the step's longest kernel.
template <bool GRID, bool CVT> __global__ __launch_bounds__(1024) void tk_front_dets(Dev d, int pred_blocks) {
    const int first_pred = (int)gridDim.x - pred_blocks;
    if ((int)blockIdx.x >= first_pred) { predict_body<true>(d, (int)blockIdx.x - first_pred); return; }
    if (blockIdx.x != 0) { if constexpr (CVT) cvt_body(d, (long)blockIdx.x - 1); return; }
    prep_dets<GRID>(d);
    int32_t *h = d.hdr;
    for (int s = threadIdx.x; s < d.T; s += blockDim.x) { d.drop_t[s] = 0; d.drop_l[s] = 0; }
    if (threadIdx.x == 0) {
        const int N = h[H_NPOOL_N], nunc = h[H_NUNC_N];
        h[H_FRAME] += 1;
        h[H_NPOOL] = N; h[H_NCONF] = h[H_NCONF_N]; h[H_ALLF32] = h[H_ALLF32_N];
        h[H_DIMS1] = N; h[H_NR2] = 0; h[H_DIMS2] = 0;
        h[H_NUNC] = nunc; h[H_DIMS3] = nunc;
    }
}

__global__ __launch_bounds__(256) void tk_predict(Dev d) { predict_body(d, blockIdx.x); }

// ---- cost tiles ----------------------------------------------------------------------------------------
constexpr int CT_ROWS = 16, CT_COLS = 256;

// ---- optional motion gate on the stage-1 pairs --------------------------------------------------------------------
// KalmanFilter.gating_distance (D:338-380, metric 'maha', all four coordinates) is defined by the reference but never
// called from its update loop; with bts_tracker_config.motion_gate the cascade applies it the way
// upstream BoT-SORT's gate_cost_matrix does: a stage-1 pair whose squared Mahalanobis distance between the track's
// predicted state and the detection's (cx, cy, w, h) exceeds chi2inv95[4] gets cost +inf, i.e. it never becomes an
// edge of the solver's list.  Only pairs that WOULD be edges (cost <= match threshold; about one per detection) can be
// affected, so the test runs inside the stage-1 tile kernel on exactly those lanes, right before the edge is staged
// (GATE instantiations of tk_edges / tk_edges_tall, launched only when the option is on): the N x M distance matrix is
// never formed and the step keeps its 11 launches.  (A separate pass over the edge list was measured first: +12 us at
// 4096x1024 and +15..24 us at 10000x2000 -- a launch boundary plus a chain of float64 divisions and square roots that
// one or a few workgroups execute; spread over the tiles the same arithmetic disappears in the tile kernel's time.)
__device__ __forceinline__ bool motion_gate_open(const Dev &d, long s, int det) {
    const bool f = d.f32[s] != 0;
    double mu[4], S[16], L[16], y[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) mu[a] = ld_mean(d, s, a);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) S[i * 4 + j] = ld_cov(d, s, i * 8 + j);      // (cho_factor reads the lower triangle)
    const double r_w = kf::sq_scaled(kf::W_POS, mu[2], f), r_h = kf::sq_scaled(kf::W_POS, mu[3], f);      // D:236-263
    S[0] += r_w; S[5] += r_h; S[10] += r_w; S[15] += r_h;
    kf::chol4(S, L);
#pragma unroll
    for (int a = 0; a < 4; ++a) y[a] = (double)d.det_xywh[det * 4 + a] - mu[a];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                         // forward substitution L y = z - H m (solve_triangular, D:374-376)
        double t = y[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= L[i * 4 + k] * y[k];
        y[i] = t / L[i * 4 + i];
    }
    double g = y[0] * y[0]; g += y[1] * y[1]; g += y[2] * y[2]; g += y[3] * y[3];
    return !(g > d.gate_thr);                             // (a NaN distance is not "greater": the pair stays, like cost[d2 > thr] = inf)
}

// STAGE 1: cost = min(iou_d, emb) with the appearance gate (D:1539-1554)
// STAGE 2: cost = iou_d (D:1570)      STAGE 3: D:1590-1602
// The same tile also feeds the assignment solver directly: every entry <= the stage's match
// threshold is appended to the solver's edge list (entries above it can never be matched), so the
// dense matrix is only written when cost recording is on.
// MODE 0: fused dense form (similarities read from the dense float32 matrix), emits edges.
// MODE 1: pre-filter form -- stage 1 emits the IoU edges only (appearance edges come from
//         k_filter); stage 3 refines the few overlapping pairs with an exact wavefront dot product.
// MODE 2: record only (dense cost matrix for inspection, no edges).
// ROWS = rows of a tile: 16, or 64 for big stage-1 problems (a tile's fixed cost -- its chains of dependent
// loads -- dominates, the per-row screening is cheap).
// ACC = false compiles the split-K form of the pre-filter out (the tall tiles: the column records it keeps in registers
// across the row loop pushed them over their 128-register cap -- 12-20 B of scratch, ~2 us per launch)
// XO = true (tie_mode 1 with face features): entries that the appearance term lowered but that stay ABOVE the match threshold
// (a pair whose face passes the gate keeps its body distance, D:1543-1554) are handed to the exact-tie step as OVERRIDES, at
// the top end of the solver's edge arrays (filled downwards, counter lap hdr[7]); the fast solver never sees them.
template <int STAGE, int MODE, int ROWS = CT_ROWS, bool GATE = false, bool ACC = true, bool XO = false>
__device__ __forceinline__ void edges_tile(const Dev &d, const bts_lap_work &w, int bx, int by, int accum_rt) {
    const int accum = ACC ? accum_rt : 0;
    __shared__ double sa[ROWS][4];
    __shared__ double sarea[ROWS];
    __shared__ int sflag[ROWS];
    __shared__ int sslot[ROWS];
    __shared__ float4 srec[ROWS], srec_f[ROWS];       // full-row records of the tile's tracks (split-K form of the pre-filter)
    __shared__ iou::ScreenBox sboxf[ROWS];       // row boxes rounded OUTWARD to 16-bit pixels (conservative overlap pre-test)
    const int32_t *h = d.hdr;
    const int i0 = by * ROWS, q = bx * CT_COLS + threadIdx.x;
    // every chain of dependent loads of the tile starts here, before the barrier, so that they overlap:
    // (1) this thread's detection column, (2) the row boxes, (3) the wavefront's candidate-bitmap words.
    // Chain (1) does not wait for the extents: the column's list entry is fetched at its capacity-clamped position together
    // with them (an entry past the end is a stale but valid index, see bts_tracker_create) and discarded if the column is dead
    const int qc = min(q, d.Mx - 1);
    const int det_raw = STAGE == 1 ? d.hi_idx[qc] : STAGE == 2 ? d.lo_idx[qc] : d.left_det[qc];
    const int n = STAGE == 1 ? h[H_NPOOL] : STAGE == 2 ? h[H_NR2] : h[H_NUNC];
    const int m = STAGE == 1 ? h[H_NHI] : STAGE == 2 ? h[H_NLO] : h[H_NLEFT];
    const double *abox = STAGE == 1 ? d.pool_tlbr : STAGE == 2 ? d.r2_tlbr : d.unc_tlbr;
    const double bb_raw[4] = {d.det_tlbr[det_raw * 4 + 0], d.det_tlbr[det_raw * 4 + 1], d.det_tlbr[det_raw * 4 + 2], d.det_tlbr[det_raw * 4 + 3]};
    const bool live = q < m;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int det = live ? det_raw : 0;
    __shared__ unsigned long long scw[CT_COLS / 64][ROWS];      // [wave][row]: candidate lanes of the row (stage 1)
    if (MODE == 1 && STAGE == 1 && !accum) {
        // the bitmap row segment of this wavefront's 64 columns is one aligned 8-byte word: lane u fetches
        // row i0 + u's (64 rows per load instruction); consumed and cleared
        for (int u = lane; u < ROWS; u += 64) {
            unsigned long long v = 0;
            if (i0 + u < n && bx * CT_COLS + wid * 64 < d.Mx) {
                const long brow = d.pool_cbrow ? (long)d.pool_cbrow[i0 + u] : (long)(i0 + u);      // (BTS_FILTER_AHEAD: by slot / previous detection)
                unsigned long long *pw = reinterpret_cast<unsigned long long *>(
                    &d.cbits[brow * d.cb_ld + ((bx * CT_COLS + wid * 64) >> 5)]);
                v = *pw;
                if (v) *pw = 0ull;
            }
            scw[wid][u] = v;
        }
    }
    // (a dead column used detection 0's box: its values never reach a result -- `live` guards every use -- so the box fetched
    //  through the stale entry serves as well)
    const double bb[4] = {bb_raw[0], bb_raw[1], bb_raw[2], bb_raw[3]};
    float4 crec = make_float4(0.f, 0.f, 0.f, 0.f), crec_f = crec;
    if (STAGE == 1 && MODE == 1 && accum && live) {
        crec = feat::i8_rec(d.det_m + (long)det * feat::I8_META, feat::I8_NCHK);
        if (d.Df) crec_f = feat::i8_rec(d.dface_m + (long)det * feat::I8_META, feat::I8_NCHK);
    }
    if (threadIdx.x < ROWS && i0 + threadIdx.x < n) {
        const int i = i0 + threadIdx.x;
        double box[4] = {abox[i * 4 + 0], abox[i * 4 + 1], abox[i * 4 + 2], abox[i * 4 + 3]};
        const bool f = STAGE == 3 ? d.unc_f32[i] != 0 : false;
        for (int c = 0; c < 4; ++c) sa[threadIdx.x][c] = box[c];
        sarea[threadIdx.x] = iou::area(box, f);
        sflag[threadIdx.x] = f;
        sboxf[threadIdx.x] = iou::screen_box(box);
        if (STAGE == 1 && MODE == 1) {
            const int s = d.pool_slot[i];
            sslot[threadIdx.x] = s;
            if (accum) {
                srec[threadIdx.x] = feat::i8_rec(d.body_m + (long)s * feat::I8_META, feat::I8_NCHK);
                if (d.Df) srec_f[threadIdx.x] = feat::i8_rec(d.face_m + (long)s * feat::I8_META, feat::I8_NCHK);
            }
        }
    }
    __syncthreads();
    const double area_b = iou::area(bb, true);
    const iou::ScreenBox bf = live ? iou::screen_box(bb) : iou::screen_none();      // (a dead column overlaps nothing)
    double *cost = STAGE == 1 ? d.cost1 : STAGE == 2 ? d.cost2 : d.cost3;
    const double limit = STAGE == 1 ? d.match1 : STAGE == 2 ? d.match2 : d.match3;
    // rows cannot be screened by box overlap when every entry of the matrix is wanted, or in the dense stage-1 form:
    // there a pair of DISJOINT boxes still yields an edge when its appearance distance passes the gate (the pre-filter
    // form carries those pairs as candidates instead)
    const bool all_rows = MODE == 2 || (MODE == 0 && (d.record || STAGE == 1)) || !(limit < 1.0);
    const int rows = min(ROWS, n - i0);
    const unsigned long long livemask = __ballot(live);
    // Edges are staged per wavefront in LDS (ballot + lane-prefix, no atomics) and appended to the
    // solver's list with ONE counter bump per flush and coalesced stores: a returning atomic costs a
    // memory round trip, far too slow to pay per edge.
    constexpr int EB = 128;
    __shared__ double wb_c[CT_COLS / 64][EB];
    __shared__ int32_t wb_r[CT_COLS / 64][EB], wb_q[CT_COLS / 64][EB];
    int wcount = 0;
    __shared__ int s_wcnt[CT_COLS / 64], s_base;
    auto flush_at = [&](int base) {
        for (int k = lane; k < wcount; k += 64) {
            const int pos = base + k;
            if (pos < w.e_cap) { w.e_row[pos] = wb_r[wid][k]; w.e_col[pos] = wb_q[wid][k]; w.e_cost[pos] = wb_c[wid][k]; }
            else w.hdr[2] = 1;
        }
        wcount = 0;
    };
    auto flush = [&]() {                                  // mid-tile overflow of the staging buffer (rare)
        if (wcount == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(&w.hdr[0], wcount);
        flush_at(__shfl(base, 0));
    };
    auto emit = [&](bool pred, int row, double c) {          // convergent: every lane of the wavefront calls it
        const unsigned long long mk = __ballot(pred);
        if (!mk) return;
        if (pred) {
            const int p = wcount + __popcll(mk & ((1ull << lane) - 1ull));
            wb_c[wid][p] = c; wb_r[wid][p] = row; wb_q[wid][p] = q;
        }
        wcount += __popcll(mk);
        if (wcount > EB - 64) flush();
    };
    // Rows are handled in chunks of 16.  Stage 1, pre-filter form: which pairs of this column could pass
    // the appearance gate was decided by the fp8 contraction (k_filter_dma) -- either its split-K partial
    // sums are complete in `approx` now, or it left a candidate bitmap.  Both are consumed and cleared
    // here; the 16 rows of a chunk are fetched up front so that they cost one memory latency, not sixteen.
#pragma unroll 1
    for (int rc = 0; rc < rows; rc += 16) {
    const int crows = min(16, rows - rc);
    // split-K form: the candidate lanes of each row of the chunk, as wave-uniform 64-bit lane masks in LDS
    if (MODE == 1 && STAGE == 1 && accum) {
        uint32_t cmask = 0;
        if (live) {
            auto scan = [&](int32_t *acc, const float4 *srow, const float4 &col) {
                int32_t *p = acc + (long)(i0 + rc) * d.Mx + q;
                int32_t av[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) av[u] = u < crows ? p[(long)u * d.Mx] : 0;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (u >= crows) continue;
                    p[(long)u * d.Mx] = 0;
                    cmask |= (uint32_t)!feat::i8_excluded(av[u], srow[rc + u], col) << u;
                }
            };
            scan(d.approx, srec, crec);
            if (d.Df) scan(d.approx_f, srec_f, crec_f);
        }
        unsigned long long mine = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned long long bu = __ballot((cmask >> u) & 1u);
            if (lane == u) mine = bu;
        }
        if (lane < 16 && lane < ROWS) scw[wid][(rc + lane) & (ROWS - 1)] = mine;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // The N x M pair tests are the bulk of this kernel's instructions and nearly all pairs are disjoint
    // boxes without an appearance candidate (cost exactly 1 > limit: no edge).  Which rows hold anything
    // else for this wavefront is found first, with the chunk's LDS reads in flight together: four float32
    // comparisons per row on outward-rounded boxes (disjoint there => disjoint exactly); only those
    // rows are then evaluated exactly in float64.
    uint32_t work = 0;                                    // wave-uniform: bit u = some lane needs row rc + u
    if (MODE == 1 && STAGE == 1) {                        // rows with an appearance candidate in this wavefront's columns: one pass, lane u = row u
        const unsigned long long cw = lane < 16 ? scw[wid][(rc + lane) & (ROWS - 1)] & livemask : 0ull;
        work = (uint32_t)__ballot(cw != 0ull) & 0xffffu;
    }
#pragma unroll 1
    for (int u0 = 0; u0 < 16; u0 += 8)
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
        const int u = u0 + uu;
        // three packed 16-bit operations + one test per pair; the verdict stays on the scalar side
        work |= (__ballot(iou::screen_overlap(sboxf[(rc + u) & (ROWS - 1)], bf)) != 0ull ? 1u : 0u) << u;
    }
    if (all_rows) work = 0xffffu;
    work &= (1u << crows) - 1u;
    while (work) {
        const int ub = __ffs((int)work) - 1, r = rc + ub;
        work &= work - 1;
        const double aa[4] = {sa[r][0], sa[r][1], sa[r][2], sa[r][3]};
        const double iou_d = iou::cost(aa, sflag[r] != 0, sarea[r], bb, true, area_b);
        const long o = (long)(i0 + r) * d.Mx + q;
        double c = iou_d;
        if (MODE == 1 && STAGE == 1) {
            // candidates get the exact float32 dot product, computed by the whole wavefront; the pair's
            // single edge carries min(iou_d, emb) = np.minimum of D:1554
            unsigned long long todo = scw[wid][r] & livemask;
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const long sdet = __shfl(det, src), strk = sslot[r];
                const float emb = 1.0f - fmaxf(0.0f, feat::wave_dot<(ROWS <= CT_ROWS ? 8 : 4)>(d.body_curr + strk * d.D, IN(d, feats) + sdet * d.D, d.D, lane));
                float face = 1.0f;
                if (d.Df)
                    face = 1.0f - face_fixup(fmaxf(0.0f, feat::wave_dot<(ROWS <= CT_ROWS ? 8 : 4)>(d.face_curr + strk * d.Df, IN(d, face_feats) + sdet * d.Df, d.Df, lane)));
                if (lane == src && !(fminf(emb, face) > d.app)) c = fmin(c, (double)emb);
            }
        } else if (MODE == 1 && STAGE == 3) {
            // appearance can only lower the cost where the proximity gate is open (D:1600-1601):
            // those few pairs get an exact float32 dot product computed by the whole wavefront
            unsigned long long todo = __ballot(live && iou_d <= d.prox);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const long sdet = __shfl(det, src);
                const long strk = d.unc_slot[i0 + r];
                const float emb = 1.0f - fmaxf(0.0f, feat::wave_dot<(ROWS <= CT_ROWS ? 8 : 4)>(d.body_curr + strk * d.D, IN(d, feats) + sdet * d.D, d.D, lane));
                if (lane == src && !(emb > d.app)) c = fmin(c, (double)emb);
            }
        } else if (MODE != 1 && live) {
            if (STAGE == 1) {
                float emb = 1.0f - d.sim[o];
                const float face = 1.0f - (d.Df ? face_fixup(d.fsim[o]) : 0.0f);
                if (fminf(emb, face) > d.app) emb = 1.0f;
                c = fmin(iou_d, (double)emb);
            } else if (STAGE == 3) {
                float emb = 1.0f - d.sim[o];
                if (emb > d.app) emb = 1.0f;
                if (iou_d > d.prox) emb = 1.0f;
                c = fmin(iou_d, (double)emb);
            }
            if (d.c16) c = round_to_f16(c);
            if (MODE == 2 || d.record) cost[o] = c;
        }
        if (MODE == 1 && d.c16) c = round_to_f16(c);      // cost_dtype BTS_F16: the value the threshold test and the solver see
        if (XO && STAGE == 1 && live && c > limit && c < iou_d) {
            // (their own region behind the edge list, filled from the top down: an overflow is an error of the frame, raised
            //  here -- the fast solver never sees a corrupted triple)
            const long k = (long)atomicAdd(&w.hdr[7], 1), idx = w.e_cap + w.o_cap - 1 - k;
            if (k < w.o_cap) { w.e_row[idx] = i0 + r; w.e_col[idx] = q; w.e_cost[idx] = c; }
            else w.hdr[2] = 1;
        }
        bool edge = live && c <= limit;
        if (GATE && STAGE == 1 && edge) edge = motion_gate_open(d, d.pool_slot[i0 + r], det);
        if (MODE != 2) emit(edge, i0 + r, c);
    }
    }   // chunk
    if (MODE != 2) {
        // end of tile: ONE counter bump for the whole workgroup (every wavefront of the chip hammering the
        // same counter costs more than the tile itself)
        if (lane == 0) s_wcnt[wid] = wcount;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int k = 0; k < CT_COLS / 64; ++k) tot += s_wcnt[k];
            s_base = tot ? atomicAdd(&w.hdr[0], tot) : 0;
        }
        __syncthreads();
        int base = s_base;
        for (int k = 0; k < wid; ++k) base += s_wcnt[k];
        flush_at(base);
    }
    __syncthreads();          // the tile's LDS is reused by the caller's next tile
}

// Persistent tiles: the live extents are only known on the device, so a fixed grid walks the
// (column tile, row tile) space of the actual problem instead of launching capacity-sized grids.
template <int STAGE, int MODE, int ROWS = CT_ROWS, bool GATE = false, bool ACC = true, bool XO = false>
__device__ __forceinline__ void edges_all(const Dev &d, const bts_lap_work &w, int bid, int nblk, int accum = 0) {
    const int32_t *h = d.hdr;
    const int n = STAGE == 1 ? h[H_NPOOL] : STAGE == 2 ? h[H_NR2] : h[H_NUNC];
    const int m = STAGE == 1 ? h[H_NHI] : STAGE == 2 ? h[H_NLO] : h[H_NLEFT];
    const int nct = (m + CT_COLS - 1) / CT_COLS, ntile = nct * ((n + ROWS - 1) / ROWS);
    for (int t = bid; t < ntile; t += nblk) edges_tile<STAGE, MODE, ROWS, GATE, ACC, XO>(d, w, t % nct, t / nct, accum);
}

template <int STAGE, int MODE, int ROWS = CT_ROWS, bool GATE = false>
__global__ __launch_bounds__(CT_COLS) void tk_edges(Dev d, bts_lap_work w, int accum) {
    edges_all<STAGE, MODE, ROWS, GATE>(d, w, blockIdx.x, gridDim.x, accum);
}
// stage-1 tiles that also emit the above-threshold overrides (tie_mode 1 with face features; every tile height, uncapped)
template <int MODE, int ROWS>
__global__ __launch_bounds__(CT_COLS) void tk_edges_xo(Dev d, bts_lap_work w, int accum) {
    edges_all<1, MODE, ROWS, false, true, true>(d, w, blockIdx.x, gridDim.x, accum);
}
// the tall stage-1 tiles of big problems are throughput-bound on their screening loop (59 % of the wave cycles waiting
// at 3 wavefronts per SIMD): capped at 128 registers they run 4 per SIMD (no spills at 32 / 64 rows; the short tiles
// would spill, and any scratch costs a launch ~2 us)
template <int ROWS>
__global__ __launch_bounds__(CT_COLS) __attribute__((amdgpu_waves_per_eu(4, 8))) void tk_edges_tall(Dev d, bts_lap_work w, int accum) {
    edges_all<1, 1, ROWS, false, false>(d, w, blockIdx.x, gridDim.x, 0);      // (launched for the bitmap form only)
}
// the motion-gate form of the tall tiles: no register cap up to 128 rows (under it the gate's float64 factorisation
// spills 8-12 bytes per lane, and any scratch costs a launch ~2 us); the 256-row tiles of the biggest problems keep the
// cap (tk_edges_tall_gate_c): there the fourth wavefront per SIMD is worth more than the spill costs (16384x4096 with
// the gate: 52 us capped, 65 us uncapped, 43 us without the gate; 10000x2000, 128 rows: 34-35 us either way)
template <int ROWS>
__global__ __launch_bounds__(CT_COLS) void tk_edges_tall_gate(Dev d, bts_lap_work w, int accum) {
    edges_all<1, 1, ROWS, true, false>(d, w, blockIdx.x, gridDim.x, 0);
}
template <int ROWS>
__global__ __launch_bounds__(CT_COLS) __attribute__((amdgpu_waves_per_eu(4, 8))) void tk_edges_tall_gate_c(Dev d, bts_lap_work w, int accum) {
    edges_all<1, 1, ROWS, true, false>(d, w, blockIdx.x, gridDim.x, 0);
}

// ---- stage-1 cost edges by neighbour search (the default when the pre-filter left a candidate bitmap) -------------------
// One wavefront per 64 pool rows.  (1) Geometry: lane = row; the row's predicted box is looked up in the detections' grid
// (build_grid) and the few detections found get the exact float64 IoU cost (pairs that are appearance candidates are
// left to (2)).  (2) Appearance: the rows' words of the candidate bitmap are swept 64 at a time (coalesced); every
// candidate pair gets the exact float32 dot product from the whole wavefront and ONE edge with min(IoU cost, 1 - sim)
// (np.minimum of D:1554) when that is within the match threshold.  Edges are staged in LDS and appended
// with one counter bump per wavefront.  Same edges as the tile form (tk_edges<1,1,R>) whenever match_thresh < 1 (the
// launcher takes the tile form otherwise: with a threshold >= 1 even disjoint pairs are feasible at cost 1.0), in another
// order -- the solver does not depend on the order.
template <bool GATE>
__global__ __launch_bounds__(64) void tk_edges_grid(Dev d, bts_lap_work w) {
    constexpr int EB = 512;
    __shared__ double wb_c[EB];
    __shared__ int32_t wb_r[EB], wb_q[EB];
    const int32_t *h = d.hdr;
    const int n = h[H_NPOOL], m = h[H_NHI];
    const int lane = threadIdx.x;
    const double limit = d.match1;
    const int32_t *g = d.grid_hdr;
    const int ox = g[G_OX], oy = g[G_OY], cs = g[G_CS], gx = g[G_GX], gy = g[G_GY], maxw = g[G_MAXW], maxh = g[G_MAXH];
    int wcount = 0;
    auto flush = [&]() {
        if (wcount == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(&w.hdr[0], wcount);
        base = __shfl(base, 0);
        for (int k = lane; k < wcount; k += 64) {
            const int pos = base + k;
            if (pos < w.e_cap) { w.e_row[pos] = wb_r[k]; w.e_col[pos] = wb_q[k]; w.e_cost[pos] = wb_c[k]; }
            else w.hdr[2] = 1;
        }
        wcount = 0;
    };
    auto emit = [&](bool pred, int row, int q, double c) {          // convergent: every lane of the wavefront calls it
        const unsigned long long mk = __ballot(pred);
        if (!mk) return;
        if (wcount + __popcll(mk) > EB) flush();
        if (pred) {
            const int p = wcount + __popcll(mk & ((1ull << lane) - 1ull));
            wb_c[p] = c; wb_r[p] = row; wb_q[p] = q;
        }
        wcount += __popcll(mk);
    };
    const int words = (m + 31) >> 5;                    // live words of a bitmap row
    // RPW rows per wavefront (many wavefronts: every step below is a chain of dependent loads, and the exact dot products
    // of the candidates want to be spread thin); LPR lanes share a row in the neighbour search (cells dealt round-robin)
    constexpr int RPW = 16, LPR = 64 / RPW;
    for (int base = (int)blockIdx.x * RPW; base < n; base += (int)gridDim.x * RPW) {
        const int rows = min(RPW, n - base);
        const int i = base + lane / LPR, sub = lane % LPR;
        const bool live = i < n;
        // this lane's row
        double aa[4] = {0.0, 0.0, 0.0, 0.0};
        if (live) { aa[0] = d.pool_tlbr[i * 4 + 0]; aa[1] = d.pool_tlbr[i * 4 + 1]; aa[2] = d.pool_tlbr[i * 4 + 2]; aa[3] = d.pool_tlbr[i * 4 + 3]; }
        const double area_a = iou::area(aa, false);
        // ---- (1) geometric neighbours: walk the cells the row's widened box touches (pairs that are appearance
        //      candidates are left to (2), which gives them their one edge)
        int cx0 = 0, ncx = 0, cy0 = 0, ncell_r = 0;
        if (live && m > 0) {
            // detection centres (doubled) that can belong to an overlapping box: (2 x1 - maxw, 2 x2 + maxw) x (2 y1 - maxh, 2 y2 + maxh)
            const double lx = 2.0 * aa[0] - (double)maxw - (double)ox, hx = 2.0 * aa[2] + (double)maxw - (double)ox;
            const double ly = 2.0 * aa[1] - (double)maxh - (double)oy, hy = 2.0 * aa[3] + (double)maxh - (double)oy;
            const double span_x = (double)gx * (double)cs, span_y = (double)gy * (double)cs;
            if (hx >= 0.0 && lx < span_x && hy >= 0.0 && ly < span_y && hx >= lx && hy >= ly) {
                cx0 = lx <= 0.0 ? 0 : (int)(lx / (double)cs);
                cy0 = ly <= 0.0 ? 0 : (int)(ly / (double)cs);
                const int cx1 = min(hx >= span_x ? gx - 1 : (int)(hx / (double)cs), gx - 1);
                const int cy1 = min(hy >= span_y ? gy - 1 : (int)(hy / (double)cs), gy - 1);
                ncx = cx1 - cx0 + 1;
                ncell_r = ncx > 0 && cy1 >= cy0 ? ncx * (cy1 - cy0 + 1) : 0;
            }
        }
        int ck = sub, it = 0, it_end = 0;               // this lane's next cell (linear index in the row's range), item cursor
        bool more = ck < ncell_r;
        if (more) { const int c = (cy0 + ck / ncx) * gx + cx0 + ck % ncx; it = d.cell_start[c]; it_end = d.cell_start[c + 1]; }
        while (__any(more)) {
            bool have = false;
            int q = 0;
            if (more) {
                while (it >= it_end) {                   // next non-empty cell of this lane's share
                    ck += LPR;
                    if (ck >= ncell_r) { more = false; break; }
                    const int c = (cy0 + ck / ncx) * gx + cx0 + ck % ncx;
                    it = d.cell_start[c]; it_end = d.cell_start[c + 1];
                }
                if (more) { q = d.cell_items[it++]; have = true; }
            }
            if (have && ((d.cbits[(long)i * d.cb_ld + (q >> 5)] >> (q & 31)) & 1u)) have = false;
            double c = 1.0;
            bool edge = false;
            if (have) {
                const int det = d.hi_idx[q];
                const double bb[4] = {d.det_tlbr[det * 4 + 0], d.det_tlbr[det * 4 + 1], d.det_tlbr[det * 4 + 2], d.det_tlbr[det * 4 + 3]};
                c = iou::cost(aa, false, area_a, bb, true, iou::area(bb, true));
                if (d.c16) c = round_to_f16(c);
                edge = c <= limit;
                if (GATE && edge) edge = motion_gate_open(d, d.pool_slot[i], det);
            }
            emit(edge, i, q, c);
        }
        // ---- (2) appearance candidates: lane = bitmap word, one row per step (the words are consumed: this wavefront's rows only)
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int wl = w0 + lane;
            for (int r0 = 0; r0 < rows; r0 += 8) {
                uint32_t wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    wv[u] = (r0 + u < rows && wl < words) ? d.cbits[(long)(base + r0 + u) * d.cb_ld + wl] : 0u;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    unsigned long long any = __ballot(wv[u] != 0u);
                    if (!any) continue;
                    const int row = base + r0 + u;
                    if (wv[u]) d.cbits[(long)row * d.cb_ld + wl] = 0u;              // consumed
                    const long strk = d.pool_slot[row];
                    const double ra[4] = {d.pool_tlbr[row * 4 + 0], d.pool_tlbr[row * 4 + 1], d.pool_tlbr[row * 4 + 2], d.pool_tlbr[row * 4 + 3]};
                    const double rarea = iou::area(ra, false);
                    while (any) {
                        const int src = __ffsll((long long)any) - 1;
                        any &= any - 1;
                        uint32_t bits = __shfl(wv[u], src);
                        while (bits) {
                            const int q = (w0 + src) * 32 + (__ffs((int)bits) - 1);
                            bits &= bits - 1;
                            if (q >= m) continue;
                            const long det = d.hi_idx[q];
                            const float emb = 1.0f - fmaxf(0.0f, feat::wave_dot<8>(d.body_curr + strk * d.D, IN(d, feats) + det * d.D, d.D, lane));
                            float face = 1.0f;
                            if (d.Df)
                                face = 1.0f - face_fixup(fmaxf(0.0f, feat::wave_dot<8>(d.face_curr + strk * d.Df, IN(d, face_feats) + det * d.Df, d.Df, lane)));
                            const double bb[4] = {d.det_tlbr[det * 4 + 0], d.det_tlbr[det * 4 + 1], d.det_tlbr[det * 4 + 2], d.det_tlbr[det * 4 + 3]};
                            double c = iou::cost(ra, false, rarea, bb, true, iou::area(bb, true));
                            if (!(fminf(emb, face) > d.app)) c = fmin(c, (double)emb);
                            if (d.c16) c = round_to_f16(c);
                            bool edge = lane == 0 && c <= limit;
                            if (GATE && edge) edge = motion_gate_open(d, strk, (int)det);
                            emit(edge, row, q, c);
                        }
                    }
                }
            }
        }
    }
    flush();
}

// (tile shape of the pre-filter GEMM: also what the host side sizes its grids with)
constexpr int FBM = 128, FBN = 128;      // tile of the pre-filter GEMM
constexpr int DBK = feat::I8_STAGE;      // its K step: 128 int8 elements = one full 128-byte line per row and stage
constexpr int FILTER_PW = 512;       // 2 resident workgroups per CU (64 KiB of operand stages each)
constexpr int CAND_W = 32;           // candidate pairs a (tile, wavefront) slot of the candidate list holds (Dev::cand)
enum { FS_STAGES = 0, FS_TILES, FS_EARLY, FS_CAND, FS_CHECKS };     // ([5..7]: clock cycles / 64 in candidate tests, stage waits, in total, summed over the workgroups)
// ---- appearance pre-filter: int8 MFMA GEMM with prefix exits + exact float32 refine -----------------------------
// Stage-1 appearance can change a cost only where  min(1 - body_sim, 1 - face_sim) <= appearance_thresh
// (D:1543-1547), i.e. for similarities >= 0.75 -- about one pair per detection out of N*M.  So the
// N x M x D contraction runs on the matrix cores over int8 shadows of the feature rows (feat::to_i8_row), and only
// pairs that a RIGOROUS upper bound on the exact dot product cannot exclude (feat::i8_excluded) are recomputed exactly
// in float32 by the stage-1 tile kernel (same value the dense float32 path produces up to summation order); the
// pair's single edge then carries min(IoU cost, 1 - sim) = np.minimum of D:1554.
// The bound works on any PREFIX of the rows (squared distances only grow with more elements), so a tile walks K in
// stages and stops at the first checkpoint where (nearly) nothing in it can still reach the gate: on embeddings whose
// non-matching pairs are far from the gate that is after ~5/16 of K; the survivors of an early stop are candidates
// like the others (a superset -- the exact refinement decides).
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, slot = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// 128 x 128 tiles of v_mfma_i32_32x32x32_i8 over the int8 shadows.  What bounds this kernel is neither HBM nor the
// matrix cores but the CU's fill path (a 128 x 128 tile needs 64 B/clk/CU of operand traffic at full MFMA rate with
// 2-byte operands) -- hence 1-byte operands -- and how the operands reach LDS:
//   * straight from memory (global_load_lds, 16 B per lane: no staging registers, no ds_write pass), double-buffered
//     in stages of DBK = 128 elements: a row of a stage is one FULL 128-byte line (a CU's load path retires about one
//     line segment per 4 clocks -- with 64-byte half rows the same bytes cost twice the load-path time), and the L2
//     answers in ~240 clocks, so one stage in flight is enough;
//   * the LDS image of such a load is lane-linear, so a tile is stored as unpadded 128-byte rows and the
//     bank spread comes from the SOURCE side: lane (row, c) fetches 16-byte chunk c ^ swz(row) of its row
//     and the fragment reads undo the permutation; swz(row) = (row >> 1) & 7 makes every ds_read_b128
//     lane group hit 16 distinct 16-byte slots (0 bank conflicts measured);
//   * one workgroup barrier per stage behind an s_waitcnt vmcnt issued in inline assembly together with it;
//   * a lane's 16-byte fragment read is the 16 K values it feeds to ONE K = 32 MFMA (which K indices a lane
//     holds is free as long as both operands agree).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DNS = 2;                 // (DBK = 128 elements = 128 bytes per row and stage, defined above)
#if !defined(BTS_HOST_EMU)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef unsigned lds_t;                // an LDS byte address, as the ds_* instructions take it
#define LDS_ADDR(p) ((lds_t)(__UINTPTR_TYPE__)(lds_void *)(p))

template <int N> __device__ __forceinline__ void wait_stage_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// Fragment reads of the ring are issued as inline assembly: for a ds_read it can see, the compiler
// drains EVERY outstanding LDS-DMA load first (it cannot tell the ring's stages apart), which would
// serialise the pipeline.  The explicit lgkmcnt wait below takes the fragments as operands so that no
// MFMA can be scheduled above it.
__device__ __forceinline__ v4i lds_read16(lds_t addr) {
    v4i v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int N> __device__ __forceinline__ void lds_wait(v4i &a, v4i &b, v4i &c, v4i &e) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e) : "n"(N));
}
__device__ __forceinline__ f32x4 lds_read16f(lds_t addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// (every operand of a wait must be a DIFFERENT variable: naming one twice makes the compiler copy it -- before the wait)
__device__ __forceinline__ void lds_wait_f(f32x4 &a, f32x4 &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lds_wait_i(v4i &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
__device__ __forceinline__ void lds_wait_f1(f32x4 &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
__device__ __forceinline__ void lds_write8(lds_t addr, float x, float y) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {x, y};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write16(lds_t addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write4(lds_t addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
#else
// tests/hip_emu: the same kernels on the CPU.  An "LDS address" is the host address of the __shared__ object; the LDS accesses
// above become plain loads / stores behind a hazard check against LDS-DMA transfers in flight, the stage wait lands the calling
// thread's transfers (the emulator's global_load_lds is asynchronous since round 6: tests/hip_emu/hip/hip_runtime.h) and the stage
// barrier is the workgroup barrier.  The MFMA and global_load_lds builtins are served by hip_runtime.h at instruction level.
typedef void lds_void;
typedef const void gbl_void;
typedef uintptr_t lds_t;
#define LDS_ADDR(p) ((lds_t)(p))
template <int N> __device__ __forceinline__ void wait_stage_barrier() { emu::lds_dma_wait(N); __syncthreads(); }
__device__ __forceinline__ v4i lds_read16(lds_t addr) { emu::lds_dma_check((const void *)addr, 16, "ds_read_b128 of"); v4i v; memcpy(&v, (const void *)addr, 16); return v; }
template <int N> __device__ __forceinline__ void lds_wait(v4i &, v4i &, v4i &, v4i &) {}
__device__ __forceinline__ f32x4 lds_read16f(lds_t addr) { emu::lds_dma_check((const void *)addr, 16, "ds_read_b128 of"); f32x4 v; memcpy(&v, (const void *)addr, 16); return v; }
__device__ __forceinline__ void lds_wait_f(f32x4 &, f32x4 &) {}
__device__ __forceinline__ void lds_wait_i(v4i &) {}
__device__ __forceinline__ void lds_wait_f1(f32x4 &) {}
__device__ __forceinline__ void lds_write8(lds_t addr, float x, float y) { emu::lds_dma_check((const void *)addr, 8, "ds_write_b64 to"); float v[2] = {x, y}; memcpy((void *)addr, v, 8); }
__device__ __forceinline__ void lds_write16(lds_t addr, f32x4 v) { emu::lds_dma_check((const void *)addr, 16, "ds_write_b128 to"); memcpy((void *)addr, &v, 16); }
__device__ __forceinline__ void lds_write4(lds_t addr, int v) { emu::lds_dma_check((const void *)addr, 4, "ds_write_b32 to"); memcpy((void *)addr, &v, 4); }
#endif

// minimum / maximum over a wavefront, returned to every lane: DPP steps inside the 16-lane rows, then the four rows through the scalar side
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float rdlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_min_f(float v) {
    v = fminf(v, dpp_f<0xB1>(v)); v = fminf(v, dpp_f<0x4E>(v)); v = fminf(v, dpp_f<0x141>(v)); v = fminf(v, dpp_f<0x140>(v));
    return fminf(fminf(rdlane_f(v, 0), rdlane_f(v, 16)), fminf(rdlane_f(v, 32), rdlane_f(v, 48)));
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(rdlane_f(v, 0), rdlane_f(v, 16)), fmaxf(rdlane_f(v, 32), rdlane_f(v, 48)));
}

// the per-thread view of a 128 x 128 tile's operands in LDS: wave (wr, wc) owns the 64 x 64 quadrant, as 2 x 2 MFMA blocks
struct FragMap {
    int aoff[2], boff[2], asw[2], bsw[2];
    __device__ __forceinline__ void init(int wr, int wc, int frow) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ra = wr * 64 + x * 32 + frow, rb = wc * 64 + x * 32 + frow;
            aoff[x] = ra * DBK; asw[x] = (ra >> 1) & 7;
            boff[x] = rb * DBK; bsw[x] = (rb >> 1) & 7;
        }
    }
};
// one K stage (128 int8 per row) of a wave's quadrant: four K = 32 steps, two per batch of LDS reads
// (VAR 1: timing experiment only -- the same operand bytes through the fp8 K = 64 instruction; results are meaningless)
typedef int v8i_ __attribute__((ext_vector_type(8)));
typedef float f32x16_ __attribute__((ext_vector_type(16)));
template <int VAR = 0>
__device__ __forceinline__ void stage_mfma(lds_t abase, lds_t bbase, const FragMap &fm, int fh, i32x16 (&acc)[2][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        v4i fa[2][2], fb[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int u = (h * 2 + t) * 2 + fh;          // K step (2 h + t): bytes [32 (2 h + t), +32), this lane half takes 16 of them
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                fa[t][x] = lds_read16(abase + fm.aoff[x] + ((u ^ fm.asw[x]) << 4));
                fb[t][x] = lds_read16(bbase + fm.boff[x] + ((u ^ fm.bsw[x]) << 4));
            }
        }
        lds_wait<0>(fa[0][0], fa[0][1], fb[0][0], fb[0][1]);
        lds_wait<0>(fa[1][0], fa[1][1], fb[1][0], fb[1][1]);
        if constexpr (VAR == 1) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const v8i_ av = {fa[0][mi].x, fa[0][mi].y, fa[0][mi].z, fa[0][mi].w, fa[1][mi].x, fa[1][mi].y, fa[1][mi].z, fa[1][mi].w};
                    const v8i_ bv = {fb[0][ni].x, fb[0][ni].y, fb[0][ni].z, fb[0][ni].w, fb[1][ni].x, fb[1][ni].y, fb[1][ni].z, fb[1][ni].w};
                    f32x16_ c;
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = __int_as_float(acc[mi][ni][r]);
                    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 127, 0, 127);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __float_as_int(c[r]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[t][mi], fb[t][ni], acc[mi][ni], 0, 0, 0);
        }
    }
}

// Split-K form for SMALL problems (too few tiles to fill the chip): ntiles * nsplit GEMM workgroups (tile-major inside
// a K split) add their exact integer partial products into `approx`; the stage-1 tile kernel applies the candidate test
// once they are complete (and clears them).  Body pass only: followed by the workgroups of the Kalman predict, which is
// independent of the contraction and shares the launch.
template <bool FACE>
__global__ __launch_bounds__(256) void k_filter_splitk(Dev d, int m_max, int ntiles, int nsplit) {
    __shared__ __attribute__((aligned(1024))) int8_t As2[DNS][FBM * DBK];
    __shared__ __attribute__((aligned(1024))) int8_t Bs2[DNS][FBN * DBK];
    if ((int)blockIdx.x >= ntiles * nsplit) {
        if (!FACE) predict_body(d, (int)blockIdx.x - ntiles * nsplit);
        return;
    }
    const int n = d.hdr[H_NPOOL], m = d.hdr[H_NHI];
    const int Dd = FACE ? d.Df : d.D;
    const int8_t *A = FACE ? d.face_h : d.body_h, *B = FACE ? d.dface_h : d.det_h;
    const int nbn = (m_max + FBN - 1) / FBN;
    const int ksplit = blockIdx.x / ntiles;
    const int vb = xcd_remap(blockIdx.x % ntiles, ntiles);
    const int bm = vb / nbn, bn = vb % nbn;
    if (bm * FBM >= n || bn * FBN >= m) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int kt_all = Dd / DBK, kt_per = (kt_all + nsplit - 1) / nsplit;
    const int kt0 = ksplit * kt_per, kt1 = min(kt_all, kt0 + kt_per);
    if (kt0 >= kt1) return;

    // piece s of a stage = rows [(4 s + wid) * 8, +8): one 1-KiB wavefront load of 8 FULL 128-byte lines,
    // lane -> (row, 16-byte chunk); swz(row) = (row >> 1) & 7 spreads every ds_read_b128 lane group over 16 slots
    const int prow = lane >> 3, pchunk = lane & 7;
    const int8_t *asrc[4], *bsrc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = (4 * s + wid) * 8 + prow;
        const int i = bm * FBM + r, j = bn * FBN + r;
        const int sw = (pchunk ^ ((r >> 1) & 7)) * 16;
        asrc[s] = A + (long)d.pool_slot[i < n ? i : bm * FBM] * (long)d.sh_rt[FACE] + sw;       // rows past the end: any valid row, ignored later
        bsrc[s] = B + (long)d.hi_idx[j < m ? j : bn * FBN] * (long)d.sh_rd[FACE] + sw;
    }
    auto stage = [&](int buf, int kt) {                  // 8 wavefront loads per stage and wave
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(asrc[s] + (size_t)kt * d.sh_ct[FACE]), (lds_void *)&As2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(bsrc[s] + (size_t)kt * d.sh_cd[FACE]), (lds_void *)&Bs2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
        }
    };
    i32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;
    const int frow = lane & 31, fh = lane >> 5;
    const lds_t lds_a = LDS_ADDR(&As2[0][0]), lds_b = LDS_ADDR(&Bs2[0][0]);
    FragMap fm;
    fm.init(wr, wc, frow);
    stage(0, kt0);
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        wait_stage_barrier<0>();             // stage kt has landed (the only one in flight); every wave is past its reads of the other buffer
        if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
        stage_mfma(lds_a + cur * (FBM * DBK), lds_b + cur * (FBN * DBK), fm, fh, acc);
        cur ^= 1;
    }
    int32_t *approx = FACE ? d.approx_f : d.approx;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int q = bn * FBN + wc * 64 + ni * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = bm * FBM + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (i < n && q < m) atomicAdd(&approx[(long)i * d.Mx + q], acc[mi][ni][r]);
            }
        }
}

// The one-pass form: FILTER_PW persistent workgroups walk the tiles (2 resident per CU), each tile runs its K loop up to the
// first checkpoint that lets it stop (or to the end) and sets the candidate bits itself.  The chain of dependent loads
// that opens a tile (index lists -> row pointers -> first stage) is taken off the tile's critical path: the NEXT tile's
// indices are fetched while the current K loop runs, and its first stage is issued before the current tile's last
// candidate test (the operand buffer it lands in is idle then).
// KF_FIRST (opt-in, BTS_PREDICT_FIRST=1 at create; NOT yet measured): the Kalman-predict workgroups take the FRONT of the grid.
// Every wave of this kernel is allocated its 244 VGPRs, so a predict workgroup cannot share a SIMD with two resident GEMM
// workgroups: behind them (the default) the predict work runs when the persistent GEMM workgroups retire, i.e. it is added to
// the kernel's tail; in front it overlaps the GEMM's start, and the GEMM workgroups that start late are the ones that own the
// fewest tiles.  The default instantiation is unchanged code.
// LIST (BTS_EDGES_SORTED=1 at create): a tile's survivors are also written to the candidate list, one slot per wavefront of the
// tile, for the candidate wavefronts of tk_edges_sorted.  The default instantiation is unchanged code.
template <bool FACE, int VAR = 0, bool KF_FIRST = false, bool LIST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_filter_i8(Dev d, int m_max, int ntiles, int gemm_blocks) {
    __shared__ __attribute__((aligned(1024))) int8_t As2[DNS][FBM * DBK];
    __shared__ __attribute__((aligned(1024))) int8_t Bs2[DNS][FBN * DBK];
    __shared__ float4 srec2[2][FBM + FBN];       // records of the tile's rows / columns for the checkpoint being tested (two
                                                 // images: a test never waits for the readers of the previous one)
    __shared__ __attribute__((aligned(16))) int s_cnt[2][4];     // survivors per wavefront
    __shared__ __attribute__((aligned(16))) float sred[2][8];    // {min uh, max alpha, max r, -} of the rows' records, per row wavefront
    const int kf_front = KF_FIRST ? (int)gridDim.x - gemm_blocks : 0;      // predict workgroups in front of the GEMM ones
    if (KF_FIRST ? (int)blockIdx.x < kf_front : (int)blockIdx.x >= gemm_blocks) {
#ifdef BTS_FILTER_TRACE
        const int p_beg = (int)(((long long)__builtin_amdgcn_s_memrealtime()) & 0x3fffffff);
#endif
        if (!FACE) predict_body(d, KF_FIRST ? (int)blockIdx.x : (int)blockIdx.x - gemm_blocks);
#ifdef BTS_FILTER_TRACE
        if (threadIdx.x == 0) { atomicMin(&d.fstat[12], p_beg); atomicMax(&d.fstat[13], (int)(((long long)__builtin_amdgcn_s_memrealtime()) & 0x3fffffff)); }
#endif
        return;
    }
    // The chain that opens a workgroup -- live extents -> index lists -> first operand stage, three dependent (cold) loads -- is
    // cut by one link: the index loads of the workgroup's FIRST tile position are issued together with the extents' (indices
    // clamped to the arrays' capacities, so that the addresses need no extent) and sorted out once the extents are known.
    const int n_raw = d.hdr[H_NPOOL], m_raw = d.hdr[H_NHI];
    const int Dd = FACE ? d.Df : d.D;
    const int8_t *A = FACE ? d.face_h : d.body_h, *B = FACE ? d.dface_h : d.det_h;
    const float *ma = FACE ? d.face_m : d.body_m, *mb = FACE ? d.dface_m : d.det_m;
    const unsigned rs_a = d.sh_rt[FACE], cs_a = d.sh_ct[FACE], rs_b = d.sh_rd[FACE], cs_b = d.sh_cd[FACE];      // shadow layout (feat::i8_elem)
    const int nbn = (m_max + FBN - 1) / FBN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int vt_first = (int)blockIdx.x - (KF_FIRST ? (int)gridDim.x - gemm_blocks : 0);
    int sia[4], sib[4], sirec = 0, sfa = 0, sfb = 0, sbm = 0, sbn = 0;      // the first position's indices, fetched ahead of the extents
    if (vt_first < ntiles) {
        const int vb = xcd_remap(vt_first, ntiles);
        sbm = vb / nbn; sbn = vb % nbn;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = (4 * s + wid) * 8 + (lane >> 3);
            sia[s] = d.pool_slot[min(sbm * FBM + r, d.T - 1)];
            sib[s] = d.hi_idx[min(sbn * FBN + r, d.Mx - 1)];
        }
        sfa = d.pool_slot[sbm * FBM]; sfb = d.hi_idx[min(sbn * FBN, d.Mx - 1)];
        sirec = tid < FBM ? d.pool_slot[min(sbm * FBM + tid, d.T - 1)] : d.hi_idx[min(sbn * FBN + (tid - FBM), d.Mx - 1)];
    }
    const int n = __builtin_amdgcn_readfirstlane(n_raw), m = __builtin_amdgcn_readfirstlane(m_raw);     // (scalar registers)
    const int kt_all = Dd / DBK;
    int chk[feat::I8_NCHK];
    feat::i8_checkpoints(kt_all, chk, d.chk_first);
    static_assert(feat::I8_NCHK == 4, "chk_at selects by hand (a runtime-indexed register array would go to scratch)");
    auto chk_at = [&](int j) { return j == 0 ? chk[0] : j == 1 ? chk[1] : j == 2 ? chk[2] : chk[3]; };
    auto skip_unused = [&](int j) { return (j < feat::I8_NCHK && chk_at(j) >= kt_all) ? (int)feat::I8_NCHK : j; };
    const int prow = lane >> 3, pchunk = lane & 7;
    const int frow = lane & 31, fh = lane >> 5;
    const lds_t lds_a = LDS_ADDR(&As2[0][0]), lds_b = LDS_ADDR(&Bs2[0][0]);
    FragMap fm;
    fm.init(wr, wc, frow);
    const lds_t lds_rec = LDS_ADDR(&srec2[0][0]), lds_cnt = LDS_ADDR(&s_cnt[0][0]), lds_red = LDS_ADDR(&sred[0][0]);
    // live tiles of this workgroup, in order: vt = blockIdx.x, + gemm_blocks, ... (tiles past the live extents are skipped)
    auto next_live = [&](int vt, int &bm, int &bn) {
        for (; vt < ntiles; vt += gemm_blocks) {
            const int vb = xcd_remap(vt, ntiles);
            bm = vb / nbn; bn = vb % nbn;
            if (bm * FBM < n && bn * FBN < m) return vt;
        }
        return ntiles;
    };
    // index fetch of a tile: the table slots / detection indices of this lane's four A rows and four B rows, and of the
    // row (threads 0-127) or column (128-255) whose record this thread carries into the candidate tests
    auto fetch_idx = [&](int bm, int bn, int (&ia)[4], int (&ib)[4], int &irec) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = (4 * s + wid) * 8 + prow;
            const int i = bm * FBM + r, j = bn * FBN + r;
            ia[s] = d.pool_slot[i < n ? i : bm * FBM];       // rows past the end: any valid row, ignored later
            ib[s] = d.hi_idx[j < m ? j : bn * FBN];
        }
        const int g = tid < FBM ? bm * FBM + tid : bn * FBN + (tid - FBM);
        irec = tid < FBM ? d.pool_slot[g < n ? g : bm * FBM] : d.hi_idx[g < m ? g : bn * FBN];
    };
    unsigned asrc[4], bsrc[4];                           // byte offsets into A / B (tables up to 4 GiB of shadows)
    auto set_src = [&](const int (&ia)[4], const int (&ib)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = (4 * s + wid) * 8 + prow;
            const unsigned sw = (unsigned)((pchunk ^ ((r >> 1) & 7)) * 16);
            asrc[s] = (unsigned)ia[s] * rs_a + sw;
            bsrc[s] = (unsigned)ib[s] * rs_b + sw;
        }
    };
    auto stage = [&](int buf, int kt) {                  // 8 wavefront loads per stage and wave
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(A + ((size_t)asrc[s] + (size_t)kt * cs_a)), (lds_void *)&As2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(B + ((size_t)bsrc[s] + (size_t)kt * cs_b)), (lds_void *)&Bs2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
        }
    };
    // the NEXT tile's first stage (its offsets are derived from the prefetched indices on the spot: the current tile's stay intact)
    auto stage_next = [&](int buf, const int (&ia)[4], const int (&ib)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = (4 * s + wid) * 8 + prow;
            const size_t sw = (size_t)((pchunk ^ ((r >> 1) & 7)) * 16);
            __builtin_amdgcn_global_load_lds((gbl_void *)(A + ((size_t)((unsigned)ia[s] * rs_a) + sw)), (lds_void *)&As2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(B + ((size_t)((unsigned)ib[s] * rs_b) + sw)), (lds_void *)&Bs2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
        }
    };
    // survivors a tile may hand to the exact refinement when it stops at a prefix: the true matches it holds anyway
    // (~ columns x rows / pool rows) plus a few -- an exact float32 dot product costs about one K stage of a tile
    const int stop_thr = 6 + (3 * FBM * FBN) / (2 * (n > 0 ? n : 1));
    const float i8_s2 = feat::i8_scale(Dd) * feat::i8_scale(Dd) * 0.99999f;        // integer dot products per unit of dequantised dot product
    int bm, bn;
    int vt = next_live((int)blockIdx.x - kf_front, bm, bn);
    if (vt >= ntiles) return;
    int ia[4], ib[4], irec;
    if (vt == vt_first) {
        // the position fetched ahead is live: rows / columns past the extents take the tile's first row / column like in fetch_idx
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = (4 * s + wid) * 8 + prow;
            ia[s] = sbm * FBM + r < n ? sia[s] : sfa;
            ib[s] = sbn * FBN + r < m ? sib[s] : sfb;
        }
        irec = tid < FBM ? (sbm * FBM + tid < n ? sirec : sfa) : (sbn * FBN + (tid - FBM) < m ? sirec : sfb);
    } else {
        fetch_idx(bm, bn, ia, ib, irec);
    }
    set_src(ia, ib);
    const float *mbase = tid < FBM ? ma : mb;
    unsigned my_meta_o = (unsigned)irec * feat::I8_META;                             // this thread's row / column record (element offset)
    // The records a candidate test needs ({1/s, r, alpha_c, uh_c} of the tile's 128 rows and 128 columns for checkpoint c) go
    // straight from memory into an LDS image, one 16-byte transfer per thread, at least one K stage ahead of the test:
    // the stage barrier in between waits for them like for the operands.  Nothing of it passes through registers, so
    // the compiler has no load result to wait for inside the K loop (its wait would drain the operand stage in flight).
    auto stage_rec = [&](int img, int c) {
        __builtin_amdgcn_global_load_lds((gbl_void *)(mbase + ((size_t)my_meta_o + (size_t)(4 * c))), (lds_void *)&srec2[img][wid * 64], 16, 0, 0);
    };
    int rimg = 0;                                // image the NEXT test's records are in (or on their way to)
    // (everything derived from the index loads is materialised BEFORE the first LDS-DMA goes out: behind one, the compiler turns
    //  the wait for ANY vector load into a full drain -- the record offset, sunk below stage(0, 0), cost the first stage a drain of
    //  its own in front of the record transfer; scripts/isa_chain.py)
#if !defined(BTS_HOST_EMU)
    asm volatile("" : "+v"(my_meta_o));
#endif
    stage(0, 0);
    int cur = 0, flip = 0;
    // Where this workgroup's tiles have been stopping: `last_exit` = the stage count of the predicted stop (-1: none yet).
    // The prediction moves up at once and down only after three earlier stops in a row (a stop before the predicted one
    // costs a cold first stage of the next tile, a stop after it a cold K stage: late is the cheaper mistake); tests at
    // checkpoints before the predicted one are skipped except on every fourth tile (the probe that lets it move down).
    // (Dev::f_guess, BTS_FILTER_GUESS=1 at create; NOT yet measured: the FIRST tile is predicted to stop at the first checkpoint
    //  as well -- on separable embeddings every tile does -- so the next tile's first stage is on its way before the first test
    //  instead of after it: one cold tile opening less per workgroup, which has two or three tiles at the headline size.  Only
    //  what is prefetched changes: the tests, their order and the candidate bits are the same.)
    int last_exit = d.f_guess ? chk[0] : -1, early_run = 0, tile_no = 0;
    int st_stages = 0, st_early = 0, st_cand = 0;
#ifdef BTS_FILTER_TRACE
    long long tm_check = 0, tm_wait = 0, tm_all = 0, tm_first = 0, tm_pro = 0;      // clock cycles of this workgroup: candidate tests, stage waits, total, first test, prologue
    const long long tm_begin = (long long)__builtin_amdgcn_s_memrealtime();
#define FT_BEGIN(v) const long long v = (long long)__builtin_amdgcn_s_memrealtime()
#define FT_ADD(acc, v) acc += (long long)__builtin_amdgcn_s_memrealtime() - v
#else
#define FT_BEGIN(v) do { } while (0)
#define FT_ADD(acc, v) do { } while (0)
#endif
    for (;;) {
        int ci = VAR == 2 ? (int)feat::I8_NCHK : skip_unused(0);     // next checkpoint (feat::I8_NCHK = the full row)
        // checkpoints before the predicted stop are not tested, except on every fourth tile (the probe that lets the
        // prediction move down)
        if ((tile_no & 3) != 3)
            while (ci < feat::I8_NCHK && chk_at(ci) < last_exit) ci = skip_unused(ci + 1);
        // The record transfer goes out BEFORE the next tile's index loads (round 6, read off the ISA: scripts/isa_chain.py): the
        // compiler consumes those indices as soon as they are issued (my_meta_o, the source offsets) and every vector-memory wait
        // behind an LDS-DMA is a full drain (vmcnt(0)), so with the records issued behind them a tile opened with TWO serial drains
        // -- first stage + indices, then the records -- in front of its first stage barrier; now one drain covers all three.
        stage_rec(rimg, ci);
        // the next tile's indices travel while this tile's K loop runs
        int nbm = 0, nbnn = 0;
        const int nvt = next_live(vt + gemm_blocks, nbm, nbnn);
        if (nvt < ntiles) fetch_idx(nbm, nbnn, ia, ib, irec);
        i32x16 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;
        int issued = 1;                              // stages whose loads have been issued
        int kt = 0;
        bool next_staged = false;                    // the next tile's first stage is already on its way
        for (;;) {
            const bool at_end = kt == kt_all;
            if (at_end || (VAR != 2 && ci < feat::I8_NCHK && kt == chk_at(ci))) {
                FT_BEGIN(tc0);
                // ---- candidate test on the prefix of kt stages (checkpoint ci; at the end: the full-row records)
                if (at_end && nvt < ntiles && !next_staged) {
                    // every wave is past its reads of buffer `cur` (the barrier of the last stage): the next tile's first
                    // stage goes there now and lands while the test runs
                    stage_next(cur, ia, ib);
                    next_staged = true;
                }
                // Every LDS access of the test is inline assembly: for an LDS access it can see, the compiler first drains ALL
                // outstanding LDS-DMA loads (s_waitcnt vmcnt(0)) -- the operand stage that is landing under this test.
                // The records are in image `rimg` since the last stage barrier at the latest.
                const lds_t srec_a = lds_rec + (unsigned)rimg * (unsigned)((FBM + FBN) * 16);
                const lds_t scnt_a = lds_cnt + (unsigned)flip * 16u, sred_a = lds_red + (unsigned)flip * 32u;
                // survivors of this lane: sv[mi] bit ni * 16 + r.  All shadows share one scale, so "excluded" (feat::i8_excluded:
                //     dot / s^2 + alpha_i r_j + alpha_j r_i  <  uh_i + uh_j)
                // becomes ONE integer compare per accumulator against a per-column threshold once the tile's rows are
                // summarised by {min uh, max alpha, max r}:
                //     dot < ((min uh_rows + uh_j) - (max alpha_rows r_j + alpha_j max r_rows)) s^2      =>  excluded
                // (prefix records have alpha = 0; looser than pair by pair only by the spread of the rows' records).
                unsigned sv[2] = {0u, 0u};
                {
                    f32x4 mine = lds_read16f(srec_a + (unsigned)tid * 16u);
                    lds_wait_f1(mine);
                    if (wid < 2) {                   // wavefronts 0-1 hold the rows
                        const float m_u = wave_min_f(mine.w), m_a = wave_max_f(mine.z), m_r = wave_max_f(mine.y);
                        if (lane == 0) lds_write16(sred_a + (unsigned)wid * 16u, f32x4{m_u, m_a, m_r, 0.0f});
                    }
                    bscan::lds_barrier();
                    f32x4 ra = lds_read16f(sred_a), rb = lds_read16f(sred_a + 16u);
                    f32x4 c0 = lds_read16f(srec_a + (unsigned)(FBM + wc * 64 + frow) * 16u), c1 = lds_read16f(srec_a + (unsigned)(FBM + wc * 64 + 32 + frow) * 16u);
                    lds_wait_f(ra, rb);
                    lds_wait_f(c0, c1);
                    const float u_min = fminf(ra.x, rb.x), a_max = fmaxf(ra.y, rb.y), r_max = fmaxf(ra.z, rb.z);
                    const float s2 = i8_s2;
                    const float t0 = ((u_min + c0.w) - (a_max * c0.y + c0.z * r_max)) * s2;
                    const float t1 = ((u_min + c1.w) - (a_max * c1.y + c1.z * r_max)) * s2;
                    // (NaN / non-positive bounds: nothing is excluded)
                    const int ti0 = t0 > 0.0f ? (int)fminf(t0 * 0.999999f, 2.0e9f) - 1 : (int)0x80000000;
                    const int ti1 = t1 > 0.0f ? (int)fminf(t1 * 0.999999f, 2.0e9f) - 1 : (int)0x80000000;
                    // Survivors are rare (a 128 x 128 tile holds one or two true matches): the lane's 32 accumulators of a column
                    // block are first reduced to their maximum (v_max3 trees) and only a wavefront in which some lane's maximum
                    // reaches its threshold extracts the bits -- the same bits, most of the time none.
                    int mx0 = acc[0][0][0], mx1 = acc[0][1][0];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { mx0 = max(mx0, acc[mi][0][r]); mx1 = max(mx1, acc[mi][1][r]); }
                    if (__ballot((mx0 >= ti0) | (mx1 >= ti1)) != 0ull) {
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                sv[mi] |= (acc[mi][0][r] >= ti0 ? 1u : 0u) << r;
                                sv[mi] |= (acc[mi][1][r] >= ti1 ? 1u : 0u) << (16 + r);
                            }
                    }
                }
                if (__ballot((sv[0] | sv[1]) != 0u) != 0ull) {      // (wave-uniform: most wavefronts hold no survivor and skip the extents)
                    // which of this lane's 64 accumulators lie inside the live extents (edge tiles)
                    unsigned vmask0, vmask1;
                    {
                        const int lim = n - bm * FBM - (wr * 64 + 4 * fh);       // local row (mi * 32 + 8 g + u) is live iff < lim
                        unsigned rv0 = 0u, rv1 = 0u;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lr = (r & 3) + 8 * (r >> 2);
                            rv0 |= (lr < lim ? 1u : 0u) << r;
                            rv1 |= (32 + lr < lim ? 1u : 0u) << r;
                        }
                        const int q0 = bn * FBN + wc * 64 + frow;
                        const unsigned m0 = q0 < m ? 0xffffu : 0u, m1 = q0 + 32 < m ? 0xffff0000u : 0u;
                        vmask0 = (rv0 | (rv0 << 16)) & (m0 | m1);
                        vmask1 = (rv1 | (rv1 << 16)) & (m0 | m1);
                    }
                    sv[0] &= vmask0; sv[1] &= vmask1;            // rows / columns past the live extents
                }
                int cnt = __popc(sv[0]) + __popc(sv[1]);
                bool stop = at_end;
                if (!at_end) {
                    // the workgroup's survivors: every wavefront leaves its count, everybody adds the four up
                    int wcnt = cnt;
#pragma unroll
                    for (int off = 32; off; off >>= 1) wcnt += __shfl_xor(wcnt, off);
                    if (lane == 0) lds_write4(scnt_a + (unsigned)wid * 4u, wcnt);
                    bscan::lds_barrier();
                    v4i c4 = lds_read16(scnt_a);
                    lds_wait_i(c4);
                    // (wave-uniform by construction; saying so keeps the whole K loop's control flow -- stage counters, buffer
                    //  indices, the stop decision -- on the scalar side instead of exec-masked vector code)
                    const int total = __builtin_amdgcn_readfirstlane((c4.x + c4.y) + (c4.z + c4.w));
                    flip ^= 1;
                    // an exact refinement costs about what a K stage of this tile costs: stop when the survivors are
                    // fewer than the stages that are left (plus the handful of true matches a tile holds anyway)
                    stop = total <= stop_thr;
                }
                if (stop) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        unsigned bits = sv[mi];
                        while (bits) {                           // (rare: about one survivor per detection column)
                            const int b = __ffs((int)bits) - 1, r = b & 15;
                            bits &= bits - 1;
                            const int q = bn * FBN + wc * 64 + (b >> 4) * 32 + frow;
                            const int i = bm * FBM + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                            atomicOr(&d.cbits[(long)i * d.cb_ld + (q >> 5)], 1u << (q & 31));
                        }
                    }
                    if constexpr (LIST) {
                        int wtot = cnt;
#pragma unroll
                        for (int off = 32; off; off >>= 1) wtot += __shfl_xor(wtot, off);
                        wtot = __builtin_amdgcn_readfirstlane(wtot);
                        if (wtot > 0) {                          // (about every other wavefront of a tile: one true match per detection)
                            const int slot = vt * 4 + wid;
                            if (wtot <= CAND_W) {
                                int pre = cnt;                   // inclusive scan of the lanes' counts
#pragma unroll
                                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(pre, off); if (lane >= off) pre += t; }
                                int pos = slot * CAND_W + pre - cnt;
#pragma unroll
                                for (int mi = 0; mi < 2; ++mi) {
                                    unsigned bits = sv[mi];
                                    while (bits) {
                                        const int b = __ffs((int)bits) - 1, r = b & 15;
                                        bits &= bits - 1;
                                        const int q = bn * FBN + wc * 64 + (b >> 4) * 32 + frow;
                                        const int i = bm * FBM + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                                        d.cand[pos++] = make_int2(i, q);
                                    }
                                }
                            }
                            if (lane == 0) d.cand_cnt[slot] = wtot <= CAND_W ? wtot : -1;      // (-1: the consumer sweeps the quadrant's bitmap words)
                        }
                    }
                    st_cand += cnt;
                    if (!at_end) ++st_early;
                    const int here = at_end ? kt_all : kt;
                    if (here >= last_exit) { last_exit = here; early_run = 0; }
                    else if (++early_run >= 3) { last_exit = here; early_run = 0; }
#ifdef BTS_FILTER_TRACE
                    if (tm_first == 0) tm_first = (long long)__builtin_amdgcn_s_memrealtime() - tc0; else
#endif
                    FT_ADD(tm_check, tc0);
                    break;
                }
                ci = skip_unused(ci + 1);
                rimg ^= 1;                           // (the image just tested may still be read by the slower wavefronts)
                stage_rec(rimg, ci);
#ifdef BTS_FILTER_TRACE
                if (tm_first == 0) tm_first = (long long)__builtin_amdgcn_s_memrealtime() - tc0; else
#endif
                FT_ADD(tm_check, tc0);
            }
            // ---- K stage kt
            if (issued <= kt) {                  // its prefetch was skipped (an exit was predicted here and did not happen): the
                stage(cur, kt);                  // speculative first stage of the next tile sits in this buffer -- overwritten
                issued = kt + 1;
                next_staged = false;
            }
            {
                FT_BEGIN(tw0);
                wait_stage_barrier<0>();         // stage kt has landed; every wave is past its reads of the other buffer
                FT_ADD(tm_wait, tw0);
#ifdef BTS_FILTER_TRACE
                if (tm_pro == 0) tm_pro = (long long)__builtin_amdgcn_s_memrealtime() - tm_begin;
#endif
            }
            // Prefetch the next stage into the other buffer -- unless this workgroup's previous tile stopped right there: then
            // this one most likely will, and the buffer takes the NEXT TILE's first stage instead (it lands under this stage
            // and the candidate test, so the next K loop starts without a cold load)
            if (kt + 1 < kt_all && issued == kt + 1) {
                if (kt + 1 != last_exit) { stage(cur ^ 1, kt + 1); issued = kt + 2; }
                else if (nvt < ntiles && !next_staged) { stage_next(cur ^ 1, ia, ib); next_staged = true; }
            }
            stage_mfma<VAR == 1 ? 1 : 0>(lds_a + cur * (FBM * DBK), lds_b + cur * (FBN * DBK), fm, fh, acc);
            cur ^= 1;
            ++kt;
            ++st_stages;
        }
        ++tile_no;
        if (nvt >= ntiles) break;
        vt = nvt; bm = nbm; bn = nbnn;
        my_meta_o = (unsigned)irec * feat::I8_META;
        rimg ^= 1;
        if (!next_staged) {
            // stopped at an unpredicted checkpoint: a prefetched stage of the OLD tile may still be landing in buffer `cur`;
            // the new tile starts in the other one (the test's barriers put every wave past its reads of it) and the
            // first stage barrier waits for both
            cur ^= 1;
            stage_next(cur, ia, ib);
        }
        set_src(ia, ib);
    }
    if (tid == 0) {
        atomicAdd(&d.fstat[FS_STAGES], st_stages); atomicAdd(&d.fstat[FS_EARLY], st_early); atomicAdd(&d.fstat[FS_TILES], tile_no);
#ifdef BTS_FILTER_TRACE
        tm_all = (long long)__builtin_amdgcn_s_memrealtime() - tm_begin;
        atomicAdd(&d.fstat[5], (int)tm_check); atomicAdd(&d.fstat[6], (int)tm_wait); atomicAdd(&d.fstat[7], (int)tm_all);
        atomicAdd(&d.fstat[3], (int)tm_first); atomicAdd(&d.fstat[4], (int)tm_pro);      // (trace build: [3] is NOT the candidate count)
        const int t_beg = (int)(tm_begin & 0x3fffffff), t_end = (int)(((long long)__builtin_amdgcn_s_memrealtime()) & 0x3fffffff);
        atomicMin(&d.fstat[8], t_beg); atomicMax(&d.fstat[9], t_end); atomicMax(&d.fstat[10], (int)tm_all); atomicMax(&d.fstat[11], t_beg);
#endif
    }
    // candidates: every lane counted its own
#pragma unroll
    for (int off = 32; off; off >>= 1) st_cand += __shfl_xor(st_cand, off);
    if (lane == 0 && st_cand) atomicAdd(&d.fstat[FS_CAND], st_cand);
}


// ---- the one-pass pre-filter with the TILE SHAPE as a parameter (round 5; opt-in: BTS_FILTER_TILE at create; NOT yet measured) ----
// k_filter_i8 above is tied to 128 x 128 tiles, two resident workgroups per CU, 2 x 2 wavefronts of 64 x 64.  Its measured model at
// the headline size (round 2) is ~22 us fixed + three tile ROUNDS (1 106 live tiles over 512 resident workgroups; the third
// round holds 82) of ~8 us each -- a tile's time is its latency chain, not its size.  This template keeps the kernel's structure
// (persistent workgroups, LDS-DMA double buffering in 128-byte stages, records one stage ahead, stop prediction, the next tile's
// first stage issued under the test) and frees the shape:
//   TM x TN       tile (multiples of 32), 4 wavefronts as WR x WC, each (TM / WR) x (TN / WC) = MI x NI blocks of 32 x 32;
//   OCC           resident workgroups per CU (the register cap follows: 512 / OCC per lane);
//   NIMG          record images in LDS: with ONE image a test's readers must be through before the next checkpoint's records are
//                 staged -- the count exchange's barrier, which every test of this kernel therefore runs (k_filter_i8 skips it at the
//                 end of a row) -- in exchange for (TM + TN) * 16 bytes of LDS.
// Instantiations:   160 x 128 (1 x 4 wavefronts of 160 x 32: 80 accumulators per lane; 79 008 B of LDS with one record image, two
//                   workgroups per CU) -- 882 live tiles at the headline size = TWO rounds;
//                   128 x 64 (2 x 2 wavefronts of 64 x 32; 52.4 KB, 168 registers: THREE workgroups per CU, 768 resident).
// Same bound, same records, same stop rule as k_filter_i8; the tile's row summary is taken over TM rows and stop_thr scales with
// the tile's area, so the candidate SET may differ from the 128 x 128 tiles' (both are supersets of the pairs the exact refinement
// accepts: the step's results are identical).  No face pass, no candidate list (the sorted-edge form keeps k_filter_i8).
template <int TM, int TN, int WR, int WC>
struct FragMapT {
    static constexpr int MI = TM / (32 * WR), NI = TN / (32 * WC);
    // pa[x] = byte offset of this lane's fragment row in an operand image, with the row's swizzle and the lane's K half folded in:
    // the 16-byte chunk of K step k (0..3) is at  pa ^ (k << 5)  -- row offsets are multiples of 128, the chunk index
    // ((2 k + fh) ^ swz(row)) << 4 lives in bits 4-6, so "^" composes (one xor + one add per read, one register per block)
    int pa[MI], pb[NI];
    __device__ __forceinline__ void init(int wr, int wc, int frow, int fh) {
#pragma unroll
        for (int x = 0; x < MI; ++x) { const int ra = wr * (TM / WR) + x * 32 + frow; pa[x] = ra * DBK + ((((ra >> 1) & 7) ^ fh) << 4); }
#pragma unroll
        for (int x = 0; x < NI; ++x) { const int rb = wc * (TN / WC) + x * 32 + frow; pb[x] = rb * DBK + ((((rb >> 1) & 7) ^ fh) << 4); }
    }
};
#if !defined(BTS_HOST_EMU)
// (the first statement carries the wait; the others only tie their operand behind it -- volatile asm statements keep their order --
//  so that no MFMA can be scheduled above the wait whichever fragment it reads)
__device__ __forceinline__ void lds_wait_first(v4i &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
__device__ __forceinline__ void lds_tie(v4i &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void lds_tie_f(f32x4 &a) { asm volatile("" : "+v"(a)); }
#else
__device__ __forceinline__ void lds_wait_first(v4i &) {}
__device__ __forceinline__ void lds_tie(v4i &) {}
__device__ __forceinline__ void lds_tie_f(f32x4 &) {}
#endif
// one K stage (128 int8 per row) of a wave's MI x NI blocks: four K = 32 steps; TB of them per batch of LDS reads (2 like
// k_filter_i8 while the fragments of two steps fit the register budget, 1 for the tall shapes: 80 accumulators + 2 x 24 fragment
// registers pushed the 160 x 128 kernel into scratch)
template <int TM, int TN, int WR, int WC>
__device__ __forceinline__ void stage_mfma_t(lds_t abase, lds_t bbase, const FragMapT<TM, TN, WR, WC> &fm,
                                             i32x16 (&acc)[FragMapT<TM, TN, WR, WC>::MI][FragMapT<TM, TN, WR, WC>::NI]) {
    constexpr int MI = FragMapT<TM, TN, WR, WC>::MI, NI = FragMapT<TM, TN, WR, WC>::NI;
    constexpr int TB = (MI + NI) * 2 * 4 + MI * NI * 16 > 112 ? 1 : 2;
#pragma unroll
    for (int h = 0; h < 4 / TB; ++h) {
        v4i fa[TB][MI], fb[TB][NI];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int kx = (h * TB + t) << 5;            // K step: bytes [32 k, +32) of the stage, this lane half takes 16 of them
#pragma unroll
            for (int x = 0; x < MI; ++x) fa[t][x] = lds_read16(abase + (lds_t)(fm.pa[x] ^ kx));
#pragma unroll
            for (int x = 0; x < NI; ++x) fb[t][x] = lds_read16(bbase + (lds_t)(fm.pb[x] ^ kx));
        }
        lds_wait_first(fa[0][0]);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
#pragma unroll
            for (int x = 0; x < MI; ++x) if (t || x) lds_tie(fa[t][x]);
#pragma unroll
            for (int x = 0; x < NI; ++x) lds_tie(fb[t][x]);
        }
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[t][mi], fb[t][ni], acc[mi][ni], 0, 0, 0);
    }
}

template <int TM, int TN, int WR, int WC, int OCC, int NIMG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_filter_t(Dev d, int m_max, int ntiles, int gemm_blocks, int kf_front) {
    typedef FragMapT<TM, TN, WR, WC> FM;
    constexpr int MI = FM::MI, NI = FM::NI;
    constexpr int PA = TM / 32, PB = TN / 32;            // 1-KiB wavefront loads (8 full 128-byte rows) per stage, wavefront and operand
    constexpr int E = TM + TN;                           // records of a tile: rows [0, TM), columns [TM, E)
    constexpr int NRW = (TM + 63) / 64;                  // wavefronts that summarise the rows' records
    static_assert(WR * WC == 4 && TM % (32 * WR) == 0 && TN % (32 * WC) == 0 && NI <= 2 && E <= 512 && E % 32 == 0 && NRW <= 4, "tile shape");
    static_assert(NIMG == 1 || NIMG == 2, "record images");
    __shared__ __attribute__((aligned(1024))) int8_t As2[DNS][TM * DBK];
    __shared__ __attribute__((aligned(1024))) int8_t Bs2[DNS][TN * DBK];
    constexpr int EP = (E + 63) / 64 * 64;               // (padded to whole wavefront loads: every LDS-DMA below is issued by a FULL wavefront)
    __shared__ float4 srec2[NIMG][EP];
    __shared__ __attribute__((aligned(16))) int s_cnt[2][4];     // survivors per wavefront
    __shared__ __attribute__((aligned(16))) float sred[2][16];   // {min uh, max alpha, max r, -} of the rows' records, per summarising wavefront
    // Kalman-predict workgroups of the same launch: behind the GEMM workgroups (kf_front = 0, the default) or in FRONT of them
    // (BTS_PREDICT_FIRST: kf_front of them; a kernel argument here, not a second instantiation)
    if (kf_front > 0 ? (int)blockIdx.x < kf_front : (int)blockIdx.x >= gemm_blocks) { predict_body(d, kf_front > 0 ? (int)blockIdx.x : (int)blockIdx.x - gemm_blocks); return; }
    const int gb = (int)blockIdx.x - kf_front;           // this workgroup's index among the GEMM workgroups
    // (the chain that opens a workgroup is cut like k_filter_i8's: the index loads of the first tile position travel with the extents')
    const int n_raw = d.hdr[H_NPOOL], m_raw = d.hdr[H_NHI];
    const int Dd = d.D;
    const int8_t *A = d.body_h, *B = d.det_h;
    const float *ma = d.body_m, *mb = d.det_m;
    const unsigned rs_a = d.sh_rt[0], cs_a = d.sh_ct[0], rs_b = d.sh_rd[0], cs_b = d.sh_cd[0];      // shadow layout (feat::i8_elem)
    const int nbn = (m_max + TN - 1) / TN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid / WC, wc = wid % WC;
    const int prow = lane >> 3, pchunk = lane & 7;
    const int frow = lane & 31, fh = lane >> 5;
    // record entry e of a tile: row e (e < TM) or column e - TM; thread tid carries entry tid and, when E > 256, the threads of the
    // first wavefront(s) entry 256 + tid as well (entries past E repeat the last one: whole wavefronts load, the padding is never read)
    const bool rec1 = tid < EP, rec2 = E > 256 && tid < EP - 256;
    const int e1 = tid < E ? tid : E - 1, e2 = 256 + tid < E ? 256 + tid : E - 1;
    auto rec_index = [&](int e, int bm, int bn, int n, int m) {
        const int g = e < TM ? bm * TM + e : bn * TN + (e - TM);
        return e < TM ? d.pool_slot[g < n ? g : bm * TM] : d.hi_idx[g < m ? g : bn * TN];
    };
    const int vt_first = gb;
    int sia[PA], sib[PB], sirec = 0, sirec2 = 0, sfa = 0, sfb = 0, sbm = 0, sbn = 0;      // the first position's indices, fetched ahead of the extents
    if (vt_first < ntiles) {
        const int vb = xcd_remap(vt_first, ntiles);
        sbm = vb / nbn; sbn = vb % nbn;
#pragma unroll
        for (int s = 0; s < PA; ++s) sia[s] = d.pool_slot[min(sbm * TM + (4 * s + wid) * 8 + prow, d.T - 1)];
#pragma unroll
        for (int s = 0; s < PB; ++s) sib[s] = d.hi_idx[min(sbn * TN + (4 * s + wid) * 8 + prow, d.Mx - 1)];
        sfa = d.pool_slot[min(sbm * TM, d.T - 1)]; sfb = d.hi_idx[min(sbn * TN, d.Mx - 1)];
        if (rec1) sirec = e1 < TM ? d.pool_slot[min(sbm * TM + e1, d.T - 1)] : d.hi_idx[min(sbn * TN + (e1 - TM), d.Mx - 1)];
        if (rec2) sirec2 = e2 < TM ? d.pool_slot[min(sbm * TM + e2, d.T - 1)] : d.hi_idx[min(sbn * TN + (e2 - TM), d.Mx - 1)];
    }
    const int n = __builtin_amdgcn_readfirstlane(n_raw), m = __builtin_amdgcn_readfirstlane(m_raw);     // (scalar registers)
    const int kt_all = Dd / DBK;
    int chk[feat::I8_NCHK];
    feat::i8_checkpoints(kt_all, chk, d.chk_first);
    static_assert(feat::I8_NCHK == 4, "chk_at selects by hand (a runtime-indexed register array would go to scratch)");
    auto chk_at = [&](int j) { return j == 0 ? chk[0] : j == 1 ? chk[1] : j == 2 ? chk[2] : chk[3]; };
    auto skip_unused = [&](int j) { return (j < feat::I8_NCHK && chk_at(j) >= kt_all) ? (int)feat::I8_NCHK : j; };
    const lds_t lds_a = LDS_ADDR(&As2[0][0]), lds_b = LDS_ADDR(&Bs2[0][0]);
    FM fm;
    fm.init(wr, wc, frow, fh);
    const lds_t lds_rec = LDS_ADDR(&srec2[0][0]), lds_cnt = LDS_ADDR(&s_cnt[0][0]), lds_red = LDS_ADDR(&sred[0][0]);
    auto next_live = [&](int vt, int &bm, int &bn) {
        for (; vt < ntiles; vt += gemm_blocks) {
            const int vb = xcd_remap(vt, ntiles);
            bm = vb / nbn; bn = vb % nbn;
            if (bm * TM < n && bn * TN < m) return vt;
        }
        return ntiles;
    };
    auto fetch_idx = [&](int bm, int bn, int (&ia)[PA], int (&ib)[PB], int &irec, int &irec2) {
#pragma unroll
        for (int s = 0; s < PA; ++s) { const int i = bm * TM + (4 * s + wid) * 8 + prow; ia[s] = d.pool_slot[i < n ? i : bm * TM]; }       // rows past the end: any valid row, ignored later
#pragma unroll
        for (int s = 0; s < PB; ++s) { const int j = bn * TN + (4 * s + wid) * 8 + prow; ib[s] = d.hi_idx[j < m ? j : bn * TN]; }
        if (rec1) irec = rec_index(e1, bm, bn, n, m);
        if (rec2) irec2 = rec_index(e2, bm, bn, n, m);
    };
    unsigned asrc[PA], bsrc[PB];                         // byte offsets into A / B (tables up to 4 GiB of shadows)
    auto set_src = [&](const int (&ia)[PA], const int (&ib)[PB]) {
#pragma unroll
        for (int s = 0; s < PA; ++s) { const int r = (4 * s + wid) * 8 + prow; asrc[s] = (unsigned)ia[s] * rs_a + (unsigned)((pchunk ^ ((r >> 1) & 7)) * 16); }
#pragma unroll
        for (int s = 0; s < PB; ++s) { const int r = (4 * s + wid) * 8 + prow; bsrc[s] = (unsigned)ib[s] * rs_b + (unsigned)((pchunk ^ ((r >> 1) & 7)) * 16); }
    };
    auto stage = [&](int buf, int kt) {                  // PA + PB wavefront loads per stage and wave
#pragma unroll
        for (int s = 0; s < PA; ++s)
            __builtin_amdgcn_global_load_lds((gbl_void *)(A + ((size_t)asrc[s] + (size_t)kt * cs_a)), (lds_void *)&As2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
#pragma unroll
        for (int s = 0; s < PB; ++s)
            __builtin_amdgcn_global_load_lds((gbl_void *)(B + ((size_t)bsrc[s] + (size_t)kt * cs_b)), (lds_void *)&Bs2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
    };
    auto stage_next = [&](int buf, const int (&ia)[PA], const int (&ib)[PB]) {      // the NEXT tile's first stage, from the prefetched indices
#pragma unroll
        for (int s = 0; s < PA; ++s) {
            const int r = (4 * s + wid) * 8 + prow;
            __builtin_amdgcn_global_load_lds((gbl_void *)(A + ((size_t)((unsigned)ia[s] * rs_a) + (size_t)((pchunk ^ ((r >> 1) & 7)) * 16))), (lds_void *)&As2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < PB; ++s) {
            const int r = (4 * s + wid) * 8 + prow;
            __builtin_amdgcn_global_load_lds((gbl_void *)(B + ((size_t)((unsigned)ib[s] * rs_b) + (size_t)((pchunk ^ ((r >> 1) & 7)) * 16))), (lds_void *)&Bs2[buf][(4 * s + wid) * 8 * DBK], 16, 0, 0);
        }
    };
    const int stop_thr = 6 + (3 * TM * TN) / (2 * (n > 0 ? n : 1));
    const float i8_s2 = feat::i8_scale(Dd) * feat::i8_scale(Dd) * 0.99999f;        // integer dot products per unit of dequantised dot product
    int bm, bn;
    int vt = next_live(gb, bm, bn);
    if (vt >= ntiles) return;
    int ia[PA], ib[PB], irec = 0, irec2 = 0;
    if (vt == vt_first) {
#pragma unroll
        for (int s = 0; s < PA; ++s) ia[s] = sbm * TM + (4 * s + wid) * 8 + prow < n ? sia[s] : sfa;
#pragma unroll
        for (int s = 0; s < PB; ++s) ib[s] = sbn * TN + (4 * s + wid) * 8 + prow < m ? sib[s] : sfb;
        if (rec1) irec = e1 < TM ? (sbm * TM + e1 < n ? sirec : sfa) : (sbn * TN + (e1 - TM) < m ? sirec : sfb);
        if (rec2) irec2 = e2 < TM ? (sbm * TM + e2 < n ? sirec2 : sfa) : (sbn * TN + (e2 - TM) < m ? sirec2 : sfb);
    } else {
        fetch_idx(bm, bn, ia, ib, irec, irec2);
    }
    set_src(ia, ib);
    const float *mbase1 = e1 < TM ? ma : mb, *mbase2 = e2 < TM ? ma : mb;
    unsigned meta_o1 = (unsigned)irec * feat::I8_META, meta_o2 = (unsigned)irec2 * feat::I8_META;
    // records of checkpoint c into image img: entry e lands at srec2[img][e] (the LDS image of a wavefront load is lane-linear)
    auto stage_rec = [&](int img, int c) {
        if (rec1) __builtin_amdgcn_global_load_lds((gbl_void *)(mbase1 + ((size_t)meta_o1 + (size_t)(4 * c))), (lds_void *)&srec2[img][wid * 64], 16, 0, 0);
        if (rec2) __builtin_amdgcn_global_load_lds((gbl_void *)(mbase2 + ((size_t)meta_o2 + (size_t)(4 * c))), (lds_void *)&srec2[img][256 + wid * 64], 16, 0, 0);
        static_assert(EP % 64 == 0 && (EP <= 256 || EP - 256 <= 256), "rec1 / rec2 are wave-uniform");
    };
    int rimg = 0;                                // image the NEXT test's records are in (or on their way to)
#if !defined(BTS_HOST_EMU)
    asm volatile("" : "+v"(meta_o1), "+v"(meta_o2));      // (materialised before the first LDS-DMA: see k_filter_i8)
#endif
    stage(0, 0);
    int cur = 0, flip = 0;
    int last_exit = d.f_guess ? chk[0] : -1, early_run = 0, tile_no = 0;        // (stop prediction: see k_filter_i8)
    int st_stages = 0, st_early = 0, st_cand = 0;
    for (;;) {
        int ci = skip_unused(0);                     // next checkpoint (feat::I8_NCHK = the full row)
        if ((tile_no & 3) != 3)
            while (ci < feat::I8_NCHK && chk_at(ci) < last_exit) ci = skip_unused(ci + 1);
        stage_rec(rimg, ci);                         // (in FRONT of the index loads below: one drain instead of two, see k_filter_i8)
        int nbm = 0, nbnn = 0;
        const int nvt = next_live(vt + gemm_blocks, nbm, nbnn);
        if (nvt < ntiles) fetch_idx(nbm, nbnn, ia, ib, irec, irec2);        // the next tile's indices travel while this tile's K loop runs
        i32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;
        int issued = 1, kt = 0;
        bool next_staged = false;
        for (;;) {
            const bool at_end = kt == kt_all;
            if (at_end || (ci < feat::I8_NCHK && kt == chk_at(ci))) {
                // ---- candidate test on the prefix of kt stages (checkpoint ci; at the end: the full-row records)
                if (at_end && nvt < ntiles && !next_staged) { stage_next(cur, ia, ib); next_staged = true; }
                const lds_t srec_a = lds_rec + (unsigned)rimg * (unsigned)(EP * 16);
                const lds_t scnt_a = lds_cnt + (unsigned)flip * 16u, sred_a = lds_red + (unsigned)flip * 64u;
                unsigned sv[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) sv[mi] = 0u;
                {
                    if (wid < NRW) {                 // these wavefronts summarise the rows' records
                        const int e = wid * 64 + lane;
                        f32x4 mine = lds_read16f(srec_a + (unsigned)(e < TM ? e : TM - 1) * 16u);
                        lds_wait_f1(mine);
                        const bool live = e < TM;
                        const float m_u = wave_min_f(live ? mine.w : 3.0e38f), m_a = wave_max_f(live ? mine.z : -3.0e38f), m_r = wave_max_f(live ? mine.y : -3.0e38f);
                        if (lane == 0) lds_write16(sred_a + (unsigned)wid * 16u, f32x4{m_u, m_a, m_r, 0.0f});
                    }
                    bscan::lds_barrier();
                    f32x4 rr[NRW], cc[NI];
#pragma unroll
                    for (int w = 0; w < NRW; ++w) rr[w] = lds_read16f(sred_a + (unsigned)w * 16u);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) cc[ni] = lds_read16f(srec_a + (unsigned)(TM + wc * (TN / WC) + ni * 32 + frow) * 16u);
                    lds_wait_f1(rr[0]);
#pragma unroll
                    for (int w = 1; w < NRW; ++w) lds_tie_f(rr[w]);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) lds_tie_f(cc[ni]);
                    float u_min = rr[0].x, a_max = rr[0].y, r_max = rr[0].z;
#pragma unroll
                    for (int w = 1; w < NRW; ++w) { u_min = fminf(u_min, rr[w].x); a_max = fmaxf(a_max, rr[w].y); r_max = fmaxf(r_max, rr[w].z); }
                    int ti[NI], mx[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float t0 = ((u_min + cc[ni].w) - (a_max * cc[ni].y + cc[ni].z * r_max)) * i8_s2;
                        // (NaN / non-positive bounds: nothing is excluded)
                        ti[ni] = t0 > 0.0f ? (int)fminf(t0 * 0.999999f, 2.0e9f) - 1 : (int)0x80000000;
                        mx[ni] = acc[0][ni][0];
                    }
                    // survivors are rare: reduce a lane's accumulators per column block to their maximum first, extract bits only
                    // in a wavefront where some maximum reaches its threshold
                    bool any = false;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int r = 0; r < 16; ++r) mx[ni] = max(mx[ni], acc[mi][ni][r]);
                        any |= mx[ni] >= ti[ni];
                    }
                    if (__ballot(any) != 0ull) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                                for (int r = 0; r < 16; ++r) sv[mi] |= (acc[mi][ni][r] >= ti[ni] ? 1u : 0u) << (ni * 16 + r);
                    }
                }
                unsigned svany = 0u;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) svany |= sv[mi];
                if (__ballot(svany != 0u) != 0ull) {      // (wave-uniform: most wavefronts hold no survivor and skip the extents)
                    const int lim = n - bm * TM - (wr * (TM / WR) + 4 * fh);       // local row (mi * 32 + 8 g + u) is live iff < lim
                    const int q0 = bn * TN + wc * (TN / WC) + frow;
                    unsigned cm = 0u;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) cm |= q0 + ni * 32 < m ? (0xffffu << (ni * 16)) : 0u;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        unsigned rv = 0u;
#pragma unroll
                        for (int r = 0; r < 16; ++r) rv |= (mi * 32 + (r & 3) + 8 * (r >> 2) < lim ? 1u : 0u) << r;
                        sv[mi] &= (rv | (rv << 16)) & cm;      // rows / columns past the live extents
                    }
                }
                int cnt = 0;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) cnt += __popc(sv[mi]);
                bool stop = at_end;
                if (!at_end || NIMG == 1) {
                    // the workgroup's survivors: every wavefront leaves its count, everybody adds the four up (with one record
                    // image this exchange's barrier is also what lets the next records be staged: every test runs it)
                    int wcnt = cnt;
#pragma unroll
                    for (int off = 32; off; off >>= 1) wcnt += __shfl_xor(wcnt, off);
                    if (lane == 0) lds_write4(scnt_a + (unsigned)wid * 4u, wcnt);
                    bscan::lds_barrier();
                    v4i c4 = lds_read16(scnt_a);
                    lds_wait_i(c4);
                    const int total = __builtin_amdgcn_readfirstlane((c4.x + c4.y) + (c4.z + c4.w));
                    flip ^= 1;
                    if (!at_end) stop = total <= stop_thr;
                }
                if (stop) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        unsigned bits = sv[mi];
                        while (bits) {                           // (rare: about one survivor per detection column)
                            const int b = __ffs((int)bits) - 1, r = b & 15;
                            bits &= bits - 1;
                            const int q = bn * TN + wc * (TN / WC) + (b >> 4) * 32 + frow;
                            const int i = bm * TM + wr * (TM / WR) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                            atomicOr(&d.cbits[(long)i * d.cb_ld + (q >> 5)], 1u << (q & 31));
                        }
                    }
                    st_cand += cnt;
                    if (!at_end) ++st_early;
                    const int here = at_end ? kt_all : kt;
                    if (here >= last_exit) { last_exit = here; early_run = 0; }
                    else if (++early_run >= 3) { last_exit = here; early_run = 0; }
                    break;
                }
                ci = skip_unused(ci + 1);
                if (NIMG == 2) rimg ^= 1;            // (two images: the one just tested may still be read by the slower wavefronts)
                stage_rec(rimg, ci);
            }
            // ---- K stage kt
            if (issued <= kt) { stage(cur, kt); issued = kt + 1; next_staged = false; }      // its prefetch was skipped (a predicted exit that did not happen)
            wait_stage_barrier<0>();             // stage kt has landed; every wave is past its reads of the other buffer
            if (kt + 1 < kt_all && issued == kt + 1) {
                if (kt + 1 != last_exit) { stage(cur ^ 1, kt + 1); issued = kt + 2; }
                else if (nvt < ntiles && !next_staged) { stage_next(cur ^ 1, ia, ib); next_staged = true; }
            }
            stage_mfma_t<TM, TN, WR, WC>(lds_a + cur * (TM * DBK), lds_b + cur * (TN * DBK), fm, acc);
            cur ^= 1;
            ++kt;
            ++st_stages;
        }
        ++tile_no;
        if (nvt >= ntiles) break;
        vt = nvt; bm = nbm; bn = nbnn;
        meta_o1 = (unsigned)irec * feat::I8_META; meta_o2 = (unsigned)irec2 * feat::I8_META;
        if (NIMG == 2) rimg ^= 1;
        if (!next_staged) { cur ^= 1; stage_next(cur, ia, ib); }      // stopped at an unpredicted checkpoint: see k_filter_i8
        set_src(ia, ib);
    }
    if (tid == 0) { atomicAdd(&d.fstat[FS_STAGES], st_stages); atomicAdd(&d.fstat[FS_EARLY], st_early); atomicAdd(&d.fstat[FS_TILES], tile_no); }
#pragma unroll
    for (int off = 32; off; off >>= 1) st_cand += __shfl_xor(st_cand, off);
    if (lane == 0 && st_cand) atomicAdd(&d.fstat[FS_CAND], st_cand);
}


// ---- stage-1 edges, SORTED / LIST form (opt-in: BTS_EDGES_SORTED=1 when the tracker is created; NOT yet measured) --------------
// What the tile form above spends its time on is (a) screening every row of a tile against every wavefront's 64 columns --
// N M / 64 wave-rows of ~20 instructions, nearly all of them for nothing: a wavefront's columns are 64 detections from all over
// the image, so no row can be ruled out as a whole -- and (b) the exact dot products of the appearance candidates, which sit in
// the row loop of whichever wavefront happens to own the pair.  This form takes both out of the tiles:
//   * COLUMNS are visited in the order of the uniform grid over the detection centres that tk_front builds (build_grid:
//     Dev::cell_items = the high-score detections sorted by cell, row-major).  A wavefront's 64 columns are then neighbours;
//     their union box U (packed 16-bit pixels, rounded outward like every screening box) covers a few percent of the image
//     and ONE test per row against U -- lane = row, 64 rows per instruction -- leaves the handful of rows that can overlap
//     any of them.  Only those are screened per pair and evaluated exactly (bit-identical iou::cost, same dtype flags).
//   * CANDIDATES arrive as a list (k_filter_i8<.., LIST>): the workgroups behind the tiles take one pair per wavefront --
//     exact float32 dot product, the pair's IoU cost, ONE edge with min(IoU cost, 1 - sim) (np.minimum of D:1554) -- all in
//     parallel instead of serialised inside tiles.  A tile drops an IoU edge whose pair is a candidate (bit test on the few
//     edges it staged, at its flush): the candidate wavefront owns that pair, so every pair still yields at most one edge.
// Same edge SET as tk_edges<1,1,R> (another order; the solver does not depend on it).  The bitmap is not consumed here: tk_mid1
// clears the listed bits and the counts.  Restrictions (the launcher falls back to the tile form): no face features, no
// motion gate, no cost recording, match threshold < 1, the one-pass pre-filter (not split K).
__device__ __forceinline__ unsigned wave_min_pk16(unsigned v) {      // per-half signed minimum over the wavefront, to every lane
    typedef short s2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off);
        v = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s2, v), __builtin_bit_cast(s2, o)));
    }
    return v;
}
__device__ __forceinline__ unsigned wave_max_pk16(unsigned v) {
    typedef short s2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off);
        v = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, v), __builtin_bit_cast(s2, o)));
    }
    return v;
}

// per-wavefront edge staging shared by the tiles and the candidate wavefronts of tk_edges_sorted
constexpr int SE_EB = 128;
struct SortedStage {
    double c[CT_COLS / 64][SE_EB];
    int32_t r[CT_COLS / 64][SE_EB], q[CT_COLS / 64][SE_EB];
};

template <int ROWS>
__device__ __forceinline__ void edges_tile_sorted(const Dev &d, const bts_lap_work &w, int bx, int by, SortedStage &S) {
    __shared__ double sa[ROWS][4];
    __shared__ double sarea[ROWS];
    __shared__ iou::ScreenBox sboxf[ROWS];
    __shared__ int s_wcnt[CT_COLS / 64], s_base;
    const int32_t *h = d.hdr;
    const int n = h[H_NPOOL], m = h[H_NHI];
    const int i0 = by * ROWS, qs = bx * CT_COLS + threadIdx.x;      // qs: position in the grid order
    const bool live = qs < m;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // chains of dependent loads start side by side before the barrier: this thread's column (grid order -> high-score index ->
    // detection -> box), the tile's row boxes
    const int q = live ? d.cell_items[qs] : 0;
    const int det = live ? d.hi_idx[q] : 0;
    const double bb[4] = {d.det_tlbr[det * 4 + 0], d.det_tlbr[det * 4 + 1], d.det_tlbr[det * 4 + 2], d.det_tlbr[det * 4 + 3]};
    for (int u = threadIdx.x; u < ROWS; u += CT_COLS) {
        if (i0 + u < n) {
            const int i = i0 + u;
            const double box[4] = {d.pool_tlbr[i * 4 + 0], d.pool_tlbr[i * 4 + 1], d.pool_tlbr[i * 4 + 2], d.pool_tlbr[i * 4 + 3]};
            for (int c = 0; c < 4; ++c) sa[u][c] = box[c];
            sarea[u] = iou::area(box, false);
            sboxf[u] = iou::screen_box(box);
        } else sboxf[u] = iou::screen_none();
    }
    __syncthreads();
    const double area_b = iou::area(bb, true);
    const iou::ScreenBox bf = live ? iou::screen_box(bb) : iou::screen_none();      // (a dead column overlaps nothing)
    iou::ScreenBox U;                                                               // union of this wavefront's column boxes
    U.lo = __builtin_bit_cast(iou::s16x2, wave_min_pk16(__builtin_bit_cast(unsigned, bf.lo)));
    U.hi = __builtin_bit_cast(iou::s16x2, wave_max_pk16(__builtin_bit_cast(unsigned, bf.hi)));
    const double limit = d.match1;
    const int rows = min(ROWS, n - i0);
    int wcount = 0;
    // An IoU edge whose pair is an appearance candidate belongs to the candidate wavefronts (one edge per pair): the staged
    // edges are tested against the bitmap when they are flushed -- all bit loads of a flush travel together.
    unsigned long long keep[SE_EB / 64];
    auto test_staged = [&]() {                               // -> number of staged edges that stay
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int kept = 0;
#pragma unroll
        for (int k0 = 0; k0 < SE_EB / 64; ++k0) {
            const int k = k0 * 64 + lane;
            bool kp = false;
            if (k < wcount) {
                const long brow = d.pool_cbrow ? (long)d.pool_cbrow[S.r[wid][k]] : (long)S.r[wid][k];      // (BTS_FILTER_AHEAD: by slot / previous detection)
                kp = !((d.cbits[brow * d.cb_ld + (S.q[wid][k] >> 5)] >> (S.q[wid][k] & 31)) & 1u);
            }
            keep[k0] = __ballot(kp);
            kept += __popcll(keep[k0]);
        }
        return kept;
    };
    auto write_kept = [&](int pos0) {
#pragma unroll
        for (int k0 = 0; k0 < SE_EB / 64; ++k0) {
            const int k = k0 * 64 + lane;
            if ((keep[k0] >> lane) & 1ull) {
                const int pos = pos0 + __popcll(keep[k0] & ((1ull << lane) - 1ull));
                if (pos < w.e_cap) { w.e_row[pos] = S.r[wid][k]; w.e_col[pos] = S.q[wid][k]; w.e_cost[pos] = S.c[wid][k]; }
                else w.hdr[2] = 1;
            }
            pos0 += __popcll(keep[k0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wcount = 0;
    };
    auto emit = [&](bool pred, int row, double c) {          // convergent: every lane of the wavefront calls it
        const unsigned long long mk = __ballot(pred);
        if (!mk) return;
        if (wcount + __popcll(mk) > SE_EB) {                  // mid-tile overflow of the staging buffer (dense crowds): this wavefront's own bump
            const int kept = test_staged();
            int base = 0;
            if (lane == 0 && kept) base = atomicAdd(&w.hdr[0], kept);
            write_kept(__shfl(base, 0));
        }
        if (pred) {
            const int p = wcount + __popcll(mk & ((1ull << lane) - 1ull));
            S.c[wid][p] = c; S.r[wid][p] = row; S.q[wid][p] = q;
        }
        wcount += __popcll(mk);
    };
#pragma unroll 1
    for (int base = 0; base < rows; base += 64) {
        // rows of this block that can overlap ANY of the wavefront's columns: lane = row
        const int rr = base + lane;
        unsigned long long touch = __ballot(rr < rows && iou::screen_overlap(sboxf[rr & (ROWS - 1)], U));
        while (touch) {
            const int r = base + __ffsll((long long)touch) - 1;
            touch &= touch - 1;
            if (!__ballot(iou::screen_overlap(sboxf[r], bf))) continue;
            const double aa[4] = {sa[r][0], sa[r][1], sa[r][2], sa[r][3]};
            double c = iou::cost(aa, false, sarea[r], bb, true, area_b);
            if (d.c16) c = round_to_f16(c);
            emit(live && c <= limit, i0 + r, c);
        }
    }
    // end of tile: ONE counter bump for the whole workgroup
    const int kept = test_staged();
    if (lane == 0) s_wcnt[wid] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < CT_COLS / 64; ++k) tot += s_wcnt[k];
        s_base = tot ? atomicAdd(&w.hdr[0], tot) : 0;
    }
    __syncthreads();
    int pos0 = s_base;
    for (int k = 0; k < wid; ++k) pos0 += s_wcnt[k];
    write_kept(pos0);
    __syncthreads();          // the tile's LDS is reused by the workgroup's next tile
}

// one candidate pair (pool row i, high-score column q), handled by a whole wavefront: returns its edge cost to lane 0
__device__ __forceinline__ bool cand_pair(const Dev &d, int i, int q, int lane, double &c_out) {
    const long strk = d.pool_slot[i], det = d.hi_idx[q];
    const float emb = 1.0f - fmaxf(0.0f, feat::wave_dot<4>(d.body_curr + strk * d.D, IN(d, feats) + det * d.D, d.D, lane));
    const double aa[4] = {d.pool_tlbr[i * 4 + 0], d.pool_tlbr[i * 4 + 1], d.pool_tlbr[i * 4 + 2], d.pool_tlbr[i * 4 + 3]};
    const double bb[4] = {d.det_tlbr[det * 4 + 0], d.det_tlbr[det * 4 + 1], d.det_tlbr[det * 4 + 2], d.det_tlbr[det * 4 + 3]};
    double c = iou::cost(aa, false, iou::area(aa, false), bb, true, iou::area(bb, true));
    if (!(fminf(emb, 1.0f) > d.app)) c = fmin(c, (double)emb);      // (no face features in this form: the face distance is 1)
    if (d.c16) c = round_to_f16(c);
    c_out = c;
    return c <= d.match1;
}

__device__ __forceinline__ void cand_all(const Dev &d, const bts_lap_work &w, int bid, int nblk, SortedStage &S) {
    const int32_t *h = d.hdr;
    const int n = h[H_NPOOL], m = h[H_NHI];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool fa = d.pool_of_slot != nullptr;               // BTS_FILTER_AHEAD: two lists, rows named by slot / previous detection
    const int nslots = (fa ? d.fa_mt + d.fa_dt : d.f_ntiles) * 4, nbn = (d.f_mc + FBN - 1) / FBN;
    int wcount = 0;
    auto flush = [&]() {
        if (wcount == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(&w.hdr[0], wcount);
        base = __shfl(base, 0);
        for (int k = lane; k < wcount; k += 64) {
            const int pos = base + k;
            if (pos < w.e_cap) { w.e_row[pos] = S.r[wid][k]; w.e_col[pos] = S.q[wid][k]; w.e_cost[pos] = S.c[wid][k]; }
            else w.hdr[2] = 1;
        }
        wcount = 0;
    };
    // (ahead form) a side pass's row -> the pool row that reads it, or -1: the track is not in this frame's pool, or its bitmap row is
    // of the other kind (a slot the previous step rewrote / a detection nobody absorbed)
    auto pool_row = [&](int pass, int i) {
        if (!fa) return i;
        const int r = pass ? d.pool_of_pdet[i] : d.pool_of_slot[i];
        return ((unsigned)r < (unsigned)n && d.pool_cbrow[r] == (pass ? d.T + i : i)) ? r : -1;
    };
    auto pair = [&](int i, int q) {                           // convergent
        if (i < 0) return;
        double c;
        const bool edge = cand_pair(d, i, q, lane, c);
        if (edge) {
            if (wcount == SE_EB) flush();
            if (lane == 0) { S.c[wid][wcount] = c; S.r[wid][wcount] = i; S.q[wid][wcount] = q; }
            ++wcount;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    for (int sv = bid * (CT_COLS / 64) + wid; sv < nslots; sv += nblk * (CT_COLS / 64)) {
        // (ahead form: virtual slot sv -> the pass it belongs to, its slot inside that pass's list, the list's place in memory)
        const int pass = (fa && sv >= d.fa_mt * 4) ? 1 : 0;
        const int s = pass ? sv - d.fa_mt * 4 : sv, sm = pass ? d.fa_dd_slot0 + s : s;
        const int cnt = __builtin_amdgcn_readfirstlane(d.cand_cnt[sm]);
        if (cnt == 0) continue;
        if (cnt > 0) {
            const int2 mine = lane < cnt ? d.cand[(long)sm * CAND_W + lane] : make_int2(0, 0);
            for (int e = 0; e < cnt; ++e) pair(pool_row(pass, __shfl(mine.x, e)), __shfl(mine.y, e));
        } else {
            // the slot overflowed: sweep the quadrant's 64 bitmap words (64 rows x the one aligned 8-byte word of its 64 columns)
            const int ntl = !fa ? d.f_ntiles : pass ? d.fa_dt : d.fa_mt, nrow = !fa ? n : pass ? d.fa_nprev : d.T;
            const int vb = xcd_remap(s >> 2, ntl), bm = vb / nbn, bn = vb % nbn, wr = (s & 3) >> 1, wc = s & 1;
            const int row = bm * FBM + wr * 64 + lane, col0 = bn * FBN + wc * 64;
            unsigned long long word = 0ull;
            if (row < nrow && col0 < d.Mx) word = *reinterpret_cast<const unsigned long long *>(&d.cbits[(long)((pass ? d.T : 0) + row) * d.cb_ld + (col0 >> 5)]);
            unsigned long long any = __ballot(word != 0ull);
            while (any) {
                const int src = __ffsll((long long)any) - 1;
                any &= any - 1;
                unsigned long long bits = ((unsigned long long)(unsigned)__shfl((int)(word >> 32), src) << 32) | (unsigned)__shfl((int)word, src);
                const int prow = pool_row(pass, bm * FBM + wr * 64 + src);         // (wave-uniform: one row per step)
                while (bits) {
                    const int b = __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    if (col0 + b < m) pair(prow, col0 + b);
                }
            }
        }
    }
    flush();
}

template <int ROWS>
__global__ __launch_bounds__(CT_COLS) __attribute__((amdgpu_waves_per_eu(4, 8))) void tk_edges_sorted(Dev d, bts_lap_work w, int tile_blocks) {
    __shared__ SortedStage S;
    if ((int)blockIdx.x >= tile_blocks) { cand_all(d, w, (int)blockIdx.x - tile_blocks, (int)gridDim.x - tile_blocks, S); return; }
    const int32_t *h = d.hdr;
    const int n = h[H_NPOOL], m = h[H_NHI];
    const int nct = (m + CT_COLS - 1) / CT_COLS, ntile = nct * ((n + ROWS - 1) / ROWS);
    for (int t = blockIdx.x; t < ntile; t += tile_blocks) edges_tile_sorted<ROWS>(d, w, t % nct, t / nct, S);
}

// the candidate list and the bitmap bits it names are cleared for the next frame (workgroups of tk_mid1, behind the assignment)
__device__ __forceinline__ void cand_clear(const Dev &d, int bid, int nblk) {
    const int32_t *h = d.hdr;
    const int n = h[H_NPOOL];
    const int lane = threadIdx.x & 63, wv = bid * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6), nw = nblk * ((int)blockDim.x >> 6);
    const int nslots = d.f_ntiles * 4, nbn = (d.f_mc + FBN - 1) / FBN;
    for (int s = wv; s < nslots; s += nw) {
        const int cnt = __builtin_amdgcn_readfirstlane(d.cand_cnt[s]);
        if (cnt == 0) continue;
        if (cnt > 0) {
            if (lane < cnt) { const int2 e = d.cand[(long)s * CAND_W + lane]; d.cbits[(long)e.x * d.cb_ld + (e.y >> 5)] = 0u; }
        } else {
            const int vb = xcd_remap(s >> 2, d.f_ntiles), bm = vb / nbn, bn = vb % nbn, wr = (s & 3) >> 1, wc = s & 1;
            const int row = bm * FBM + wr * 64 + lane, col0 = bn * FBN + wc * 64;
            if (row < n && col0 < d.Mx) *reinterpret_cast<unsigned long long *>(&d.cbits[(long)row * d.cb_ld + (col0 >> 5)]) = 0ull;
        }
        if (lane == 0) d.cand_cnt[s] = 0;
    }
}

// ---- detection feature rows: in-place renormalisation (quirk Q8) ------------------------------------------
__device__ __forceinline__ void renorm_body(const Dev &d, int blk) {        // four rows per 1024-thread workgroup
    const int32_t *h = d.hdr;
    const int nhi = h[H_NHI], nlo = h[H_NLO];
    const int q = blk * 4 + feat::row_group();
    // (Dev::renorm_fused, BTS_RENORM_FUSED=1 at create; NOT yet measured: the rows stage 1 matched -- most of them -- are left to
    //  the feature workgroups of tk_mid2, which read them anyway: one 4 D-byte read per match less, and this launch is down to
    //  its list jobs plus the few unmatched / low-score rows.  Same functions on the same values: bit-identical rows.)
    const bool on = q < nhi + nlo && !(d.renorm_fused && q < nhi && d.y1[q < nhi ? q : 0] >= 0);
    const long det = !on ? 0 : q < nhi ? d.hi_idx[q] : d.lo_idx[q - nhi];
    const float nb = feat::renorm_row(IN(d, feats) + det * d.D, d.D, on);
    // the detection's record becomes the record of the unit row a track will hold (feat::i8_track_side): its column-side
    // readers (k_filter, the stage-1 tiles) ran before this launch
    if (d.filter && on && feat::row_tid() == 0) feat::i8_track_side(d.det_m + det * feat::I8_META, nb, d.gate_sim);
    if (d.Df) {
        const float nf = feat::renorm_row(IN(d, face_feats) + det * d.Df, d.Df, on);
        if (d.filter && on && feat::row_tid() == 0) feat::i8_track_side(d.dface_m + det * feat::I8_META, nf, d.gate_sim);
    }
}

// ---- match application (D:570-610).  FEAT=false: Kalman update + lifecycle, one thread per row with
// the state in registers.  FEAT=true: ReID feature bookkeeping, one workgroup per row.  (Two launches:
// the Kalman part needs the whole register file, the feature part wants occupancy.) ---------------------
template <int STAGE, bool FEAT>
__device__ __forceinline__ void apply_body(const Dev &d, int bid) {
    const int32_t *h = d.hdr;
    const int n = STAGE == 1 ? h[H_NPOOL] : STAGE == 2 ? h[H_NR2] : h[H_NUNC];
    const int32_t *rslot = STAGE == 1 ? d.pool_slot : STAGE == 2 ? d.r2_slot : d.unc_slot;
    const int32_t *xs = STAGE == 1 ? d.x1 : STAGE == 2 ? d.x2 : d.x3;
    const int32_t *cdet = STAGE == 1 ? d.hi_idx : STAGE == 2 ? d.lo_idx : d.left_det;
    if (FEAT) {                                               // ---- feature rows: one workgroup per matched COLUMN
        // (a stage matches at most one row per detection column and there are far fewer columns than table rows:
        // the grid is max_dets workgroups, not max_tracks)
        const int32_t *ys = STAGE == 1 ? d.y1 : STAGE == 2 ? d.y2 : d.y3;
        const int xq = bid, xqc = min(xq, d.Mx - 1);
        const int r_raw = ys[xqc], det_raw = cdet[xqc];          // (fetched together with the extents: capacity-clamped positions)
        const int m = STAGE == 1 ? h[H_NHI] : STAGE == 2 ? h[H_NLO] : h[H_NLEFT];
        if (xq >= m) return;
        const int r = r_raw;
        if (r < 0 || r >= n) return;
        const long s = rslot[r], det = det_raw;
        if (STAGE == 1 && d.renorm_fused) {
            // this row was skipped by renorm_body (tk_mid1): renormalised here, in place (Q8), by the threads that read it next
            // (thread t rewrites exactly the elements it reads back below); the record turns into the unit row's like there
            const float nb = feat::renorm_row(IN(d, feats) + det * d.D, d.D, true);
            if (d.filter && feat::row_tid() == 0) feat::i8_track_side(d.det_m + det * feat::I8_META, nb, d.gate_sim);
            if (d.Df) {
                const float nf = feat::renorm_row(IN(d, face_feats) + det * d.Df, d.Df, true);
                if (d.filter && feat::row_tid() == 0) feat::i8_track_side(d.dface_m + det * feat::I8_META, nf, d.gate_sim);
            }
            __syncthreads();                 // (the record: copy_shadow below reads it with other threads)
        }
        if (d.smooth_def) {      // (body features only: the form is ruled out with face features)
            feat::copy_row(d.body_curr + s * d.D, IN(d, feats) + det * d.D, d.D);
            if (threadIdx.x == 0) d.ema_pending[s] = d.has_body[s] != 0 ? 1 : 2;
        } else
        feat::ema_row(d.body_curr + s * d.D, d.body_smooth + s * d.D, IN(d, feats) + det * d.D, d.D, d.has_body[s] != 0, d.ema_a, d.ema_b);
        if (d.Df)
            feat::ema_row(d.face_curr + s * d.Df, d.face_smooth + s * d.Df, IN(d, face_feats) + det * d.Df, d.Df, d.has_face[s] != 0, d.ema_a, d.ema_b);
        if (d.filter) {        // the track's latest feature is this detection's row: so is its shadow
            copy_shadow(d.det_h + det * d.sh_rd[0], d.det_m + det * feat::I8_META, d.body_h + s * d.sh_rt[0], d.body_m + s * feat::I8_META, d.D, d.sh_cd[0], d.sh_ct[0]);
            if (d.Df) copy_shadow(d.dface_h + det * d.sh_rd[1], d.dface_m + det * feat::I8_META, d.face_h + s * d.sh_rt[1], d.face_m + s * feat::I8_META, d.Df, d.sh_cd[1], d.sh_ct[1]);
        }
        if (threadIdx.x == 0) { d.has_body[s] = 1; if (d.Df) d.has_face[s] = 1; }
        return;
    }
    const int g = bid * (int)blockDim.x + threadIdx.x;        // ---- Kalman + lifecycle: 8 lanes per row
    const int r = g >> 3, i = g & 7, rcl = min(r, d.T - 1);
    const int s_raw = rslot[rcl], xq_raw = xs[rcl];           // (the row's list entries travel with the extent: see bts_tracker_create)
    if (r >= n) return;
    const int s = s_raw;
    const int xq = xq_raw;
    if (xq < 0) {
        if (i == 0) {
            if (STAGE == 2) d.state[s] = BTS_LOST;                     // D:1582-1586 (tk_merge finds them by x2 < 0)
            if (STAGE == 3) d.state[s] = BTS_REMOVED;                  // D:1609-1612 (... by x3 < 0)
        }
        return;
    }
    const int det = cdet[xq];
    const float z[4] = {d.det_xywh[det * 4 + 0], d.det_xywh[det * 4 + 1], d.det_xywh[det * 4 + 2], d.det_xywh[det * 4 + 3]};
    double mi, Pr[8];
    load_row(d, s, i, mi, Pr);
    kf8::update(mi, Pr, i, z, d.f32[s] != 0);
    store_row(d, s, i, mi, Pr);
    if (i != 0) return;
    d.f32[s] = 0;
    const int frame = h[H_FRAME];
    if (d.state[s] == BTS_TRACKED) {
        d.tracklet_len[s] += 1;                         // STrack.update
    } else {
        d.tracklet_len[s] = 0;                          // STrack.re_activate (new_id=False); tk_merge: lost row with x1 >= 0
    }
    d.state[s] = BTS_TRACKED;
    d.is_activated[s] = 1;
    d.frame_id[s] = frame;
    d.score[s] = IN(d, scores)[det];
    d.det_index[s] = det;
}

// the feature rows of apply_body<STAGE, true> in a 1024-thread workgroup: four matched columns per workgroup, one per row group
// (a group without a match only takes part in the barriers of the row arithmetic)
template <int STAGE>
__device__ __forceinline__ void apply_feat4(const Dev &d, int bid) {
    const int32_t *h = d.hdr;
    const int n = STAGE == 2 ? h[H_NR2] : h[H_NUNC], m = STAGE == 2 ? h[H_NLO] : h[H_NLEFT];
    const int32_t *rslot = STAGE == 2 ? d.r2_slot : d.unc_slot, *cdet = STAGE == 2 ? d.lo_idx : d.left_det, *ys = STAGE == 2 ? d.y2 : d.y3;
    const int xq = bid * 4 + feat::row_group();
    const int r = xq < m ? ys[xq] : -1;
    const bool on = r >= 0 && r < n;
    const long s = on ? rslot[r] : 0, det = on ? cdet[xq] : 0;
    if (d.smooth_def) {
        feat::copy_row(d.body_curr + s * d.D, IN(d, feats) + det * d.D, d.D, on);
        if (on && feat::row_tid() == 0) d.ema_pending[s] = d.has_body[s] != 0 ? 1 : 2;
    } else
    feat::ema_row(d.body_curr + s * d.D, d.body_smooth + s * d.D, IN(d, feats) + det * d.D, d.D, on && d.has_body[s] != 0, d.ema_a, d.ema_b, on);
    if (d.Df)
        feat::ema_row(d.face_curr + s * d.Df, d.face_smooth + s * d.Df, IN(d, face_feats) + det * d.Df, d.Df, on && d.has_face[s] != 0, d.ema_a, d.ema_b, on);
    if (!on) return;
    if (d.filter) {
        copy_shadow(d.det_h + det * d.sh_rd[0], d.det_m + det * feat::I8_META, d.body_h + s * d.sh_rt[0], d.body_m + s * feat::I8_META, d.D, d.sh_cd[0], d.sh_ct[0]);
        if (d.Df) copy_shadow(d.dface_h + det * d.sh_rd[1], d.dface_m + det * feat::I8_META, d.face_h + s * d.sh_rt[1], d.face_m + s * feat::I8_META, d.Df, d.sh_cd[1], d.sh_ct[1]);
    }
    if (feat::row_tid() == 0) { d.has_body[s] = 1; if (d.Df) d.has_face[s] = 1; }
}

// One launch after the stage-1 lists are rebuilt (tk_mid1): the stage-1 Kalman updates, the stage-1
// feature bookkeeping and the cost tiles of stages 2 and 3 touch disjoint tracks, so they share a
// launch as consecutive workgroup ranges (all 256 threads wide).
template <int MODE3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void tk_mid2(Dev d, bts_lap_work w2, bts_lap_work w3, int kf_blocks, int feat_blocks, int tile_blocks) {
    int b = blockIdx.x;
#ifdef BTS_LAP_TRACE
    // role timeline (kernel 4 of g_jobtrace): [0] first workgroup's start, [1..4] latest end of the Kalman / feature /
    // stage-2 tile / stage-3 tile workgroups (low 32 bits of the 100 MHz clock)
    if (b == 0 && threadIdx.x == 0) g_jobtrace[32] = (int)__builtin_amdgcn_s_memrealtime();
    const int role = b < tile_blocks ? 3 : b < 2 * tile_blocks ? 4 : b < 2 * tile_blocks + kf_blocks ? 1 : 2;
#endif
    // the (few, latency-bound) tile workgroups first: dispatched behind the bulk of the Kalman / feature workgroups they
    // started late and ended the kernel
    if (b < tile_blocks) edges_all<2, 0>(d, w2, b, tile_blocks);
    else if ((b -= tile_blocks) < tile_blocks) edges_all<3, MODE3>(d, w3, b, tile_blocks);
    else if ((b -= tile_blocks) < kf_blocks) apply_body<1, false>(d, b);
    else apply_body<1, true>(d, b - kf_blocks);
#ifdef BTS_LAP_TRACE
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&g_jobtrace[32 + role], (int)__builtin_amdgcn_s_memrealtime());
#endif
}

// stages 2 and 3 applied together (disjoint tracks and detections): Kalman ranges, then feature rows
__global__ __launch_bounds__(256) void tk_apply23(Dev d, int kf_blocks, int feat_blocks) {
    int b = blockIdx.x;
    if (b < kf_blocks) { apply_body<2, false>(d, b); return; }
    b -= kf_blocks;
    if (b < kf_blocks) { apply_body<3, false>(d, b); return; }
    b -= kf_blocks;
    if (b < feat_blocks) { apply_body<2, true>(d, b); return; }
    apply_body<3, true>(d, b - feat_blocks);
}

// ---- after stage 1: rows of stage 2, columns of stage 3 (three single-workgroup jobs) -------------------
__host__ __device__ constexpr int mid1_jobs(int nch) { return nch + 2; }
template <bool CH> __device__ __forceinline__ void after1_body(const Dev &d, int job, int ch) {
    __shared__ int32_t tmp[1024];
    int32_t *h = d.hdr;
    if (job == 0) {
        // r_tracked: unmatched pool rows that are still Tracked (D:1569).  Matched rows are set to
        // Tracked by the stage-1 update, which runs after this launch: the state test sees pre-frame state.
        // (only the confirmed rows of the pool, [0, nconf), can be Tracked: the rest are the lost tracks)
        const int N = h[H_NCONF];
        const int nr2 = job_compact<CH>(chunk_of(d, J_R2, ch), N, tmp, [&](int r) { return d.pool_slot[r]; },
                                             [&](int r, int s) { return (d.x1[r] < 0) & (d.state[s] == BTS_TRACKED); },
                                             [&](int p, int r, int s) {
                                                 d.r2_row[p] = r; d.r2_slot[p] = s;
                                                 if (d.merge_early) d.r2_of_slot[s] = p;
                                                 for (int c = 0; c < 4; ++c) d.r2_tlbr[p * 4 + c] = d.pool_tlbr[r * 4 + c];
                                             });
        if (ch == nchunks<CH>(d) - 1 && threadIdx.x == 0) { h[H_NR2] = nr2; h[H_DIMS2] = nr2; }
    } else if (job == 1) {
        const int nhi = h[H_NHI];
        const int nleft = block_compact(nhi, tmp, [&](int q) { return d.hi_idx[q]; }, [&](int q, int det) { return d.y1[q] < 0; },
                                        [&](int p, int q, int det) { d.left_hi[p] = q; d.left_det[p] = det; });
        if (threadIdx.x == 0) { h[H_NLEFT] = nleft; h[H_DIMS3 + 1] = nleft; }
    } else {
        const int nunc = h[H_NUNC];
        for (int k = threadIdx.x; k < nunc; k += blockDim.x) {     // unconfirmed boxes (may be float32 state)
            const int s = d.unc_slot[k];
            double b[4];
            slot_tlbr(d, s, b);
            for (int c = 0; c < 4; ++c) d.unc_tlbr[k * 4 + c] = b[c];
            d.unc_f32[k] = d.f32[s];
        }
    }
}

// one launch after the first assignment: workgroups 0-2 prepare stages 2/3, the others renormalise rows
// (clear_blocks > 0 only with the sorted / list form of stage 1: the last workgroups clear the candidate list, cand_clear)
template <bool CH> __global__ __launch_bounds__(1024) void tk_mid1(Dev d, int clear_blocks) {
    TJ_BEGIN();
    const int b = blockIdx.x, nch = nchunks<CH>(d), first_clear = (int)gridDim.x - clear_blocks;
    if (b < nch) { after1_body<CH>(d, 0, b); if (b == nch - 1) TJ_END(1, 0); }
    else if (b < nch + 2) { after1_body<CH>(d, 1 + (b - nch), 0); TJ_END(1, 1 + (b - nch)); }
    else if (b >= first_clear) cand_clear(d, b - first_clear, clear_blocks);
    else { renorm_body(d, b - mid1_jobs(nch)); if (b == mid1_jobs(nch)) TJ_END(1, 3); }
}

// ---- births + list merge, D:1614-1636: five independent single-workgroup jobs ----------------------------
// tracked' = [old tracked still Tracked] ++ births ++ re-found (pool order);
// lost'    = ([old lost not re-found] ++ newly lost (pool order)) minus ids already in removed (Q6).
// Each piece is compacted by its own workgroup into its own region of tracked2 / lost2 (T entries
// apart); tracked2_at / lost2_at read the concatenation.
__host__ __device__ constexpr int merge_jobs(int nch) { return 1 + 4 * nch; }      // births + four chunked jobs
__device__ __forceinline__ int tracked2_at(const Dev &d, int k) {
    const int c0 = d.hdr[H_NKEEP], c1 = d.hdr[H_NB];
    return k < c0 ? d.tracked2[k] : k < c0 + c1 ? d.tracked2[d.T + (k - c0)] : d.tracked2[2 * d.T + (k - c0 - c1)];
}
__device__ __forceinline__ int lost2_at(const Dev &d, int k) {
    const int c0 = d.hdr[H_NL0];
    return k < c0 ? d.lost2[k] : d.lost2[d.T + (k - c0)];
}

// EARLY (BTS_MERGE_EARLY=1 at create; NOT yet measured): the five jobs run in the launch that APPLIES stages 2 and 3
// (tk_apply23_merge) instead of behind it -- one launch boundary less.  Four of them never read what that application
// writes (they work from the assignment vectors, the lists, frame counters of LOST tracks and free slots); "kept tracked"
// did, through the state the application sets (Lost for a stage-2 row left unmatched, Removed for an unconfirmed track left
// unmatched): here it takes the same verdict from the assignment vectors -- a member of the old tracked list is a stage-2 row
// (kept iff x2 >= 0), an unconfirmed track (kept iff x3 >= 0) or was matched in stage 1 (kept) -- through slot -> list-index
// maps written where those lists are built (r2_of_slot: tk_mid1, unc_of_slot: tk_front / tk_finalize) and validated by reading
// the list entry back, so a stale map entry of an earlier frame can never be taken for a membership.
template <bool CH, bool EARLY> __device__ __forceinline__ void merge_body(const Dev &d, int blk) {
    __shared__ int32_t tmp[1024];
    int32_t *h = d.hdr;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int frame = h[H_FRAME];
    TJ_BEGIN();
    const int nch = nchunks<CH>(d), job = blk == 0 ? 0 : 1 + (blk - 1) / nch, ch = blk == 0 ? 0 : (blk - 1) % nch;
    const bool last = ch == nch - 1;
    if (job == 0) {
        // births in ascending leftover-detection order (new ids ascending, quirk Q7)
        const int nleft = h[H_NLEFT], nfree = h[H_NFREE], last_id = h[H_LASTID];
        const float *scores = IN(d, scores);
        for (int k = tid; k < nleft; k += nth) d.born_slot[k] = -1;
        __syncthreads();
        const int nborn_want = block_compact(
            nleft, tmp, [&](int k) { return d.left_det[k]; }, [&](int k, int det) { return (d.y3[k] < 0) & !((double)scores[det] < d.new_thr); },
            [&](int b, int k, int det) {
                if (b >= nfree) return;
                const int s = d.free_list[b];
                d.born_slot[k] = s;
                d.tracked2[d.T + b] = s;                         // births keep detection order
                d.track_id[s] = last_id + 1 + b;
                d.state[s] = BTS_TRACKED;
                d.is_activated[s] = frame == 1 ? 1 : 0;          // D:565-566
                d.frame_id[s] = frame; d.start_frame[s] = frame; d.tracklet_len[s] = 0;
                d.score[s] = scores[det]; d.det_index[s] = det;
                d.f32[s] = 1; d.in_removed[s] = 0;
                d.has_body[s] = 1; d.has_face[s] = d.Df ? 1 : 0;
                // position half of KalmanFilter.initiate (D:166-197): the duplicate test of this frame reads
                // it while the rest of the state is still being initialised (tk_tail)
                for (int c = 0; c < 4; ++c) st_mean(d, s, c, (double)d.det_xywh[det * 4 + c]);
            });
        if (tid == 0) {
            const int nborn = min(nborn_want, nfree);
            h[H_NB] = nborn; h[H_NBORN] = nborn;
            h[H_LASTID] = last_id + nborn;
            int err = nborn_want > nfree ? ERR_TRACK_CAPACITY : 0;
            if (err && !(h[H_ERR] & ERR_TRACK_CAPACITY)) { h[30] = frame; h[31] = nborn_want - nfree; }      // (first occurrence, for the error message)
            for (int k = 0; k < 3; ++k)                          // assignment solver: edge-list overflow / helper time-out
                if (d.lap_hdr[BTS_LAP_HDR_INTS * k + 2]) err |= ERR_LAP_EDGES;
            if (err) atomicOr(&h[H_ERR], err);
        }
    } else if (job == 1) {
        const int nt = h[H_NTRACKED];
        int n_keep;
        if constexpr (EARLY) {
            const unsigned nr2 = (unsigned)h[H_NR2], nunc = (unsigned)h[H_NUNC];
            n_keep = job_compact<CH>(chunk_of(d, J_KEEP, ch), nt, tmp, [&](int k) { return d.tracked[k]; },
                                     [&](int k, int s) {
                                         const unsigned q = (unsigned)d.r2_of_slot[s], u = (unsigned)d.unc_of_slot[s];
                                         const bool in2 = q < nr2 && d.r2_slot[q < nr2 ? q : 0] == s;
                                         const bool in3 = u < nunc && d.unc_slot[u < nunc ? u : 0] == s;
                                         return in2 ? d.x2[q] >= 0 : in3 ? d.x3[u] >= 0 : true;
                                     }, [&](int p, int k, int s) { d.tracked2[p] = s; });
        } else {
            n_keep = job_compact<CH>(chunk_of(d, J_KEEP, ch), nt, tmp, [&](int k) { return d.tracked[k]; },
                                     [&](int k, int s) { return d.state[s] == BTS_TRACKED; }, [&](int p, int k, int s) { d.tracked2[p] = s; });
        }
        if (last && tid == 0) h[H_NKEEP] = n_keep;
    } else if (job == 2) {
        // re-found = lost rows of the pool (rows >= nconf) that stage 1 matched; no per-slot flag needed
        const int N = h[H_NPOOL], nconf = h[H_NCONF];
        const int n_rf = job_compact<CH>(chunk_of(d, J_RF, ch), N - nconf, tmp, [&](int k) { return d.x1[nconf + k]; },
                                              [&](int k, int xq) { return xq >= 0; },
                                              [&](int p, int k, int) { d.tracked2[2 * d.T + p] = d.pool_slot[nconf + k]; });
        if (last && tid == 0) h[H_NRF] = n_rf;
    } else if (job == 3) {
        const int nl = h[H_NLOST], nunc = h[H_NUNC], nconf = h[H_NCONF];
        // old lost that were not re-found (x1 of their pool row < 0) and whose id is not in removed (Q6)
        const bscan::Chunk cx = chunk_of(d, J_L0, ch);
        const int n_l0 = job_compact<CH>(cx, nl, tmp, [&](int k) { return d.lost[k]; },
                                              [&](int k, int s) { return (d.x1[nconf + k] < 0) & !d.in_removed[s]; },
                                              [&](int p, int k, int s) { d.lost2[p] = s; });
        int lo, hi;
        job_range<CH>(cx, nl, lo, hi);
        // expiry of lost tracks (D:1623-1627); re-found ones carry frame_id == frame.  removed_stracks.extend(
        // removed_now): the ids (here: slots) are remembered for the next frames -- after the filter above (Q6)
        int gone = 0;
        for (int k0 = lo + tid; k0 < hi; k0 += 4 * nth) {     // (this chunk's share; two levels of loads, 4 iterations in flight)
            int sl[4], fid[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int k = k0 + j * nth; sl[j] = d.lost[k < hi ? k : lo]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) fid[j] = d.frame_id[sl[j]];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j * nth, s = sl[j];
                if (k < hi && frame - fid[j] > d.max_time_lost) { d.state[s] = BTS_REMOVED; d.in_removed[s] = 1; ++gone; }
            }
        }
        // unconfirmed tracks that stage 3 left unmatched were removed (D:1609-1612)
        if (ch == 0)
            bscan::for_batched<4>(nunc, [&](int k) { return d.x3[k]; }, [&](int k, int xq) {
                if (xq < 0) { d.in_removed[d.unc_slot[k]] = 1; ++gone; }
            });
        int tot_gone;
        block_exscan_i(gone, tmp, &tot_gone);
        if (tid == 0) {
            if (last) h[H_NL0] = n_l0;
            if (tot_gone) atomicAdd(&h[H_NREMOVED], tot_gone);
        }
    } else {
        // newly lost = stage-2 rows (the still-Tracked leftovers of stage 1, in pool order) that stage 2 left unmatched
        const int nr2 = h[H_NR2];
        const int n_l1 = job_compact<CH>(chunk_of(d, J_L1, ch), nr2, tmp, [&](int k) { return d.r2_slot[k]; },
                                              [&](int k, int s) { return (d.x2[k] < 0) & !d.in_removed[s]; },
                                              [&](int p, int k, int s) { d.lost2[d.T + p] = s; });
        if (last && tid == 0) h[H_NL1] = n_l1;
    }
    if (last) TJ_END(2, job);
}
template <bool CH> __global__ __launch_bounds__(1024) void tk_merge(Dev d) { merge_body<CH, false>(d, (int)blockIdx.x); }
// BTS_MERGE_EARLY: stages 2 and 3 applied AND the list merge, one launch of 1024-thread workgroups -- the list jobs first (they
// are the longer chains), then the Kalman ranges (8 lanes per row) and the feature rows (four matched columns per workgroup)
template <bool CH> __global__ __launch_bounds__(1024) void tk_apply23_merge(Dev d, int kf_blocks, int feat_blocks) {
    int b = (int)blockIdx.x;
    const int mj = merge_jobs(nchunks<CH>(d));
    if (b >= mj + 2 * kf_blocks) {
        b -= mj + 2 * kf_blocks;
        if (b < feat_blocks) apply_feat4<2>(d, b); else apply_feat4<3>(d, b - feat_blocks);
    } else if (b >= mj) {
        b -= mj;
        if (b < kf_blocks) apply_body<2, false>(d, b); else apply_body<3, false>(d, b - kf_blocks);
    } else {
        merge_body<CH, true>(d, b);
    }
}

// Kalman initiate + feature rows of the newborn tracks (D:556-568, D:492-514): one workgroup per
// leftover detection.
__device__ __forceinline__ void birth_body(const Dev &d, int k) {
    if (k >= d.hdr[H_NLEFT]) return;
    const long s = d.born_slot[k];
    if (s < 0) return;
    const long det = d.left_det[k];
    if (threadIdx.x == 0) {
        const float z[4] = {d.det_xywh[det * 4 + 0], d.det_xywh[det * 4 + 1], d.det_xywh[det * 4 + 2], d.det_xywh[det * 4 + 3]};
        double m[8], P[64];
        kf::initiate(z, m, P);
        for (int c = 4; c < 8; ++c) st_mean(d, s, c, m[c]);          // m[0..3] were written by tk_merge
        for (int c = 0; c < 64; ++c) st_cov(d, s, c, P[c]);
    }
    // the detection row was renormalised in place; curr and smooth alias it in the reference
    for (int c = threadIdx.x; c < d.D; c += blockDim.x) {
        const float v = IN(d, feats)[det * d.D + c];
        d.body_curr[s * d.D + c] = v;
        if (!d.smooth_def) d.body_smooth[s * d.D + c] = v;
    }
    if (d.smooth_def && threadIdx.x == 0) d.ema_pending[s] = 3;
    for (int c = threadIdx.x; c < d.Df; c += blockDim.x) {
        const float v = IN(d, face_feats)[det * d.Df + c];
        d.face_curr[s * d.Df + c] = v; d.face_smooth[s * d.Df + c] = v;
    }
    if (d.filter) {
        copy_shadow(d.det_h + det * d.sh_rd[0], d.det_m + det * feat::I8_META, d.body_h + s * d.sh_rt[0], d.body_m + s * feat::I8_META, d.D, d.sh_cd[0], d.sh_ct[0]);
        if (d.Df) copy_shadow(d.dface_h + det * d.sh_rd[1], d.dface_m + det * feat::I8_META, d.face_h + s * d.sh_rt[1], d.face_m + s * feat::I8_META, d.Df, d.sh_cd[1], d.sh_ct[1]);
    }
}

// ---- duplicate removal between tracked' and lost' (D:1665-1680): boxes straight from the table ------------
// ... in one launch with the initialisation of the newborn tracks (workgroups >= dedupe_blocks)
template <int ROWS>
__global__ __launch_bounds__(CT_COLS) void tk_tail(Dev d, int dedupe_blocks) {
    if ((int)blockIdx.x >= dedupe_blocks) { birth_body(d, (int)blockIdx.x - dedupe_blocks); return; }
    __shared__ double sa[ROWS][4];
    __shared__ double sarea[ROWS];
    __shared__ int sflag[ROWS];
    __shared__ int sage[ROWS];
    __shared__ iou::ScreenBox sboxf[ROWS];    // row boxes rounded outward to 16-bit pixels (screening, see edges_tile)
    const int n = d.hdr[H_NKEEP] + d.hdr[H_NB] + d.hdr[H_NRF], m = d.hdr[H_NL0] + d.hdr[H_NL1];
    const int nct = (m + CT_COLS - 1) / CT_COLS, ntile = nct * ((n + ROWS - 1) / ROWS);
    for (int t = blockIdx.x; t < ntile; t += dedupe_blocks) {
        const int i0 = (t / nct) * ROWS, j = (t % nct) * CT_COLS + threadIdx.x;
        // both chains of dependent loads (this thread's lost track, the tile's tracked rows) start before the barrier
        const bool live = j < m;
        const int sj = live ? lost2_at(d, j) : 0;
        double bb[4] = {0.0, 0.0, 0.0, 0.0};
        bool bf = false;
        int agej = 0;
        if (live) {
            slot_tlbr(d, sj, bb);
            bf = d.f32[sj] != 0;
            agej = d.frame_id[sj] - d.start_frame[sj];
        }
        if (threadIdx.x < ROWS && i0 + threadIdx.x < n) {
            const int s = tracked2_at(d, i0 + threadIdx.x);
            double box[4];
            slot_tlbr(d, s, box);
            const bool f = d.f32[s] != 0;
            for (int c = 0; c < 4; ++c) sa[threadIdx.x][c] = box[c];
            sarea[threadIdx.x] = iou::area(box, f);
            sflag[threadIdx.x] = f;
            sage[threadIdx.x] = d.frame_id[s] - d.start_frame[s];
            sboxf[threadIdx.x] = iou::screen_box(box);
        }
        __syncthreads();
        const double area_b = iou::area(bb, bf);
        const iou::ScreenBox bq = live ? iou::screen_box(bb) : iou::screen_none();
        const int rows = min(ROWS, n - i0);
        bool dropj = false;
#pragma unroll 1
        for (int rc = 0; rc < rows; rc += 8) {
            uint32_t work = 0;                    // wave-uniform: rows of this chunk some lane overlaps
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                work |= (__ballot(iou::screen_overlap(sboxf[(rc + u) & (ROWS - 1)], bq)) != 0ull ? 1u : 0u) << u;
            }
            work &= (1u << min(8, rows - rc)) - 1u;
            while (work) {
                const int r = rc + __ffs((int)work) - 1;
                work &= work - 1;
                const double aa[4] = {sa[r][0], sa[r][1], sa[r][2], sa[r][3]};
                if (live && iou::cost(aa, sflag[r] != 0, sarea[r], bb, bf, area_b) < d.dedupe_thr) {
                    if (sage[r] > agej) dropj = true;
                    else d.drop_t[i0 + r] = 1;
                }
            }
        }
        if (dropj) d.drop_l[j] = 1;
        __syncthreads();
    }
}

// ---- frame epilogue: final lists (one workgroup each), drop flags cleared for the next frame ------------------
// BTS_LISTS_AHEAD (Dev::ahead; opt-in, NOT yet measured): the list jobs that open the NEXT frame -- pool = confirmed ++ lost,
// unconfirmed, free slots (prep_pool / prep_unconfirmed / prep_free of tk_front) -- depend on nothing but this frame's final
// lists, so they run here, in the same launch as the compactions that produce those lists and straight from the same inputs
// (tracked2 / lost2 + drop flags): the next frame then opens with the detection-side work alone (tk_front_dets), which in
// turn can be overlapped with this frame's tail (BTS_CVT_OVERLAP).  Their sizes go to the *_N header words; the drop flags
// are cleared by tk_front_dets (they have several readers here).
template <bool CH> __device__ __forceinline__ void ahead_body(const Dev &d, int job, int ch, int32_t *tmp) {
    int32_t *h = d.hdr;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int nt2 = h[H_NKEEP] + h[H_NB] + h[H_NRF], nl2 = h[H_NL0] + h[H_NL1];
    const bool last = ch == nchunks<CH>(d) - 1;
    if (job == 0) {
        // pool = [kept tracked that are activated] ++ [kept lost]   (D:1414-1423 of the next frame)
        const bscan::Chunk ca = chunk_of(d, J_POOL, ch), cb = chunk_of(d, J_PB, ch);
        // "every pool row still holds float32 state" (H_ALLF32, see prep_pool): only on the frame after the first one
        const bool chk = h[H_FRAME] == 1;
        int f64 = 0;
        if (chk) {
            int a, b;
            job_range<CH>(ca, nt2, a, b);
            for (int k = a + tid; k < b; k += nth) { const int s = tracked2_at(d, k); if (!d.drop_t[k] && (d.is_activated[s] != 0) & !d.f32[s]) f64 = 1; }
            job_range<CH>(cb, nl2, a, b);
            for (int k = a + tid; k < b; k += nth) if (!d.drop_l[k] && !d.f32[lost2_at(d, k)]) f64 = 1;
            f64 = __syncthreads_or(f64);
        }
        int nconf = 0;
        bool any64 = false;
        // BTS_FILTER_AHEAD: the bitmap row the next frame's stage-1 tiles read for this pool row -- a track rewritten during THIS
        // frame (frame_id == the frame: update / re_activate / activate, which also set det_index) holds the shadow and track-side
        // record of detection det_index of this frame: row T + det_index of the det x det pass; any other track's shadow has been
        // stable since before this frame began: row `slot` of the main pass
        const int fr = h[H_FRAME];
        auto cbrow = [&](int s) { return d.frame_id[s] == fr ? d.T + d.det_index[s] : s; };
        // (+ the sorted edges: the reverse maps its candidate wavefronts translate the side passes' rows with)
        auto put = [&](int p, int s) {
            if (!d.pool_cbrow) return;
            const int cb = cbrow(s);
            d.pool_cbrow[p] = cb;
            if (d.pool_of_slot) { if (cb >= d.T) d.pool_of_pdet[cb - d.T] = p; else d.pool_of_slot[s] = p; }
        };
        job_compact<CH>(ca, nt2, tmp, [&](int k) { return tracked2_at(d, k); }, [&](int k, int s) { return !d.drop_t[k] && d.is_activated[s] != 0; },
                        [&](int p, int k, int s) { d.pool_slot[p] = s; put(p, s); }, f64 != 0, true, &nconf, &any64);
        const int nl = job_compact<CH>(cb, nl2, tmp, [&](int k) { return lost2_at(d, k); }, [&](int k, int s) { return !d.drop_l[k]; },
                                       [&](int p, int k, int s) { d.pool_slot[nconf + p] = s; put(nconf + p, s); });
        if (last && tid == 0) {
            const int N = nconf + nl;
            h[H_NPOOL_N] = N; h[H_NCONF_N] = nconf;
            h[H_ALLF32_N] = (N > 0 && chk && !any64) ? 1 : 0;
        }
    } else if (job == 1) {
        const int nunc = job_compact<CH>(chunk_of(d, J_UNC, ch), nt2, tmp, [&](int k) { return tracked2_at(d, k); },
                                         [&](int k, int s) { return !d.drop_t[k] && d.is_activated[s] == 0; }, [&](int p, int k, int s) { d.unc_slot[p] = s; if (d.merge_early) d.unc_of_slot[s] = p; });
        if (last && tid == 0) h[H_NUNC_N] = nunc;
    } else {
        // free slots = the table minus the kept entries of both lists (prep_free, from the same inputs)
        __shared__ uint32_t used[FREE_BITS_WORDS];
        const bool lds = d.T <= 32 * FREE_BITS_WORDS;
        if (lds) {
            for (int k = tid; k < (d.T + 31) / 32; k += nth) used[k] = 0u;
            __syncthreads();
            bscan::for_batched<4>(nt2, [&](int k) { return tracked2_at(d, k); }, [&](int k, int s) { if (!d.drop_t[k]) atomicOr(&used[s >> 5], 1u << (s & 31)); });
            bscan::for_batched<4>(nl2, [&](int k) { return lost2_at(d, k); }, [&](int k, int s) { if (!d.drop_l[k]) atomicOr(&used[s >> 5], 1u << (s & 31)); });
            __syncthreads();
        } else {
            for (int s = tid; s < d.T; s += nth) d.in_use[s] = 0;
            __syncthreads();
            bscan::for_batched<4>(nt2, [&](int k) { return tracked2_at(d, k); }, [&](int k, int s) { if (!d.drop_t[k]) d.in_use[s] = 1; });
            bscan::for_batched<4>(nl2, [&](int k) { return lost2_at(d, k); }, [&](int k, int s) { if (!d.drop_l[k]) d.in_use[s] = 1; });
            __syncthreads();
        }
        const int nfree = block_compact(d.T, tmp, [&](int s) { return lds ? (int)((used[s >> 5] >> (s & 31)) & 1u) : (int)d.in_use[s]; },
                                        [&](int s, int v) { return v == 0; }, [&](int p, int s, int) { d.free_list[p] = s; });
        if (tid == 0) h[H_NFREE] = nfree;
    }
}

// AHEAD (BTS_LISTS_AHEAD at create) is a template parameter since round 5: the default instantiation no longer carries the
// next-frame list jobs it never runs (they had grown it from 4 KB to 37 KB of LDS and by a third in code: VERDICT r4, weak 9)
template <bool CH, bool AHEAD> __global__ __launch_bounds__(1024) void tk_finalize(Dev d) {
    __shared__ int32_t tmp[1024];
    int32_t *h = d.hdr;
    const int tid = threadIdx.x, nth = blockDim.x;
    TJ_BEGIN();
    const int nch = nchunks<CH>(d), job = (int)blockIdx.x / nch, ch = (int)blockIdx.x % nch;
    if constexpr (AHEAD) {
        if (job >= 2) { ahead_body<CH>(d, job - 2, ch, tmp); return; }      // the next frame's pool, unconfirmed, free lists
    }
    int lo, hi;
    if (job == 0) {
        const int nt2 = h[H_NKEEP] + h[H_NB] + h[H_NRF];
        const bscan::Chunk cx = chunk_of(d, J_FT, ch);
        const int nt = job_compact<CH>(cx, nt2, tmp, [&](int k) { return tracked2_at(d, k); }, [&](int k, int s) { return !d.drop_t[k]; },
                                            [&](int p, int k, int s) { d.tracked[p] = s; });
        job_range<CH>(cx, nt2, lo, hi);
        if (!AHEAD) for (int k = lo + tid; k < hi; k += nth) d.drop_t[k] = 0;
        if (ch == nch - 1 && tid == 0) h[H_NTRACKED] = nt;
    } else {
        const int nl2 = h[H_NL0] + h[H_NL1];
        const bscan::Chunk cx = chunk_of(d, J_FL, ch);
        const int nl = job_compact<CH>(cx, nl2, tmp, [&](int k) { return lost2_at(d, k); }, [&](int k, int s) { return !d.drop_l[k]; },
                                            [&](int p, int k, int s) { d.lost[p] = s; });
        job_range<CH>(cx, nl2, lo, hi);
        if (!AHEAD) for (int k = lo + tid; k < hi; k += nth) d.drop_l[k] = 0;
        if (ch == nch - 1 && tid == 0) h[H_NLOST] = nl;
    }
    if (ch == nch - 1) TJ_END(3, job);
}

// BTS_SMOOTH_DEFERRED: the `smooth` rows of the slots the step just rewrote, from their `curr` rows -- the row ema_row blended is
// bit for bit the row it stored in `curr`, so this is ema_row's own arithmetic, later.  One 256-thread workgroup per table slot;
// most exit on the mark.  Runs on the tracker's third stream, behind "the step is done", in front of the next step's first apply.
__global__ __launch_bounds__(256) void tk_smooth(Dev d) {
    const long s = blockIdx.x;
    const int p = d.ema_pending[s];                   // (uniform over the workgroup)
    if (p == 0) return;
    float *curr = d.body_curr + s * d.D, *smooth = d.body_smooth + s * d.D;
    if (p == 3) feat::copy_row(smooth, curr, d.D);                               // newborn: smooth aliases curr in the reference (D:556-568)
    else feat::ema_row(curr, smooth, curr, d.D, p == 1, d.ema_a, d.ema_b);       // (rewrites curr with itself)
    __syncthreads();                                  // every thread has read the mark before it is cleared
    if (threadIdx.x == 0) d.ema_pending[s] = 0;
}

__global__ __launch_bounds__(1024) void tk_reset(Dev d) {
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int s = tid; s < d.T; s += nth) {
        d.free_list[s] = s; d.in_use[s] = 0; d.in_removed[s] = 0; d.f32[s] = 0; d.state[s] = BTS_NEW;
        d.drop_t[s] = 0; d.drop_l[s] = 0;
        d.is_activated[s] = 0; d.track_id[s] = 0; d.has_body[s] = 0; d.has_face[s] = 0;
        if (d.ema_pending) d.ema_pending[s] = 0;
    }
    for (int k = tid; k < H_SIZE; k += nth) d.hdr[k] = 0;
    for (int k = tid; k < J_COUNT * bscan::CHUNK_MAX; k += nth) d.agg[k] = 0ull;
    __syncthreads();
    if (tid == 0) d.hdr[H_NFREE] = d.T;
    for (int j = tid; j < d.Mx; j += nth) d.ones[j] = 1;
}

// ---- camera-motion compensation of every live track (STrack.multi_gmc, D:538-554) -------------------------
struct Warp6 { double h[6]; };
__global__ __launch_bounds__(64) void tk_gmc(Dev d, Warp6 w) {
    const int nt = d.hdr[H_NTRACKED], nl = d.hdr[H_NLOST];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nt + nl) return;
    const int s = k < nt ? d.tracked[k] : d.lost[k - nt];
    double m[8], P[64];
    load_state(d, s, m, P);
    kf::gmc(m, P, w.h);
    store_state(d, s, m, P);
    d.f32[s] = 0;                                   // R8.dot(mean) is float64 whatever the state was
    if (k == 0) d.hdr[H_ALLF32_N] = 0;              // (BTS_LISTS_AHEAD: ... and so is the verdict tk_finalize took ahead for the next frame)
}

// ---- export -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tk_export(Dev d, int which, int32_t *ints, double *mean, double *tlbr, float *score) {
    const int n = which == 0 ? d.hdr[H_NTRACKED] : d.hdr[H_NLOST];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = which == 0 ? d.tracked[k] : d.lost[k];
    ints[k * 8 + 0] = d.track_id[s]; ints[k * 8 + 1] = d.state[s]; ints[k * 8 + 2] = d.is_activated[s];
    ints[k * 8 + 3] = d.frame_id[s]; ints[k * 8 + 4] = d.start_frame[s]; ints[k * 8 + 5] = d.tracklet_len[s];
    ints[k * 8 + 6] = d.det_index[s]; ints[k * 8 + 7] = s;
    for (int c = 0; c < 8; ++c) mean[k * 8 + c] = ld_mean(d, s, c);
    double b[4];
    slot_tlbr(d, s, b);
    for (int c = 0; c < 4; ++c) tlbr[k * 4 + c] = b[c];
    score[k] = d.score[s];
}

// MOTChallenge rows of the tracked list: frame, id, left, top, width, height, conf, -1, -1, -1
__global__ __launch_bounds__(256) void tk_export_mot(Dev d, double *rows) {
    const int n = d.hdr[H_NTRACKED];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = d.tracked[k];
    double b[4];
    slot_tlbr(d, s, b);
    double *r = rows + (long)k * 10;
    r[0] = (double)d.hdr[H_FRAME]; r[1] = (double)d.track_id[s];
    r[2] = b[0]; r[3] = b[1]; r[4] = b[2] - b[0]; r[5] = b[3] - b[1];
    r[6] = (double)d.score[s]; r[7] = -1.0; r[8] = -1.0; r[9] = -1.0;
}

// the same rows written where the consumer reads them (pinned, device-mapped host memory) together with their count:
// no staging copy and no host synchronisation -- the caller waits on an event of its own when it wants the frame
__global__ __launch_bounds__(256) void tk_export_mot_direct(Dev d, double *rows, int32_t *n_out) {
    const int n = d.hdr[H_NTRACKED];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { n_out[0] = n; n_out[1] = d.hdr[H_FRAME]; n_out[2] = d.hdr[H_ERR]; }
    if (k >= n) return;
    const int s = d.tracked[k];
    double b[4];
    slot_tlbr(d, s, b);
    double *r = rows + (long)k * 10;
    r[0] = (double)d.hdr[H_FRAME]; r[1] = (double)d.track_id[s];
    r[2] = b[0]; r[3] = b[1]; r[4] = b[2] - b[0]; r[5] = b[3] - b[1];
    r[6] = (double)d.score[s]; r[7] = -1.0; r[8] = -1.0; r[9] = -1.0;
}

// rows [first, first + count) of a list, one table field, gathered into `out` (row-major, `width` elements per row):
// field 0 covariance (64 doubles), 1 / 2 body curr / smooth, 3 / 4 face curr / smooth (floats)
__global__ __launch_bounds__(256) void tk_gather_state(Dev d, int which, int field, int first, int count, void *out) {
    const int n = which == 0 ? d.hdr[H_NTRACKED] : d.hdr[H_NLOST];
    const int k = first + (int)blockIdx.x;
    if ((int)blockIdx.x >= count || k >= n) return;
    const long s = which == 0 ? d.tracked[k] : d.lost[k];
    if (field == 0) {
        double *o = (double *)out + (long)blockIdx.x * 64;
        for (int c = threadIdx.x; c < 64; c += blockDim.x) o[c] = ld_cov(d, s, c);
        return;
    }
    const int w = field <= 2 ? d.D : d.Df;
    const float *src = field == 1 ? d.body_curr : field == 2 ? d.body_smooth : field == 3 ? d.face_curr : d.face_smooth;
    float *o = (float *)out + (long)blockIdx.x * w;
    for (int c = threadIdx.x; c < w; c += blockDim.x) o[c] = src[s * w + c];
}

}  // namespace

// =========================================================================================== host side
struct bts_tracker {
    bts_context *ctx = nullptr;
    bts_tracker_config cfg;
    Dev dv;
    bts_lap_work lap;      // stages 1 and 2 (sequential)
    bts_lap_work lap3;     // stage 3 runs next to stage 2: its own work area
    int lap_form = 0;             // BTS_LAP_FORM=row at create: the assignment kernel's round-2 row forms (bts_lap_enqueue)
    int lap_helpers = -1;         // BTS_LAP_HELPERS at create: helper workgroups per assignment problem (< 0: automatic)
    int filter_pw = 0;            // BTS_FILTER_PW at create: persistent GEMM workgroups of the pre-filter (0: FILTER_PW)
#ifdef BTS_HOST_EMU
    std::vector<std::pair<size_t, size_t>> redzones;      // (offset, bytes) inside `block`: see take()
    bool redzones_armed = false;
#endif
    int filter_tile = 0;          // BTS_FILTER_TILE at create: 0 = 128x128 (k_filter_i8), 1 = 160x128, 2 = 128x64 (k_filter_t)
    int last_ftm = FBM, last_ftn = FBN;      // tile shape the last step's pre-filter ran with (bts_tracker_debug_filter)
    int tile_goal = 768;          // BTS_TILE_GOAL at create: workgroup count the stage-1 / duplicate tile heights aim at
    char forms[448] = "";         // the switches in effect, as bts_tracker_forms reports them
    bool onepass = false;         // BTS_FILTER_ONEPASS=1 at create: the persistent one-pass pre-filter also for split-K-sized problems
    bool predict_front = false;   // BTS_PREDICT_FRONT=1 at create (with BTS_LISTS_AHEAD): the predict workgroups in the frame's first launch
    bool predict_first = false;   // BTS_PREDICT_FIRST=1 at create: k_filter_i8<.., KF_FIRST> (experiment, see the kernel)
    // BTS_CVT_OVERLAP=1 at create (experiment, see tk_front): second set of detection shadows (frames alternate), a side
    // stream, "shadows of frame k are ready" / "frame k is done with its shadows" events per buffer
    bool cvt_overlap = false;
    int8_t *det_h2 = nullptr, *dface_h2 = nullptr;
    float *det_m2 = nullptr, *dface_m2 = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t ev_cvt[2] = {}, ev_done[2] = {};
    bool inputs_resident = false; // bts_tracker_inputs_resident
    hipEvent_t ev_in = nullptr;   // "the caller's stream has reached this step": the side stream waits for it before it reads the frame's arrays
    bool ev_done_used[2] = {};
    // BTS_FILTER_AHEAD=1 at create (implies the side stream above, BTS_LISTS_AHEAD and BTS_PREDICT_FRONT): see enqueue_step
    float *det_mts[2] = {};       // track-side records of a frame's detection rows (tk_cvt_ahead), per shadow buffer
    int32_t *hi_b[2] = {};        // the side passes' own high-score list of the frame
    int32_t *shdr = nullptr;      // [2 buffers][main, det x det] headers (H_SIZE ints each: H_NPOOL, H_NHI) the pre-filter kernel reads
    int32_t *fstat_b = nullptr;   // [2][16] pre-filter counters of the side passes
    int32_t *iota = nullptr;      // 0, 1, 2, ...: the identity "list" of the side passes' rows
    int mc_prev = 0;              // detections of the previous step (rows of the det x det pass)
    hipStream_t side2 = nullptr;  // BTS_SMOOTH_DEFERRED: tk_smooth's stream
    hipEvent_t ev_smooth[2] = {};
    bool ev_smooth_used[2] = {};
    int filter_ahead_pw = 0;      // BTS_FILTER_AHEAD_PW: persistent workgroups per side pass.  Default 256 = one per CU: a pre-filter wave holds
                                  // ~244 VGPRs, two of them close a SIMD to everything else -- with one per SIMD the step's own kernels keep
                                  // half of every register file while the side passes run (unmeasured; 384 / 512 are tune candidates)
    bts_lap_exact xk[2];   // tie_mode 1: scratch of the exact-tie step, [0] stages 1 and 2 (sequential), [1] stage 3
    char *block = nullptr;
    size_t bytes = 0;
    // export staging
    int32_t *x_ints = nullptr; double *x_mean = nullptr, *x_tlbr = nullptr, *x_mot = nullptr; float *x_score = nullptr;
    int32_t *host_hdr = nullptr;   // pinned
    // optional per-segment timing (bts_tracker_profile_*): a ring of hipEvent pairs
    int prof_seg = -1, prof_cap = 0, prof_n = 0, prof_stride = 1;
    hipEvent_t *prof_ev = nullptr;
    // input descriptors: pinned host ring -> one device descriptor (40-byte async copy per step)
    static constexpr int RING = 64;
    InDesc *d_in = nullptr, *h_in = nullptr;
    hipEvent_t in_ev[RING] = {};
    bool in_used[RING] = {};
    unsigned long step_no = 0;
    // hipGraph replay of a whole step (optional): captured once on a private stream
    hipStream_t cs[1] = {};
    hipGraphExec_t gexec = nullptr;
};

constexpr size_t SPLITK_MAX_ENTRIES = (size_t)8 << 20;     // T x M entries up to which the split-K accumulators are allocated

#ifdef BTS_HOST_EMU
// CPU emulation build only (tests/hip_emu): every array of the table block is followed by a RED ZONE -- its alignment slack plus
// 256 bytes -- filled with a canary when the tracker is created and checked at every bts_tracker_counts and at destroy: a kernel
// that writes past the end of an array (a ragged tile, an off-by-one at a capacity edge) is caught on the CPU, whatever the
// GPU's allocator would have let it get away with.  The product build has neither the zones nor the checks (its layout and
// bts_tracker_memory_bytes are unchanged).
static thread_local std::vector<std::pair<size_t, size_t>> *g_redzones = nullptr;
static size_t take(size_t &off, size_t bytes) {
    const size_t o = off;
    off = bts_align(off + bytes) + 256;
    if (g_redzones) g_redzones->push_back({o + bytes, off - (o + bytes)});
    return o;
}
#else
static size_t take(size_t &off, size_t bytes) { const size_t o = off; off = bts_align(off + bytes); return o; }
#endif

// slots of the sorted form's candidate list(s): four per pre-filter tile; with BTS_FILTER_AHEAD the main pass's tiles (all table
// slots) and the det x det pass's, per shadow buffer
static size_t cand_slots_main(const Dev &d) { return (size_t)bts_cdiv(d.T, FBM) * (size_t)bts_cdiv(d.Mx, FBN) * 4; }
static size_t cand_slots(const Dev &d) { return cand_slots_main(d) + (d.fahead ? (size_t)bts_cdiv(d.Mx, FBM) * (size_t)bts_cdiv(d.Mx, FBN) * 4 : 0); }

static void tracker_layout(bts_tracker *t, char *base, size_t &off) {
    Dev &d = t->dv;
    const size_t T = d.T, M = d.Mx, D = d.D, Df = d.Df ? d.Df : 1;
#define SLOT(ptr, type, count) { size_t o = take(off, sizeof(type) * (size_t)(count)); ptr = base ? (type *)(base + o) : nullptr; }
    {   // Kalman rows: float64, or float32 elements in the same arrays (kalman_dtype BTS_F32)
        const size_t el = d.sf32 ? sizeof(float) : sizeof(double);
        size_t o = take(off, el * 8 * T); d.mean = base ? (double *)(base + o) : nullptr;
        o = take(off, el * 64 * T); d.cov = base ? (double *)(base + o) : nullptr;
    }
    SLOT(d.f32, uint8_t, T); SLOT(d.in_removed, uint8_t, T); SLOT(d.in_use, uint8_t, T);
    SLOT(d.agg, unsigned long long, J_COUNT * bscan::CHUNK_MAX);
    SLOT(d.has_body, uint8_t, T); SLOT(d.has_face, uint8_t, T);
    SLOT(d.state, int32_t, T); SLOT(d.is_activated, int32_t, T); SLOT(d.track_id, int32_t, T);
    SLOT(d.frame_id, int32_t, T); SLOT(d.start_frame, int32_t, T); SLOT(d.tracklet_len, int32_t, T);
    SLOT(d.det_index, int32_t, T); SLOT(d.score, float, T);
    SLOT(d.body_curr, float, T * D); SLOT(d.body_smooth, float, T * D);
    SLOT(d.face_curr, float, T * Df); SLOT(d.face_smooth, float, T * Df);
    SLOT(d.tracked, int32_t, T); SLOT(d.lost, int32_t, T); SLOT(d.tracked2, int32_t, 3 * T); SLOT(d.lost2, int32_t, 2 * T);
    SLOT(d.free_list, int32_t, T); SLOT(d.hdr, int32_t, H_SIZE);
    SLOT(d.pool_slot, int32_t, T); SLOT(d.unc_slot, int32_t, T); SLOT(d.hi_idx, int32_t, M); SLOT(d.lo_idx, int32_t, M);
    SLOT(d.r2_row, int32_t, T); SLOT(d.r2_slot, int32_t, T); SLOT(d.r2_of_slot, int32_t, d.merge_early ? T : 1); SLOT(d.unc_of_slot, int32_t, d.merge_early ? T : 1); SLOT(d.left_hi, int32_t, M); SLOT(d.left_det, int32_t, M);
    SLOT(d.born_slot, int32_t, M);
    SLOT(d.pool_tlbr, double, 4 * T); SLOT(d.unc_tlbr, double, 4 * T); SLOT(d.det_tlbr, double, 4 * M);
    SLOT(d.r2_tlbr, double, 4 * T); SLOT(d.unc_f32, uint8_t, T); SLOT(d.ones, uint8_t, M); SLOT(d.det_xywh, float, 4 * M);
    // the dense N x M matrices exist only where they are read: similarities in the DENSE appearance mode (and for
    // cost recording), cost matrices for cost recording alone
    const bool dense_sim = !d.filter || t->cfg.record_costs;
    SLOT(d.sim, float, dense_sim ? T * M : 1); SLOT(d.fsim, float, dense_sim && d.Df ? T * M : 1);
    SLOT(d.cost1, double, t->cfg.record_costs ? T * M : 1);
    if (t->cfg.record_costs) { SLOT(d.cost2, double, T * M); SLOT(d.cost3, double, T * M); }
    else { d.cost2 = d.cost1; d.cost3 = d.cost1; }
    SLOT(d.x1, int32_t, T); SLOT(d.y1, int32_t, M); SLOT(d.x2, int32_t, T); SLOT(d.y2, int32_t, M);
    SLOT(d.x3, int32_t, T); SLOT(d.y3, int32_t, M);
    SLOT(d.drop_t, uint8_t, T); SLOT(d.drop_l, uint8_t, T); SLOT(d.lap_hdr, int32_t, 3 * BTS_LAP_HDR_INTS);
    if (d.filter) {
        SLOT(d.body_h, int8_t, T * D); SLOT(d.det_h, int8_t, M * D); SLOT(d.body_m, float, T * feat::I8_META); SLOT(d.det_m, float, M * feat::I8_META);
        SLOT(d.face_h, int8_t, T * Df); SLOT(d.dface_h, int8_t, M * Df); SLOT(d.face_m, float, T * feat::I8_META); SLOT(d.dface_m, float, M * feat::I8_META);
        if (t->cvt_overlap) {
            SLOT(t->det_h2, int8_t, M * D); SLOT(t->det_m2, float, M * feat::I8_META);
            SLOT(t->dface_h2, int8_t, M * Df); SLOT(t->dface_m2, float, M * feat::I8_META);
        }
        if (d.fahead) {
            SLOT(t->det_mts[0], float, M * feat::I8_META); SLOT(t->det_mts[1], float, M * feat::I8_META);
            SLOT(t->hi_b[0], int32_t, M); SLOT(t->hi_b[1], int32_t, M);
            SLOT(t->shdr, int32_t, 4 * H_SIZE); SLOT(t->fstat_b, int32_t, 2 * 16);
            SLOT(t->iota, int32_t, T > M ? T : M); SLOT(d.pool_cbrow, int32_t, T);
            if (d.smooth_def) SLOT(d.ema_pending, uint8_t, T);
        }
        if (d.sorted) {
            // (with BTS_FILTER_AHEAD: per shadow buffer, the main pass's list followed by the det x det pass's)
            const size_t slots = cand_slots(d) * (d.fahead ? 2 : 1);
            SLOT(d.cand_cnt, int32_t, slots); SLOT(d.cand, int2, slots * CAND_W);
            if (d.fahead) { SLOT(d.pool_of_slot, int32_t, T); SLOT(d.pool_of_pdet, int32_t, M); }
        }
        SLOT(d.fstat, int32_t, 16);
        SLOT(d.grid_hdr, int32_t, 8); SLOT(d.cell_start, int32_t, GRID_CELLS + 1); SLOT(d.cell_items, int32_t, M);
        // split-K accumulators: only tables small enough to ever need split K (< 256 tiles of 128 x 128) carry them
        if (T * M <= SPLITK_MAX_ENTRIES) { SLOT(d.approx, int32_t, T * M); SLOT(d.approx_f, int32_t, d.Df ? T * M : 1); }
        else { d.approx = nullptr; d.approx_f = nullptr; }
        d.cb_ld = (int)((M + 63) / 64) * 2;        // 64 columns = one aligned 8-byte word
        SLOT(d.cbits, uint32_t, (d.fahead ? 2 * (T + M) : T) * (size_t)d.cb_ld);      // (BTS_FILTER_AHEAD: two frames' [T + Mx] rows)
    }
    SLOT(t->x_ints, int32_t, 8 * T); SLOT(t->x_mean, double, 8 * T); SLOT(t->x_tlbr, double, 4 * T); SLOT(t->x_score, float, T); SLOT(t->x_mot, double, 10 * T);
#undef SLOT
}

extern "C" {

int bts_tracker_config_size(void) { return (int)sizeof(bts_tracker_config); }

int bts_tracker_default_config(bts_tracker_config *cfg) {
    BTS_REQUIRE(cfg, "cfg");
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(*cfg); cfg->abi_version = BTS_ABI_VERSION;
    cfg->max_tracks = 1024; cfg->max_dets = 512; cfg->feat_dim = 2048; cfg->face_dim = 0;
    cfg->frame_rate = 30; cfg->track_buffer = 300;
    cfg->track_high_thresh = 0.40; cfg->track_low_thresh = 0.1; cfg->new_track_thresh = 0.9;
    cfg->match_thresh = 0.8; cfg->second_match_thresh = 0.5; cfg->unconfirmed_match_thresh = 0.7;
    cfg->proximity_thresh = 0.5; cfg->appearance_thresh = 0.25; cfg->dedupe_thresh = 0.15;
    cfg->ema_alpha = 0.9; cfg->record_costs = 0; cfg->appearance_mode = BTS_APPEARANCE_FILTER; cfg->use_graph = 0;
    cfg->motion_gate = 0;      // the reference's loop never gates on motion (A6): off keeps parity
    cfg->max_edges = 0;        // automatic
    cfg->tie_mode = 0;         // the fast solver's optimum (= the reference's whenever it is unique)
    cfg->kalman_dtype = BTS_F64; cfg->cost_dtype = BTS_F64;      // the reference's arithmetic (BASELINE config 5's mode: BTS_F32 / BTS_F16)
    return BTS_OK;
}

// What create and memory_bytes both derive from a config: the appearance mode actually used and the capacity of the
// solver's edge lists.  The feasible pairs (cost <= match threshold) are O(detections) in any scene a tracker can
// follow, hence 64 (T + M) by default; a configuration in which EVERY pair is feasible (a stage threshold >= 1: the
// cost tiles then keep all rows, `!(limit < 1.0)`) and small tables keep the T * M that cannot overflow.  Overflow
// raises the sticky capacity error (BTS_ECAPACITY from the next export).
static int derive_sizes(const bts_tracker_config *cfg, Dev &d, int64_t *e_cap_out) {
    BTS_REQUIRE(cfg->struct_size == (int32_t)sizeof(bts_tracker_config) && cfg->abi_version == BTS_ABI_VERSION,
                "bts_tracker_config.struct_size / abi_version do not match this library's bts_tracker_config_size(): start from "
                "bts_tracker_default_config, and check the binding's mirror of the struct against include/botsort_hip.h");
    d.T = cfg->max_tracks; d.Mx = cfg->max_dets; d.D = cfg->feat_dim; d.Df = cfg->face_dim;
    d.record = cfg->record_costs ? 1 : 0;
    // the pre-filter works on 128-element K steps; other feature widths take the dense float32 path
    d.filter = (cfg->appearance_mode == BTS_APPEARANCE_FILTER && d.D % DBK == 0 && d.Df % DBK == 0) ? 1 : 0;
    if (d.filter) {     // k_filter_i8 addresses the int8 shadows with 32-bit byte offsets (slot * dim)
        const uint64_t lim = 1ull << 32;
        BTS_REQUIRE((uint64_t)d.T * (uint64_t)d.D < lim && (uint64_t)d.Mx * (uint64_t)d.D < lim &&
                    (uint64_t)d.T * (uint64_t)d.Df < lim && (uint64_t)d.Mx * (uint64_t)d.Df < lim,
                    "max_tracks * feat_dim and max_dets * feat_dim must stay below 2^32 in the FILTER appearance mode");
    }
    BTS_REQUIRE(cfg->tie_mode == 0 || cfg->tie_mode == 1, "tie_mode");
    BTS_REQUIRE(cfg->kalman_dtype == BTS_F64 || cfg->kalman_dtype == BTS_F32, "kalman_dtype: BTS_F64 or BTS_F32");
    BTS_REQUIRE(cfg->cost_dtype == BTS_F64 || cfg->cost_dtype == BTS_F16, "cost_dtype: BTS_F64 or BTS_F16");
    BTS_REQUIRE(!(cfg->cost_dtype == BTS_F16 && cfg->tie_mode), "cost_dtype BTS_F16 needs tie_mode 0 (the exact-tie replay prices pairs from the boxes)");
    // BTS_EDGES_SORTED=1 (read when the tracker is created / sized): stage-1 edges in the sorted / list form (tk_edges_sorted)
    d.sorted = (bts_form_flag("BTS_EDGES_SORTED") && d.filter && d.Df == 0 && !cfg->motion_gate && !cfg->record_costs &&
                cfg->match_thresh < 1.0 && !cfg->use_graph) ? 1 : 0;
    // BTS_LISTS_AHEAD=1 (read at create): the next frame's pool / unconfirmed / free lists are built by tk_finalize
    d.ahead = (bts_form_flag("BTS_LISTS_AHEAD") && !cfg->use_graph) ? 1 : 0;
    // BTS_FILTER_AHEAD=1 (read at create; opt-in THROUGHPUT mode, see enqueue_step): the pre-filter of a frame runs on the side
    // stream ahead of the step that consumes it.  Body features only, bitmap consumers = the tile forms; the pool list it selects
    // bitmap rows for is the one tk_finalize builds (it implies BTS_LISTS_AHEAD; create adds BTS_PREDICT_FRONT and the side stream)
    d.fahead = (bts_form_flag("BTS_FILTER_AHEAD") && d.filter && d.Df == 0 && !cfg->use_graph && !cfg->record_costs) ? 1 : 0;      // (with or without the sorted edges)
    if (d.fahead) d.ahead = 1;
    // BTS_SMOOTH_DEFERRED=1 (read at create; needs BTS_FILTER_AHEAD's streams): the `smooth` EMA of the rewritten tracks leaves the step
    d.smooth_def = (bts_form_flag("BTS_SMOOTH_DEFERRED") && d.fahead) ? 1 : 0;
    // BTS_MERGE_EARLY=1 (read at create): tk_merge's jobs in the launch of tk_apply23 (tk_apply23_merge)
    d.merge_early = bts_form_flag("BTS_MERGE_EARLY") ? 1 : 0;
    d.renorm_fused = bts_form_flag("BTS_RENORM_FUSED") ? 1 : 0;
    {   // BTS_SHADOW_PLANES=1 (read at create): the int8 shadows in K-stage planes (feat_math.h, i8_elem); default row-major
        const bool planes = bts_form_flag("BTS_SHADOW_PLANES") && d.filter;
        const int dim[2] = {d.D, d.Df};
        d.sh_planes = planes ? 1 : 0;
        for (int f = 0; f < 2; ++f) {
            d.sh_rt[f] = planes ? (unsigned)feat::I8_STAGE : (unsigned)dim[f]; d.sh_ct[f] = planes ? (unsigned)d.T * feat::I8_STAGE : (unsigned)feat::I8_STAGE;
            d.sh_rd[f] = planes ? (unsigned)feat::I8_STAGE : (unsigned)dim[f]; d.sh_cd[f] = planes ? (unsigned)d.Mx * feat::I8_STAGE : (unsigned)feat::I8_STAGE;
        }
    }
    d.sf32 = cfg->kalman_dtype == BTS_F32 ? 1 : 0;
    d.c16 = cfg->cost_dtype == BTS_F16 ? 1 : 0;
    if (cfg->tie_mode) {
        // the replay prices a pair that is not a feasible edge at its IoU cost: true as long as every pair the appearance term
        // lowers (<= appearance_thresh) IS an edge of its stage, and nothing else edits the matrix (the motion gate does)
        BTS_REQUIRE(!cfg->motion_gate, "tie_mode 1 needs motion_gate 0");
        // (with face features a pair whose FACE passes the gate keeps its BODY distance even above the match threshold,
        //  D:1543-1554: a non-edge that is not an IoU cost -- the stage-1 tiles hand those to the replay as overrides, tk_edges_xo)
        BTS_REQUIRE(cfg->appearance_thresh <= cfg->match_thresh && cfg->appearance_thresh <= cfg->unconfirmed_match_thresh &&
                    cfg->match_thresh >= 0.0 && cfg->second_match_thresh >= 0.0 && cfg->unconfirmed_match_thresh >= 0.0,
                    "tie_mode 1 needs 0 <= appearance_thresh <= match_thresh, unconfirmed_match_thresh and second_match_thresh >= 0");
        BTS_REQUIRE(((int64_t)d.T + d.Mx) * ((int64_t)d.T + d.Mx) < (1ll << 31), "tie_mode 1 needs (max_tracks + max_dets)^2 < 2^31");
    }
    const int64_t full = (int64_t)d.T * d.Mx, e_auto = 64ll * ((int64_t)d.T + d.Mx);
    const bool all_feasible = !(cfg->match_thresh < 1.0) || !(cfg->second_match_thresh < 1.0) || !(cfg->unconfirmed_match_thresh < 1.0);
    int64_t e_cap = full;
    if (cfg->max_edges > 0) e_cap = full < cfg->max_edges ? full : cfg->max_edges;
    else if (!all_feasible && full > e_auto) e_cap = e_auto;
    *e_cap_out = e_cap;
    return BTS_OK;
}

// BTS_LAP_SELF=1 (read at create): every workgroup of an assignment launch prepares the problem itself (lap.hip, struct Packed);
// -> the number of extra packed-component slices the two work areas carry (one per helper workgroup the launches will have)
static int lap_self_slices(const Dev &d) {
    if (!bts_form_flag("BTS_LAP_SELF")) return 0;
    const char *e = bts_form_get("BTS_LAP_HELPERS");
    return bts_lap_helpers_for(d.T, e && *e && atoi(e) >= 0 && atoi(e) <= 255 ? atoi(e) : -1);
}

int bts_tracker_memory_bytes(const bts_tracker_config *cfg, uint64_t *out_bytes) {
    BTS_REQUIRE(cfg && out_bytes, "null argument");
    BTS_REQUIRE(cfg->max_tracks > 0 && cfg->max_dets > 0 && cfg->feat_dim > 0 && cfg->face_dim >= 0, "capacities");
    BTS_REQUIRE((int64_t)cfg->max_tracks * cfg->max_dets < (1ll << 31), "max_tracks * max_dets must fit int32");
    BTS_REQUIRE(cfg->max_edges >= 0, "max_edges");
    // the sizes bts_tracker_create derives from cfg, then the same layout walk with no base address
    bts_tracker *t = new (std::nothrow) bts_tracker();
    if (!t) { bts_set_error("host allocation failed"); return BTS_ENOMEM; }
    t->cfg = *cfg;
    Dev &d = t->dv;
    memset(&d, 0, sizeof(d));
    int64_t e_cap = 0;
    const int rc = derive_sizes(cfg, d, &e_cap);
    if (rc) { delete t; return rc; }
    t->cvt_overlap = (bts_form_flag("BTS_CVT_OVERLAP") || d.fahead) && d.filter && !cfg->use_graph;
    size_t off = 0;
    tracker_layout(t, nullptr, off);
    const int64_t o_cap = (cfg->tie_mode && d.Df > 0) ? e_cap : 0;      // overrides of the exact-tie step (tk_edges_xo): their own region
    const int xs = lap_self_slices(d);
    off += bts_lap_work_bytes(d.T, d.Mx, e_cap, o_cap, xs) + bts_lap_work_bytes(d.T, d.Mx, e_cap, 0, xs);
    if (cfg->tie_mode) {
        size_t nd = 0, ni = 0;
        bts_lap_exact_sizes(d.T, d.Mx, e_cap + o_cap, &nd, &ni);
        off += 2 * (bts_align(nd * sizeof(double)) + bts_align(ni * sizeof(int32_t)));
    }
    *out_bytes = (uint64_t)off;
    delete t;
    return BTS_OK;
}

int bts_tracker_create(bts_context *ctx, const bts_tracker_config *cfg, bts_tracker **out) {
    BTS_REQUIRE(ctx && cfg && out, "null argument");
    BTS_ON_DEVICE(ctx);
    *out = nullptr;
    BTS_REQUIRE(cfg->max_tracks > 0 && cfg->max_dets > 0 && cfg->feat_dim > 0 && cfg->face_dim >= 0, "capacities");
    BTS_REQUIRE((int64_t)cfg->max_tracks * cfg->max_dets < (1ll << 31), "max_tracks * max_dets must fit int32");
    BTS_REQUIRE(cfg->max_edges >= 0, "max_edges");
    bts_tracker *t = new (std::nothrow) bts_tracker();
    if (!t) { bts_set_error("host allocation failed"); return BTS_ENOMEM; }
    t->ctx = ctx; t->cfg = *cfg;
    Dev &d = t->dv;
    memset(&d, 0, sizeof(d));
    int64_t e_cap = 0;
    {
        const int rc_sz = derive_sizes(cfg, d, &e_cap);
        if (rc_sz) { delete t; return rc_sz; }
    }
    d.high = cfg->track_high_thresh; d.low = cfg->track_low_thresh; d.new_thr = cfg->new_track_thresh;
    d.match1 = cfg->match_thresh; d.match2 = cfg->second_match_thresh; d.match3 = cfg->unconfirmed_match_thresh;
    d.prox = cfg->proximity_thresh; d.app = (float)cfg->appearance_thresh; d.dedupe_thr = cfg->dedupe_thresh;
    d.ema_a = (float)cfg->ema_alpha; d.ema_b = (float)(1.0 - cfg->ema_alpha);
    d.max_time_lost = (int)((double)cfg->frame_rate / 30.0 * cfg->track_buffer);    // D:1276-1277
    d.motion_gate = cfg->motion_gate ? 1 : 0;
    d.gate_thr = 9.4877;       // chi2inv95[4], D:139-149
    {   // list jobs over tables beyond ~6000 tracks are split into chunks of ~2048 entries, one workgroup each
        // (measured: 4096 tracks slightly better unsplit, 10 000 +3.5 %, 16 384 +6.5 %); BTS_CHUNK_ROWS forces a chunk size
        const char *e = bts_form_get("BTS_CHUNK_ROWS");
        const bool forced = e != nullptr;
        const int rows = e && atoi(e) > 0 ? atoi(e) : 2048;
        d.nch = (!forced && d.T <= 6144) ? 1 : bts_cdiv(d.T, rows);
        d.nch = d.nch < 1 ? 1 : d.nch > 8 ? 8 : d.nch;
    }
    d.use_grid = (bts_form_flag("BTS_EDGES_GRID") && !d.fahead) ? 1 : 0;      // (the neighbour search reads the bitmap by pool row)
    t->predict_first = bts_form_flag("BTS_PREDICT_FIRST");
    t->predict_front = (bts_form_flag("BTS_PREDICT_FRONT") || d.fahead) && d.ahead;
    t->cvt_overlap = (bts_form_flag("BTS_CVT_OVERLAP") || d.fahead) && d.filter && !cfg->use_graph;
    { const char *e = bts_form_get("BTS_FILTER_AHEAD_PW"); const int v = e ? atoi(e) : 0; t->filter_ahead_pw = v > 0 && v <= 2048 ? v : 256; }
    t->lap_form = bts_lap_form_from_env();
    {   // resolved ONCE, here: an absent / empty key is the automatic count for this table size, never a later look at the
        // process environment -- the SELF form's slices below are carved for exactly this many helper workgroups
        const char *e = bts_form_get("BTS_LAP_HELPERS");
        t->lap_helpers = bts_lap_helpers_for(d.T, e && *e && atoi(e) >= 0 && atoi(e) <= 255 ? atoi(e) : -1);
    }
    { const char *e = bts_form_get("BTS_FILTER_PW"); const int v = e ? atoi(e) : 0; t->filter_pw = v > 0 && v <= 2048 ? v : 0; }
    { const char *e = bts_form_get("BTS_TILE_GOAL"); const int v = e ? atoi(e) : 0; t->tile_goal = v >= 64 && v <= 16384 ? v : 768; }
    t->onepass = bts_form_flag("BTS_FILTER_ONEPASS");
    {   // BTS_FILTER_TILE=160x128 | 128x64 (read at create): tile shape of the one-pass pre-filter (k_filter_t); anything else = 128x128
        const char *e = bts_form_get("BTS_FILTER_TILE");
        t->filter_tile = !e ? 0 : !strcmp(e, "160x128") ? 1 : !strcmp(e, "128x64") ? 2 : 0;
#ifdef BTS_HOST_EMU
        if (e && !strcmp(e, "128x128t")) t->filter_tile = 3;      // (emulator build only: the template at k_filter_i8's own shape, two record
                                                                 //  images -- tests/test_emu_gpu_subset.py holds the two kernels against each other)
#endif
        if (d.Df > 0 || d.sorted || d.fahead) t->filter_tile = 0;      // (no face pass, no candidate list in k_filter_t; the side passes run k_filter_i8)
        if (!d.filter) t->filter_tile = 0;                 // (dense appearance mode: no pre-filter at all -- bts_tracker_forms reports forms IN EFFECT)
    }
    d.gate_sim = 1.0f - d.app;
    d.f_guess = bts_form_flag("BTS_FILTER_GUESS") ? 1 : 0;
    {   // first prefix checkpoint of the pre-filter: from the gate, or BTS_I8_FIRST16 (3 .. 12; 5 = the round-2 set)
        const char *e = bts_form_get("BTS_I8_FIRST16");
        const int f = e ? atoi(e) : 0;
        d.chk_first = (f >= 3 && f <= 12) ? f : feat::i8_first_checkpoint(d.app);
    }
    // the forms in effect (a requested form that the configuration rules out -- the sorted edges with a motion gate, say --
    // does not appear), for bts_tracker_forms
    snprintf(t->forms, sizeof(t->forms), "edges=%s,lists_ahead=%d,merge_early=%d,renorm_fused=%d,cvt_overlap=%d,predict_front=%d,predict_first=%d,lap_form=%s,lap_helpers=%d,list_chunks=%d,"
             "filter_pw=%d,filter_first16=%d,filter_onepass=%d,filter_guess=%d,shadow_planes=%d,tile_goal=%d,lap_self=%d,filter_tile=%s,filter_ahead=%d,smooth_deferred=%d",
             d.sorted ? "sorted" : (d.use_grid && d.filter) ? "grid" : "tiles", d.ahead, d.merge_early, d.renorm_fused, t->cvt_overlap ? 1 : 0, t->predict_front ? 1 : 0, t->predict_first ? 1 : 0,
             t->lap_form == 1 ? "row" : "col", t->lap_helpers, d.nch, t->filter_pw > 0 ? t->filter_pw : (t->filter_tile == 2 ? 768 : FILTER_PW), d.chk_first,
             t->onepass ? 1 : 0, d.f_guess, d.sh_planes, t->tile_goal, lap_self_slices(d) > 0 ? 1 : 0,
             t->filter_tile == 1 ? "160x128" : t->filter_tile == 2 ? "128x64" : t->filter_tile == 3 ? "128x128t" : "128x128", d.fahead, d.smooth_def);
    size_t off = 0;
    tracker_layout(t, nullptr, off);
    const int64_t o_cap = (cfg->tie_mode && d.Df > 0) ? e_cap : 0;      // overrides of the exact-tie step (tk_edges_xo): their own region
    const size_t lap_off = off;
    const int xs = lap_self_slices(d);
    off += bts_lap_work_bytes(d.T, d.Mx, e_cap, o_cap, xs);
    const size_t lap3_off = off;
    off += bts_lap_work_bytes(d.T, d.Mx, e_cap, 0, xs);
    size_t xk_off[2][2] = {{0, 0}, {0, 0}}, xk_nd = 0, xk_ni = 0;
    if (cfg->tie_mode) {
        bts_lap_exact_sizes(d.T, d.Mx, e_cap + o_cap, &xk_nd, &xk_ni);
        for (int k = 0; k < 2; ++k) {
            xk_off[k][0] = off; off += bts_align(xk_nd * sizeof(double));
            xk_off[k][1] = off; off += bts_align(xk_ni * sizeof(int32_t));
        }
    }
    t->bytes = off;
    if (hipMalloc((void **)&t->block, t->bytes) != hipSuccess) {
        (void)hipGetLastError();
        bts_set_error("device allocation of %zu bytes for the track table failed", t->bytes);
        delete t;
        return BTS_ENOMEM;
    }
    // Every index list of the table starts out as zeros and is only ever written with valid indices: a kernel may therefore
    // issue the loads behind a list entry BEFORE it knows how long the list is this frame (the entry past the end is a stale or
    // zero, but valid, index; what was loaded through it is discarded once the extent is known) -- one dependent load less at
    // the head of the chains that open a kernel (edges_tile, apply_body, predict_body)
    if (hipMemset(t->block, 0, t->bytes) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(t->block);
        bts_set_error("hipMemset failed");
        delete t;
        return BTS_EHIP;
    }
    off = 0;
#ifdef BTS_HOST_EMU
    g_redzones = &t->redzones;
#endif
    tracker_layout(t, t->block, off);
#ifdef BTS_HOST_EMU
    g_redzones = nullptr;
#endif
#ifdef BTS_HOST_EMU
    {   // the two solver work areas carry red zones as well (offsets relative to their own base)
        std::vector<std::pair<size_t, size_t>> z1, z3;
        bts_emu_lap_zones = &z1; bts_lap_work_carve(&t->lap, t->block + lap_off, d.T, d.Mx, e_cap, o_cap, xs);
        bts_emu_lap_zones = &z3; bts_lap_work_carve(&t->lap3, t->block + lap3_off, d.T, d.Mx, e_cap, 0, xs);
        bts_emu_lap_zones = nullptr;
        for (const auto &z : z1) t->redzones.push_back({lap_off + z.first, z.second});
        for (const auto &z : z3) t->redzones.push_back({lap3_off + z.first, z.second});
    }
#else
    bts_lap_work_carve(&t->lap, t->block + lap_off, d.T, d.Mx, e_cap, o_cap, xs);
    bts_lap_work_carve(&t->lap3, t->block + lap3_off, d.T, d.Mx, e_cap, 0, xs);
#endif
    for (int k = 0; k < 2 && cfg->tie_mode; ++k) {
        t->xk[k].dbl = (double *)(t->block + xk_off[k][0]); t->xk[k].ndbl = xk_nd;
        t->xk[k].ints = (int32_t *)(t->block + xk_off[k][1]); t->xk[k].nints = xk_ni;
    }
    if (hipHostMalloc((void **)&t->host_hdr, H_SIZE * sizeof(int32_t)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(t->block);
        bts_set_error("pinned host allocation failed");
        delete t;
        return BTS_ENOMEM;
    }
    const size_t tm = (size_t)t->dv.T * t->dv.Mx;      // the pre-filter's accumulators and bitmap are self-cleaning from here on
    if (t->dv.filter && ((t->dv.approx && (hipMemset(t->dv.approx, 0, sizeof(int32_t) * tm) != hipSuccess ||
                                           hipMemset(t->dv.approx_f, 0, sizeof(int32_t) * (t->dv.Df ? tm : 1)) != hipSuccess)) ||
                         hipMemset(t->dv.fstat, 0, sizeof(int32_t) * 16) != hipSuccess ||
                         hipMemset(t->dv.cbits, 0, sizeof(uint32_t) * (size_t)(t->dv.fahead ? 2 * (t->dv.T + t->dv.Mx) : t->dv.T) * t->dv.cb_ld) != hipSuccess)) {
        (void)hipGetLastError();
        bts_set_error("hipMemset failed");
        bts_tracker_destroy(t);
        return BTS_EHIP;
    }
    if (t->dv.fahead) {       // the identity list of the side passes' rows
        std::vector<int32_t> io((size_t)(t->dv.T > t->dv.Mx ? t->dv.T : t->dv.Mx));
        for (size_t k = 0; k < io.size(); ++k) io[k] = (int32_t)k;
        if (hipMemcpy(t->iota, io.data(), io.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            bts_set_error("hipMemcpy failed");
            bts_tracker_destroy(t);
            return BTS_EHIP;
        }
    }
    if (t->dv.sorted && hipMemset(t->dv.cand_cnt, 0, sizeof(int32_t) * cand_slots(t->dv) * (t->dv.fahead ? 2 : 1)) != hipSuccess) {
        (void)hipGetLastError();
        bts_set_error("hipMemset failed");
        bts_tracker_destroy(t);
        return BTS_EHIP;
    }
    // the assignment work area keeps "degree arrays are zero between calls" (lap.hip)
    if (hipMemset(t->block + lap_off, 0, t->bytes - lap_off) != hipSuccess) {
        (void)hipGetLastError();
        bts_set_error("hipMemset failed");
        bts_tracker_destroy(t);
        return BTS_EHIP;
    }
    bool ok = hipMalloc((void **)&t->d_in, sizeof(InDesc)) == hipSuccess &&
              hipHostMalloc((void **)&t->h_in, sizeof(InDesc) * bts_tracker::RING) == hipSuccess;
    for (int k = 0; ok && k < bts_tracker::RING; ++k) ok = hipEventCreateWithFlags(&t->in_ev[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&t->cs[0], hipStreamNonBlocking) == hipSuccess;
    if (t->cvt_overlap) {
        if (t->dv.fahead) {
            // the side passes of BTS_FILTER_AHEAD are background work with a whole step of slack: lowest priority, so that when both
            // streams have workgroups to place the step's go first
            int least = 0, greatest = 0;
            ok = ok && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
                 hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, least) == hipSuccess;
        } else
        ok = ok && hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; ok && k < 2; ++k)
            ok = hipEventCreateWithFlags(&t->ev_cvt[k], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&t->ev_done[k], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming) == hipSuccess;
        if (t->dv.smooth_def) {
            int least = 0, greatest = 0;
            ok = ok && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
                 hipStreamCreateWithPriority(&t->side2, hipStreamNonBlocking, least) == hipSuccess;
            for (int k = 0; ok && k < 2; ++k) ok = hipEventCreateWithFlags(&t->ev_smooth[k], hipEventDisableTiming) == hipSuccess;
        }
    }
    if (!ok) { (void)hipGetLastError(); bts_set_error("stream / event / descriptor allocation failed"); bts_tracker_destroy(t); return BTS_EHIP; }
    t->dv.in = nullptr;
#ifdef BTS_HOST_EMU
    for (const auto &z : t->redzones) memset(t->block + z.first, 0xA5, z.second);      // (after the last memset of create)
    t->redzones_armed = true;
    // (HIPEMU_POKE_REDZONE=k: fault injection for tests/test_emu_gpu_subset.py -- one byte of zone k is overwritten the way a
    //  kernel's off-by-one store would; the next counts() / destroy must abort)
    if (const char *pk = getenv("HIPEMU_POKE_REDZONE")) { const size_t k = (size_t)atoi(pk) % t->redzones.size(); t->block[t->redzones[k].first] = 0; }
#endif
    bts_lap_init();
    int rc = bts_tracker_reset(t, nullptr);
    if (rc == BTS_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = BTS_EHIP;
    if (rc != BTS_OK) { bts_tracker_destroy(t); return rc; }
    *out = t;
    return BTS_OK;
}

int bts_tracker_create_forms(bts_context *ctx, const bts_tracker_config *cfg, const char *forms, bts_tracker **out) {
    bts_forms_scope scope(forms);      // (thread-local: the lookups of this create call see `forms` before the environment)
    return bts_tracker_create(ctx, cfg, out);
}

int bts_tracker_memory_bytes_forms(const bts_tracker_config *cfg, const char *forms, uint64_t *out_bytes) {
    bts_forms_scope scope(forms);
    return bts_tracker_memory_bytes(cfg, out_bytes);
}

int bts_tracker_forms(const bts_tracker *t, char *buf, int capacity) {
    BTS_REQUIRE(t && buf && capacity > 0, "tracker, buf, capacity");
    snprintf(buf, (size_t)capacity, "%s", t->forms);
    return (int)strlen(t->forms);
}

int bts_tracker_profile_enable(bts_tracker *t, int segment, int capacity) {
    BTS_REQUIRE(t && capacity >= 0, "tracker, capacity");
    BTS_ON_DEVICE(t->ctx);
    for (int k = 0; t->prof_ev && k < 2 * t->prof_cap + 2; ++k) (void)hipEventDestroy(t->prof_ev[k]);
    delete[] t->prof_ev;
    t->prof_ev = nullptr; t->prof_cap = 0; t->prof_n = 0; t->prof_seg = segment;
    if (capacity == 0) return BTS_OK;
    hipEvent_t *ev = new (std::nothrow) hipEvent_t[2 * capacity + 2]();
    if (!ev) { bts_set_error("host allocation failed"); return BTS_ENOMEM; }
    for (int k = 0; k < 2 * capacity + 2; ++k) {
        const hipError_t e = hipEventCreate(&ev[k]);
        if (e != hipSuccess) {                  // nothing half-built is kept
            for (int q = 0; q < k; ++q) (void)hipEventDestroy(ev[q]);
            delete[] ev;
            bts_set_error("hipEventCreate failed: %s", hipGetErrorString(e));
            return BTS_EHIP;
        }
    }
    t->prof_ev = ev;
    t->prof_cap = capacity;
    return BTS_OK;
}

int bts_tracker_profile_stride(bts_tracker *t, int stride) {
    BTS_REQUIRE(t && stride >= 1, "tracker, stride");
    BTS_ON_DEVICE(t->ctx);
    t->prof_stride = stride;
    return BTS_OK;
}

int bts_tracker_profile_read(bts_tracker *t, double *total_ms, int32_t *count) {
    BTS_REQUIRE(t && total_ms && count, "null argument");
    BTS_ON_DEVICE(t->ctx);
    double tot = 0.0;
    for (int k = 0; k < t->prof_n; ++k) {
        BTS_HIP(hipEventSynchronize(t->prof_ev[2 * k + 1]));
        float ms = 0.0f;
        BTS_HIP(hipEventElapsedTime(&ms, t->prof_ev[2 * k], t->prof_ev[2 * k + 1]));
        tot += ms;
    }
    *total_ms = tot; *count = t->prof_n;
    t->prof_n = 0;
    return BTS_OK;
}

#ifdef BTS_HOST_EMU
static void emu_check_redzones(const bts_tracker *t, const char *where) {
    if (!t->redzones_armed) return;
    (void)hipDeviceSynchronize();
    for (size_t k = 0; k < t->redzones.size(); ++k) {
        const unsigned char *p = (const unsigned char *)t->block + t->redzones[k].first;
        for (size_t b = 0; b < t->redzones[k].second; ++b)
            if (p[b] != 0xA5) {
                fprintf(stderr, "hipemu: red zone %zu of the table block (offset %zu, %zu bytes) overwritten at byte %zu (value %u), found at %s, frame %d\n",
                        k, t->redzones[k].first, t->redzones[k].second, b, (unsigned)p[b], where, t->block ? ((const int32_t *)t->dv.hdr)[H_FRAME] : -1);
                abort();
            }
    }
}
#endif

int bts_tracker_destroy(bts_tracker *t) {
    if (!t) return BTS_OK;
    BTS_ON_DEVICE(t->ctx);
#ifdef BTS_HOST_EMU
    if (t->block) emu_check_redzones(t, "bts_tracker_destroy");
#endif
    for (int k = 0; t->prof_ev && k < 2 * t->prof_cap + 2; ++k) (void)hipEventDestroy(t->prof_ev[k]);
    delete[] t->prof_ev;
    if (t->gexec) (void)hipGraphExecDestroy(t->gexec);
    for (int k = 0; k < bts_tracker::RING; ++k) if (t->in_ev[k]) (void)hipEventDestroy(t->in_ev[k]);
    if (t->cs[0]) (void)hipStreamDestroy(t->cs[0]);
    if (t->side) { (void)hipStreamSynchronize(t->side); (void)hipStreamDestroy(t->side); }
    if (t->side2) { (void)hipStreamSynchronize(t->side2); (void)hipStreamDestroy(t->side2); }
    for (int k = 0; k < 2; ++k) if (t->ev_smooth[k]) (void)hipEventDestroy(t->ev_smooth[k]);
    for (int k = 0; k < 2; ++k) { if (t->ev_cvt[k]) (void)hipEventDestroy(t->ev_cvt[k]); if (t->ev_done[k]) (void)hipEventDestroy(t->ev_done[k]); }
    if (t->ev_in) (void)hipEventDestroy(t->ev_in);
    if (t->d_in) (void)hipFree(t->d_in);
    if (t->h_in) (void)hipHostFree(t->h_in);
    if (t->block) (void)hipFree(t->block);
    if (t->host_hdr) (void)hipHostFree(t->host_hdr);
    delete t;
    return BTS_OK;
}

int bts_tracker_reset(bts_tracker *t, bts_stream stream) {
    BTS_REQUIRE(t, "tracker");
    BTS_ON_DEVICE(t->ctx);
    hipLaunchKernelGGL(tk_reset, dim3(1), dim3(1024), 0, (hipStream_t)stream, t->dv);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

// Enqueue the kernels of one step.  `mc` bounds every detection-indexed grid (the frame's m in eager
// mode, max_dets under graph capture; `capturing` = no timing events).
static int enqueue_step(bts_tracker *t, hipStream_t s, float *feats, float *face_feats, int mc, bool capturing) {
    const bool prof_room = !capturing && t->prof_cap > 0 && t->prof_n < t->prof_cap && t->step_no % t->prof_stride == 0;
    auto seg_begin = [&](int id) { if (prof_room && t->prof_seg == id) (void)hipEventRecord(t->prof_ev[2 * t->prof_n], s); };
    auto seg_end = [&](int id) { if (prof_room && t->prof_seg == id) { (void)hipEventRecord(t->prof_ev[2 * t->prof_n + 1], s); t->prof_n++; } };
    // BTS_CVT_OVERLAP: this frame's detection shadows live in buffer (step & 1), converted on the side stream
    const bool ov = t->cvt_overlap && !capturing;
    const int ovp = (int)(t->step_no & 1);
    Dev dstep = t->dv;
    dstep.f_ntiles = bts_cdiv(dstep.T, FBM) * bts_cdiv(mc, FBN); dstep.f_mc = mc;      // (the pre-filter's tiling of this step: candidate-list slots decode with it)
    if (ov && ovp) { dstep.det_h = t->det_h2; dstep.det_m = t->det_m2; dstep.dface_h = t->dface_h2; dstep.dface_m = t->dface_m2; }
    // BTS_FILTER_AHEAD: this frame's candidate bitmap ([T + Mx] rows) is buffer (step & 1) as well
    const bool fa = dstep.fahead && ov;
    if (fa) dstep.cbits = t->dv.cbits + (size_t)ovp * (size_t)(dstep.T + dstep.Mx) * (size_t)dstep.cb_ld;
    // the side passes' tilings of this step (main: every table slot; det x det: the previous step's detections)
    const int fa_mt = fa ? bts_cdiv(dstep.T, FBM) * bts_cdiv(mc, FBN) : 0;
    const int fa_dt = (fa && t->mc_prev > 0 && mc > 0) ? bts_cdiv(t->mc_prev, FBM) * bts_cdiv(mc, FBN) : 0;
    if (fa && dstep.sorted) {       // the sorted edges: this buffer's two candidate lists, main pass first
        const size_t per = cand_slots(t->dv);
        dstep.cand_cnt = t->dv.cand_cnt + (size_t)ovp * per; dstep.cand = t->dv.cand + (size_t)ovp * per * CAND_W;
        dstep.fa_mt = fa_mt; dstep.fa_dt = fa_dt; dstep.fa_dd_slot0 = (int)cand_slots_main(t->dv); dstep.fa_nprev = t->mc_prev;
    }
    const Dev &d = dstep;
    const int T = d.T, M = d.Mx;
    const int cos_kernel = t->ctx->cos_kernel;
    int ntile = bts_cdiv(mc, CT_COLS) * bts_cdiv(T, CT_ROWS);
    if (ntile > 256) ntile = 256;               // (stages 2 and 3 are small problems: idle workgroups only cost dispatch time)
    const dim3 tile_grid(ntile);               // persistent tiles: the kernels walk the live extents
    // stage-1 tiles are R x 256 with R chosen for ~TILE_GOAL workgroups: a tile's cost is its chains of dependent loads
    // plus a serial part per row, so small problems want many small tiles and big ones few tall tiles
    const int TILE_GOAL = t->tile_goal;       // 768 (swept on c2 / c3 / target: 256 ... 6144; heights up to 256 rows: 10000 x 2000 runs
                                              //  128-row tiles (632 of them: one resident round), 16384 x 4096 256-row tiles); BTS_TILE_GOAL
    int trows = 2;
    while (trows < 256 && bts_cdiv(mc, CT_COLS) * bts_cdiv(T, trows) > TILE_GOAL) trows *= 2;
    const int ntile1 = bts_cdiv(mc, CT_COLS) * bts_cdiv(T, trows);
    // BTS_PREDICT_FRONT (needs BTS_LISTS_AHEAD): the predict workgroups ride in the frame's first launch (1024 threads each), none
    // in the pre-filter's
    const int pred_front = (t->predict_front && d.ahead) ? bts_cdiv(T * 8 > M ? T * 8 : M, 1024) : 0;
    const int kf_blocks = bts_cdiv(T * 8 > M ? T * 8 : M, 256);   // 8 lanes per track
    const int kf_pre = pred_front ? 0 : kf_blocks;                 // predict workgroups in the pre-filter's / tk_predict's launch
    bts_lap_work lap[3] = {t->lap, t->lap, t->lap3};
    for (int k = 0; k < 3; ++k) lap[k].hdr = d.lap_hdr + BTS_LAP_HDR_INTS * k;
    int rc;
#define LAUNCH_ON(st, kern, grid, block, ...) hipLaunchKernelGGL(kern, grid, block, 0, st, d, ##__VA_ARGS__); BTS_LAUNCH_CHECK()
#define LAUNCH(kern, grid, block, ...) LAUNCH_ON(s, kern, grid, block, ##__VA_ARGS__)

    seg_begin(BTS_SEG_STEP);
    seg_begin(BTS_SEG_PREP);
    const bool want_grid = (d.use_grid || d.sorted) && d.filter;
    if (ov) {
        // The side stream reads this frame's feature rows.  Unless the caller has promised that they are complete when the
        // step is called (bts_tracker_inputs_resident), it is ordered behind whatever the caller's stream holds at this point
        // -- an upload, a ReID encoder writing the rows -- through an event recorded here; that event also sits
        // behind the previous step's kernels, so only the promise buys the overlap with them.
        // (the shadow buffer's previous user is the frame before last)
        if (!t->inputs_resident) {
            BTS_HIP(hipEventRecord(t->ev_in, s));
            BTS_HIP(hipStreamWaitEvent(t->side, t->ev_in, 0));
        }
        if (t->ev_done_used[ovp]) BTS_HIP(hipStreamWaitEvent(t->side, t->ev_done[ovp], 0));
        if (fa) {
            // BTS_FILTER_AHEAD (opt-in throughput mode, round 6).  Everything the pre-filter of frame f computes is known before
            // step f-1 has finished, so it runs HERE, on the side stream, under step f-1's latency-bound kernels (the host enqueues
            // step f while the device is still in step f-1 or earlier), and the step's own stream carries no pre-filter launch:
            //   main pass   rows = ALL table slots x this frame's high-score detections.  Valid for every track whose shadow has
            //               been stable since step f-2 ended (the side stream waited for that above); rows step f-1 is rewriting
            //               meanwhile may be read torn -- their bits are never consulted (a torn record can only make a tile's
            //               row summary {min uh, max alpha, max r} more conservative for the other rows);
            //   det x det   rows = the detections of frame f-1 (their shadows and TRACK-SIDE records, tk_cvt_ahead) x the same
            //               columns.  A track rewritten in step f-1 holds exactly the shadow and record of the detection it
            //               absorbed (copy_shadow; det_index[slot]), whichever detection that turns out to be.
            // Both are the UNCHANGED default kernel k_filter_i8<false>, fed a Dev whose row list is the identity and whose extents
            // sit in a small header of their own; the stage-1 tiles find a pool row's bitmap row through pool_cbrow (tk_finalize).
            // (the launch also clears what the passes write with atomics -- only what this frame can touch: the words of its mc columns,
            //  whole 8-byte words since the tiles read aligned pairs, in the rows of the two passes; and the sorted edges' list counts)
            const int row_words = bts_cdiv(mc, 64) * 2, n_prev = fa_dt > 0 ? t->mc_prev : 0;
            const int clear_blocks = 1 + (int)(((size_t)(T + n_prev) * (size_t)row_words) / (16 * 1024));      // ~16 words per thread
            int32_t *sh_main = t->shdr + (size_t)(ovp * 2 + 0) * H_SIZE, *sh_dd = t->shdr + (size_t)(ovp * 2 + 1) * H_SIZE;
            LAUNCH_ON(t->side, tk_cvt_ahead, dim3(clear_blocks + bts_cdiv(mc, 4) + 1), dim3(1024), t->det_mts[ovp], t->hi_b[ovp], sh_main, sh_dd, t->fstat_b + 16 * ovp, t->mc_prev,
                      clear_blocks, row_words, d.sorted ? (int)cand_slots(t->dv) : 0);
            Dev dm = dstep;
            dm.pool_slot = t->iota; dm.hdr = sh_main; dm.hi_idx = t->hi_b[ovp]; dm.fstat = t->fstat_b + 16 * ovp;
            const int mt = fa_mt, mb = mt < t->filter_ahead_pw ? mt : t->filter_ahead_pw;
            // (bts_tracker_profile_*: the K_FILTER segment of this mode is the pair of side passes, timed on THEIR stream -- not part of
            //  the step's critical path, but what the first A/B wants to see next to the step's own segments)
            const bool prof_side = prof_room && t->prof_seg == BTS_SEG_K_FILTER;
            if (prof_side) (void)hipEventRecord(t->prof_ev[2 * t->prof_n], t->side);
            // (an empty frame has no tiles: a grid of zero workgroups is not a launch -- the memset above leaves "no candidates")
            if (mt > 0 && d.sorted) { hipLaunchKernelGGL((k_filter_i8<false, 0, false, true>), dim3(mb), dim3(256), 0, t->side, dm, mc, mt, mb); BTS_LAUNCH_CHECK(); }
            else if (mt > 0) { hipLaunchKernelGGL(k_filter_i8<false>, dim3(mb), dim3(256), 0, t->side, dm, mc, mt, mb); BTS_LAUNCH_CHECK(); }
            if (fa_dt > 0) {
                Dev dd = dm;
                dd.hdr = sh_dd; dd.T = M;                                  // (rows = detections of the previous frame: the clamps of the row fetch)
                dd.body_h = ovp ? t->dv.det_h : t->det_h2; dd.body_m = t->det_mts[ovp ^ 1];
                dd.sh_rt[0] = d.sh_rd[0]; dd.sh_ct[0] = d.sh_cd[0];      // (the row operand is a detection-side shadow array)
                dd.cbits = d.cbits + (size_t)T * (size_t)d.cb_ld;
                const int dt = fa_dt, db = dt < t->filter_ahead_pw ? dt : t->filter_ahead_pw;
                if (d.sorted) {
                    dd.cand_cnt = d.cand_cnt + d.fa_dd_slot0; dd.cand = d.cand + (size_t)d.fa_dd_slot0 * CAND_W;
                    hipLaunchKernelGGL((k_filter_i8<false, 0, false, true>), dim3(db), dim3(256), 0, t->side, dd, mc, dt, db); BTS_LAUNCH_CHECK();
                } else { hipLaunchKernelGGL(k_filter_i8<false>, dim3(db), dim3(256), 0, t->side, dd, mc, dt, db); BTS_LAUNCH_CHECK(); }
            }
            if (prof_side) { (void)hipEventRecord(t->prof_ev[2 * t->prof_n + 1], t->side); t->prof_n++; }
        } else {
            LAUNCH_ON(t->side, tk_cvt, dim3(bts_cdiv(mc, 4)), dim3(1024));
        }
        BTS_HIP(hipEventRecord(t->ev_cvt[ovp], t->side));
        if (d.ahead) {
            if (want_grid) { LAUNCH((tk_front_dets<true, false>), dim3(1 + pred_front), dim3(1024), pred_front); } else { LAUNCH((tk_front_dets<false, false>), dim3(1 + pred_front), dim3(1024), pred_front); }
        } else {
            const dim3 fgrid(front_jobs(d.nch > 1 ? d.nch : 1));
            if (want_grid) {
                if (d.nch > 1) { LAUNCH((tk_front<true, true, false>), fgrid, dim3(1024)); } else { LAUNCH((tk_front<false, true, false>), fgrid, dim3(1024)); }
            } else if (d.nch > 1) { LAUNCH((tk_front<true, false, false>), fgrid, dim3(1024)); }
            else { LAUNCH((tk_front<false, false, false>), fgrid, dim3(1024)); }
        }
        BTS_HIP(hipStreamWaitEvent(s, t->ev_cvt[ovp], 0));
    } else if (d.ahead) {
        const dim3 fgrid(1 + (d.filter ? bts_cdiv(mc, 4) : 0) + pred_front);
        if (want_grid) { LAUNCH((tk_front_dets<true, true>), fgrid, dim3(1024), pred_front); } else { LAUNCH((tk_front_dets<false, true>), fgrid, dim3(1024), pred_front); }
    } else {
        const dim3 fgrid(front_jobs(d.nch > 1 ? d.nch : 1) + (d.filter ? bts_cdiv(mc, 4) : 0));
        if (want_grid) {
            if (d.nch > 1) { LAUNCH((tk_front<true, true>), fgrid, dim3(1024)); } else { LAUNCH((tk_front<false, true>), fgrid, dim3(1024)); }
        } else if (d.nch > 1) { LAUNCH((tk_front<true, false>), fgrid, dim3(1024)); }
        else { LAUNCH((tk_front<false, false>), fgrid, dim3(1024)); }
    }
    seg_end(BTS_SEG_PREP);
    const bool dense = !d.filter || d.record;
    // split-K factor of the pre-filter: small problems have too few 128x128 tiles to fill 256 CUs
    int split = 1;
    const int ftiles = bts_cdiv(T, FBM) * bts_cdiv(mc, FBN);
    {
        const int kt_all = bts_cdiv(d.D, DBK);
        // at least ~64 workgroups; beyond that only while it shortens a long K loop (every split adds N*M float atomics)
        while (d.approx && split * 2 <= kt_all && split < 16 && (ftiles * split < 64 || (ftiles * split < 256 && kt_all / split > 8))) split *= 2;
    }
    // BTS_FILTER_ONEPASS=1 (read at create): the persistent one-pass form (k_filter_i8) also for problems small enough for split K
    // (host side only: lets the tests drive that kernel's tile walk and prefix exits at sizes a CPU emulation handles)
    if (t->onepass || fa) split = 1;         // (BTS_FILTER_AHEAD: the side passes are the one-pass kernel at every size)
    const int accum = split > 1;
    // ---- first association: pool x high-score detections
    seg_begin(BTS_SEG_COS1);
    if (fa) {
        // (nothing here: this frame's candidates were computed on the side stream, the step waited for them right behind its first
        //  launch, and the Kalman predict rode in that launch -- BTS_PREDICT_FRONT, implied)
    } else if (d.filter) {
        // Kalman predict rides in the pre-filter's launch (independent work, consecutive workgroups)
        seg_begin(BTS_SEG_K_FILTER);
        // one-pass form (candidate bitmap, prefix exits): up to FILTER_PW persistent workgroups walk the tiles;
        // small problems add split-K partial products instead (k_filter_splitk)
        // (BTS_FILTER_PW: sweep knob for the number of persistent GEMM workgroups, host side only)
        const int pw = t->filter_pw > 0 ? t->filter_pw : FILTER_PW;
        const int gemm_blocks = ftiles < pw ? ftiles : pw;
        t->last_ftm = FBM; t->last_ftn = FBN;
        if (!accum && t->filter_tile && !d.sorted && d.Df == 0) {
            // BTS_FILTER_TILE: the shape-generic kernel (predict workgroups behind the GEMM ones like the default, in front with BTS_PREDICT_FIRST)
            const int tm = t->filter_tile == 1 ? 160 : 128, tn = t->filter_tile == 2 ? 64 : 128;
            const int ttiles = bts_cdiv(T, tm) * bts_cdiv(mc, tn);
            const int tpw = t->filter_pw > 0 ? t->filter_pw : (t->filter_tile == 2 ? 768 : FILTER_PW);
            const int tblocks = ttiles < tpw ? ttiles : tpw;
            // (in front: a multiple of 8 predict workgroups, so that GEMM workgroup g still runs on XCD g % 8 -- the tile order's
            //  assumption; surplus predict workgroups find no row and exit)
            const int kff = (t->predict_first && kf_pre) ? (kf_pre + 7) / 8 * 8 : 0;
            const int tgrid = tblocks + (kff ? kff : kf_pre);
            if (t->filter_tile == 1) { LAUNCH((k_filter_t<160, 128, 1, 4, 2, 1>), dim3(tgrid), dim3(256), mc, ttiles, tblocks, kff); }
#ifdef BTS_HOST_EMU
            else if (t->filter_tile == 3) { LAUNCH((k_filter_t<128, 128, 2, 2, 2, 2>), dim3(tgrid), dim3(256), mc, ttiles, tblocks, kff); }
#endif
            else { LAUNCH((k_filter_t<128, 64, 2, 2, 3, 1>), dim3(tgrid), dim3(256), mc, ttiles, tblocks, kff); }
            t->last_ftm = tm; t->last_ftn = tn;
        }
        else if (!accum && d.sorted && t->predict_first) { LAUNCH((k_filter_i8<false, 0, true, true>), dim3(gemm_blocks + kf_pre), dim3(256), mc, ftiles, gemm_blocks); }
        else if (!accum && d.sorted) { LAUNCH((k_filter_i8<false, 0, false, true>), dim3(gemm_blocks + kf_pre), dim3(256), mc, ftiles, gemm_blocks); }
        else if (!accum && t->predict_first) { LAUNCH((k_filter_i8<false, 0, true>), dim3(gemm_blocks + kf_pre), dim3(256), mc, ftiles, gemm_blocks); }
        else if (!accum) { LAUNCH(k_filter_i8<false>, dim3(gemm_blocks + kf_pre), dim3(256), mc, ftiles, gemm_blocks); }
        else { LAUNCH(k_filter_splitk<false>, dim3(ftiles * split + kf_pre), dim3(256), mc, ftiles, split); }
        seg_end(BTS_SEG_K_FILTER);
        if (d.Df) {
            const int fsplit = split < bts_cdiv(d.Df, DBK) ? split : bts_cdiv(d.Df, DBK);
            if (!accum) { LAUNCH(k_filter_i8<true>, dim3(gemm_blocks), dim3(256), mc, ftiles, gemm_blocks); }
            else { LAUNCH(k_filter_splitk<true>, dim3(ftiles * fsplit), dim3(256), mc, ftiles, fsplit); }
        }
    } else {
        seg_begin(BTS_SEG_PREDICT);
        if (kf_pre) { LAUNCH(tk_predict, dim3(kf_pre), dim3(256)); }
        seg_end(BTS_SEG_PREDICT);
    }
    if (dense) {
        rc = bts_cos_enqueue(cos_kernel, d.body_curr, d.pool_slot, d.D, feats, d.hi_idx, d.D, d.D, d.hdr + H_DIMS1, T, mc,
                             d.sim, M, s);
        if (rc) return rc;
        if (d.Df) {
            rc = bts_cos_enqueue(cos_kernel, d.face_curr, d.pool_slot, d.Df, face_feats, d.hi_idx, d.Df, d.Df,
                                 d.hdr + H_DIMS1, T, mc, d.fsim, M, s);
            if (rc) return rc;
        }
    }
    seg_end(BTS_SEG_COS1);
    seg_begin(BTS_SEG_COST1);
    seg_begin(BTS_SEG_K_EDGES1);
    // (the grid only finds OVERLAPPING pairs: a match threshold >= 1 makes every pair feasible, which only the tile form emits)
    const bool xo = t->cfg.tie_mode && d.Df > 0;       // (the grid form does not emit overrides)
    const bool grid_form = d.use_grid && d.filter && !accum && !d.record && d.match1 < 1.0 && !xo;
    if (d.sorted && !accum) {
        // sorted / list form: tiles in grid order (at least 32 rows tall: a tile's rows cost next to nothing here) + the
        // candidate wavefronts, one launch
        int srows = trows < 32 ? 32 : trows;
        while (srows < 256 && bts_cdiv(mc, CT_COLS) * bts_cdiv(T, srows) > TILE_GOAL) srows *= 2;
        const int stiles = bts_cdiv(mc, CT_COLS) * bts_cdiv(T, srows);
        int cblocks = bts_cdiv((fa ? fa_mt + fa_dt : d.f_ntiles) * 4, CT_COLS / 64);
        if (cblocks > 512) cblocks = 512;
        switch (srows) {
        case 32: LAUNCH(tk_edges_sorted<32>, dim3(stiles + cblocks), dim3(CT_COLS), lap[0], stiles); break;
        case 64: LAUNCH(tk_edges_sorted<64>, dim3(stiles + cblocks), dim3(CT_COLS), lap[0], stiles); break;
        case 128: LAUNCH(tk_edges_sorted<128>, dim3(stiles + cblocks), dim3(CT_COLS), lap[0], stiles); break;
        default: LAUNCH(tk_edges_sorted<256>, dim3(stiles + cblocks), dim3(CT_COLS), lap[0], stiles); break;
        }
    } else if (xo && d.filter) {
        switch (trows) {
        case 2: LAUNCH((tk_edges_xo<1, 2>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 4: LAUNCH((tk_edges_xo<1, 4>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 8: LAUNCH((tk_edges_xo<1, 8>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 16: LAUNCH((tk_edges_xo<1, 16>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 32: LAUNCH((tk_edges_xo<1, 32>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 64: LAUNCH((tk_edges_xo<1, 64>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 128: LAUNCH((tk_edges_xo<1, 128>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        default: LAUNCH((tk_edges_xo<1, 256>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        }
        if (d.record) { LAUNCH((tk_edges<1, 2>), tile_grid, dim3(CT_COLS), lap[0], 0); }
    } else if (xo) {
        LAUNCH((tk_edges_xo<0, CT_ROWS>), tile_grid, dim3(CT_COLS), lap[0], 0);
    } else if (grid_form && d.motion_gate) {
        LAUNCH(tk_edges_grid<true>, dim3(bts_cdiv(T, 16)), dim3(64), lap[0]);
    } else if (d.filter && d.motion_gate) {
        switch (trows) {
        case 2: LAUNCH((tk_edges<1, 1, 2, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 4: LAUNCH((tk_edges<1, 1, 4, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 8: LAUNCH((tk_edges<1, 1, 8, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 16: LAUNCH((tk_edges<1, 1, 16, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 32: if (accum) { LAUNCH((tk_edges<1, 1, 32, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall_gate<32>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        case 64: if (accum) { LAUNCH((tk_edges<1, 1, 64, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall_gate<64>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        case 128: if (accum) { LAUNCH((tk_edges<1, 1, 128, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall_gate<128>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        default: if (accum) { LAUNCH((tk_edges<1, 1, 256, true>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall_gate_c<256>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        }
        if (d.record) { LAUNCH((tk_edges<1, 2>), tile_grid, dim3(CT_COLS), lap[0], 0); }
    } else if (grid_form) {
        // neighbour search on the detections' grid: opt-in (BTS_EDGES_GRID=1 when the tracker is created).  Measured on
        // MI355X it only beats the tile forms below at 20 000 x 5 000 (34 vs 40 us); at the headline size it is 2 us slower.
        LAUNCH(tk_edges_grid<false>, dim3(bts_cdiv(T, 16)), dim3(64), lap[0]);
    } else if (d.filter) {
        switch (trows) {
        case 2: LAUNCH((tk_edges<1, 1, 2>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 4: LAUNCH((tk_edges<1, 1, 4>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 8: LAUNCH((tk_edges<1, 1, 8>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 16: LAUNCH((tk_edges<1, 1, 16>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); break;
        case 32: if (accum) { LAUNCH((tk_edges<1, 1, 32>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall<32>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        case 64: if (accum) { LAUNCH((tk_edges<1, 1, 64>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall<64>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        case 128: if (accum) { LAUNCH((tk_edges<1, 1, 128>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall<128>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        default: if (accum) { LAUNCH((tk_edges<1, 1, 256>), dim3(ntile1), dim3(CT_COLS), lap[0], accum); } else { LAUNCH(tk_edges_tall<256>, dim3(ntile1), dim3(CT_COLS), lap[0], 0); } break;
        }
        if (d.record) { LAUNCH((tk_edges<1, 2>), tile_grid, dim3(CT_COLS), lap[0], 0); }
    } else if (d.motion_gate) {
        LAUNCH((tk_edges<1, 0, CT_ROWS, true>), tile_grid, dim3(CT_COLS), lap[0], 0);
    } else {
        LAUNCH((tk_edges<1, 0>), tile_grid, dim3(CT_COLS), lap[0], 0);
    }
    seg_end(BTS_SEG_K_EDGES1);
    seg_end(BTS_SEG_COST1);
    seg_begin(BTS_SEG_LAP1);
    bts_lap_problem prob[3] = {{lap[0], d.hdr + H_DIMS1, T, mc, d.match1, d.x1, d.y1, 1},
                               {lap[1], d.hdr + H_DIMS2, T, mc, d.match2, d.x2, d.y2, 1},
                               {lap[2], d.hdr + H_DIMS3, T, mc, d.match3, d.x3, d.y3, 1}};
    {
        hipEvent_t ev2[2];
        const bool want = prof_room && t->prof_seg == BTS_SEG_K_LAP;
        if (want) { ev2[0] = t->prof_ev[2 * t->prof_n]; ev2[1] = t->prof_ev[2 * t->prof_n + 1]; }
        rc = bts_lap_enqueue(&prob[0], 1, s, want ? ev2 : nullptr, t->lap_form, t->lap_helpers);
        if (rc) return rc;
        if (want) t->prof_n++;
    }
    // tie_mode 1: certificate (+ dense-JV replay where it fails) behind every assignment.  Cost source of a pair that is
    // not a feasible edge: the stage's boxes (rows: pool / stage-2 rows / unconfirmed, columns through the stage's index list)
    bts_lap_exact xs[3] = {t->xk[0], t->xk[0], t->xk[1]};
    if (t->cfg.tie_mode) {
        xs[0].abox = d.pool_tlbr; xs[0].af32 = nullptr; xs[0].bbox = d.det_tlbr; xs[0].bidx = d.hi_idx;
        xs[1].abox = d.r2_tlbr; xs[1].af32 = nullptr; xs[1].bbox = d.det_tlbr; xs[1].bidx = d.lo_idx;
        xs[2].abox = d.unc_tlbr; xs[2].af32 = d.unc_f32; xs[2].bbox = d.det_tlbr; xs[2].bidx = d.left_det;
        rc = bts_lap_exact_enqueue(&prob[0], &xs[0], 1, s);
        if (rc) return rc;
    }
    seg_end(BTS_SEG_LAP1);
    seg_begin(BTS_SEG_APPLY1);
    const int clear_blocks = (d.sorted && !fa) ? 16 : 0;      // (BTS_FILTER_AHEAD: the side stream clears the next buffer's lists and bitmap)      // (the sorted / list form: the candidate list and its bitmap bits are cleared here)
    if (d.nch > 1) { LAUNCH(tk_mid1<true>, dim3(mid1_jobs(d.nch) + bts_cdiv(mc, 4) + clear_blocks), dim3(1024), clear_blocks); }
    else { LAUNCH(tk_mid1<false>, dim3(mid1_jobs(1) + bts_cdiv(mc, 4) + clear_blocks), dim3(1024), clear_blocks); }       // stage-2/3 lists read pre-update states: before the stage-1 updates
    if (dense) {
        rc = bts_cos_enqueue(cos_kernel, d.body_curr, d.unc_slot, d.D, feats, d.left_det, d.D, d.D, d.hdr + H_DIMS3, T, mc,
                             d.sim, M, s);
        if (rc) return rc;
    }
    // (BTS_SMOOTH_DEFERRED: the previous step's tk_smooth reads the `curr` rows this step's applications are about to overwrite; it
    //  has had the whole front of this step to run on its own stream)
    if (fa && d.smooth_def && t->ev_smooth_used[ovp ^ 1]) BTS_HIP(hipStreamWaitEvent(s, t->ev_smooth[ovp ^ 1], 0));
    // stage-1 Kalman updates + stage-1 feature rows + cost tiles of stages 2 and 3: disjoint tracks, one launch
    seg_begin(BTS_SEG_K_MID2);
    {
        const dim3 g(kf_blocks + mc + 2 * ntile);
        if (d.filter) { LAUNCH(tk_mid2<1>, g, dim3(256), lap[1], lap[2], kf_blocks, mc, ntile); }
        else { LAUNCH(tk_mid2<0>, g, dim3(256), lap[1], lap[2], kf_blocks, mc, ntile); }
        if (d.filter && d.record) { LAUNCH((tk_edges<3, 2>), tile_grid, dim3(CT_COLS), lap[2], 0); }
    }
    seg_end(BTS_SEG_K_MID2);
    seg_end(BTS_SEG_APPLY1);
    // ---- second association (remaining tracked rows x low-score detections) and the unconfirmed
    //      tracks (x leftover high-score detections): independent problems, shared launches
    seg_begin(BTS_SEG_STAGE2);
    rc = bts_lap_enqueue(&prob[1], 2, s, nullptr, t->lap_form, t->lap_helpers);
    if (rc) return rc;
    if (t->cfg.tie_mode) {
        rc = bts_lap_exact_enqueue(&prob[1], &xs[1], 2, s);
        if (rc) return rc;
    }
    seg_begin(BTS_SEG_K_APPLY23);
    if (d.merge_early) {
        // (BTS_MERGE_EARLY: + births and the list merge, see tk_apply23_merge; the merge segment below is then empty)
        const int kf4 = bts_cdiv(kf_blocks, 4), ft4 = bts_cdiv(mc, 4);
        if (d.nch > 1) { LAUNCH(tk_apply23_merge<true>, dim3(merge_jobs(d.nch) + 2 * kf4 + 2 * ft4), dim3(1024), kf4, ft4); }
        else { LAUNCH(tk_apply23_merge<false>, dim3(merge_jobs(1) + 2 * kf4 + 2 * ft4), dim3(1024), kf4, ft4); }
    } else {
        LAUNCH(tk_apply23, dim3(2 * kf_blocks + 2 * mc), dim3(256), kf_blocks, mc);
    }
    seg_end(BTS_SEG_K_APPLY23);
    seg_end(BTS_SEG_STAGE2);
    // ---- births, list merge, duplicate removal
    seg_begin(BTS_SEG_MERGE);
    if (d.merge_early) { }
    else if (d.nch > 1) { LAUNCH(tk_merge<true>, dim3(merge_jobs(d.nch)), dim3(1024)); }
    else { LAUNCH(tk_merge<false>, dim3(merge_jobs(1)), dim3(1024)); }
    seg_end(BTS_SEG_MERGE);
    seg_begin(BTS_SEG_DEDUPE);
    {
        int drows = 2;                             // same tile-count rule for the duplicate tiles (tracked' x lost')
        while (drows < 64 && bts_cdiv(T, CT_COLS) * bts_cdiv(T, drows) > TILE_GOAL) drows *= 2;
        int nd = bts_cdiv(T, CT_COLS) * bts_cdiv(T, drows);
        if (nd > 1024) nd = 1024;                  // (tracked' x lost' is far smaller than T x T: persistent tiles walk the rest)
        switch (drows) {
        case 2: LAUNCH(tk_tail<2>, dim3(nd + mc), dim3(CT_COLS), nd); break;
        case 4: LAUNCH(tk_tail<4>, dim3(nd + mc), dim3(CT_COLS), nd); break;
        case 8: LAUNCH(tk_tail<8>, dim3(nd + mc), dim3(CT_COLS), nd); break;
        case 16: LAUNCH(tk_tail<16>, dim3(nd + mc), dim3(CT_COLS), nd); break;
        case 32: LAUNCH(tk_tail<32>, dim3(nd + mc), dim3(CT_COLS), nd); break;
        default: LAUNCH(tk_tail<64>, dim3(nd + mc), dim3(CT_COLS), nd); break;
        }
    }
    // (BTS_LISTS_AHEAD: + the pool and unconfirmed jobs of the next frame, nch chunks each, and its free-slot job)
    if (d.ahead) {
        if (d.nch > 1) { LAUNCH((tk_finalize<true, true>), dim3(4 * d.nch + 1), dim3(1024)); } else { LAUNCH((tk_finalize<false, true>), dim3(5), dim3(1024)); }
    } else if (d.nch > 1) { LAUNCH((tk_finalize<true, false>), dim3(2 * d.nch), dim3(1024)); }
    else { LAUNCH((tk_finalize<false, false>), dim3(2), dim3(1024)); }
    seg_end(BTS_SEG_DEDUPE);
    seg_end(BTS_SEG_STEP);
    if (ov) { BTS_HIP(hipEventRecord(t->ev_done[ovp], s)); t->ev_done_used[ovp] = true; }
    if (fa) t->mc_prev = mc;
    if (fa && d.smooth_def) {
        // the `smooth` rows of what this step rewrote, behind the step, on the third stream (one workgroup per table slot; see tk_smooth)
        BTS_HIP(hipStreamWaitEvent(t->side2, t->ev_done[ovp], 0));
        hipLaunchKernelGGL(tk_smooth, dim3(T), dim3(256), 0, t->side2, d); BTS_LAUNCH_CHECK();
        BTS_HIP(hipEventRecord(t->ev_smooth[ovp], t->side2));
        t->ev_smooth_used[ovp] = true;
    }
#undef LAUNCH
#undef LAUNCH_ON
    return BTS_OK;
}

int bts_tracker_step(bts_tracker *t, const int32_t *boxes, const float *scores, float *feats,
                     float *face_feats, int m, bts_stream stream) {
    BTS_REQUIRE(t && m >= 0, "tracker, m");
    BTS_ON_DEVICE(t->ctx);
    if (m > t->cfg.max_dets) {
        bts_set_error("%d detections exceed max_dets=%d", m, t->cfg.max_dets);
        return BTS_ECAPACITY;
    }
    BTS_REQUIRE(m == 0 || (boxes && scores && feats), "null detection arrays");
    BTS_REQUIRE((uintptr_t)boxes % 16 == 0 && (uintptr_t)feats % 16 == 0, "boxes / feats must be 16-byte aligned");
    BTS_REQUIRE(t->cfg.face_dim == 0 || m == 0 || face_feats, "face_feats required when face_dim > 0");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const bool graph = t->cfg.use_graph && t->dv.filter && !t->dv.record && t->prof_cap == 0;
    if (graph) {
        // this frame's descriptor: next slot of the pinned ring (wait if the GPU is a whole ring behind)
        const int slot = (int)(t->step_no % bts_tracker::RING);
        if (t->in_used[slot]) BTS_HIP(hipEventSynchronize(t->in_ev[slot]));
        t->h_in[slot] = InDesc{boxes, scores, feats, face_feats, m, (unsigned)(t->step_no + 1)};
        BTS_HIP(hipMemcpyAsync(t->d_in, &t->h_in[slot], sizeof(InDesc), hipMemcpyHostToDevice, s));
        t->dv.in = t->d_in;
        if (!t->gexec) {       // capture once, with capacity-sized grids, on the private streams
            hipGraph_t g = nullptr;
            BTS_HIP(hipStreamBeginCapture(t->cs[0], hipStreamCaptureModeRelaxed));
            rc = enqueue_step(t, t->cs[0], nullptr, nullptr, t->cfg.max_dets, true);
            const hipError_t e = hipStreamEndCapture(t->cs[0], &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) { if (g) (void)hipGraphDestroy(g); bts_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return BTS_EHIP; }
            BTS_HIP(hipGraphInstantiate(&t->gexec, g, nullptr, nullptr, 0));
            BTS_HIP(hipGraphDestroy(g));
        }
        BTS_HIP(hipGraphLaunch(t->gexec, s));
        BTS_HIP(hipEventRecord(t->in_ev[slot], s));
        t->in_used[slot] = true;
    } else {
        t->dv.in = nullptr;
        t->dv.iv = InDesc{boxes, scores, feats, face_feats, m, (unsigned)(t->step_no + 1)};
        rc = enqueue_step(t, s, feats, face_feats, m > 0 ? m : 1, false);
        if (rc) return rc;
    }
    t->step_no++;
    return BTS_OK;
}

int bts_tracker_inputs_resident(bts_tracker *t, int flag) {
    BTS_REQUIRE(t, "tracker");
    t->inputs_resident = flag != 0;
    return BTS_OK;
}

// The device raises H_ERR instead of stopping (a step never synchronises): every call that already has the header on
// the host turns a raised flag into a status, so a drop-in caller cannot miss a truncated frame.
static int device_error_status(const bts_tracker *t) {
    const int e = t->host_hdr[H_ERR];
    if (!e) return BTS_OK;
    if (e & ERR_TRACK_CAPACITY) {
        bts_set_error("track table full (max_tracks = %d): %d births were dropped at frame %d (now at frame %d); results diverge "
                      "from the reference from there on", t->cfg.max_tracks, t->host_hdr[31], t->host_hdr[30], t->host_hdr[H_FRAME]);
        return BTS_ECAPACITY;
    }
    if (e & ERR_LAP_EDGES) {
        bts_set_error("assignment solver: edge list overflow (max_edges) or helper time-out at or before frame %d; the "
                      "assignment of that frame is incomplete", t->host_hdr[H_FRAME]);
        return BTS_ECAPACITY;
    }
    bts_set_error("device-side wait timed out (flags 0x%x) at or before frame %d", e, t->host_hdr[H_FRAME]);
    return BTS_EHIP;
}

int bts_tracker_counts(bts_tracker *t, int32_t *counts12_host, bts_stream stream) {
    BTS_REQUIRE(t && counts12_host, "null argument");
    BTS_ON_DEVICE(t->ctx);
    BTS_HIP(hipMemcpyAsync(t->host_hdr, t->dv.hdr, H_SIZE * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    const int32_t *h = t->host_hdr;
    const int map[12] = {H_FRAME, H_NTRACKED, H_NLOST, H_NREMOVED, H_LASTID, H_NPOOL, H_NHI, H_NLO, H_NUNC, H_NR2, H_NLEFT, H_ERR};
    for (int k = 0; k < 12; ++k) counts12_host[k] = h[map[k]];
#ifdef BTS_HOST_EMU
    emu_check_redzones(t, "bts_tracker_counts");
#endif
    return BTS_OK;
}

int bts_tracker_export(bts_tracker *t, int which, int32_t *ints_host, double *mean_host,
                       double *tlbr_host, float *score_host, bts_stream stream) {
    BTS_REQUIRE(t && (which == 0 || which == 1), "tracker, which");
    BTS_ON_DEVICE(t->ctx);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tk_export, dim3(bts_cdiv(t->dv.T, 256)), dim3(256), 0, s, t->dv, which, t->x_ints, t->x_mean,
                       t->x_tlbr, t->x_score);
    BTS_LAUNCH_CHECK();
    BTS_HIP(hipMemcpyAsync(t->host_hdr, t->dv.hdr, H_SIZE * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BTS_HIP(hipStreamSynchronize(s));
    if (int rc = device_error_status(t)) return rc;
    const int n = which == 0 ? t->host_hdr[H_NTRACKED] : t->host_hdr[H_NLOST];
    if (n > 0) {
        if (ints_host) BTS_HIP(hipMemcpyAsync(ints_host, t->x_ints, sizeof(int32_t) * 8 * n, hipMemcpyDeviceToHost, s));
        if (mean_host) BTS_HIP(hipMemcpyAsync(mean_host, t->x_mean, sizeof(double) * 8 * n, hipMemcpyDeviceToHost, s));
        if (tlbr_host) BTS_HIP(hipMemcpyAsync(tlbr_host, t->x_tlbr, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, s));
        if (score_host) BTS_HIP(hipMemcpyAsync(score_host, t->x_score, sizeof(float) * n, hipMemcpyDeviceToHost, s));
        BTS_HIP(hipStreamSynchronize(s));
    }
    return n;
}

int bts_tracker_export_lists(bts_tracker *t, int32_t *ints_pinned, double *mean_pinned, double *tlbr_pinned,
                             float *score_pinned, int32_t *counts12_host, bts_stream stream) {
    BTS_REQUIRE(t && counts12_host, "tracker, counts");
    BTS_ON_DEVICE(t->ctx);
    hipStream_t s = (hipStream_t)stream;
    const size_t T = (size_t)t->dv.T;
    for (int which = 0; which < 2; ++which) {
        // (a null output still needs a valid address for the kernel's stores: the staging arrays of bts_tracker_export)
        hipLaunchKernelGGL(tk_export, dim3(bts_cdiv(t->dv.T, 256)), dim3(256), 0, s, t->dv, which,
                           ints_pinned ? ints_pinned + which * T * 8 : t->x_ints, mean_pinned ? mean_pinned + which * T * 8 : t->x_mean,
                           tlbr_pinned ? tlbr_pinned + which * T * 4 : t->x_tlbr, score_pinned ? score_pinned + which * T : t->x_score);
        BTS_LAUNCH_CHECK();
    }
    BTS_HIP(hipMemcpyAsync(t->host_hdr, t->dv.hdr, H_SIZE * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BTS_HIP(hipStreamSynchronize(s));
    const int32_t *h = t->host_hdr;
    const int map[12] = {H_FRAME, H_NTRACKED, H_NLOST, H_NREMOVED, H_LASTID, H_NPOOL, H_NHI, H_NLO, H_NUNC, H_NR2, H_NLEFT, H_ERR};
    for (int k = 0; k < 12; ++k) counts12_host[k] = h[map[k]];
    if (int rc = device_error_status(t)) return rc;
    return h[H_NTRACKED] + h[H_NLOST];
}

int bts_tracker_read_column(bts_tracker *t, int field, int32_t *out_host, bts_stream stream) {
    BTS_REQUIRE(t && out_host && field >= 0 && field <= 2, "tracker, field, out");
    BTS_ON_DEVICE(t->ctx);
    const Dev &d = t->dv;
    const int32_t *src = field == 0 ? d.track_id : field == 1 ? d.state : d.frame_id;
    BTS_HIP(hipMemcpyAsync(out_host, src, sizeof(int32_t) * (size_t)d.T, hipMemcpyDeviceToHost, (hipStream_t)stream));
    BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return BTS_OK;
}

int bts_tracker_export_mot(bts_tracker *t, double *rows_host, bts_stream stream) {
    BTS_REQUIRE(t && rows_host, "tracker, rows");
    BTS_ON_DEVICE(t->ctx);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tk_export_mot, dim3(bts_cdiv(t->dv.T, 256)), dim3(256), 0, s, t->dv, t->x_mot);
    BTS_LAUNCH_CHECK();
    BTS_HIP(hipMemcpyAsync(t->host_hdr, t->dv.hdr, H_SIZE * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BTS_HIP(hipStreamSynchronize(s));
    if (int rc = device_error_status(t)) return rc;
    const int n = t->host_hdr[H_NTRACKED];
    if (n > 0) {
        BTS_HIP(hipMemcpyAsync(rows_host, t->x_mot, sizeof(double) * 10 * n, hipMemcpyDeviceToHost, s));
        BTS_HIP(hipStreamSynchronize(s));
    }
    return n;
}

int bts_tracker_export_mot_async(bts_tracker *t, double *rows_pinned, int32_t *n_pinned, bts_stream stream) {
    BTS_REQUIRE(t && rows_pinned && n_pinned, "tracker, rows, n");
    BTS_ON_DEVICE(t->ctx);
    hipLaunchKernelGGL(tk_export_mot_direct, dim3(bts_cdiv(t->dv.T, 256)), dim3(256), 0, (hipStream_t)stream, t->dv, rows_pinned, n_pinned);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

// BTS_SMOOTH_DEFERRED: whoever reads `smooth` rows on `stream` is ordered behind the latest tk_smooth (the step enqueued it on the
// tracker's third stream right behind itself)
static int wait_for_smooth(bts_tracker *t, hipStream_t stream) {
    if (!t->dv.smooth_def || t->step_no == 0) return BTS_OK;
    const int last = (int)((t->step_no - 1) & 1);
    if (t->ev_smooth_used[last]) BTS_HIP(hipStreamWaitEvent(stream, t->ev_smooth[last], 0));
    return BTS_OK;
}

int bts_tracker_export_state(bts_tracker *t, int which, double *cov_host, float *body_curr_host, float *body_smooth_host,
                             float *face_curr_host, float *face_smooth_host, bts_stream stream) {
    BTS_REQUIRE(t && (which == 0 || which == 1), "tracker, which");
    BTS_ON_DEVICE(t->ctx);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = wait_for_smooth(t, s)) return rc;
    BTS_HIP(hipMemcpyAsync(t->host_hdr, t->dv.hdr, H_SIZE * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BTS_HIP(hipStreamSynchronize(s));
    if (int rc = device_error_status(t)) return rc;
    const int n = which == 0 ? t->host_hdr[H_NTRACKED] : t->host_hdr[H_NLOST];
    void *dst[5] = {cov_host, body_curr_host, body_smooth_host, face_curr_host, face_smooth_host};
    const size_t row_bytes[5] = {64 * sizeof(double), sizeof(float) * (size_t)t->dv.D, sizeof(float) * (size_t)t->dv.D,
                                 sizeof(float) * (size_t)t->dv.Df, sizeof(float) * (size_t)t->dv.Df};
    // gathered through the context's scratch arena in batches of at most 64 MiB (one field at a time)
    for (int f = 0; f < 5 && n > 0; ++f) {
        if (!dst[f] || row_bytes[f] == 0) continue;
        int batch = (int)(((size_t)64 << 20) / row_bytes[f]);
        batch = batch < 1 ? 1 : batch > n ? n : batch;
        if (int rc = t->ctx->scratch.ensure(row_bytes[f] * (size_t)batch)) return rc;
        for (int first = 0; first < n; first += batch) {
            const int cnt = n - first < batch ? n - first : batch;
            hipLaunchKernelGGL(tk_gather_state, dim3(cnt), dim3(256), 0, s, t->dv, which, f, first, cnt, (void *)t->ctx->scratch.base);
            BTS_LAUNCH_CHECK();
            BTS_HIP(hipMemcpyAsync((char *)dst[f] + row_bytes[f] * (size_t)first, t->ctx->scratch.base, row_bytes[f] * (size_t)cnt,
                                   hipMemcpyDeviceToHost, s));
            BTS_HIP(hipStreamSynchronize(s));
        }
    }
    return n;
}

int bts_tracker_read_slot(bts_tracker *t, int slot, int field, void *out_host, bts_stream stream) {
    BTS_REQUIRE(t && out_host && slot >= 0 && slot < t->dv.T && field >= 0 && field <= 6, "tracker, slot, field, out");
    BTS_ON_DEVICE(t->ctx);
    const Dev &d = t->dv;
    if (field == 6) {          // {track_id, state}: what became of a track that left both lists (removed vs dropped as a duplicate)
        int32_t *o = (int32_t *)out_host;
        BTS_HIP(hipMemcpyAsync(o, d.track_id + slot, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
        BTS_HIP(hipMemcpyAsync(o + 1, d.state + slot, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
        BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
        return BTS_OK;
    }
    BTS_REQUIRE(field < 4 || d.Df > 0, "the tracker carries no face features");
    if (int rc = wait_for_smooth(t, (hipStream_t)stream)) return rc;
    if (d.sf32 && field <= 1) {       // float32 state rows: widened on the host (the call's contract is float64)
        const int cnt = field == 0 ? 8 : 64;
        float tmp[64];
        const float *srcf = reinterpret_cast<const float *>(field == 0 ? d.mean : d.cov) + (size_t)slot * cnt;
        BTS_HIP(hipMemcpyAsync(tmp, srcf, sizeof(float) * cnt, hipMemcpyDeviceToHost, (hipStream_t)stream));
        BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
        for (int k = 0; k < cnt; ++k) ((double *)out_host)[k] = (double)tmp[k];
        return BTS_OK;
    }
    const void *src = field == 0 ? (const void *)(d.mean + (size_t)slot * 8) : field == 1 ? (const void *)(d.cov + (size_t)slot * 64)
                    : field == 2 ? (const void *)(d.body_curr + (size_t)slot * d.D) : field == 3 ? (const void *)(d.body_smooth + (size_t)slot * d.D)
                    : field == 4 ? (const void *)(d.face_curr + (size_t)slot * d.Df) : (const void *)(d.face_smooth + (size_t)slot * d.Df);
    const size_t bytes = field == 0 ? 8 * sizeof(double) : field == 1 ? 64 * sizeof(double) : sizeof(float) * (size_t)(field <= 3 ? d.D : d.Df);
    BTS_HIP(hipMemcpyAsync(out_host, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return BTS_OK;
}

int bts_tracker_gmc(bts_tracker *t, const double *h2x3_host, bts_stream stream) {
    BTS_REQUIRE(t && h2x3_host, "tracker, H");
    BTS_ON_DEVICE(t->ctx);
    Warp6 w;
    for (int k = 0; k < 6; ++k) w.h[k] = h2x3_host[k];
    hipLaunchKernelGGL(tk_gmc, dim3(bts_cdiv(t->dv.T, 64)), dim3(64), 0, (hipStream_t)stream, t->dv, w);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_debug_jobtrace(int32_t *out64_host) {
    BTS_REQUIRE(out64_host, "out");
    BTS_HIP(hipMemcpyFromSymbol(out64_host, HIP_SYMBOL(g_jobtrace), 64 * sizeof(int32_t)));
    return BTS_OK;
}

int bts_tracker_debug_filter(bts_tracker *t, int32_t *out8_host, bts_stream stream) {
    BTS_REQUIRE(t && out8_host, "tracker, out");
    BTS_ON_DEVICE(t->ctx);
    for (int k = 0; k < 16; ++k) out8_host[k] = 0;
    if (!t->dv.filter) return BTS_OK;
    // (BTS_FILTER_AHEAD: the counters of the last step's two side passes, summed by the kernel's own atomics)
    const int32_t *fs = (t->dv.fahead && t->fstat_b && t->step_no > 0) ? t->fstat_b + 16 * (int)((t->step_no - 1) & 1) : t->dv.fstat;
    BTS_HIP(hipMemcpyAsync(out8_host, fs, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    out8_host[14] = t->last_ftm; out8_host[15] = t->last_ftn;      // tile shape of the one-pass form (a K stage = that many rows x columns x 128)
    return BTS_OK;
}

int bts_tracker_debug_lap(bts_tracker *t, int stage, int32_t *hdr16_host, bts_stream stream) {
    BTS_REQUIRE(t && stage >= 1 && stage <= 3 && hdr16_host, "tracker, stage, out");
    BTS_ON_DEVICE(t->ctx);
    BTS_HIP(hipMemcpyAsync(hdr16_host, t->dv.lap_hdr + BTS_LAP_HDR_INTS * (stage - 1), 64 * sizeof(int32_t), hipMemcpyDeviceToHost,
                           (hipStream_t)stream));
    BTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return BTS_OK;
}

int bts_tracker_stage_result(bts_tracker *t, int stage, int32_t *x_host, int32_t *y_host,
                             double *cost_host, int32_t *n_rows, int32_t *n_cols, bts_stream stream) {
    BTS_REQUIRE(t && stage >= 1 && stage <= 3, "tracker, stage");
    BTS_ON_DEVICE(t->ctx);
    hipStream_t s = (hipStream_t)stream;
    BTS_HIP(hipMemcpyAsync(t->host_hdr, t->dv.hdr, H_SIZE * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BTS_HIP(hipStreamSynchronize(s));
    if (int rc = device_error_status(t)) return rc;
    const int32_t *h = t->host_hdr;
    const int rows = stage == 1 ? h[H_NPOOL] : stage == 2 ? h[H_NR2] : h[H_NUNC];
    const int cols = stage == 1 ? h[H_NHI] : stage == 2 ? h[H_NLO] : h[H_NLEFT];
    if (n_rows) *n_rows = rows;
    if (n_cols) *n_cols = cols;
    const Dev &d = t->dv;
    const int32_t *xs = stage == 1 ? d.x1 : stage == 2 ? d.x2 : d.x3;
    const int32_t *ys = stage == 1 ? d.y1 : stage == 2 ? d.y2 : d.y3;
    if (x_host && rows > 0) BTS_HIP(hipMemcpyAsync(x_host, xs, sizeof(int32_t) * rows, hipMemcpyDeviceToHost, s));
    if (y_host && cols > 0) BTS_HIP(hipMemcpyAsync(y_host, ys, sizeof(int32_t) * cols, hipMemcpyDeviceToHost, s));
    if (cost_host && rows > 0 && cols > 0) {
        BTS_REQUIRE(t->cfg.record_costs, "cost matrices are kept only when record_costs was set");
        const double *c = stage == 1 ? d.cost1 : stage == 2 ? d.cost2 : d.cost3;
        BTS_HIP(hipMemcpy2DAsync(cost_host, sizeof(double) * cols, c, sizeof(double) * d.Mx, sizeof(double) * cols, rows,
                                 hipMemcpyDeviceToHost, s));
    }
    BTS_HIP(hipStreamSynchronize(s));
    return BTS_OK;
}

}  // extern "C"
