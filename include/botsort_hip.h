/*
 * botsort_hip.h -- C ABI of libbotsort_hip.so: the MI355X (gfx950) BoT-SORT tracker hot path.
 *
 * The reference (PINTO0309/BoT-SORT-ONNX-TensorRT, demo_bottrack_onnx_tflite.py, cited D:line) has
 * no FFI layer: its boundary is the Python call surface of one module.  Each entry point below
 * names the reference function it replaces; the Python facade in bot-sort-onnx-tensorrt_amd/
 * binds them with ctypes (INTEGRATION.md shows the stub a maintainer would add to the reference).
 *
 * Conventions
 *   - every function returns int: 0 = BTS_OK, negative = bts_status; no C++ exception crosses;
 *     bts_last_error() gives a thread-local message for the last failure.
 *   - all array arguments are RAW DEVICE POINTERS owned by the caller (e.g. a PyTorch-ROCm
 *     tensor's data_ptr()), row-major, with explicit sizes / leading dimensions.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Calls only enqueue
 *     work; nothing synchronises unless the function says so.
 *   - Kalman state and association costs are float64 (what the reference computes, SURVEY.md
 *     8(a)); ReID features and similarities are float32; detections' boxes are float32 / int32.
 *   - `*_f32` flag arrays (uint8, may be NULL = all 0) mark rows whose values were float32 in the
 *     reference (never-predicted tracks, detections): the kernels then evaluate the same
 *     sub-expressions in float32 that NumPy scalar arithmetic does, so results are bit-identical.
 *   - one bts_context per (GPU, host thread); one bts_tracker per video stream. Neither is
 *     thread-safe; different objects are independent.  A context's scratch arena is shared by the
 *     standalone calls made on it (bts_lapjv, bts_associate_parts, bts_tracker_export_state): issue
 *     them on ONE stream (or synchronise between streams).
 *   - every call runs on its context's device and restores the caller's current device on return.
 *   - int32 box arrays are read as 16-byte rows and feature rows as float4: 16-byte alignment is
 *     required (checked; any hipMalloc / PyTorch allocation satisfies it).
 *   - device-side failures of an enqueued step (track table full, solver edge-list overflow, an
 *     in-kernel wait that timed out) cannot stop the stream; they raise a sticky flag that every
 *     synchronising call (bts_tracker_export*, _stage_result) returns as BTS_ECAPACITY / BTS_EHIP and
 *     bts_tracker_counts reports in counts[11].  Callers that never export must poll counts.
 */
#ifndef BOTSORT_HIP_H
#define BOTSORT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bts_context bts_context;
typedef struct bts_tracker bts_tracker;
typedef void *bts_stream;

enum bts_status {
    BTS_OK = 0,
    BTS_EINVAL = -1,    /* bad argument */
    BTS_EHIP = -2,      /* HIP runtime error (see bts_last_error) */
    BTS_ENOMEM = -3,    /* device/host allocation failed */
    BTS_ECAPACITY = -4, /* tracker capacity (tracks or detections) exceeded */
    BTS_ENODEVICE = -5  /* no usable gfx950 device */
};

/* track states, D:382-387 */
enum bts_track_state { BTS_NEW = 0, BTS_TRACKED = 1, BTS_LOST = 2, BTS_LONGLOST = 3, BTS_REMOVED = 4 };

/* which cosine-similarity kernel bts_cos_similarity / the tracker use */
enum bts_cos_kernel { BTS_COS_MFMA_F32 = 0, BTS_COS_VALU_F32 = 1 };

const char *bts_version(void);
const char *bts_last_error(void);
const char *bts_status_string(int status);

int bts_context_create(bts_context **out, int device_ordinal);
int bts_context_destroy(bts_context *ctx);
int bts_context_set_cos_kernel(bts_context *ctx, int which);

/* Pinned (page-locked, device-mapped) host memory for the asynchronous export below. */
int bts_pinned_alloc(void **out, size_t bytes);
int bts_pinned_free(void *p);

/* ------------------------------------------------------------------ Kalman filter (D:118-380) */

/* KalmanFilter.initiate, D:166-197. meas[n,4] = (cx,cy,w,h) float32 -> mean[n,8], cov[n,8,8].
 * The std-devs are evaluated in float32 as NumPy 2 does for a float32 measurement. */
int bts_kf_initiate(bts_context *ctx, const float *meas_xywh, int n, double *mean, double *cov,
                    bts_stream stream);

/* KalmanFilter.multi_predict (D:265-302) fused with STrack.multi_predict's rule (D:524-536):
 * rows with zero_vel[i] != 0 (nullable) get mean[6]=mean[7]=0 first.  In place.  Rows flagged in
 * row_f32 (nullable) evaluate the process noise in float32, as NumPy does for a float32 batch.
 * tlbr_out[n,4] (nullable) receives the predicted boxes (STrack.tlbr, D:624-642). */
int bts_kf_multi_predict(bts_context *ctx, double *mean, double *cov, const uint8_t *zero_vel,
                         const uint8_t *row_f32, int n, double *tlbr_out, bts_stream stream);

/* KalmanFilter.project, D:236-263 -> pmean[n,4], pcov[n,4,4]. */
int bts_kf_project(bts_context *ctx, const double *mean, const double *cov, const uint8_t *row_f32,
                   int n, double *pmean, double *pcov, bts_stream stream);

/* KalmanFilter.update (D:304-336) over a match list: for q in [0,k) the track row trk_idx[q]
 * (of mean/cov) is corrected in place with measurement meas_xywh[meas_idx[q]] (float32 cx,cy,w,h).
 * trk_idx entries must be distinct.  4x4 innovation Cholesky + two triangular solves per row. */
int bts_kf_update(bts_context *ctx, double *mean, double *cov, const uint8_t *row_f32,
                  const int32_t *trk_idx, const float *meas_xywh, const int32_t *meas_idx, int k,
                  bts_stream stream);

/* STrack.multi_gmc (D:538-554): camera-motion warp of n states in place, H = 2x3 affine (HOST
 * pointer to 6 doubles, row-major): x <- kron(I4, H[:, :2]) x, centre += H[:, 2], P <- R8 P R8^T. */
int bts_kf_multi_gmc(bts_context *ctx, double *mean, double *cov, int n, const double *h2x3_host, bts_stream stream);

/* KalmanFilter.gating_distance (D:338-380) for every (track, measurement) pair -> out[n,m].
 * metric: 0 = 'maha', 1 = 'gaussian'.  meas[m,4] float64. */
int bts_kf_gating(bts_context *ctx, const double *mean, const double *cov, int n, const double *meas,
                  int m, int only_position, int metric, double *out, bts_stream stream);

/* -------------------------------------------------------------- association costs (D:1539-1761) */

/* iou_distance / bbox_ious / bbox_iou, D:1695-1761: cost[i*ld+j] = 1 - IoU(a[i], b[j]), tlbr boxes. */
int bts_iou_distance(bts_context *ctx, const double *a_tlbr, const uint8_t *a_f32, int n,
                     const double *b_tlbr, const uint8_t *b_f32, int m, double *cost, int ld,
                     bts_stream stream);

/* sim[i*ld+j] = max(0, <trk_feats[i,:], det_feats[j,:]>) in float32 -- the D:1599 definition that
 * also stands in for the ReID ONNX graphs' similarity output (D:1453-1483). Rows have stride d. */
int bts_cos_similarity(bts_context *ctx, const float *trk_feats, int n, const float *det_feats, int m,
                       int d, float *sim, int ld, bts_stream stream);

/* Fused first-association cost, D:1539-1554 (face_sim nullable = all zeros; the D:1481-1483
 * "0.9999999" fix-up is applied to face_sim on the fly). */
int bts_fuse_cost_stage1(bts_context *ctx, const double *iou_d, int ld_iou, const float *body_sim,
                         const float *face_sim, int ld_sim, int n, int m, double proximity_thresh,
                         double appearance_thresh, double *cost, int ld_cost, bts_stream stream);

/* Fused unconfirmed-track cost, D:1590-1602. */
int bts_fuse_cost_stage3(bts_context *ctx, const double *iou_d, int ld_iou, const float *body_sim,
                         int ld_sim, int n, int m, double proximity_thresh, double appearance_thresh,
                         double *cost, int ld_cost, bts_stream stream);

/* linear_assignment -> lap.lapjv(cost, extend_cost=True, cost_limit), D:1682-1693.
 * x[n] = column of row i or -1; y[m] = row of column j or -1. Exact (optimal) solution. */
int bts_lapjv(bts_context *ctx, const double *cost, int n, int m, int ld, double cost_limit,
              int32_t *x, int32_t *y, bts_stream stream);

/* The same call with the reference's behaviour on EXACT COST TIES as well (where the optimum is not unique lap.lapjv's
 * answer is an artefact of its dense Jonker-Volgenant scan orders on the (n+m)^2 extended matrix): bts_lapjv's solver, then
 * a uniqueness certificate and -- only where it fails, or with force_replay != 0 -- lap 0.4.0's dense JV replayed on the
 * implicit extended matrix (csrc/lap_exact.h).  cost_limit >= 0; (n+m)^2 < 2^31.
 * Cost scale: "tied" means another assignment within an ABSOLUTE 1e-9 of the optimum (lapx::TIE_TOL), chosen for the
 * tracker's costs, which live in [0, 1]; matrices on another scale should be rescaled by the caller: for much SMALLER
 * costs the tolerance is loose and only triggers replays that were not needed (the replay is exact); for much LARGER ones
 * rounding can hide a tie from the certificate, and the result is then the fast solver's optimum (one of the optima, as
 * bts_lapjv returns it) rather than lap's.
 * info3_host (nullable; synchronises when given): {certificate said "tied", replay ran, error flags (0 = none)}. */
int bts_lapjv_exact(bts_context *ctx, const double *cost, int n, int m, int ld, double cost_limit,
                    int32_t *x, int32_t *y, int force_replay, int32_t *info3_host, bts_stream stream);

/* STrack.update_body_features / update_face_features, D:492-514, for k (track,det) pairs
 * (trk_idx / det_idx nullable = identity; rows of length d, float32):
 * curr[trk] = det row; smooth[trk] = normalise(alpha*smooth + (1-alpha)*det), or normalise(det)
 * when has_smooth[trk]==0 (then set to 1).  alpha is the python float of D:473 (0.9). */
int bts_feature_ema(bts_context *ctx, float *curr, float *smooth, uint8_t *has_smooth,
                    const int32_t *trk_idx, const float *det_feats, const int32_t *det_idx, int k, int d,
                    double alpha, bts_stream stream);

/* The in-place L2 renormalisation the reference applies to a detection's feature row when its
 * STrack record is built (first update_*_features call aliases and normalises, D:496-502, Q8):
 * feats[idx[q], :] /= ||.||_2 for q in [0,k) (idx nullable = identity). */
int bts_feature_renorm(bts_context *ctx, float *feats, const int32_t *idx, int k, int d, bts_stream stream);

/* remove_duplicate_stracks, D:1665-1680: drop_a[n] / drop_b[m] flags (uint8, overwritten) for
 * pairs with 1-IoU < thresh; age_* = frame_id - start_frame. */
int bts_dedupe_pairs(bts_context *ctx, const double *a_tlbr, const uint8_t *a_f32, const int32_t *age_a,
                     int n, const double *b_tlbr, const uint8_t *b_f32, const int32_t *age_b, int m,
                     double thresh, uint8_t *drop_a, uint8_t *drop_b, bts_stream stream);

/* Diagnostics of the appearance pre-filter's int8 shadows (the pre-filter never decides a cost -- it selects the pairs
 * that get the exact float32 dot product -- but its bound must hold; tests/test_gpu_kernels.py):
 * bts_debug_i8_rows: x[n,d] float32 -> q[n,d] int8 shadows and meta[n,20] records (scale, measured residual, prefix norms:
 *   csrc/feat_math.h) for the similarity `gate` (= 1 - appearance_thresh);
 * bts_debug_i8_pairs: for pair p = (ia[p], ib[p]) and every checkpoint c of the K loop (the last = the full row)
 *   excluded[p,5] = 1 when the prefix's integer dot product proves <a, b> < gate, dots[p,5] = that dot product;
 * bts_debug_i8_layout: out[0..3] checkpoint stages for dimension d, [4] stages, [5] elements per stage, [6] floats per
 *   record, [7] checkpoints. */
int bts_debug_i8_rows(bts_context *ctx, const float *x, int n, int d, double gate, int8_t *q, float *meta, bts_stream stream);
int bts_debug_i8_pairs(bts_context *ctx, const int8_t *qa, const float *meta_a, const int8_t *qb, const float *meta_b,
                       const int32_t *ia, const int32_t *ib, int n_pairs, int d, uint8_t *excluded, int32_t *dots, bts_stream stream);
int bts_debug_i8_layout(int d, int32_t *out8);
/* Launch-floor probe (scripts/launch_floor.py): `launches` back-to-back launches of a kernel whose workgroups follow
   `links` dependent loads through buf (int32[>= 8192], zero-filled). */
/* Durations of the tracker's single-workgroup list jobs (library built with `make TRACE=1`; zeros otherwise):
   100 MHz ticks at out[kernel * 8 + job], kernels 0..3 = tk_front, tk_mid1, tk_merge, tk_finalize. */
int bts_debug_jobtrace(int32_t *out64_host);
int bts_debug_launch_floor(bts_context *ctx, int32_t *buf, int launches, int blocks, int threads, int links, bts_stream stream);

/* Detection-side part association: find_most_relevant_object (D:1763-1791, IoU of integer boxes per
 * bbox_iou_by_box D:1715-1729) applied to each base box in order, `picks` times per base (1 for
 * head->body / face->head D:1379-1393, 2 for the hand1/hand2 picks D:1401-1409).  Each pick takes the
 * unused target with the largest IoU (> 0), ties by the smaller centre distance, then by list order,
 * and marks it in target_used[nt] (uint8, in/out: targets already taken by an earlier pass stay taken).
 * boxes are int32 (x1, y1, x2, y2) rows, centres int32 (cx, cy) rows (the Box.cx / Box.cy the detector
 * stored), all 16-/8-byte aligned device arrays; out_idx[nb * picks] = target index or -1. */
int bts_associate_parts(bts_context *ctx, const int32_t *base_boxes, const int32_t *base_centres, int nb,
                        const int32_t *target_boxes, const int32_t *target_centres, int nt,
                        uint8_t *target_used, int picks, int32_t *out_idx, bts_stream stream);

/* -------------------------------------------------- whole-frame tracker, BoTSORT.update D:1414-1639 */

/* How the stage-1/3 appearance term (D:1539-1554, D:1590-1602) is evaluated inside the tracker:
 * FILTER: int8-MFMA pre-filter over 8-bit shadows of the features (a rigorous upper bound on every pair's dot product,
 *         evaluated on a prefix of the rows where that already decides) + exact float32 recomputation of every pair
 *         that could pass the appearance gate (results identical to DENSE up to float32 summation order; no
 *         N x M matrix is materialised);
 * DENSE:  the dense float32 similarity and cost matrices, as the reference forms them. */
enum bts_appearance_mode { BTS_APPEARANCE_FILTER = 0, BTS_APPEARANCE_DENSE = 1 };

#define BTS_ABI_VERSION 5
typedef struct bts_tracker_config {
    int32_t struct_size;     /* = bts_tracker_config_size() of the library the caller was written against: filled in by
                                bts_tracker_default_config; create / memory_bytes reject any other value (BTS_EINVAL), so a
                                binding whose mirror of this struct is stale fails at the first call instead of reading
                                fields at the wrong offsets */
    int32_t abi_version;     /* = BTS_ABI_VERSION of the header the caller was built against (set by
                                bts_tracker_default_config; checked like struct_size: a field that changes MEANING without
                                changing the size bumps it) */
    int32_t max_tracks;      /* capacity of the resident track table (tracked + lost + unconfirmed) */
    int32_t max_dets;        /* max detections per frame */
    int32_t feat_dim;        /* body ReID feature dimension D */
    int32_t face_dim;        /* face ReID dimension, 0 = no face features */
    int32_t frame_rate;      /* D:1258; buffer_size = int(frame_rate/30*track_buffer) */
    int32_t track_buffer;    /* D:1272, default 300 */
    double track_high_thresh;   /* 0.40 D:1268 */
    double track_low_thresh;    /* 0.1  D:1269 */
    double new_track_thresh;    /* 0.9  D:1270 */
    double match_thresh;        /* 0.8  D:1271 */
    double second_match_thresh; /* 0.5  D:1571 */
    double unconfirmed_match_thresh; /* 0.7 D:1604 */
    double proximity_thresh;    /* 0.5  D:1274 */
    double appearance_thresh;   /* 0.25 D:1275 */
    double dedupe_thresh;       /* 0.15 D:1667 */
    double ema_alpha;           /* 0.9  D:473 */
    int32_t record_costs;       /* keep the three cost matrices of the last frame for inspection */
    int32_t appearance_mode;    /* enum bts_appearance_mode */
    int32_t use_graph;          /* replay each step as one hipGraph (captured on the first step; grids sized for
                                   max_dets; needs appearance_mode FILTER and record_costs 0) */
    int32_t motion_gate;        /* 0 (default, = the reference: KalmanFilter.gating_distance D:338-380 is never called by
                                   its update loop) | 1: stage-1 pairs whose squared Mahalanobis distance between the
                                   track's predicted state and the detection's (cx,cy,w,h) exceeds chi2inv95[4] = 9.4877
                                   (D:139-149) cannot be matched (cost = +inf, upstream BoT-SORT's gate_cost_matrix);
                                   applied to the sparse edge list, the N x M distance matrix is never formed
                                   (record_costs shows the cost matrix before the gate) */
    int32_t max_edges;          /* capacity of the assignment solver's edge lists (pairs with cost <= the stage's match
                                   threshold); 0 = automatic: min(max_tracks * max_dets, 64 * (max_tracks + max_dets)), or
                                   max_tracks * max_dets when a stage threshold is >= 1 (every pair feasible).
                                   Overflow raises the sticky error flag (BTS_ECAPACITY) */
    int32_t tie_mode;           /* 0 (default): the fast solver's optimum -- equal to lap.lapjv's whenever the optimum is
                                   unique, one of the optima on exact cost ties.
                                   1: the reference's result on ties as well -- every assignment is followed by a
                                   uniqueness certificate and, where it fails, by the dense Jonker-Volgenant code of
                                   lap 0.4.0 replayed on the implicit (N+M)^2 extended matrix (csrc/lap_exact.h): two more
                                   launches per frame, and a frame WITH ties costs what the dense algorithm costs.
                                   Needs motion_gate 0 and appearance_thresh <= match_thresh, unconfirmed_match_thresh */
    int32_t kalman_dtype;       /* enum bts_dtype: BTS_F64 (default) = the reference's float64 state, bit-parity tests apply.
                                   BTS_F32 = the arithmetic mode BASELINE config 5 names ("fp32 Kalman"): mean / covariance are
                                   STORED as float32 (half the Kalman bytes of a step); the filter arithmetic of a predict /
                                   update still runs in float64 registers on the widened values and rounds once at the store.
                                   Not bit-identical to the reference: parity is an ID-match RATE plus the 1e-4 bound on the
                                   state (tests/test_zz_gpu_round4.py) */
    int32_t cost_dtype;         /* BTS_F64 (default) | BTS_F16: every association cost is rounded to IEEE binary16 (round to
                                   nearest even, straight from float64) before it is compared with the stage's threshold and
                                   handed to the solver -- config 5's "fp16 cost matrix" (this design has no matrix to store:
                                   the rounding is what the mode means for the results).  Needs tie_mode 0 */
} bts_tracker_config;
enum bts_dtype { BTS_F64 = 0, BTS_F32 = 1, BTS_F16 = 2 };

/* sizeof(bts_tracker_config) in THIS library.  A binding in another language checks its mirror of the struct against it
   BEFORE it hands a buffer to bts_tracker_default_config (which writes that many bytes). */
int bts_tracker_config_size(void);

/* Fill cfg with the reference's hard-coded values (D:1268-1277); sets cfg->struct_size. */
int bts_tracker_default_config(bts_tracker_config *cfg);

/* Device bytes bts_tracker_create(cfg) allocates for the track table and its per-frame work arrays (one block; the
   shared scratch arena of the context is not included).  Host-only: needs no GPU and no context -- for sizing how many
   streams fit one GPU's 288 GB.  Reads the same environment switches as create. */
int bts_tracker_memory_bytes(const bts_tracker_config *cfg, uint64_t *out_bytes);

int bts_tracker_create(bts_context *ctx, const bts_tracker_config *cfg, bts_tracker **out);

/* EXECUTION FORMS.  Several kernels / launch structures of a step exist in more than one form with bit-identical RESULTS
 * (which one is faster depends on the table size and the scene): `forms` = "KEY=VALUE,KEY=VALUE,..." selects them for this
 * tracker; a key that is absent falls back to the process environment variable of the same name, then to the default;
 * "KEY=" (empty value) switches a form off whatever the environment says, and so do the values 0 / false / off / no of a
 * boolean form.  NULL / "" = bts_tracker_create.
 *   BTS_EDGES_SORTED=1   stage-1 edges: detections in grid order + candidate list (tk_edges_sorted) instead of the tile form
 *   BTS_EDGES_GRID=1     stage-1 edges by neighbour search on the detection grid
 *   BTS_LISTS_AHEAD=1    the next frame's pool / unconfirmed / free lists are built in this frame's last launch
 *   BTS_MERGE_EARLY=1    births and the list merge in the launch that applies stages 2 and 3 (one launch less)
 *   BTS_RENORM_FUSED=1   a detection row matched in stage 1 is renormalised by the workgroup that absorbs it into its track
 *   BTS_CVT_OVERLAP=1    the detection-row conversion of a frame runs on a side stream (see bts_tracker_inputs_resident)
 *   BTS_PREDICT_FIRST=1  the Kalman-predict workgroups in front of the pre-filter's persistent workgroups
 *   BTS_PREDICT_FRONT=1  (with BTS_LISTS_AHEAD) the Kalman predict in the frame's first launch, none in the pre-filter's
 *   BTS_SHADOW_PLANES=1  the int8 shadows of the feature rows in K-stage planes (a pre-filter tile's 128 rows x 128 bytes of a
 *                        stage are one contiguous block for consecutive slots) instead of row-major
 *   BTS_FILTER_GUESS=1   a pre-filter workgroup predicts its FIRST tile, too, to stop at the first checkpoint (prefetch only)
 *   BTS_LAP_FORM=row|col component solvers of the assignment kernel (col: default)
 *   BTS_LAP_SELF=1       every workgroup of an assignment launch prepares the problem itself (no hand-over between workgroups)
 *   BTS_LAP_HELPERS=n    helper workgroups per assignment problem (default 15 / 63)
 *   BTS_CHUNK_ROWS=n     list jobs split into chunks of n entries (default: unsplit up to 6144 tracks, then 2048)
 *   BTS_FILTER_PW=n      persistent workgroups of the pre-filter (default 512)
 *   BTS_FILTER_TILE=160x128|128x64  tile shape of the one-pass pre-filter (default 128x128): 160 x 128 = two tile rounds instead of
 *                        three at 10 000 x 2 000; 128 x 64 = three resident workgroups per CU (768 persistent workgroups)
 *   BTS_FILTER_AHEAD=1   THROUGHPUT mode (round 6; body features, tile or sorted consumers, eager launches; implies BTS_LISTS_AHEAD,
 *                        BTS_PREDICT_FRONT and the side stream of BTS_CVT_OVERLAP): the appearance pre-filter of frame f runs on
 *                        the side stream AHEAD of step f -- one pass over all table slots (valid for every track step f-1 does
 *                        not rewrite) and one pass detections(f-1) x detections(f) (a track rewritten in step f-1 holds exactly
 *                        the shadow and record of the detection it absorbed) -- and the step's own stream carries no pre-filter
 *                        launch.  Pays only when the caller enqueues steps back to back and has promised residency
 *                        (bts_tracker_inputs_resident); results are identical either way.  BTS_FILTER_AHEAD_PW=n: persistent
 *                        workgroups per side pass (default 256: one per CU)
 *   BTS_SMOOTH_DEFERRED=1  (with BTS_FILTER_AHEAD) a step writes a rewritten track's `curr` feature row only; the `smooth` EMA row --
 *                        which nothing in a step reads -- is brought up to date by a kernel on a third stream behind the step, in
 *                        front of the next step's first feature write.  bts_tracker_export_state / bts_tracker_read_slot are
 *                        ordered behind it.  Same arithmetic on the same values, later
 *   BTS_I8_FIRST16=k     first prefix checkpoint of the pre-filter, in sixteenths of the feature width (3..12)
 *   BTS_TILE_GOAL=n      workgroup count the stage-1 tile height aims at (default 768)
 * None of them changes a result (tests/test_zz_gpu_round4.py, test_zz_gpu_round5.py: identical assignment indices, lists and state per frame);
 * the Python package's `tune` module times them on the caller's own frames and keeps what is fastest AND identical.
 * bts_tracker_forms writes the forms IN EFFECT (a requested form the configuration rules out does not appear) as text into
 * buf (NUL-terminated, truncated to capacity) and returns its full length.  filter_tile is the shape of the ONE-PASS pre-filter
 * (128x128 when the tracker has no pre-filter: dense appearance mode, face features, sorted edges); steps small enough for the
 * split-K form do not run that kernel -- bts_tracker_debug_filter's out[14], out[15] say what the last step ran.
 * lap_helpers is the resolved count (BTS_LAP_HELPERS, or the automatic one for the table size), fixed for the tracker's life. */
int bts_tracker_create_forms(bts_context *ctx, const bts_tracker_config *cfg, const char *forms, bts_tracker **out);
int bts_tracker_memory_bytes_forms(const bts_tracker_config *cfg, const char *forms, uint64_t *out_bytes);
int bts_tracker_forms(const bts_tracker *trk, char *buf, int capacity);
int bts_tracker_destroy(bts_tracker *trk);
/* Back to frame 0, no tracks, ID counter 0 (what constructing a new BoTSORT does, D:1261-1266). */
int bts_tracker_reset(bts_tracker *trk, bts_stream stream);

/* One BoTSORT.update on pre-extracted arrays (tracking half, D:1414-1639), fully on device:
 *   boxes[m,4]  int32 tlbr pixels (what the detector emits, D:1009-1012)
 *   scores[m]   float32
 *   feats[m,D]  float32, L2-normalised; rows of used detections are renormalised IN PLACE (Q8)
 *   face_feats[m,Df] float32 or NULL
 * Enqueues the whole association cascade on `stream`; no host synchronisation. */
int bts_tracker_step(bts_tracker *trk, const int32_t *boxes, const float *scores, float *feats,
                     float *face_feats, int m, bts_stream stream);

/* A promise about the arrays handed to the FOLLOWING bts_tracker_step calls: flag != 0 = they are complete in device memory
 * when the call is made (no upload or encoder kernel still pending on any stream writes them), flag == 0 (default) = they
 * are ordered on `stream` like any other work.  Only matters to the opt-in pipelining of the detection-row conversion
 * (BTS_CVT_OVERLAP=1 when the tracker is created): with the promise the side stream converts frame t + 1's rows under frame
 * t's kernels; without it the side stream first waits for an event recorded on `stream` at the step's entry. */
int bts_tracker_inputs_resident(bts_tracker *trk, int flag);

/* Sizes after the last enqueued step (synchronises `stream`):
 * counts[0]=frame_id [1]=n_tracked [2]=n_lost [3]=n_removed_total [4]=last_track_id
 * [5]=pool rows N of the last frame [6]=high dets [7]=low dets [8]=unconfirmed rows
 * [9]=stage-2 rows [10]=stage-3 cols [11]=error flags (0 = none) */
int bts_tracker_counts(bts_tracker *trk, int32_t *counts12_host, bts_stream stream);

/* Copy the tracked (which=0) or lost (which=1) list, in list order, to HOST buffers (each
 * nullable, sized for max_tracks rows).  Synchronises `stream`.  Returns the row count or <0.
 * ints[row,8] = track_id, state, is_activated, frame_id, start_frame, tracklet_len, det_index, slot
 * mean[row,8] float64; tlbr[row,4] float64 (STrack.tlbr); score[row] float32. */
int bts_tracker_export(bts_tracker *trk, int which, int32_t *ints_host, double *mean_host,
                       double *tlbr_host, float *score_host, bts_stream stream);

/* BOTH lists in one call, for a per-frame drop-in caller (the BoTSORT facade's mirror of tracked_stracks / lost_stracks,
 * D:1629-1639): the export kernels write straight into PINNED host buffers of the caller (bts_pinned_alloc; each
 * nullable), list 0 (tracked) in rows [0, max_tracks), list 1 (lost) in rows [max_tracks, 2 max_tracks); ONE stream
 * synchronisation, no staging copies.  Layouts per row as in bts_tracker_export.  counts12_host as bts_tracker_counts.
 * The sticky device error is turned into a status exactly as by bts_tracker_export.  Returns n_tracked + n_lost or <0. */
int bts_tracker_export_lists(bts_tracker *trk, int32_t *ints_pinned, double *mean_pinned, double *tlbr_pinned,
                             float *score_pinned, int32_t *counts12_host, bts_stream stream);

/* One int32 column of the WHOLE table (max_tracks entries, slot order) to HOST: field 0 track_id, 1 state, 2 frame_id.
 * What became of tracks that left both lists (removed vs dropped as a duplicate, D:1609-1637) for many slots at once --
 * the bulk form of bts_tracker_read_slot(field 6).  Synchronises. */
int bts_tracker_read_column(bts_tracker *trk, int field, int32_t *out_host, bts_stream stream);

/* Batched consumer format (what the reference's main loop reads per track, D:2135-2185):
 * one MOTChallenge row per entry of the tracked list -- frame, id, left, top, width, height, conf, -1, -1, -1
 * (float64 [rows,10], HOST buffer sized for max_tracks rows).  Synchronises.  Returns the row count or <0. */
int bts_tracker_export_mot(bts_tracker *trk, double *rows_host, bts_stream stream);

/* The same rows without any host synchronisation: written by the export kernel straight into PINNED host memory
 * (bts_pinned_alloc: rows_pinned sized for max_tracks x 10 doubles, n_pinned for 4 int32) behind the steps already
 * enqueued on `stream`.  n_pinned[0] = row count, [1] = frame id, [2] = error flags.  The caller orders its read with
 * an event / stream wait of its own (e.g. a ring of buffers read one frame late): a D:2135-2185 consumer does not have
 * to stall the stream per frame. */
int bts_tracker_export_mot_async(bts_tracker *trk, double *rows_pinned, int32_t *n_pinned, bts_stream stream);

/* The rest of the per-track state a caller of the reference can read (STrack.covariance D:467, body/face curr and smooth
 * features D:475-490), rows in the order of the tracked (which=0) / lost (which=1) list, to HOST buffers (each nullable;
 * sized for the list's rows x 64 doubles / x feat_dim / x face_dim floats).  Synchronises.  Returns the row count or <0. */
int bts_tracker_export_state(bts_tracker *trk, int which, double *cov_host, float *body_curr_host, float *body_smooth_host,
                             float *face_curr_host, float *face_smooth_host, bts_stream stream);

/* One table slot's row of one field to HOST (slot = ints[row,7] of bts_tracker_export; synchronises):
 * field 0 mean[8] f64, 1 covariance[64] f64, 2 body_curr[D] f32, 3 body_smooth[D], 4 face_curr[Df], 5 face_smooth[Df],
 * 6 {track_id, state} int32[2]. */
int bts_tracker_read_slot(bts_tracker *trk, int slot, int field, void *out_host, bts_stream stream);

/* Last frame's association results to HOST (synchronises): for stage s in {1,2,3}
 * x_host[rows], y_host[cols] (nullable); returns rows via *n_rows, cols via *n_cols.
 * If record_costs was set, cost_host (nullable) receives rows*cols float64, row-major. */
int bts_tracker_stage_result(bts_tracker *trk, int stage, int32_t *x_host, int32_t *y_host,
                             double *cost_host, int32_t *n_rows, int32_t *n_cols, bts_stream stream);

/* STrack.multi_gmc applied to every track of the tracked and lost lists (the reference's call
 * sites D:1534-1536 are commented out; call it between steps when a camera-motion estimate exists). */
int bts_tracker_gmc(bts_tracker *trk, const double *h2x3_host, bts_stream stream);

/* Counters of the appearance pre-filter for the last step (synchronises): out[0] K stages (128 elements) run over all
 * tiles, [1] tiles, [2] tiles that stopped at a prefix checkpoint, [3] candidate pairs handed to the exact refinement,
 * [5..7] clock cycles / 64 summed over the kernel's workgroups: in candidate tests, waiting for operand stages, in total,
 * [14], [15] rows x columns of the tiles the step's pre-filter launch used (0 = 128: BTS_FILTER_TILE shapes only the one-pass
 * kernel -- a step small enough for the split-K form, decided per step from the live detection count, reports 128 x 128). */
int bts_tracker_debug_filter(bts_tracker *trk, int32_t *out16_host, bts_stream stream);

/* Counters of the assignment solver for stage s of the last step (synchronises): out[0] feasible
 * edges, [1] components of the non-trivial part, [2] overflow flag, [6] edges that are not isolated
 * pairs, [8] form used (0 = isolated pairs only, 1 = LDS form, 2 = global form). */
int bts_tracker_debug_lap(bts_tracker *trk, int stage, int32_t *hdr16_host, bts_stream stream);

/* Measurement hook (bench.py's roofline leg): wrap one kernel group of every following
 * bts_tracker_step in a hipEvent pair recorded on the step's stream.  `capacity` pairs are kept
 * (steps beyond that are not timed); capacity 0 disables.  Events are created here, never in a step. */
enum bts_segment {
    BTS_SEG_PREP = 0, BTS_SEG_PREDICT = 1, BTS_SEG_COS1 = 2, BTS_SEG_COST1 = 3, BTS_SEG_LAP1 = 4, BTS_SEG_APPLY1 = 5,
    BTS_SEG_STAGE2 = 6, BTS_SEG_STAGE3 = 7, BTS_SEG_MERGE = 8, BTS_SEG_DEDUPE = 9,
    /* single kernels */
    BTS_SEG_K_FILTER = 20 /* pre-filter GEMM + Kalman predict */, BTS_SEG_K_LAP = 21 /* stage-1 assignment */,
    BTS_SEG_K_EDGES1 = 23 /* stage-1 cost tiles + exact refine */, BTS_SEG_K_MID2 = 24 /* stage-1 updates + stage-2/3 tiles */,
    BTS_SEG_K_APPLY23 = 25,
    BTS_SEG_STEP = 100
};
int bts_tracker_profile_enable(bts_tracker *trk, int segment, int capacity);
/* Record events only on every `stride`-th step (default 1): an event pair costs the stream a few microseconds. */
int bts_tracker_profile_stride(bts_tracker *trk, int stride);
/* Synchronises on the recorded events; returns their summed duration and count, then clears them. */
int bts_tracker_profile_read(bts_tracker *trk, double *total_ms, int32_t *count);

/* Device pointer to body_curr_feature / covariance rows etc. is intentionally not exposed:
 * the track table is owned by the library (DESIGN.md "data layout in HBM"). */

#ifdef __cplusplus
}
#endif
#endif /* BOTSORT_HIP_H */
