// assoc.hip -- association-cost kernels: N x M IoU cost, the fused stage-1 / stage-3 costs and
// the tracked-vs-lost duplicate test (reference demo_bottrack_onnx_tflite.py:1539-1554,
// 1590-1602, 1665-1680, 1695-1761).
//
// Layout: a workgroup owns a [ROWS x 256] tile of the cost matrix.  Each lane keeps "its"
// detection box in registers, the tile's track boxes sit in LDS and are read as broadcasts, and
// every store instruction writes 256 consecutive float64 of one matrix row (2 KiB, coalesced).
// These kernels are HBM-write bound: 8 B of output per pair against 64 B per *box* of input.
#include "bts_internal.h"
#include "iou_math.h"
#include "cost_math.h"

namespace {

constexpr int TILE_ROWS = 16;
constexpr int TILE_COLS = 256;

__global__ __launch_bounds__(TILE_COLS) void k_iou_distance(const double *__restrict__ a, const uint8_t *__restrict__ a32,
                                                            int n, const double *__restrict__ b,
                                                            const uint8_t *__restrict__ b32, int m,
                                                            double *__restrict__ cost, int ld) {
    __shared__ double sa[TILE_ROWS][4];
    __shared__ double sarea[TILE_ROWS];
    __shared__ int sflag[TILE_ROWS];
    const int i0 = blockIdx.y * TILE_ROWS, j = blockIdx.x * TILE_COLS + threadIdx.x;
    if (threadIdx.x < TILE_ROWS) {
        const int i = i0 + threadIdx.x;
        if (i < n) {
            double box[4] = {a[i * 4 + 0], a[i * 4 + 1], a[i * 4 + 2], a[i * 4 + 3]};
            const bool f = a32 && a32[i];
            for (int c = 0; c < 4; ++c) sa[threadIdx.x][c] = box[c];
            sarea[threadIdx.x] = iou::area(box, f);
            sflag[threadIdx.x] = f;
        }
    }
    __syncthreads();
    if (j >= m) return;
    const double bb[4] = {b[j * 4 + 0], b[j * 4 + 1], b[j * 4 + 2], b[j * 4 + 3]};
    const bool bf = b32 && b32[j];
    const double area_b = iou::area(bb, bf);
    const int rows = min(TILE_ROWS, n - i0);
    for (int r = 0; r < rows; ++r) {
        const double aa[4] = {sa[r][0], sa[r][1], sa[r][2], sa[r][3]};
        cost[(long)(i0 + r) * ld + j] = iou::cost(aa, sflag[r] != 0, sarea[r], bb, bf, area_b);
    }
}

__global__ __launch_bounds__(256) void k_fuse1(const double *__restrict__ iou_d, int ld_iou,
                                               const float *__restrict__ body, const float *__restrict__ face,
                                               int ld_sim, int n, int m, double prox, float app,
                                               double *__restrict__ cost, int ld_cost) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int i0 = blockIdx.y * TILE_ROWS, i1 = min(n, i0 + TILE_ROWS);
    for (int i = i0; i < i1; ++i) {
        const float fs = face ? face_fixup(face[(long)i * ld_sim + j]) : 0.0f;
        cost[(long)i * ld_cost + j] = fuse1(iou_d[(long)i * ld_iou + j], body[(long)i * ld_sim + j], fs, prox, app);
    }
}

__global__ __launch_bounds__(256) void k_fuse3(const double *__restrict__ iou_d, int ld_iou,
                                               const float *__restrict__ body, int ld_sim, int n, int m,
                                               double prox, float app, double *__restrict__ cost, int ld_cost) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int i0 = blockIdx.y * TILE_ROWS, i1 = min(n, i0 + TILE_ROWS);
    for (int i = i0; i < i1; ++i)
        cost[(long)i * ld_cost + j] = fuse3(iou_d[(long)i * ld_iou + j], body[(long)i * ld_sim + j], prox, app);
}

// remove_duplicate_stracks, D:1665-1680: for every pair with cost < thresh the shorter-lived
// side is dropped (ties drop the `a` side).  Pairs are independent, so the flags are OR-ed.
__global__ __launch_bounds__(TILE_COLS) void k_dedupe(const double *__restrict__ a, const uint8_t *__restrict__ a32,
                                                      const int32_t *__restrict__ age_a, int n,
                                                      const double *__restrict__ b, const uint8_t *__restrict__ b32,
                                                      const int32_t *__restrict__ age_b, int m, double thresh,
                                                      uint8_t *__restrict__ drop_a, uint8_t *__restrict__ drop_b) {
    __shared__ double sa[TILE_ROWS][4];
    __shared__ double sarea[TILE_ROWS];
    __shared__ int sflag[TILE_ROWS];
    __shared__ int sage[TILE_ROWS];
    const int i0 = blockIdx.y * TILE_ROWS, j = blockIdx.x * TILE_COLS + threadIdx.x;
    if (threadIdx.x < TILE_ROWS) {
        const int i = i0 + threadIdx.x;
        if (i < n) {
            double box[4] = {a[i * 4 + 0], a[i * 4 + 1], a[i * 4 + 2], a[i * 4 + 3]};
            const bool f = a32 && a32[i];
            for (int c = 0; c < 4; ++c) sa[threadIdx.x][c] = box[c];
            sarea[threadIdx.x] = iou::area(box, f);
            sflag[threadIdx.x] = f;
            sage[threadIdx.x] = age_a[i];
        }
    }
    __syncthreads();
    if (j >= m) return;
    const double bb[4] = {b[j * 4 + 0], b[j * 4 + 1], b[j * 4 + 2], b[j * 4 + 3]};
    const bool bf = b32 && b32[j];
    const double area_b = iou::area(bb, bf);
    const int agej = age_b[j];
    const int rows = min(TILE_ROWS, n - i0);
    bool dropj = false;
    for (int r = 0; r < rows; ++r) {
        const double aa[4] = {sa[r][0], sa[r][1], sa[r][2], sa[r][3]};
        const double c = iou::cost(aa, sflag[r] != 0, sarea[r], bb, bf, area_b);
        if (c < thresh) {
            if (sage[r] > agej) dropj = true;
            else drop_a[i0 + r] = 1;
        }
    }
    if (dropj) drop_b[j] = 1;
}

}  // namespace

extern "C" {

int bts_iou_distance(bts_context *ctx, const double *a_tlbr, const uint8_t *a_f32, int n,
                     const double *b_tlbr, const uint8_t *b_f32, int m, double *cost, int ld,
                     bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0, "ctx, n, m");
    BTS_ON_DEVICE(ctx);
    if (n == 0 || m == 0) return BTS_OK;
    BTS_REQUIRE(a_tlbr && b_tlbr && cost && ld >= m, "null pointer / ld < m");
    hipLaunchKernelGGL(k_iou_distance, dim3(bts_cdiv(m, TILE_COLS), bts_cdiv(n, TILE_ROWS)), dim3(TILE_COLS), 0,
                       (hipStream_t)stream, a_tlbr, a_f32, n, b_tlbr, b_f32, m, cost, ld);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_fuse_cost_stage1(bts_context *ctx, const double *iou_d, int ld_iou, const float *body_sim,
                         const float *face_sim, int ld_sim, int n, int m, double proximity_thresh,
                         double appearance_thresh, double *cost, int ld_cost, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0, "ctx, n, m");
    BTS_ON_DEVICE(ctx);
    if (n == 0 || m == 0) return BTS_OK;
    BTS_REQUIRE(iou_d && body_sim && cost && ld_iou >= m && ld_sim >= m && ld_cost >= m, "null pointer / ld < m");
    hipLaunchKernelGGL(k_fuse1, dim3(bts_cdiv(m, 256), bts_cdiv(n, TILE_ROWS)), dim3(256), 0, (hipStream_t)stream,
                       iou_d, ld_iou, body_sim, face_sim, ld_sim, n, m, proximity_thresh,
                       (float)appearance_thresh, cost, ld_cost);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_fuse_cost_stage3(bts_context *ctx, const double *iou_d, int ld_iou, const float *body_sim,
                         int ld_sim, int n, int m, double proximity_thresh, double appearance_thresh,
                         double *cost, int ld_cost, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0, "ctx, n, m");
    BTS_ON_DEVICE(ctx);
    if (n == 0 || m == 0) return BTS_OK;
    BTS_REQUIRE(iou_d && body_sim && cost && ld_iou >= m && ld_sim >= m && ld_cost >= m, "null pointer / ld < m");
    hipLaunchKernelGGL(k_fuse3, dim3(bts_cdiv(m, 256), bts_cdiv(n, TILE_ROWS)), dim3(256), 0, (hipStream_t)stream,
                       iou_d, ld_iou, body_sim, ld_sim, n, m, proximity_thresh, (float)appearance_thresh,
                       cost, ld_cost);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_dedupe_pairs(bts_context *ctx, const double *a_tlbr, const uint8_t *a_f32, const int32_t *age_a,
                     int n, const double *b_tlbr, const uint8_t *b_f32, const int32_t *age_b, int m,
                     double thresh, uint8_t *drop_a, uint8_t *drop_b, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0, "ctx, n, m");
    BTS_ON_DEVICE(ctx);
    if (n > 0) { BTS_REQUIRE(drop_a, "drop_a"); BTS_HIP(hipMemsetAsync(drop_a, 0, n, (hipStream_t)stream)); }
    if (m > 0) { BTS_REQUIRE(drop_b, "drop_b"); BTS_HIP(hipMemsetAsync(drop_b, 0, m, (hipStream_t)stream)); }
    if (n == 0 || m == 0) return BTS_OK;
    BTS_REQUIRE(a_tlbr && b_tlbr && age_a && age_b, "null pointer");
    hipLaunchKernelGGL(k_dedupe, dim3(bts_cdiv(m, TILE_COLS), bts_cdiv(n, TILE_ROWS)), dim3(TILE_COLS), 0,
                       (hipStream_t)stream, a_tlbr, a_f32, age_a, n, b_tlbr, b_f32, age_b, m, thresh, drop_a, drop_b);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

}  // extern "C"
