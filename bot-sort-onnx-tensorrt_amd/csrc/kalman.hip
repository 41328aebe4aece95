// kalman.hip -- batched Kalman kernels behind the C ABI (include/botsort_hip.h).
//
// predict / update: eight lanes own one track (lane i = row i of the covariance, kf8:: in
// kf_math.h), so a wavefront reads and writes 8 contiguous 576-byte state rows -- the reference's
// own array-of-rows layout (mean[n,8], cov[n,8,8]) is already the coalesced one, and the resident
// tracker (tracker.hip) keeps its table the same way.  The low-volume kernels (initiate, project,
// gating) use one thread per track with the state in registers.
#include "bts_internal.h"
#include "kf_math.h"

namespace {

struct StateView {       // element (track t, component c) at base[t * st + c * sc]
    double *mean; long mean_st, mean_sc;
    double *cov;  long cov_st, cov_sc;
};

__device__ __forceinline__ void load_state(const StateView &v, long t, double m[8], double P[64]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = v.mean[t * v.mean_st + c * v.mean_sc];
#pragma unroll
    for (int c = 0; c < 64; ++c) P[c] = v.cov[t * v.cov_st + c * v.cov_sc];
}
__device__ __forceinline__ void store_state(const StateView &v, long t, const double m[8], const double P[64]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) v.mean[t * v.mean_st + c * v.mean_sc] = m[c];
#pragma unroll
    for (int c = 0; c < 64; ++c) v.cov[t * v.cov_st + c * v.cov_sc] = P[c];
}

// (contiguous rows -- the layout every caller of this file uses -- move as four 16-byte accesses per lane)
__device__ __forceinline__ void load_row(const StateView &v, long t, int i, double &mi, double Pr[8]) {
    mi = v.mean[t * v.mean_st + i * v.mean_sc];
    if (v.cov_sc == 1 && (v.cov_st & 1) == 0 && ((uintptr_t)v.cov & 15) == 0) {
        const double2 *p = reinterpret_cast<const double2 *>(v.cov + t * v.cov_st + i * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double2 q = p[j]; Pr[2 * j] = q.x; Pr[2 * j + 1] = q.y; }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) Pr[j] = v.cov[t * v.cov_st + (i * 8 + j) * v.cov_sc];
}
__device__ __forceinline__ void store_row(const StateView &v, long t, int i, double mi, const double Pr[8]) {
    v.mean[t * v.mean_st + i * v.mean_sc] = mi;
    if (v.cov_sc == 1 && (v.cov_st & 1) == 0 && ((uintptr_t)v.cov & 15) == 0) {
        double2 *p = reinterpret_cast<double2 *>(v.cov + t * v.cov_st + i * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = make_double2(Pr[2 * j], Pr[2 * j + 1]);
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v.cov[t * v.cov_st + (i * 8 + j) * v.cov_sc] = Pr[j];
}

// 8 lanes per track like predict / update: lane i stores mean[i] and row i of the (diagonal) covariance, so a
// wavefront writes 8 contiguous 576-byte states
__global__ __launch_bounds__(256) void k_initiate(const float *__restrict__ meas, int n, StateView v) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = g >> 3, i = g & 7;
    if (t >= n) return;
    float z[4] = {meas[t * 4 + 0], meas[t * 4 + 1], meas[t * 4 + 2], meas[t * 4 + 3]};
    double m[8], P[64];
    kf::initiate(z, m, P);                   // (72 values, of which this lane keeps 9: the rest folds away)
    double mi = 0.0, Pr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (r == i) {
            mi = m[r];
#pragma unroll
            for (int j = 0; j < 8; ++j) Pr[j] = P[r * 8 + j];
        }
    store_row(v, t, i, mi, Pr);
}

__global__ __launch_bounds__(256) void k_multi_predict(StateView v, const uint8_t *__restrict__ zero_vel,
                                                       const uint8_t *__restrict__ row_f32, int n,
                                                       double *__restrict__ tlbr_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = g >> 3, i = g & 7;
    if (t >= n) return;
    double mi, Pr[8];
    load_row(v, t, i, mi, Pr);
    if (i >= 6 && zero_vel && zero_vel[t]) mi = 0.0;           // D:529-532
    kf8::predict(mi, Pr, i, row_f32 && row_f32[t]);
    store_row(v, t, i, mi, Pr);
    if (tlbr_out) {
        double b[4];
        kf8::tlbr(mi, false, b);
        if (i < 4) tlbr_out[t * 4 + i] = b[i];
    }
}

__global__ __launch_bounds__(64) void k_project(StateView v, const uint8_t *__restrict__ row_f32, int n,
                                                double *__restrict__ pmean, double *__restrict__ pcov) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double m[8], P[64], S[16];
    load_state(v, t, m, P);
    kf::project(m, P, row_f32 && row_f32[t], S);
#pragma unroll
    for (int c = 0; c < 4; ++c) pmean[t * 4 + c] = m[c];
#pragma unroll
    for (int c = 0; c < 16; ++c) pcov[t * 16 + c] = S[c];
}

__global__ __launch_bounds__(256) void k_update(StateView v, const uint8_t *__restrict__ row_f32,
                                                const int32_t *__restrict__ trk_idx,
                                                const float *__restrict__ meas,
                                                const int32_t *__restrict__ meas_idx, int k) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = g >> 3, i = g & 7;
    if (q >= k) return;
    const int t = trk_idx ? trk_idx[q] : q;
    const int d = meas_idx ? meas_idx[q] : q;
    float z[4] = {meas[d * 4 + 0], meas[d * 4 + 1], meas[d * 4 + 2], meas[d * 4 + 3]};
    double mi, Pr[8];
    load_row(v, t, i, mi, Pr);
    kf8::update(mi, Pr, i, z, row_f32 && row_f32[t]);
    store_row(v, t, i, mi, Pr);
}

struct Warp { double h[6]; };

__global__ __launch_bounds__(64) void k_gmc(StateView v, int n, Warp w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double m[8], P[64];
    load_state(v, t, m, P);
    kf::gmc(m, P, w.h);
    store_state(v, t, m, P);
}

// gating_distance for every (track, measurement) pair, D:338-380.  The matrix is written once (8 B per pair: the
// kernel is HBM-write-bound), so lanes run along the measurements -- consecutive lanes store consecutive elements of
// one output row -- and a workgroup takes GT tracks x 256 measurements: the first GT threads factor the tracks'
// innovation covariances (4x4 Cholesky) into LDS, every thread keeps its measurement in registers.
constexpr int GT = 16;
__global__ __launch_bounds__(256) void k_gating(StateView v, int n, const double *__restrict__ meas, int m,
                                                int only_position, int metric, double *__restrict__ out) {
    __shared__ double sL[GT][16], smu[GT][4];
    const int t0 = blockIdx.y * GT, j = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < GT && t0 + threadIdx.x < n) {
        double mu[8], P[64], S[16], L[16];
        load_state(v, t0 + threadIdx.x, mu, P);
        kf::project(mu, P, false, S);
        if (only_position) {   // 2x2 problem embedded in the 4x4 routines
            S[2] = S[3] = S[6] = S[7] = S[8] = S[9] = S[12] = S[13] = S[11] = S[14] = 0.0;
            S[10] = 1.0; S[15] = 1.0;
        }
        kf::chol4(S, L);
        for (int c = 0; c < 16; ++c) sL[threadIdx.x][c] = (c % 5 == 0) ? 1.0 / L[c] : L[c];      // diagonal stored as its reciprocal
        for (int c = 0; c < 4; ++c) smu[threadIdx.x][c] = mu[c];
    }
    __syncthreads();
    if (j >= m) return;
    double z[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) z[a] = meas[(long)j * 4 + a];
    const int nt = min(GT, n - t0);
    for (int tt = 0; tt < nt; ++tt) {
        // (fully unrolled over the 4 components, the 2-D case masks the last two: no runtime-indexed arrays)
        double d[4], y[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) d[a] = z[a] - smu[tt][a];
        double r;
        if (metric == 1) {
            r = d[0] * d[0]; r += d[1] * d[1];
            if (!only_position) { r += d[2] * d[2]; r += d[3] * d[3]; }
        } else {   // forward substitution L y = d, sum y^2
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double sm = d[i];
#pragma unroll
                for (int kk = 0; kk < i; ++kk) sm -= sL[tt][i * 4 + kk] * y[kk];
                y[i] = sm * sL[tt][i * 4 + i];            // (diagonal stored as its reciprocal: one division per track, not per pair)
            }
            r = y[0] * y[0]; r += y[1] * y[1];
            if (!only_position) { r += y[2] * y[2]; r += y[3] * y[3]; }
        }
        out[(long)(t0 + tt) * m + j] = r;
    }
}

StateView aos(double *mean, double *cov) { return StateView{mean, 8, 1, cov, 64, 1}; }

}  // namespace

extern "C" {

int bts_kf_initiate(bts_context *ctx, const float *meas_xywh, int n, double *mean, double *cov,
                    bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0, "ctx, n");
    BTS_ON_DEVICE(ctx);
    if (n == 0) return BTS_OK;
    BTS_REQUIRE(meas_xywh && mean && cov, "null pointer");
    hipLaunchKernelGGL(k_initiate, dim3(bts_cdiv(n * 8, 256)), dim3(256), 0, (hipStream_t)stream, meas_xywh, n,
                       aos(mean, cov));
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_kf_multi_predict(bts_context *ctx, double *mean, double *cov, const uint8_t *zero_vel,
                         const uint8_t *row_f32, int n, double *tlbr_out, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0, "ctx, n");
    BTS_ON_DEVICE(ctx);
    if (n == 0) return BTS_OK;
    BTS_REQUIRE(mean && cov, "null pointer");
    hipLaunchKernelGGL(k_multi_predict, dim3(bts_cdiv(n * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       aos(mean, cov), zero_vel, row_f32, n, tlbr_out);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_kf_project(bts_context *ctx, const double *mean, const double *cov, const uint8_t *row_f32,
                   int n, double *pmean, double *pcov, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0, "ctx, n");
    BTS_ON_DEVICE(ctx);
    if (n == 0) return BTS_OK;
    BTS_REQUIRE(mean && cov && pmean && pcov, "null pointer");
    hipLaunchKernelGGL(k_project, dim3(bts_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream,
                       aos(const_cast<double *>(mean), const_cast<double *>(cov)), row_f32, n, pmean, pcov);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_kf_update(bts_context *ctx, double *mean, double *cov, const uint8_t *row_f32,
                  const int32_t *trk_idx, const float *meas_xywh, const int32_t *meas_idx, int k,
                  bts_stream stream) {
    BTS_REQUIRE(ctx && k >= 0, "ctx, k");
    BTS_ON_DEVICE(ctx);
    if (k == 0) return BTS_OK;
    BTS_REQUIRE(mean && cov && meas_xywh, "null pointer");
    hipLaunchKernelGGL(k_update, dim3(bts_cdiv(k * 8, 256)), dim3(256), 0, (hipStream_t)stream, aos(mean, cov),
                       row_f32, trk_idx, meas_xywh, meas_idx, k);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_kf_multi_gmc(bts_context *ctx, double *mean, double *cov, int n, const double *h2x3_host, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && h2x3_host, "ctx, n, H");
    BTS_ON_DEVICE(ctx);
    if (n == 0) return BTS_OK;
    BTS_REQUIRE(mean && cov, "null pointer");
    Warp w;
    for (int k = 0; k < 6; ++k) w.h[k] = h2x3_host[k];
    hipLaunchKernelGGL(k_gmc, dim3(bts_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, aos(mean, cov), n, w);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_kf_gating(bts_context *ctx, const double *mean, const double *cov, int n, const double *meas,
                  int m, int only_position, int metric, double *out, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0, "ctx, n, m");
    BTS_ON_DEVICE(ctx);
    BTS_REQUIRE(metric == 0 || metric == 1, "metric must be 0 (maha) or 1 (gaussian)");
    if (n == 0 || m == 0) return BTS_OK;
    BTS_REQUIRE(mean && cov && meas && out, "null pointer");
    hipLaunchKernelGGL(k_gating, dim3(bts_cdiv(m, 256), bts_cdiv(n, GT)), dim3(256), 0, (hipStream_t)stream,
                       aos(const_cast<double *>(mean), const_cast<double *>(cov)), n, meas, m,
                       only_position, metric, out);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

}  // extern "C"
