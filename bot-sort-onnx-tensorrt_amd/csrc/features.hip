// features.hip -- ReID feature bookkeeping of a track (reference demo_bottrack_onnx_tflite.py:
// 492-514, STrack.update_body_features / update_face_features):
//     curr   = f
//     smooth = f                          (first feature of the track)
//            = alpha*smooth + (1-alpha)*f (afterwards; float32, alpha = 0.9 as a python float)
//     smooth /= ||smooth||_2
// plus the in-place renormalisation the reference applies to every detection's feature row when
// the detection record is built (the first call aliases `smooth = f` and normalises in place,
// quirk Q8).  One workgroup per row: a row is read once from HBM and stays in L2 for the second
// sweep; float4 accesses, wavefront __shfl reduction of the squared norm.
#include "bts_internal.h"
#include "feat_math.h"

namespace {

__global__ __launch_bounds__(256) void k_feature_ema(float *__restrict__ curr, float *__restrict__ smooth,
                                                     uint8_t *__restrict__ has_smooth,
                                                     const int32_t *__restrict__ trk_idx,
                                                     const float *__restrict__ det_feats,
                                                     const int32_t *__restrict__ det_idx, int k, int d, float alpha,
                                                     float beta) {
    const int q = blockIdx.x;
    if (q >= k) return;
    const long t = trk_idx ? trk_idx[q] : q, s = det_idx ? det_idx[q] : q;
    feat::ema_row(curr + t * d, smooth + t * d, det_feats + s * d, d, has_smooth[t] != 0, alpha, beta);
    if (threadIdx.x == 0) has_smooth[t] = 1;
}

__global__ __launch_bounds__(256) void k_feature_renorm(float *__restrict__ feats, const int32_t *__restrict__ idx,
                                                        int k, int d) {
    const int q = blockIdx.x;
    if (q >= k) return;
    const long s = idx ? idx[q] : q;
    feat::renorm_row(feats + s * d, d);
}

// diagnostics of the appearance pre-filter's int8 shadows (feat::to_i8_row / feat::i8_excluded), exposed for the tests
__global__ __launch_bounds__(256) void k_i8_rows(const float *__restrict__ x, int n, int d, float gate, int8_t *__restrict__ q,
                                                 float *__restrict__ meta) {
    const long j = blockIdx.x;
    feat::to_i8_row(x + j * d, q + j * d, meta + j * feat::I8_META, d, gate, j < n);
}
// pair p = (ia[p], ib[p]): for every checkpoint c (the last = full row) out[p][c] = 1 when the prefix's integer dot
// product excludes the pair (feat::i8_excluded), dots[p][c] = that integer dot product
__global__ __launch_bounds__(64) void k_i8_pairs(const int8_t *__restrict__ qa, const float *__restrict__ ma, const int8_t *__restrict__ qb,
                                                 const float *__restrict__ mb, const int32_t *__restrict__ ia, const int32_t *__restrict__ ib,
                                                 int np, int d, uint8_t *__restrict__ out, int32_t *__restrict__ dots) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    int chk[feat::I8_NCHK];
    feat::i8_checkpoints(d / feat::I8_STAGE, chk);
    const int8_t *a = qa + (long)ia[p] * d, *b = qb + (long)ib[p] * d;
    const float *ra = ma + (long)ia[p] * feat::I8_META, *rb = mb + (long)ib[p] * feat::I8_META;
    int dot = 0, k = 0;
    for (int c = 0; c <= feat::I8_NCHK; ++c) {
        const int upto = c == feat::I8_NCHK ? d : (chk[c] < d / feat::I8_STAGE ? chk[c] * feat::I8_STAGE : d);
        for (; k < upto; ++k) dot += (int)a[k] * (int)b[k];
        out[p * (feat::I8_NCHK + 1) + c] = feat::i8_excluded(dot, feat::i8_rec(ra, c), feat::i8_rec(rb, c)) ? 1 : 0;
        dots[p * (feat::I8_NCHK + 1) + c] = dot;
    }
}

}  // namespace

extern "C" {

int bts_debug_i8_rows(bts_context *ctx, const float *x, int n, int d, double gate, int8_t *q, float *meta, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && d > 0, "ctx, n, d");
    BTS_ON_DEVICE(ctx);
    if (n == 0) return BTS_OK;
    BTS_REQUIRE(x && q && meta, "null pointer");
    hipLaunchKernelGGL(k_i8_rows, dim3(n), dim3(256), 0, (hipStream_t)stream, x, n, d, (float)gate, q, meta);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_debug_i8_pairs(bts_context *ctx, const int8_t *qa, const float *meta_a, const int8_t *qb, const float *meta_b,
                       const int32_t *ia, const int32_t *ib, int n_pairs, int d, uint8_t *excluded, int32_t *dots, bts_stream stream) {
    BTS_REQUIRE(ctx && n_pairs >= 0 && d > 0, "ctx, n_pairs, d");
    BTS_ON_DEVICE(ctx);
    if (n_pairs == 0) return BTS_OK;
    BTS_REQUIRE(qa && meta_a && qb && meta_b && ia && ib && excluded && dots, "null pointer");
    hipLaunchKernelGGL(k_i8_pairs, dim3(bts_cdiv(n_pairs, 64)), dim3(64), 0, (hipStream_t)stream, qa, meta_a, qb, meta_b, ia, ib, n_pairs, d,
                       excluded, dots);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_debug_i8_layout(int d, int32_t *out8) {
    BTS_REQUIRE(d > 0 && out8, "d, out");
    int chk[feat::I8_NCHK];
    feat::i8_checkpoints(d / feat::I8_STAGE, chk);
    for (int c = 0; c < feat::I8_NCHK; ++c) out8[c] = chk[c];
    out8[4] = d / feat::I8_STAGE; out8[5] = feat::I8_STAGE; out8[6] = feat::I8_META; out8[7] = feat::I8_NCHK;
    return BTS_OK;
}


// Launch-floor probe: `launches` dependent launches of a kernel that follows a chain of `links` dependent loads
// (each workgroup walks buf[0] -> buf[buf[0]] -> ...; links = 0: an empty kernel).  Timed by the caller.
__global__ void k_floor_probe(int32_t *buf, int32_t *out, int links) {
    // chain: buf[0] -> buf[64] -> buf[128] ... (one cache line each), rewritten by every launch so that the next
    // launch finds the lines freshly produced by another kernel (cold in its L2), like real producer/consumer kernels
    int v = 0;
    for (int k = 0; k < links; ++k) v = __hip_atomic_load(&buf[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && v == 12345) out[blockIdx.x] = v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) buf[threadIdx.x * 64] = (threadIdx.x + 1) * 64;
}
int bts_debug_launch_floor(bts_context *ctx, int32_t *buf, int launches, int blocks, int threads, int links, bts_stream stream) {
    BTS_REQUIRE(ctx && buf && launches >= 0 && blocks > 0 && threads > 0 && links >= 0, "arguments");
    BTS_ON_DEVICE(ctx);
    for (int k = 0; k < launches; ++k) {
        hipLaunchKernelGGL(k_floor_probe, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, buf, buf + 4096, links);
        BTS_LAUNCH_CHECK();
    }
    return BTS_OK;
}

int bts_feature_ema(bts_context *ctx, float *curr, float *smooth, uint8_t *has_smooth,
                    const int32_t *trk_idx, const float *det_feats, const int32_t *det_idx, int k, int d,
                    double alpha, bts_stream stream) {
    BTS_REQUIRE(ctx && k >= 0 && d > 0, "ctx, k, d");
    BTS_ON_DEVICE(ctx);
    if (k == 0) return BTS_OK;
    BTS_REQUIRE(curr && smooth && has_smooth && det_feats, "null pointer");
    // alpha and (1 - alpha) are python floats in the reference, demoted to float32 by the array
    const float alpha32 = (float)alpha, beta = (float)(1.0 - alpha);
    hipLaunchKernelGGL(k_feature_ema, dim3(k), dim3(256), 0, (hipStream_t)stream, curr, smooth, has_smooth,
                       trk_idx, det_feats, det_idx, k, d, alpha32, beta);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

int bts_feature_renorm(bts_context *ctx, float *feats, const int32_t *idx, int k, int d, bts_stream stream) {
    BTS_REQUIRE(ctx && k >= 0 && d > 0, "ctx, k, d");
    BTS_ON_DEVICE(ctx);
    if (k == 0) return BTS_OK;
    BTS_REQUIRE(feats, "null pointer");
    hipLaunchKernelGGL(k_feature_renorm, dim3(k), dim3(256), 0, (hipStream_t)stream, feats, idx, k, d);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

}  // extern "C"
