// cosine.hip -- ReID appearance similarity  sim[i][j] = max(0, <trk_feat[i], det_feat[j]>)  in
// float32: the reference's only in-repo definition (demo_bottrack_onnx_tflite.py:1593-1599), which
// also stands in for the similarity output of the ReID ONNX graphs (D:1453-1483).
//
// This is the one dense contraction of the tracker hot path (2*N*M*D flop): both operands are
// K-contiguous rows, gathered through optional index lists (pool rows / detection subsets), so it
// is an "A * B^T" GEMM.  Two kernels:
//   * BTS_COS_MFMA_F32: 128x128x32 LDS-tiled, v_mfma_f32_32x32x2_f32.  gfx950's float32-input
//     MFMA is bit-for-bit a k-ordered fmaf chain at the float32 vector rate, so numerics are plain
//     float32 (no reduced-precision path is involved).
//   * BTS_COS_VALU_F32: 64x64 register-tiled v_fma_f32 kernel (also the fallback for feature rows
//     that are not 16-byte aligned).
// Both accumulate k in ascending chunks with fused multiply-adds; results differ from a CPU BLAS
// sgemm only by float32 summation order (<= ~1e-6), which the parity tests bound.
#include "bts_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;   // +4 floats: conflict-free b128 reads

// XCD-aware remap of the linear workgroup id (8 XCDs, private L2 each): consecutive "virtual"
// ids -- which share an A panel -- are placed on one XCD.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, slot = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__global__ __launch_bounds__(256) void k_cos_mfma(const float *__restrict__ a, const int32_t *__restrict__ a_idx, int lda,
                                                  const float *__restrict__ b, const int32_t *__restrict__ b_idx, int ldb,
                                                  int d, const int32_t *__restrict__ dims, int n_max, int m_max,
                                                  float *__restrict__ sim, int ld) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];
    const int n = dims ? dims[0] : n_max, m = dims ? dims[1] : m_max;
    const int nbn = (m_max + BN - 1) / BN;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int bm = vb / nbn, bn = vb % nbn;
    if (bm * BM >= n || bn * BN >= m) return;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    // staging map: 128 rows x 8 float4 per operand; thread t covers rows (t>>3) + 32*s, s=0..3
    const int srow = tid >> 3, sk = (tid & 7) * 4;
    const float *arow[4], *brow[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = bm * BM + srow + 32 * s, j = bn * BN + srow + 32 * s;
        arow[s] = i < n ? a + (long)(a_idx ? a_idx[i] : i) * lda : nullptr;
        brow[s] = j < m ? b + (long)(b_idx ? b_idx[j] : j) * ldb : nullptr;
    }
    float4 ra[4], rb[4];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + sk;
            ra[s] = (arow[s] && k < d) ? *reinterpret_cast<const float4 *>(arow[s] + k) : make_float4(0, 0, 0, 0);
            rb[s] = (brow[s] && k < d) ? *reinterpret_cast<const float4 *>(brow[s] + k) : make_float4(0, 0, 0, 0);
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *reinterpret_cast<float4 *>(&As[(srow + 32 * s) * LDS_LD + sk]) = ra[s];
            *reinterpret_cast<float4 *>(&Bs[(srow + 32 * s) * LDS_LD + sk]) = rb[s];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;
    stage_load(0);
    for (int k0 = 0; k0 < d; k0 += BK) {
        __syncthreads();          // previous tile fully consumed
        stage_write();
        __syncthreads();
        if (k0 + BK < d) stage_load(k0 + BK);   // next tile's HBM latency hides under the MFMAs
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            float4 fa[2], fb[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                fa[mi] = *reinterpret_cast<const float4 *>(&As[(wr * 64 + mi * 32 + frow) * LDS_LD + t * 8 + fh * 4]);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                fb[ni] = *reinterpret_cast<const float4 *>(&Bs[(wc * 64 + ni * 32 + frow) * LDS_LD + t * 8 + fh * 4]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].x, fb[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].y, fb[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].z, fb[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].w, fb[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int j = bn * BN + wc * 64 + ni * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = bm * BM + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (i < n && j < m) sim[(long)i * ld + j] = fmaxf(0.0f, acc[mi][ni][r]);
            }
        }
}

// float32 vector-ALU variant: 64x64 tile, 256 threads, 4x4 outputs per thread, BK = 16.
__global__ __launch_bounds__(256) void k_cos_valu(const float *__restrict__ a, const int32_t *__restrict__ a_idx, int lda,
                                                  const float *__restrict__ b, const int32_t *__restrict__ b_idx, int ldb,
                                                  int d, const int32_t *__restrict__ dims, int n_max, int m_max,
                                                  float *__restrict__ sim, int ld) {
    constexpr int T = 64, K = 16;
    __shared__ float As[K][T + 1];
    __shared__ float Bs[K][T + 1];
    const int n = dims ? dims[0] : n_max, m = dims ? dims[1] : m_max;
    const int nbn = (m_max + T - 1) / T;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    if (bm * T >= n || bn * T >= m) return;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    float acc[4][4] = {};
    const int lrow = tid >> 2, lk = (tid & 3) * 4;   // 64 rows x 4 chunks of 4 k
    const int gi = bm * T + lrow, gj = bn * T + lrow;
    const float *pa = gi < n ? a + (long)(a_idx ? a_idx[gi] : gi) * lda : nullptr;
    const float *pb = gj < m ? b + (long)(b_idx ? b_idx[gj] : gj) * ldb : nullptr;
    for (int k0 = 0; k0 < d; k0 += K) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + lk + c;
            As[lk + c][lrow] = (pa && k < d) ? pa[k] : 0.0f;
            Bs[lk + c][lrow] = (pb && k < d) ? pb[k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = As[k][ty * 4 + r]; bv[r] = Bs[k][tx * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(av[r], bv[c], acc[r][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = bm * T + ty * 4 + r, j = bn * T + tx * 4 + c;
            if (i < n && j < m) sim[(long)i * ld + j] = fmaxf(0.0f, acc[r][c]);
        }
}

}  // namespace

int bts_cos_enqueue(int which, const float *a, const int32_t *a_idx, int lda, const float *b,
                    const int32_t *b_idx, int ldb, int d, const int32_t *dims, int n_max, int m_max,
                    float *sim, int ld, hipStream_t stream) {
    if (n_max <= 0 || m_max <= 0) return BTS_OK;
    const bool aligned = ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && lda % 4 == 0 && ldb % 4 == 0 && d % 4 == 0;
    if (which == BTS_COS_MFMA_F32 && aligned) {
        const int nblk = bts_cdiv(n_max, BM) * bts_cdiv(m_max, BN);
        hipLaunchKernelGGL(k_cos_mfma, dim3(nblk), dim3(256), 0, stream, a, a_idx, lda, b, b_idx, ldb, d, dims,
                           n_max, m_max, sim, ld);
    } else {
        const int nblk = bts_cdiv(n_max, 64) * bts_cdiv(m_max, 64);
        hipLaunchKernelGGL(k_cos_valu, dim3(nblk), dim3(256), 0, stream, a, a_idx, lda, b, b_idx, ldb, d, dims,
                           n_max, m_max, sim, ld);
    }
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}

extern "C" int bts_cos_similarity(bts_context *ctx, const float *trk_feats, int n, const float *det_feats, int m,
                                  int d, float *sim, int ld, bts_stream stream) {
    BTS_REQUIRE(ctx && n >= 0 && m >= 0 && d >= 0, "ctx, n, m, d");
    BTS_ON_DEVICE(ctx);
    if (n == 0 || m == 0) return BTS_OK;
    BTS_REQUIRE(trk_feats && det_feats && sim && ld >= m, "null pointer / ld < m");
    return bts_cos_enqueue(ctx->cos_kernel, trk_feats, nullptr, d, det_feats, nullptr, d, d, nullptr, n, m, sim, ld,
                           (hipStream_t)stream);
}
