// feat_math.h -- cooperative float32 feature-row arithmetic (D:492-514).
// A row is handled by a GROUP of ROW_G = 256 consecutive threads; a workgroup holds blockDim.x / 256
// groups working on different rows in lockstep (256-thread workgroups: one row; the 1024-thread list
// kernels: four rows).  Every function must be reached by all threads of the workgroup; a group
// without a row passes active = false (it only takes part in the barriers).
#pragma once
#include <hip/hip_runtime.h>
#include "block_scan.h"

namespace feat {

constexpr int ROW_G = 256;
__device__ __forceinline__ int row_tid() { return threadIdx.x & (ROW_G - 1); }
__device__ __forceinline__ int row_group() { return threadIdx.x / ROW_G; }

// sum over the group; result returned to every thread of the group (fixed order: lanes by xor
// butterfly, then the group's four wavefront partials in ascending order)
__device__ __forceinline__ float block_sum(float v) {
    __shared__ float part[16];
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bscan::lds_barrier();                 // protects `part` across back-to-back calls (LDS-only hand-off)
    if (lane == 0) part[wid] = v;
    bscan::lds_barrier();
    const int w0 = row_group() * (ROW_G / 64);
    return ((part[w0] + part[w0 + 1]) + part[w0 + 2]) + part[w0 + 3];
}

__device__ __forceinline__ bool vec4_ok(const void *p, int d) { return ((uintptr_t)p % 16 == 0) && (d % 4 == 0); }

// row /= ||row||_2   (np.linalg.norm of a float32 vector = sqrt(dot(x,x)) in float32); returns that norm to every thread of the group
__device__ __forceinline__ float renorm_row(float *row, int d, bool active = true) {
    const int t = row_tid();
    float ss = 0.0f;
    const bool v4 = active && vec4_ok(row, d);
    if (v4 && d / 4 <= 2 * ROW_G) {                       // the row stays in registers between the two sweeps
        float4 *r4 = reinterpret_cast<float4 *>(row);
        float4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            v[u] = k < d / 4 ? r4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
        }
        const float nrm = sqrtf(block_sum(ss));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            if (k < d / 4) r4[k] = make_float4(v[u].x / nrm, v[u].y / nrm, v[u].z / nrm, v[u].w / nrm);
        }
        return nrm;
    }
    if (v4) {
        float4 *r4 = reinterpret_cast<float4 *>(row);
        for (int k = t; k < d / 4; k += ROW_G) {
            const float4 v = r4[k];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        const float nrm = sqrtf(block_sum(ss));
        for (int k = t; k < d / 4; k += ROW_G) {
            float4 v = r4[k];
            v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
            r4[k] = v;
        }
        return nrm;
    } else {
        if (active) for (int k = t; k < d; k += ROW_G) ss += row[k] * row[k];
        const float nrm = sqrtf(block_sum(ss));
        if (active) for (int k = t; k < d; k += ROW_G) row[k] /= nrm;
        return nrm;
    }
}

// ||row||_2 exactly as renorm_row computes it (same per-thread expression, same block_sum order), without touching the row: the
// norm a later renorm_row of the same row will divide by (tk_cvt_ahead: a detection row's track-side record ahead of its absorption)
__device__ __forceinline__ float row_norm(const float *row, int d, bool active = true) {
    const int t = row_tid();
    float ss = 0.0f;
    const bool v4 = active && vec4_ok(row, d);
    if (v4 && d / 4 <= 2 * ROW_G) {
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            const float4 v = k < d / 4 ? r4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else if (v4) {
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
        for (int k = t; k < d / 4; k += ROW_G) {
            const float4 v = r4[k];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else if (active) {
        for (int k = t; k < d; k += ROW_G) ss += row[k] * row[k];
    }
    return sqrtf(block_sum(ss));
}

// curr = f, nothing else (BTS_SMOOTH_DEFERRED: the smooth half of ema_row runs later, from curr).  No barriers.
__device__ __forceinline__ void copy_row(float *curr, const float *f, int d, bool active = true) {
    if (!active) return;
    const int t = row_tid();
    if (vec4_ok(curr, d) && vec4_ok(f, d)) {
        const float4 *f4 = reinterpret_cast<const float4 *>(f);
        float4 *c4 = reinterpret_cast<float4 *>(curr);
        for (int k = t; k < d / 4; k += ROW_G) c4[k] = f4[k];
    } else {
        for (int k = t; k < d; k += ROW_G) curr[k] = f[k];
    }
}

// curr = f ; smooth = normalise(has ? alpha*smooth + beta*f : f)
__device__ __forceinline__ void ema_row(float *curr, float *smooth, const float *f, int d, bool has, float alpha,
                                        float beta, bool active = true) {
    const int t = row_tid();
    float ss = 0.0f;
    if (active && vec4_ok(curr, d) && vec4_ok(smooth, d) && vec4_ok(f, d) && d / 4 <= 2 * ROW_G) {      // 16 B per lane, values stay in registers
        const float4 *f4 = reinterpret_cast<const float4 *>(f);
        float4 *c4 = reinterpret_cast<float4 *>(curr), *s4 = reinterpret_cast<float4 *>(smooth);
        const int nv = d / 4;
        float4 fv[2], sv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            if (k < nv) {
                fv[u] = f4[k];
                if (has) {
                    const float4 o = s4[k];
                    sv[u].x = alpha * o.x + beta * fv[u].x; sv[u].y = alpha * o.y + beta * fv[u].y;
                    sv[u].z = alpha * o.z + beta * fv[u].z; sv[u].w = alpha * o.w + beta * fv[u].w;
                } else sv[u] = fv[u];
                ss += sv[u].x * sv[u].x + sv[u].y * sv[u].y + sv[u].z * sv[u].z + sv[u].w * sv[u].w;
            }
        }
        const float nrm = sqrtf(block_sum(ss));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            if (k < nv) {
                c4[k] = fv[u];
                s4[k] = make_float4(sv[u].x / nrm, sv[u].y / nrm, sv[u].z / nrm, sv[u].w / nrm);
            }
        }
        return;
    }
    if (active)
        for (int k = t; k < d; k += ROW_G) {
            const float fv = f[k];
            const float s = has ? alpha * smooth[k] + beta * fv : fv;
            ss += s * s;
        }
    const float nrm = sqrtf(block_sum(ss));
    if (active)
        for (int k = t; k < d; k += ROW_G) {
            const float fv = f[k];
            const float s = has ? alpha * smooth[k] + beta * fv : fv;
            curr[k] = fv;
            smooth[k] = s / nrm;
        }
}

// ---- 8-bit integer shadow rows for the appearance pre-filter ------------------------------------------------
// A float32 feature row x gets the shadow  q[k] = clamp(round(x[k] * s), +-127)  (int8, symmetric; ONE scale s for every row of
// a given dimension: 127 at I8_CLIP standard deviations of a unit vector's components, so that an integer dot product means
// the same cosine for every pair and the candidate test can compare accumulators with integer thresholds; a component
// beyond the clip saturates and shows up in the row's measured residual; qd = q / s is the dequantised row) and an
// 80-byte record of what the pre-filter's candidate test needs.  The test is a
// RIGOROUS upper bound on the exact dot product <a, b> from integer arithmetic on a PREFIX of the rows:
//     <a, b> = (|a|^2 + |b|^2 - |a - b|^2) / 2                      (any vectors)
//     |a - b| >= |a_k - b_k| >= |qd_a,k - qd_b,k| - r_a - r_b       (prefix of k elements; r = |x - qd| over the WHOLE row,
//                                                                     measured when the row is quantised -- no error model)
//     |qd_a,k - qd_b,k|^2 = A_k + B_k - 2 <q_a,k, q_b,k> / (s_a s_b)  (A_k = |qd_a,k|^2; the integer dot product is exact)
// so  <a, b> >= gate  is impossible once  |qd_a,k - qd_b,k| > r_a + r_b + sqrt(|a|^2 + |b|^2 - 2 gate).  Evaluated without
// a square root per pair: the square of the right-hand side is bounded by a ROW part plus a COLUMN part (to_i8_row), so a
// pair is excluded when  c < uh_i + uh_j  with c = <q_a,k, q_b,k> / (s_a s_b); on the full row the direct form
// <a, b> <= c + |qd_a| r_b + |b| r_a is used (the same expression with cross terms alpha * r: i8_excluded).
// At k = D the bound is the quantised dot product plus the measured residuals (a band of ~0.02 for well-spread rows,
// against 0.14 of a worst-case element model of an 8-bit float format); at a prefix it lets a tile STOP the K loop as
// soon as (nearly) nothing in it can still reach the gate.  Non-finite rows get r = +inf: every pair with them is a
// candidate and the exact float32 dot product decides.
constexpr float I8_CLIP = 6.0f;               // int8 full scale, in units of 1 / sqrt(D) (the component RMS of a unit vector)
__host__ __device__ inline float i8_scale(int d) { return 127.0f * sqrtf((float)d) / I8_CLIP; }
constexpr int I8_NCHK = 4;                    // prefix checkpoints of the K loop (stages of I8_STAGE elements) besides the full row
constexpr int I8_STAGE = 128;                 // elements per K stage of the pre-filter GEMM
constexpr int I8_META = 4 * (I8_NCHK + 1);    // floats per row record
// record layout: five float4 {1/s, r, alpha_c, uh_c}, c = 0 .. I8_NCHK (the last: full row) -- the float4 of a checkpoint is what
// a candidate test reads, and one 16-byte global -> LDS transfer per row brings it there
// The FIRST checkpoint is where a prefix test starts to exclude anything: a prefix holding the fraction f of a unit row's
// energy excludes a pair of unrelated rows once  f > 1.03 (1 - gate) + slack  (uh > 0 in to_i8_row), i.e. f > 0.2575 + slack
// at the reference's gate 0.75.  `first16` = that fraction in sixteenths: 5 is the round-2 set {5, 7, 10, 13} (what the
// standalone diagnostics use); the tracker derives it from its gate (i8_first_checkpoint: 6 at 0.75 -- at 5/16 the test of
// a 128-row tile, which works with the SMALLEST prefix energy of its rows, excludes next to nothing, so every tile ran on
// to 7/16; with {6, 8, 11, 14} it stops one stage earlier).  BTS_I8_FIRST16 overrides (A/B).
__host__ __device__ inline int i8_first_checkpoint(float appearance_thresh) {
    const int f = (int)ceilf(16.0f * (1.03f * appearance_thresh + 0.10f));
    return f < 3 ? 3 : f > 12 ? 12 : f;
}
__host__ __device__ inline void i8_checkpoints(int kt_all, int (&c)[I8_NCHK], int first16 = 5) {
    // stages after which a tile may stop: ~first16/16 of the row, then +2/16, +5/16, +8/16 (strictly increasing, at least 3
    // stages -- shorter prefixes are statistically useless --, unused ones = kt_all)
    const int num[I8_NCHK] = {first16, first16 + 2, first16 + 5, first16 + 8};
    int prev = 2;
    for (int j = 0; j < I8_NCHK; ++j) {
        int v = (kt_all * num[j] + 15) / 16;
        if (v <= prev) v = prev + 1;
        if (v >= kt_all) v = kt_all;
        c[j] = v;
        prev = v < kt_all ? v : kt_all;
    }
}

// sum and maximum over the row group in one pass of barriers
__device__ __forceinline__ void block_sum_max(float &sum, float &mx) {
    __shared__ float part[2][16];
#pragma unroll
    for (int off = 32; off; off >>= 1) { sum += __shfl_xor(sum, off); mx = fmaxf(mx, __shfl_xor(mx, off)); }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bscan::lds_barrier();
    if (lane == 0) { part[0][wid] = sum; part[1][wid] = mx; }
    bscan::lds_barrier();
    const int w0 = row_group() * (ROW_G / 64);
    sum = ((part[0][w0] + part[0][w0 + 1]) + part[0][w0 + 2]) + part[0][w0 + 3];
    mx = fmaxf(fmaxf(part[1][w0], part[1][w0 + 1]), fmaxf(part[1][w0 + 2], part[1][w0 + 3]));
}
// six sums at once (the residual and the five prefix buckets of the squared quantised row)
__device__ __forceinline__ void block_sum6(float (&v)[6]) {
    __shared__ float part[6][16];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int off = 32; off; off >>= 1) v[q] += __shfl_xor(v[q], off);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bscan::lds_barrier();
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 6; ++q) part[q][wid] = v[q];
    bscan::lds_barrier();
    const int w0 = row_group() * (ROW_G / 64);
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = ((part[q][w0] + part[q][w0 + 1]) + part[q][w0 + 2]) + part[q][w0 + 3];
}

__device__ __forceinline__ int i8_round(float v) { return (int)__builtin_rintf(fminf(fmaxf(v, -127.0f), 127.0f)); }

// x[d] -> q[d] (int8), meta[I8_META].  `gate` = the similarity the appearance term must reach (1 - appearance_thresh).
// Called by all threads of the row group; inactive groups only take part in the barriers.
// Shadow layout.  A shadow array keeps its rows ROW-MAJOR (row r = d consecutive bytes: chunk stride I8_STAGE) or -- form
// BTS_SHADOW_PLANES -- in K-STAGE PLANES: plane s holds the s-th I8_STAGE-byte chunk of every row, rows consecutive inside a
// plane, so that the 128 rows x 128 bytes a pre-filter tile loads per K stage are (for consecutive table slots) ONE contiguous
// 16-KiB block instead of 128 pieces a whole feature row apart (a power-of-two stride at the reference's widths).  `q` is the
// row's first chunk, `cs` the bytes between consecutive chunks of the row: element e lives at q + (e / I8_STAGE) cs + e % I8_STAGE.
__host__ __device__ inline size_t i8_elem(int e, unsigned cs) { return (size_t)(e / I8_STAGE) * cs + (size_t)(e % I8_STAGE); }
__device__ __forceinline__ void to_i8_row(const float *x, int8_t *q, float *meta, int d, float gate, bool active = true, int first16 = 5, unsigned cs = I8_STAGE) {
    const int t = row_tid();
    int cst[I8_NCHK];
    i8_checkpoints(d / I8_STAGE, cst, first16);
    float ss = 0.0f, mx = 0.0f;
    const bool v4 = active && vec4_ok(x, d) && (uintptr_t)q % 4 == 0 && d / 4 <= 2 * ROW_G;
    float4 v[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (v4) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            if (k < d / 4) v[u] = x4[k];
            ss += v[u].x * v[u].x; ss += v[u].y * v[u].y; ss += v[u].z * v[u].z; ss += v[u].w * v[u].w;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
        }
    } else if (active) {
        for (int k = t; k < d; k += ROW_G) { const float e = x[k]; ss += e * e; mx = fmaxf(mx, fabsf(e)); }
    }
    block_sum_max(ss, mx);
    const bool finite = ss == ss && ss < 3.0e38f;            // (NaN / inf rows: every pair with them stays a candidate)
    const float s = i8_scale(d), inv = 1.0f / s;
    (void)mx;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // [0] residual^2, [1..5] squared quantised elements per prefix bucket
    auto bucket = [&](int elem) { const int st = elem / I8_STAGE; int b = 0;
#pragma unroll
        for (int j = 0; j < I8_NCHK; ++j) b += st >= cst[j] ? 1 : 0;
        return b; };
    auto add = [&](int b, float qq) { acc[1] += b == 0 ? qq : 0.f; acc[2] += b == 1 ? qq : 0.f; acc[3] += b == 2 ? qq : 0.f;
                                      acc[4] += b == 3 ? qq : 0.f; acc[5] += b == 4 ? qq : 0.f; };
    if (v4) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            if (k >= d / 4) continue;
            const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            unsigned w = 0;
            const int b = bucket(4 * k);                      // (a float4 never straddles a stage)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int qi = finite ? i8_round(e[c] * s) : 0;
                const float dq = (float)qi * inv, res = e[c] - dq;
                acc[0] += res * res;
                add(b, dq * dq);
                w |= (unsigned)(qi & 0xff) << (8 * c);
            }
            *reinterpret_cast<unsigned *>(q + i8_elem(4 * k, cs)) = w;      // (four elements never straddle a chunk)
        }
    } else if (active) {
        for (int k = t; k < d; k += ROW_G) {
            const float e = x[k];
            const int qi = finite ? i8_round(e * s) : 0;
            const float dq = (float)qi * inv, res = e - dq;
            acc[0] += res * res;
            add(bucket(k), dq * dq);
            q[i8_elem(k, cs)] = (int8_t)qi;
        }
    }
    block_sum6(acc);
    if (active && t == 0) {
        const float inf = __builtin_huge_valf();
        // slack: float32 evaluation of the sums above and of the test (~1e-6 relative), the 1e-7 difference between a raw
        // detection row and its in-place renormalised self (the track rows are copies of the latter, D:492-502 / Q8)
        const float r = finite ? sqrtf(acc[0]) * 1.0001f + 1e-5f : inf;
        float run = 0.0f;
#pragma unroll
        for (int c = 0; c <= I8_NCHK; ++c) {
            run += acc[1 + c];
            const float a_k = c == I8_NCHK ? ((acc[1] + acc[2]) + (acc[3] + acc[4])) + acc[5] : run;
            // Prefix records: (tau + r_a + r_b)^2 <= (1 + e) tau^2 + (1 + 1/e) (r_a + r_b)^2 <= 1.03 tau^2 + 68.7 (r_a^2 + r_b^2)  with
            // e = 0.03 (tight where r ~ 0.02 tau, which is where prefix exits happen) and tau^2 = (|a|^2 - gate) + (|b|^2 - gate):
            // the right-hand side splits into a row part and a column part, so the pair is excluded when
            //     c < uh_a + uh_b,   uh = (A_k - 1.03 (|x|^2 - gate) - 68.7 r^2) / 2     (alpha = 0: no cross terms).
            // Full row: the direct form  <a, b> <= c + |qd_a| r_b + |b| r_a,  same arithmetic with alpha = max(|x|, |qd|)
            // and uh = gate / 2.
            const float alpha = c == I8_NCHK ? fmaxf(sqrtf(a_k), sqrtf(ss)) * 1.00001f : 0.0f;
            const float uh = c == I8_NCHK ? 0.5f * gate : 0.5f * (a_k - 1.03f * (ss - gate) - 68.7f * r * r);
            meta[4 * c + 0] = inv; meta[4 * c + 1] = r;
            meta[4 * c + 2] = alpha;
            meta[4 * c + 3] = finite ? uh - 1e-4f * (1.0f + ss) : -inf;
        }
    }
}

// A detection row's record, as written by to_i8_row, describes the RAW row x (that is what the frame's similarities are
// taken with, D:1453-1483).  The track that absorbs the detection keeps  u = x / |x|  (the row is renormalised in place
// before it becomes the track's curr feature, D:492-502 / Q8), while its shadow stays the shadow of x.  For the pair
// (u, b):  <u, b> >= gate  <=>  <x, b> >= gate * n,  n = |x|  -- the same test with the pair's gate G = gate * n, i.e.
// with the ROW's share of 2 G raised from gate to gate (2 n - 1):
//     prefix records:  uh += 1.03 gate (n - 1)          full row:  uh += gate (n - 1)
// (n = 1: nothing changes; n < 1: the track's true similarities are LARGER than its shadow suggests and fewer pairs may
// be dropped).  Applied by the thread that renormalises the row, to the detection's record, which copy_shadow then hands
// to the track; nothing reads the record as a COLUMN record after that point of the frame.  A zero / non-finite norm
// makes the row a candidate for everybody (the exact float32 path decides).
__host__ __device__ inline void i8_track_side(float *meta, float nrm, float gate) {
    const bool ok = nrm > 0.0f && nrm < 3.0e38f;
    for (int c = 0; c <= I8_NCHK; ++c) {
        const float k = c == I8_NCHK ? 1.0f : 1.03f;
        meta[4 * c + 3] = ok ? meta[4 * c + 3] + k * gate * (nrm - 1.0f) : -__builtin_huge_valf();
    }
}

// "no pair of these two rows can reach the gate", from the accumulated integer dot product of the prefix the records'
// checkpoint c describes: ra / rb = float4 {1/s, r, alpha_c, uh_c} of the row / column.  NaN-safe: anything that does not
// compare stays a candidate.
__device__ __forceinline__ bool i8_excluded(int dot, const float4 &ra, const float4 &rb) {
    const float lhs = fmaf((float)dot * ra.x, rb.x, fmaf(ra.z, rb.y, rb.z * ra.y));
    return lhs < ra.w + rb.w;
}
__device__ __forceinline__ float4 i8_rec(const float *meta, int c) { return *reinterpret_cast<const float4 *>(meta + 4 * c); }

// <a, b> over d floats by one wavefront, every lane gets the result.  Fixed evaluation order
// (lane-strided fma chains, then an xor butterfly), so the value is reproducible.
// B = 16-byte loads in flight per lane and operand (register footprint 8*B; the value does not depend on it).
template <int B = 8>
__device__ __forceinline__ float wave_dot(const float *a, const float *b, int d, int lane) {
    float acc = 0.0f;
    if (vec4_ok(a, d) && vec4_ok(b, d)) {
        const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
        const int nv = d / 4;
        int k = lane;
        for (; k + (B - 1) * 64 < nv; k += B * 64) {    // 2*B independent 16-byte loads in flight per lane
            float4 x[B], y[B];
#pragma unroll
            for (int t = 0; t < B; ++t) { x[t] = a4[k + t * 64]; y[t] = b4[k + t * 64]; }
#pragma unroll
            for (int t = 0; t < B; ++t) {
                acc = fmaf(x[t].x, y[t].x, acc); acc = fmaf(x[t].y, y[t].y, acc);
                acc = fmaf(x[t].z, y[t].z, acc); acc = fmaf(x[t].w, y[t].w, acc);
            }
        }
        for (; k < nv; k += 64) {
            const float4 x = a4[k], y = b4[k];
            acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
    } else {
        for (int k = lane; k < d; k += 64) acc = fmaf(a[k], b[k], acc);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    return acc;
}

}  // namespace feat
