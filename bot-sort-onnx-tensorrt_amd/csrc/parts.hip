// parts.hip -- detection-side part association: which head / face / hand box
// belongs to which body box, as the reference's find_most_relevant_object decides it (D:1763-1791,
// driven by the loops at D:1379-1411).
//
// The reference walks the bases in order; each base scans every still-unused target, keeps the one
// with the largest integer-box IoU (bbox_iou_by_box, D:1715-1729; > 0 required), breaks IoU ties
// by the smaller centre distance and remaining ties by list order, and marks it used.  The order
// dependence is real (a target taken by an earlier base is gone), so the work is split in two:
//   k_parts_candidates  one wavefront per base, all bases in parallel: the targets that overlap
//                       the base at all (IoU > 0), with IoU and squared centre distance, compacted
//                       in target order.  nb x nt box tests, embarrassingly parallel.
//   k_parts_resolve     one wavefront walks the bases in order and takes the best unused
//                       candidate of each (lane = candidate, xor-butterfly arg-max on the key
//                       (IoU desc, distance^2 asc, index asc)); the used flags live in LDS.
// A base with more than PARTS_CAND overlapping targets is resolved by scanning all targets
// in-line (same key), so the result is exact for any input.
// Distances are compared as exact integer squares: the reference compares (dx^2+dy^2)**0.5, and
// sqrt is strictly monotone on distinct integers < 2^53, so the order is the same.
#include "bts_internal.h"

namespace {

constexpr int PARTS_CAND = 64;      // candidate slots per base = one wavefront

struct Cand { double iou; long long d2; int t; };

__device__ __forceinline__ bool better(const Cand &a, const Cand &b) {     // a beats b
    if (a.iou != b.iou) return a.iou > b.iou;
    if (a.d2 != b.d2) return a.d2 < b.d2;
    return a.t < b.t;
}

__device__ __forceinline__ Cand evaluate(const int4 b, const int2 bc, const int4 *__restrict__ tb,
                                         const int2 *__restrict__ tc, int t) {
    const int4 q = tb[t];
    const int2 qc = tc[t];
    Cand c;
    c.t = t;
    const long long dx = (long long)bc.x - qc.x, dy = (long long)bc.y - qc.y;
    c.d2 = dx * dx + dy * dy;
    const int xl = max(b.x, q.x), yl = max(b.y, q.y), xh = min(b.z, q.z), yh = min(b.w, q.w);
    if (xh <= xl || yh <= yl) { c.iou = 0.0; return c; }
    const long long inter = (long long)(xh - xl) * (yh - yl);
    const long long uni = (long long)(b.z - b.x) * (b.w - b.y) + (long long)(q.z - q.x) * (q.w - q.y) - inter;
    c.iou = (double)inter / (double)uni;
    return c;
}

__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        Cand o;
        o.iou = __shfl_xor(c.iou, s);
        o.d2 = __shfl_xor(c.d2, s);
        o.t = __shfl_xor(c.t, s);
        if (better(o, c)) c = o;
    }
    return c;
}

__global__ __launch_bounds__(256) void k_parts_candidates(const int4 *__restrict__ bb, const int2 *__restrict__ bc, int nb,
                                                          const int4 *__restrict__ tb, const int2 *__restrict__ tc, int nt,
                                                          int32_t *__restrict__ cnt, int32_t *__restrict__ ct,
                                                          double *__restrict__ ciou, long long *__restrict__ cd2) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nb) return;
    const int4 box = bb[b];
    const int2 cen = bc[b];
    int n = 0;
    for (int t0 = 0; t0 < nt; t0 += 64) {
        const int t = t0 + lane;
        Cand c;
        c.iou = 0.0;
        if (t < nt) c = evaluate(box, cen, tb, tc, t);
        const bool hit = c.iou > 0.0;
        const unsigned long long mask = __ballot(hit);
        const int pos = n + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && pos < PARTS_CAND) {
            ct[(long)b * PARTS_CAND + pos] = t;
            ciou[(long)b * PARTS_CAND + pos] = c.iou;
            cd2[(long)b * PARTS_CAND + pos] = c.d2;
        }
        n += __popcll(mask);
    }
    if (lane == 0) cnt[b] = n;      // > PARTS_CAND tells the resolver to rescan this base
}

__global__ __launch_bounds__(64) void k_parts_resolve(const int4 *__restrict__ bb, const int2 *__restrict__ bc, int nb,
                                                      const int4 *__restrict__ tb, const int2 *__restrict__ tc, int nt,
                                                      const int32_t *__restrict__ cnt, const int32_t *__restrict__ ct,
                                                      const double *__restrict__ ciou, const long long *__restrict__ cd2,
                                                      uint8_t *__restrict__ used_io, int picks, int32_t *__restrict__ out) {
#if defined(BTS_HOST_EMU)
    uint8_t *const used = reinterpret_cast<uint8_t *>(emu::dynamic_lds());
#else
    extern __shared__ uint8_t used[];
#endif
    const int lane = threadIdx.x;
    for (int t = lane; t < nt; t += 64) used[t] = used_io[t];
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int n = cnt[b];
        Cand mine;
        mine.iou = 0.0; mine.d2 = 0; mine.t = 0x7fffffff;
        if (n <= PARTS_CAND && lane < n) {
            mine.t = ct[(long)b * PARTS_CAND + lane];
            mine.iou = ciou[(long)b * PARTS_CAND + lane];
            mine.d2 = cd2[(long)b * PARTS_CAND + lane];
        }
        for (int p = 0; p < picks; ++p) {
            Cand c;
            c.iou = 0.0; c.d2 = 0; c.t = 0x7fffffff;
            if (n > PARTS_CAND) {                 // crowded base: scan every target
                const int4 box = bb[b];
                const int2 cen = bc[b];
                for (int t = lane; t < nt; t += 64) {
                    if (used[t]) continue;
                    const Cand e = evaluate(box, cen, tb, tc, t);
                    if (e.iou > 0.0 && better(e, c)) c = e;
                }
            } else if (lane < n && !used[mine.t]) {
                c = mine;
            }
            c = wave_best(c);
            const int pick = c.iou > 0.0 ? c.t : -1;
            if (lane == 0) {
                out[(long)b * picks + p] = pick;
                if (pick >= 0) used[pick] = 1;
            }
            __syncthreads();
        }
    }
    for (int t = lane; t < nt; t += 64) used_io[t] = used[t];
}

}  // namespace

extern "C" int bts_associate_parts(bts_context *ctx, const int32_t *base_boxes, const int32_t *base_centres, int nb,
                                   const int32_t *target_boxes, const int32_t *target_centres, int nt,
                                   uint8_t *target_used, int picks, int32_t *out_idx, bts_stream stream) {
    BTS_REQUIRE(ctx && nb >= 0 && nt >= 0, "ctx, nb, nt");
    BTS_ON_DEVICE(ctx);
    BTS_REQUIRE(picks >= 1 && picks <= 8, "picks must be in 1..8");
    BTS_REQUIRE(nt <= 65536, "at most 65536 target boxes");
    if (nb == 0) return BTS_OK;
    BTS_REQUIRE(base_boxes && base_centres && out_idx, "null pointer");
    BTS_REQUIRE(nt == 0 || (target_boxes && target_centres && target_used), "null pointer");
    BTS_REQUIRE((uintptr_t)base_boxes % 16 == 0 && (uintptr_t)target_boxes % 16 == 0, "box arrays must be 16-byte aligned");
    BTS_REQUIRE((uintptr_t)base_centres % 8 == 0 && (uintptr_t)target_centres % 8 == 0, "centre arrays must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t per = bts_align((size_t)nb * PARTS_CAND * 8);
    const size_t need = bts_align((size_t)nb * 4) + bts_align((size_t)nb * PARTS_CAND * 4) + 2 * per;
    if (int rc = ctx->scratch.ensure(need)) return rc;
    char *p = ctx->scratch.base;
    int32_t *cnt = (int32_t *)p;  p += bts_align((size_t)nb * 4);
    int32_t *ct = (int32_t *)p;   p += bts_align((size_t)nb * PARTS_CAND * 4);
    double *ciou = (double *)p;   p += per;
    long long *cd2 = (long long *)p;
    hipLaunchKernelGGL(k_parts_candidates, dim3(bts_cdiv(nb, 4)), dim3(256), 0, s, (const int4 *)base_boxes,
                       (const int2 *)base_centres, nb, (const int4 *)target_boxes, (const int2 *)target_centres, nt, cnt,
                       ct, ciou, cd2);
    BTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_parts_resolve, dim3(1), dim3(64), bts_align((size_t)max(nt, 1), 64), s, (const int4 *)base_boxes,
                       (const int2 *)base_centres, nb, (const int4 *)target_boxes, (const int2 *)target_centres, nt, cnt, ct,
                       ciou, cd2, target_used, picks, out_idx);
    BTS_LAUNCH_CHECK();
    return BTS_OK;
}
