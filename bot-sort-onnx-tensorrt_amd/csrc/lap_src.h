// lap_src.h -- cost sources of the exact-tie replay (lap_exact.h: Src::real(i, j) = entry of a real pair that the edge list does
// not override).  __host__ __device__: k_lap_exact (lap.hip) uses them on the device, tests/host_emul/lap_box_host.hip on the CPU.
#pragma once
#include "iou_math.h"

struct DenseSrc {                       // bts_lapjv_exact: the caller's matrix is the whole truth
    static constexpr bool patched = false;
    const double *c; int ld;
    __host__ __device__ __forceinline__ double real(int i, int j) const { return c[(long)i * ld + j]; }
};
struct BoxSrc {                         // the tracker: a pair that is not a feasible edge costs its IoU distance (D:1554, D:1570, D:1602)
    static constexpr bool patched = true;
    const double *abox;                 // [rows][4] tlbr of the stage's rows (pool / stage-2 rows / unconfirmed)
    const uint8_t *af32;                // [rows] "the row's box is float32-valued" (never-predicted unconfirmed tracks), nullable = no
    const double *bbox;                 // [dets][4] tlbr of the frame's detections (always float32-valued)
    const int32_t *bidx;                // [cols] column -> detection index (hi_idx / lo_idx / left_det)
    __host__ __device__ __forceinline__ double real(int i, int j) const {
        const double *a = abox + (long)i * 4, *b = bbox + (long)bidx[j] * 4;
        const double aa[4] = {a[0], a[1], a[2], a[3]}, bb[4] = {b[0], b[1], b[2], b[3]};
        const bool fa = af32 ? af32[i] != 0 : false;
        return iou::cost(aa, fa, iou::area(aa, fa), bb, true, iou::area(bb, true));
    }
};
