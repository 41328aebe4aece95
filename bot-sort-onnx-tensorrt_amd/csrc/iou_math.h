// iou_math.h -- one (track box, detection box) IoU cost with the reference's dtype flow.
//
// Restates bbox_iou / iou_distance (demo_bottrack_onnx_tflite.py:1695-1713, 1745-1761): boxes are
// tlbr, touching or disjoint boxes give IoU exactly 0 (cost exactly 1), no "+1" pixel convention.
// The reference evaluates the expression with NumPy *scalars*, so every intermediate is float32
// exactly when both of its operands are float32 (detections, never-predicted tracks) and float64
// otherwise; max/min return one of their arguments.  (sub / mul / area / cost are __host__ __device__: the CPU emulation
// tests of the exact-tie step run the very same arithmetic, tests/host_emul/lap_box_host.hip.)  Boxes are carried as float64 plus an
// "is float32" flag (a float32 value is exactly representable), and the float32 branches below
// reproduce those roundings bit for bit.  The final `1 - iou` is always float64.
#pragma once
#include <hip/hip_runtime.h>

namespace iou {

__host__ __device__ __forceinline__ double sub(double hi, double lo, bool f32) {
    return f32 ? (double)((float)hi - (float)lo) : hi - lo;
}
__host__ __device__ __forceinline__ double mul(double p, double q, bool f32) {
    return f32 ? (double)((float)p * (float)q) : p * q;
}

__host__ __device__ __forceinline__ double area(const double b[4], bool f32) {
    return mul(sub(b[2], b[0], f32), sub(b[3], b[1], f32), f32);
}

// cost = 1 - IoU(a, b); area_a / area_b precomputed with area() above.
__host__ __device__ __forceinline__ double cost(const double a[4], bool a32, double area_a, const double b[4],
                                       bool b32, double area_b) {
    const bool tx0 = b[0] > a[0], ty0 = b[1] > a[1], tx1 = b[2] < a[2], ty1 = b[3] < a[3];
    const double x_lo = tx0 ? b[0] : a[0], y_lo = ty0 ? b[1] : a[1];
    const double x_hi = tx1 ? b[2] : a[2], y_hi = ty1 ? b[3] : a[3];
    if (x_hi <= x_lo || y_hi <= y_lo) return 1.0;   // D:1705 (also the common case: no division)
    const bool w32 = (tx0 ? b32 : a32) && (tx1 ? b32 : a32);
    const bool h32 = (ty0 ? b32 : a32) && (ty1 ? b32 : a32);
    const bool i32 = w32 && h32;
    const double inter = mul(sub(x_hi, x_lo, w32), sub(y_hi, y_lo, h32), i32);
    const bool ab32 = a32 && b32;
    const double s = ab32 ? (double)((float)area_a + (float)area_b) : area_a + area_b;
    const double den = (ab32 && i32) ? (double)((float)s - (float)inter) : s - inter;
    const double q = i32 ? (double)((float)inter / (float)den) : inter / den;
    return 1.0 - q;
}

// ---- conservative overlap pre-test on 16-bit pixel boxes -------------------------------------------------------
// The N x M pair tests are instruction-bound (nearly all pairs are disjoint), so the screening box is two packed
// dwords -- lo = (floor x1, floor y1), hi = (ceil x2 - 1, ceil y2 - 1) -- and a pair costs v_pk_max_i16, v_pk_min_i16,
// v_pk_sub_i16 and one sign test.  Never misses an overlap: floor(a.x1) <= a.x1 < b.x2 <= ceil(b.x2), and the clamps
// (lo to [-16000, 15999], hi to [-15999, 16000]) keep lo < hi for every pair that satisfied it before clamping, so
// boxes beyond the 16-bit range only make the test say "maybe".  (The exact float64 IoU decides afterwards.)
typedef short s16x2 __attribute__((ext_vector_type(2)));
struct ScreenBox { s16x2 lo, hi; };

// (round towards -inf / +inf with saturation; NaN -> 0 like the device conversion.  The host forms exist for the CPU property
//  test of the screening, tests/test_product_math_host.py: the device pass uses the hardware conversions as before.)
__host__ __device__ __forceinline__ int screen_floor(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2int_rd(v);
#else
    return v != v ? 0 : v <= -2147483648.0 ? (int)0x80000000 : v >= 2147483647.0 ? 0x7fffffff : (int)__builtin_floor(v);
#endif
}
__host__ __device__ __forceinline__ int screen_ceil(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2int_ru(v);
#else
    return v != v ? 0 : v <= -2147483648.0 ? (int)0x80000000 : v >= 2147483647.0 ? 0x7fffffff : (int)__builtin_ceil(v);
#endif
}
__host__ __device__ __forceinline__ ScreenBox screen_box(const double b[4]) {
    auto lo = [](double v) { const int i = screen_floor(v); return (short)(i < -16000 ? -16000 : i > 15999 ? 15999 : i); };
    auto hi = [](double v) { const int i = screen_ceil(v); return (short)((i < -15999 ? -15999 : i > 16000 ? 16000 : i) - 1); };
    ScreenBox s;
    s.lo = s16x2{lo(b[0]), lo(b[1])};
    s.hi = s16x2{hi(b[2]), hi(b[3])};
    return s;
}

// a box that overlaps nothing (lo beyond every hi)
__host__ __device__ __forceinline__ ScreenBox screen_none() { ScreenBox s; s.lo = s16x2{16001, 16001}; s.hi = s16x2{-16001, -16001}; return s; }

__host__ __device__ __forceinline__ bool screen_overlap(const ScreenBox &a, const ScreenBox &b) {
    const s16x2 lo = __builtin_elementwise_max(a.lo, b.lo), hi = __builtin_elementwise_min(a.hi, b.hi);
    const s16x2 dd = hi - lo;                      // both halves >= 0 <=> lo < hi + 1 in x and in y
    return (__builtin_bit_cast(unsigned, dd) & 0x80008000u) == 0u;
}

}  // namespace iou
