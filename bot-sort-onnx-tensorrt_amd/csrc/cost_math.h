// cost_math.h -- the fused association costs of one (track, detection) pair (D:1539-1554, D:1590-1602) and the face-similarity
// fix-up (D:1481-1483).  __host__ __device__: k_fuse1 / k_fuse3 (assoc.hip) on the device, tests/host_emul/cost_host.hip on
// the CPU against the golden vectors recorded from the reference (G3).
#pragma once
#include <hip/hip_runtime.h>

// np.isclose(x, 0.9999999, atol=1e-8, rtol=1e-8) evaluated in float64 (D:1481-1483)
__host__ __device__ __forceinline__ float face_fixup(float s) {
    const double d = fabs((double)s - 0.9999999);
    return d <= 1e-08 + 1e-08 * 0.9999999 ? 0.0f : s;
}

// D:1539-1554.  emb/face distances are float32 (1.0 - float32), the thresholds compare in the
// dtype NumPy uses (float32 vs 0.25, float64 vs 0.5) and the minimum is float64.
__host__ __device__ __forceinline__ double fuse1(double iou_d, float body_sim, float face_sim, double prox, float app) {
    float emb = 1.0f - body_sim;
    const float face = 1.0f - face_sim;
    const bool far = fminf(emb, face) > app;
    if (far) emb = 1.0f;
    (void)prox;   // the proximity gate only re-sets entries that are already 1 (quirk Q2)
    return fmin(iou_d, (double)emb);
}

// D:1590-1602
__host__ __device__ __forceinline__ double fuse3(double iou_d, float body_sim, double prox, float app) {
    float emb = 1.0f - body_sim;
    if (emb > app) emb = 1.0f;
    if (iou_d > prox) emb = 1.0f;
    return fmin(iou_d, (double)emb);
}


// float64 -> IEEE binary16, round to nearest even, returned widened (bts_tracker_config.cost_dtype = BTS_F16: what
// numpy's astype(float16) does to a float64 cost).  Association costs live in [0, 1] (+inf for gated pairs); the general
// cases are kept so that the function is the conversion, not an approximation of it: overflow -> inf, subnormals, NaN.
__host__ __device__ inline double round_to_f16(double v) {
    if (!(v == v)) return v;
    const double a = fabs(v);
    if (a >= 65520.0) return v < 0 ? -__builtin_huge_val() : __builtin_huge_val();      // (65504 + half an ulp rounds to inf)
    if (a == 0.0) return v;
    int e;
    (void)frexp(a, &e);                      // a = f * 2^e, f in [0.5, 1)
    int q = e - 11;                          // ulp exponent of a binary16 normal: 11 significant bits
    if (q < -24) q = -24;                    // subnormals: fixed ulp 2^-24
    const double r = ldexp(nearbyint(ldexp(a, -q)), q);      // (nearbyint: ties to even in the default rounding mode)
    return v < 0 ? -r : r;
}
