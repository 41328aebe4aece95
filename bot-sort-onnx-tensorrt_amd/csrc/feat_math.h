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

// row /= ||row||_2   (np.linalg.norm of a float32 vector = sqrt(dot(x,x)) in float32)
__device__ __forceinline__ void renorm_row(float *row, int d, bool active = true) {
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
        return;
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
    } else {
        if (active) for (int k = t; k < d; k += ROW_G) ss += row[k] * row[k];
        const float nrm = sqrtf(block_sum(ss));
        if (active) for (int k = t; k < d; k += ROW_G) row[k] /= nrm;
    }
}

// (8-bit shadows of the pre-filter: definitions further down)
constexpr float Q8_SCALE = 128.0f;
__device__ __forceinline__ unsigned q8_pack4(float a, float b, float c, float e) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, e, w, true);
    return (unsigned)w;
}

// curr = f ; smooth = normalise(has ? alpha*smooth + beta*f : f)
// unit_q8 / unit_norm (optional): also write the 8-bit shadow of f for the pre-filter, for rows that are KNOWN to be
// unit vectors (the detection rows were renormalised in place before any track consumes them): the shadow is then
// q(128 f) with no second reduction and the stored norm is 1 (the 1e-7 deviation of a renormalised row from unit
// length is far inside the slack of feat::q8_cos_error).
__device__ __forceinline__ bool ema_row(float *curr, float *smooth, const float *f, int d, bool has, float alpha,
                                        float beta, bool active = true, uint8_t *unit_q8 = nullptr, float *unit_norm = nullptr) {
    const int t = row_tid();
    float ss = 0.0f;
    if (active && vec4_ok(curr, d) && vec4_ok(smooth, d) && vec4_ok(f, d) && d / 4 <= 2 * ROW_G) {      // 16 B per lane, values stay in registers
        const float4 *f4 = reinterpret_cast<const float4 *>(f);
        float4 *c4 = reinterpret_cast<float4 *>(curr), *s4 = reinterpret_cast<float4 *>(smooth);
        if (unit_q8) {
            unsigned *h4 = reinterpret_cast<unsigned *>(unit_q8);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = t + u * ROW_G;
                if (k < d / 4) { const float4 v = f4[k]; h4[k] = q8_pack4(v.x * Q8_SCALE, v.y * Q8_SCALE, v.z * Q8_SCALE, v.w * Q8_SCALE); }
            }
            if (t == 0) *unit_norm = 1.0f;
        }
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
        return unit_q8 != nullptr;          // shadow written
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
    return false;
}

// ---- 8-bit shadow rows for the appearance pre-filter ----------------------------------------------------
// q[k] = fp8_e4m3(Q8_SCALE * f[k] / ||f||_2), *norm = ||f||_2.  The shadow is the UNIT vector (no component can
// saturate whatever the norm of the row is; 128 keeps the largest component below the maximum of both e4m3
// flavours), so the accumulated dot product of two shadows approximates the COSINE; the norms travel separately.
// Error model of one element x of a unit vector (validated on the device by tests, bts_debug_q8_roundtrip):
//     |q(x) - x| <= Q8_REL * |x| + Q8_ABS        (round to nearest, 3 mantissa bits; Q8_ABS covers the subnormal
//                                                 range even if the conversion flushes it to zero)
// => |cos_q - cos| <= (2 Q8_REL + Q8_REL^2) + 2 (1 + Q8_REL) Q8_ABS sqrt(D) + D Q8_ABS^2   (+ float32 accumulation).
constexpr float Q8_UNSCALE = 1.0f / (128.0f * 128.0f);
constexpr float Q8_REL = 0.0625f;                 // 2^-4
constexpr float Q8_ABS = 1.0f / 8192.0f;          // 2^-6 (smallest normal of e4m3fn) / 128
__host__ __device__ inline float q8_cos_error(int d) {
    return (2.0f * Q8_REL + Q8_REL * Q8_REL) + 2.0f * (1.0f + Q8_REL) * Q8_ABS * sqrtf((float)d) + (float)d * Q8_ABS * Q8_ABS +
           (float)d / 4194304.0f + 1e-3f;
}


__device__ __forceinline__ void to_q8_row(const float *f, uint8_t *h, float *norm, int d, bool active = true) {
    const int t = row_tid();
    float ss = 0.0f;
    const bool v4 = active && vec4_ok(f, d) && (uintptr_t)h % 4 == 0;
    if (v4 && d / 4 <= 2 * ROW_G) {                       // the row stays in registers between the two sweeps
        const float4 *f4 = reinterpret_cast<const float4 *>(f);
        unsigned *h4 = reinterpret_cast<unsigned *>(h);
        float4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            v[u] = k < d / 4 ? f4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += v[u].x * v[u].x; ss += v[u].y * v[u].y; ss += v[u].z * v[u].z; ss += v[u].w * v[u].w;
        }
        const float nrm = sqrtf(block_sum(ss));
        const float sc = nrm > 0.0f ? Q8_SCALE / nrm : (nrm == 0.0f ? 0.0f : nrm);       // NaN norms poison the row on purpose
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t + u * ROW_G;
            if (k < d / 4) h4[k] = q8_pack4(v[u].x * sc, v[u].y * sc, v[u].z * sc, v[u].w * sc);
        }
        if (t == 0) *norm = nrm;
        return;
    }
    if (active) for (int k = t; k < d; k += ROW_G) ss += f[k] * f[k];
    const float nrm = sqrtf(block_sum(ss));
    const float sc = nrm > 0.0f ? Q8_SCALE / nrm : (nrm == 0.0f ? 0.0f : nrm);
    if (active) {
        for (int k = t; k < d; k += ROW_G) h[k] = (uint8_t)(q8_pack4(f[k] * sc, 0.f, 0.f, 0.f) & 0xffu);
        if (t == 0) *norm = nrm;
    }
}

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
