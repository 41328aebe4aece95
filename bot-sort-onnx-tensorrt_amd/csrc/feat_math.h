// feat_math.h -- cooperative float32 feature-row arithmetic (D:492-514).
// A row is handled by a GROUP of ROW_G = 256 consecutive threads; a workgroup holds blockDim.x / 256
// groups working on different rows in lockstep (256-thread workgroups: one row; the 1024-thread list
// kernels: four rows).  Every function must be reached by all threads of the workgroup; a group
// without a row passes active = false (it only takes part in the barriers).
#pragma once
#include <hip/hip_runtime.h>

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
    __syncthreads();                      // protects `part` across back-to-back calls
    if (lane == 0) part[wid] = v;
    __syncthreads();
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

// ---- float16 shadow rows for the appearance pre-filter -----------------------------------------------
// h[k] = half(256 * f[k]) (the power-of-two scale keeps small components of a unit vector out of the
// float16 subnormal range; it is undone exactly on the accumulated dot product), *norm = ||f||_2.
constexpr float HALF_SCALE = 256.0f;
constexpr float HALF_UNSCALE = 1.0f / (256.0f * 256.0f);

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void to_half_row(const float *f, _Float16 *h, float *norm, int d, bool active = true) {
    const int t = row_tid();
    float ss = 0.0f;
    if (active && vec4_ok(f, d) && (uintptr_t)h % 8 == 0) {
        const float4 *f4 = reinterpret_cast<const float4 *>(f);
        half4 *h4 = reinterpret_cast<half4 *>(h);
        for (int k = t; k < d / 4; k += ROW_G) {
            const float4 v = f4[k];
            ss += v.x * v.x; ss += v.y * v.y; ss += v.z * v.z; ss += v.w * v.w;
            half4 o;
            o.x = (_Float16)(v.x * HALF_SCALE); o.y = (_Float16)(v.y * HALF_SCALE);
            o.z = (_Float16)(v.z * HALF_SCALE); o.w = (_Float16)(v.w * HALF_SCALE);
            h4[k] = o;
        }
    } else if (active) {
        for (int k = t; k < d; k += ROW_G) {
            const float v = f[k];
            ss += v * v;
            h[k] = (_Float16)(v * HALF_SCALE);
        }
    }
    const float tot = block_sum(ss);
    if (active && t == 0) *norm = sqrtf(tot);
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
