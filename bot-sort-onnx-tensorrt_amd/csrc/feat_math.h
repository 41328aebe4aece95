// feat_math.h -- workgroup-cooperative float32 feature-row arithmetic (D:492-514).
// Every function must be called by all threads of a workgroup (blockDim.x a multiple of 64,
// at most 1024) with workgroup-uniform arguments.
#pragma once
#include <hip/hip_runtime.h>

namespace feat {

// sum over the workgroup; result returned to every thread
__device__ __forceinline__ float block_sum(float v) {
    __shared__ float part[16];
    __shared__ float total;
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();                      // protects `part`/`total` across back-to-back calls
    if (lane == 0) part[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < nw; ++w) s += part[w];
        total = s;
    }
    __syncthreads();
    return total;
}

__device__ __forceinline__ bool vec4_ok(const void *p, int d) { return ((uintptr_t)p % 16 == 0) && (d % 4 == 0); }

// row /= ||row||_2   (np.linalg.norm of a float32 vector = sqrt(dot(x,x)) in float32)
__device__ __forceinline__ void renorm_row(float *row, int d) {
    float ss = 0.0f;
    if (vec4_ok(row, d)) {
        float4 *r4 = reinterpret_cast<float4 *>(row);
        for (int k = threadIdx.x; k < d / 4; k += blockDim.x) {
            const float4 v = r4[k];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        const float nrm = sqrtf(block_sum(ss));
        for (int k = threadIdx.x; k < d / 4; k += blockDim.x) {
            float4 v = r4[k];
            v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
            r4[k] = v;
        }
    } else {
        for (int k = threadIdx.x; k < d; k += blockDim.x) ss += row[k] * row[k];
        const float nrm = sqrtf(block_sum(ss));
        for (int k = threadIdx.x; k < d; k += blockDim.x) row[k] /= nrm;
    }
}

// curr = f ; smooth = normalise(has ? alpha*smooth + beta*f : f)
__device__ __forceinline__ void ema_row(float *curr, float *smooth, const float *f, int d, bool has, float alpha,
                                        float beta) {
    float ss = 0.0f;
    if (vec4_ok(curr, d) && vec4_ok(smooth, d) && vec4_ok(f, d)) {      // 16 B per lane, values stay in registers
        const float4 *f4 = reinterpret_cast<const float4 *>(f);
        float4 *c4 = reinterpret_cast<float4 *>(curr), *s4 = reinterpret_cast<float4 *>(smooth);
        const int nv = d / 4;
        if (nv <= 2 * (int)blockDim.x) {
            float4 fv[2], sv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = threadIdx.x + t * blockDim.x;
                if (k < nv) {
                    fv[t] = f4[k];
                    if (has) {
                        const float4 o = s4[k];
                        sv[t].x = alpha * o.x + beta * fv[t].x; sv[t].y = alpha * o.y + beta * fv[t].y;
                        sv[t].z = alpha * o.z + beta * fv[t].z; sv[t].w = alpha * o.w + beta * fv[t].w;
                    } else sv[t] = fv[t];
                    ss += sv[t].x * sv[t].x + sv[t].y * sv[t].y + sv[t].z * sv[t].z + sv[t].w * sv[t].w;
                }
            }
            const float nrm = sqrtf(block_sum(ss));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = threadIdx.x + t * blockDim.x;
                if (k < nv) {
                    c4[k] = fv[t];
                    s4[k] = make_float4(sv[t].x / nrm, sv[t].y / nrm, sv[t].z / nrm, sv[t].w / nrm);
                }
            }
            return;
        }
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const float fv = f[k];
        const float s = has ? alpha * smooth[k] + beta * fv : fv;
        ss += s * s;
    }
    const float nrm = sqrtf(block_sum(ss));
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
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

__device__ __forceinline__ void to_half_row(const float *f, _Float16 *h, float *norm, int d) {
    float ss = 0.0f;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const float v = f[k];
        ss += v * v;
        h[k] = (_Float16)(v * HALF_SCALE);
    }
    const float tot = block_sum(ss);
    if (threadIdx.x == 0) *norm = sqrtf(tot);
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
