// block_scan.h -- workgroup-wide exclusive scan / ordered compaction built from wavefront __shfl
// scans (64 lanes, no LDS traffic inside a wave) plus one LDS hop across the <=16 waves.
// All functions must be reached by every thread of the workgroup (blockDim.x multiple of 64).
#pragma once
#include <hip/hip_runtime.h>

namespace bscan {

// exclusive scan of one int per thread; *total = sum over the workgroup.  smem: >= 17 ints.
__device__ __forceinline__ int exscan(int v, int *smem, int *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __syncthreads();                       // smem may still be read from a previous call
    if (lane == 63) smem[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        int w = lane < nw ? smem[lane] : 0;
        int wi = w;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int t = __shfl_up(wi, off);
            if (lane >= off) wi += t;
        }
        if (lane < nw) smem[lane] = wi - w;          // exclusive wave offsets
        if (lane == nw - 1) smem[16] = wi;           // grand total
    }
    __syncthreads();
    *total = smem[16];
    return smem[wid] + incl - v;
}

// Ordered compaction of the indices k in [0, L) with pred(k): emit(position, k).  Each thread
// owns a contiguous chunk, so positions follow index order.  Returns the number emitted.
template <class Pred, class Emit>
__device__ __forceinline__ int compact(int L, int *smem, Pred pred, Emit emit) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const int per = (L + nth - 1) / nth;
    const int lo = min(L, tid * per), hi = min(L, lo + per);
    int total;
    if (per <= 64) {                       // predicate evaluated once: results kept in a bit mask
        unsigned long long mask = 0;
        int k = lo;
        for (; k + 3 < hi; k += 4) {       // four independent predicate evaluations: their loads overlap
            const bool b0 = pred(k), b1 = pred(k + 1), b2 = pred(k + 2), b3 = pred(k + 3);
            mask |= (unsigned long long)((b0 ? 1 : 0) | (b1 ? 2 : 0) | (b2 ? 4 : 0) | (b3 ? 8 : 0)) << (k - lo);
        }
        for (; k < hi; ++k) mask |= (unsigned long long)(pred(k) ? 1 : 0) << (k - lo);
        int pos = exscan(__popcll(mask), smem, &total);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            emit(pos++, lo + b);
        }
    } else {
        int cnt = 0;
        for (int k = lo; k < hi; ++k) cnt += pred(k) ? 1 : 0;
        int pos = exscan(cnt, smem, &total);
        for (int k = lo; k < hi; ++k)
            if (pred(k)) emit(pos++, k);
    }
    __syncthreads();
    return total;
}

}  // namespace bscan
