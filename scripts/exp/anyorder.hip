// Experiment: does hipExtAnyOrderLaunch (AQL packets without the barrier bit) let consecutive kernels of ONE stream overlap
// on gfx950, and what does a dependent hand-off through a device-scope flag cost compared with a kernel boundary?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_kernel(long long *ts, int us) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (long long)us * 100) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) { ts[0] = t0; ts[1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void stamp_kernel(long long *ts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[2] = __builtin_amdgcn_s_memrealtime();
}
// chain link k: wait until flag >= k * nblk, do a token of work, release, bump the flag
__global__ void link_kernel(int *flag, int *data, int k, int nblk, int use_flag) {
    if (use_flag) {
        if (threadIdx.x == 0) { while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k * nblk) __builtin_amdgcn_s_sleep(1); }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (agent scope by default in HIP)
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    data[i] = data[i] + 1;                         // every link increments: a race would show as a wrong total
    if (use_flag) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    long long *ts; CK(hipMalloc(&ts, 64)); CK(hipMemset(ts, 0, 64));
    // 1. overlap test
    for (int flags = 0; flags < 2; ++flags) {
        CK(hipMemset(ts, 0, 64)); CK(hipDeviceSynchronize());
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ts, 50);
        hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, ts);
        CK(hipStreamSynchronize(s));
        long long h[3]; CK(hipMemcpy(h, ts, 24, hipMemcpyDeviceToHost));
        printf("flags=%d: spin [0, %.2f us], second kernel starts at %.2f us -> %s\n", flags, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0,
               h[2] < h[1] ? "OVERLAPS" : "serialised");
    }
    // 2. chains of dependent links
    const int nblk = 256, nth = 256, links = 11, reps = 200;
    int *flag, *data; CK(hipMalloc(&flag, 4)); CK(hipMalloc(&data, nblk * nth * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(data, 0, nblk * nth * 4)); CK(hipMemset(flag, 0, 4)); CK(hipDeviceSynchronize());
        int k = 0;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                for (int l = 0; l < links; ++l, ++k)
                    hipExtLaunchKernelGGL(link_kernel, dim3(nblk), dim3(nth), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, flag, data, k, nblk, mode);
        }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int> h(nblk * nth); CK(hipMemcpy(h.data(), data, h.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int v : h) bad += v != k;
        printf("%s: %.2f us per dependent link (%d links of %d x %d threads), wrong totals: %d\n", mode ? "any-order + device flag" : "stream order (barrier bit)",
               ms * 1e3 / (reps * links), reps * links, nblk, nth, bad);
    }
    return 0;
}
