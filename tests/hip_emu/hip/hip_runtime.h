// tests/hip_emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a CPU stand-in for the subset of HIP that csrc/*.hip uses.
//
// Purpose: round 3 had no GPU.  With this header in front of the include path (and -DBTS_HOST_EMU) the kernel SOURCES of the
// product compile as plain C++ (clang++) and run on the CPU: every GPU thread is a fibre (ucontext), a workgroup's fibres are
// scheduled cooperatively, __syncthreads() and the wavefront collectives (__shfl*, __ballot, readlane, DPP, ...) are rendezvous
// points -- a wavefront's lanes that arrive at the same call site exchange values exactly as the 64 lanes of a CDNA wavefront
// do; lanes that diverged to another call site wait their turn (lower source position first).  Workgroups of a launch run one
// after the other in ascending blockIdx order, launches and copies complete at once (every stream is in order).  What it does
// NOT model: timing, memory ordering bugs, MFMA / LDS-DMA / inline assembly (the pre-filter GEMM is compiled out: the emulated
// library serves the DENSE appearance mode only).  tests/test_emu_*.py run the parity tests of the GPU suite through it.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include <functional>
#include <vector>
#include <algorithm>

// ---- qualifiers -------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __constant__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define HIP_SYMBOL(x) (&(x))
#define amdgpu_waves_per_eu(...) unused        /* __attribute__((amdgpu_waves_per_eu(a, b))) on a kernel: an occupancy hint */

// ---- vector types -----------------------------------------------------------------------------------------------------------
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
struct double4 { double x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- the emulator core (hipemu.cpp) -----------------------------------------------------------------------------------------
namespace emu {
struct Thread;                                   // one GPU thread = one fibre
Thread *cur();
unsigned tid_x();
unsigned bid(int axis);
unsigned bdim(int axis);
unsigned gdim(int axis);
void launch(dim3 grid, dim3 block, size_t dyn_lds, const std::function<void()> &body);
void barrier();                                  // __syncthreads
int barrier_or(int v);
char *dynamic_lds();
// A thread that keeps spinning on something another WORKGROUP has to write (workgroups run one at a time here): after a few
// spins the whole workgroup is abandoned and re-run from its start once the rest of the launch has run -- valid because
// everything this code base does in front of such a wait is idempotent (counting, and a publish of the same value).
void spin();
// wavefront collectives: every participating lane calls with the same `site`
enum Op { OP_SHFL, OP_SHFL_XOR, OP_SHFL_UP, OP_BALLOT, OP_READLANE, OP_READFIRST, OP_DPP, OP_WAVE_BARRIER, OP_GATHER };
uint64_t collective(Op op, int site, uint64_t value, int arg, int arg2, uint64_t old_value);
// operand exchange of a matrix instruction: every participating lane deposits `n` <= GATHER_MAX bytes; returns the wavefront's
// staging area [64][GATHER_MAX] (lanes that do not take part read as zero), valid until the wavefront's next gather
constexpr int GATHER_MAX = 64;
const char *gather(int site, const void *mine, int n);
}  // namespace emu

struct EmuIdx { struct F { int axis; int which; operator unsigned() const; } x{0, 0}, y{1, 0}, z{2, 0}; };
struct EmuThreadIdx { struct F { operator unsigned() const { return emu::tid_x(); } } x; struct Z { operator unsigned() const { return 0u; } } y, z; };
struct EmuBlockIdx { struct X { operator unsigned() const { return emu::bid(0); } } x; struct Y { operator unsigned() const { return emu::bid(1); } } y; struct Z { operator unsigned() const { return emu::bid(2); } } z; };
struct EmuBlockDim { struct X { operator unsigned() const { return emu::bdim(0); } } x; struct Y { operator unsigned() const { return emu::bdim(1); } } y; struct Z { operator unsigned() const { return emu::bdim(2); } } z; };
struct EmuGridDim { struct X { operator unsigned() const { return emu::gdim(0); } } x; struct Y { operator unsigned() const { return emu::gdim(1); } } y; struct Z { operator unsigned() const { return emu::gdim(2); } } z; };
static EmuThreadIdx threadIdx;
static EmuBlockIdx blockIdx;
static EmuBlockDim blockDim;
static EmuGridDim gridDim;
constexpr int warpSize = 64;

// ---- barriers, fences -------------------------------------------------------------------------------------------------------
static inline void __syncthreads() { emu::barrier(); }
static inline int __syncthreads_or(int v) { return emu::barrier_or(v); }
static inline void __threadfence() {}
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_sleep(n) emu::spin()      /* only used inside spin-waits on another workgroup */
#define __builtin_amdgcn_wave_barrier() ((void)emu::collective(emu::OP_WAVE_BARRIER, __COUNTER__, 0, 0, 0, 0))
static inline unsigned long long __builtin_amdgcn_s_memrealtime() { static unsigned long long t = 0; return t += 1; }
#define __ATOMIC_RELAXED_EMU 0
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))

// ---- wavefront collectives (macros: __COUNTER__ identifies the call site) ----------------------------------------------------
namespace emu {
template <class T> inline uint64_t bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T unbits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
template <class T> inline T shfl(int site, T v, int src, int width) { return unbits<T>(collective(OP_SHFL, site, bits(v), src, width, bits(v))); }
template <class T> inline T shfl_xor(int site, T v, int mask, int width) { return unbits<T>(collective(OP_SHFL_XOR, site, bits(v), mask, width, bits(v))); }
template <class T> inline T shfl_up(int site, T v, int delta, int width) { return unbits<T>(collective(OP_SHFL_UP, site, bits(v), delta, width, bits(v))); }
}  // namespace emu
#define EMU_SHFL3(v, s, w) emu::shfl(__COUNTER__, (v), (int)(s), (int)(w))
#define EMU_SHFL2(v, s) emu::shfl(__COUNTER__, (v), (int)(s), 64)
#define EMU_PICK3(_1, _2, _3, NAME, ...) NAME
#define __shfl(...) EMU_PICK3(__VA_ARGS__, EMU_SHFL3, EMU_SHFL2)(__VA_ARGS__)
#define EMU_SHFLX3(v, s, w) emu::shfl_xor(__COUNTER__, (v), (int)(s), (int)(w))
#define EMU_SHFLX2(v, s) emu::shfl_xor(__COUNTER__, (v), (int)(s), 64)
#define __shfl_xor(...) EMU_PICK3(__VA_ARGS__, EMU_SHFLX3, EMU_SHFLX2)(__VA_ARGS__)
#define EMU_SHFLU3(v, s, w) emu::shfl_up(__COUNTER__, (v), (int)(s), (int)(w))
#define EMU_SHFLU2(v, s) emu::shfl_up(__COUNTER__, (v), (int)(s), 64)
#define __shfl_up(...) EMU_PICK3(__VA_ARGS__, EMU_SHFLU3, EMU_SHFLU2)(__VA_ARGS__)
#define __ballot(p) ((unsigned long long)emu::collective(emu::OP_BALLOT, __COUNTER__, (p) ? 1u : 0u, 0, 0, 0))
#define __any(p) (emu::collective(emu::OP_BALLOT, __COUNTER__, (p) ? 1u : 0u, 0, 0, 0) != 0)
#define __all(p) (emu::collective(emu::OP_BALLOT, __COUNTER__, (p) ? 0u : 1u, 0, 0, 0) == 0)
#define __builtin_amdgcn_readlane(v, l) ((int)emu::collective(emu::OP_READLANE, __COUNTER__, (uint64_t)(uint32_t)(v), (int)(l), 0, 0))
#define __builtin_amdgcn_readfirstlane(v) ((int)emu::collective(emu::OP_READFIRST, __COUNTER__, (uint64_t)(uint32_t)(v), 0, 0, 0))
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): 32-bit values
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rmask, bmask, bctrl) \
    ((int)emu::collective(emu::OP_DPP, __COUNTER__, (uint64_t)(uint32_t)(src), (int)(ctrl), (int)(rmask) | ((bctrl) ? 0x100 : 0), (uint64_t)(uint32_t)(old)))
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
static inline int __double2int_rd(double v) { return v != v ? 0 : v <= -2147483648.0 ? (int)0x80000000 : v >= 2147483647.0 ? 0x7fffffff : (int)floor(v); }
static inline int __double2int_ru(double v) { return v != v ? 0 : v <= -2147483648.0 ? (int)0x80000000 : v >= 2147483647.0 ? 0x7fffffff : (int)ceil(v); }
using std::min;
using std::max;

// ---- atomics (fibres never pre-empt each other: plain read-modify-write) ------------------------------------------------------
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
static inline int atomicAdd(int *p, unsigned v) { int o = *p; *p = o + (int)v; return o; }
template <class T> static inline T atomicSub(T *p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
static inline unsigned atomicOr(unsigned *p, int v) { unsigned o = *p; *p = o | (unsigned)v; return o; }

// ---- matrix instructions, LDS-DMA (instruction-level stand-ins: the lanes of a wavefront meet at the call site) ---------------
// v_mfma_i32_32x32x32_i8: lane l supplies 16 K-values of A row (l & 31) and of B column (l & 31), K half (l >> 5); it receives
// D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31] in c[r] -- the C/D map of the 32 x 32 MFMAs (MI355X_MICROARCH.md; what the
// kernels' epilogues index by, validated on hardware by the round-1/2 GPU runs).  Which K indices a lane holds cancels out
// as long as both operands use the same assignment.
typedef int emu_v4i __attribute__((ext_vector_type(4)));
typedef int emu_i32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
static inline emu_i32x16 emu_mfma_i32_32x32x32_i8(int site, emu_v4i a, emu_v4i b, emu_i32x16 c) {
    struct { emu_v4i a, b; } mine = {a, b};
    const char *all = emu::gather(site, &mine, 32);
    const int lane = (int)(emu::tid_x() & 63), j = lane & 31, half = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        int sum = 0;
        for (int h = 0; h < 2; ++h) {
            const int8_t *A = (const int8_t *)(all + (i + 32 * h) * emu::GATHER_MAX), *B = (const int8_t *)(all + (j + 32 * h) * emu::GATHER_MAX + 16);
            for (int k = 0; k < 16; ++k) sum += (int)A[k] * (int)B[k];
        }
        c[r] += sum;
    }
    return c;
}
#define __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, x, y, z) emu_mfma_i32_32x32x32_i8(__COUNTER__, (a), (b), (c))
// v_mfma_f32_32x32x2_f32: lane l supplies A[l & 31][l >> 5] and B[l >> 5][l & 31]; a k-ordered float32 fma chain
static inline emu_f32x16 emu_mfma_f32_32x32x2f32(int site, float a, float b, emu_f32x16 c) {
    struct { float a, b; } mine = {a, b};
    const char *all = emu::gather(site, &mine, 8);
    const int lane = (int)(emu::tid_x() & 63), j = lane & 31, half = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, all + (i + 32 * k) * emu::GATHER_MAX, 4);
            memcpy(&bv, all + (j + 32 * k) * emu::GATHER_MAX + 4, 4);
            c[r] = fmaf(av, bv, c[r]);
        }
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32(__COUNTER__, (a), (b), (c))
// (only named inside a discarded `if constexpr` branch of a timing experiment)
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, ...) (c)
// global_load_lds (16 B per lane): the LDS image is lane-linear from the wave-uniform base -- lane l's bytes land at base + 16 l.
// ASYNCHRONOUS like the hardware's (round 6; hipemu.cpp "LDS-DMA"): the bytes are fetched at issue, the destination is POISONED, and
// the data lands only when the issuing thread's wavefront executes its s_waitcnt vmcnt (emu::lds_dma_wait -- the kernels' explicit
// waits; nothing else completes a transfer here, which is stricter than the hardware, where a compiler-inserted wait may help by
// accident).  A landed transfer is another wavefront's to read only behind the next workgroup barrier.  Every LDS access the
// kernels make through their inline-assembly wrappers calls emu::lds_dma_check: touching bytes of a transfer that has not landed
// (or landed in another wavefront with no barrier since) aborts with the block / thread / address.  HIPEMU_LDS_DMA=sync: the old
// immediate copy; fault injection (the detector must then fire): HIPEMU_LDS_DMA=dropwaits -- the waits are ignored --, =late:N -- every
// wait leaves N transfers too many in flight, a mis-counted vmcnt.
namespace emu {
void lds_dma_issue(const void *g, void *dst, int n);
void lds_dma_wait(int keep);                                   // s_waitcnt vmcnt(keep), as far as this thread's LDS-DMA goes
void lds_dma_check(const void *p, int n, const char *what);    // an LDS read / write of [p, p + n)
}
static inline void emu_global_load_lds(const void *g, void *lds_base, int size) {
    emu::lds_dma_issue(g, (char *)lds_base + (size_t)(emu::tid_x() & 63) * (size_t)size, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((const void *)(g), (void *)(l), (size))

// ---- runtime API subset -----------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct EmuStream *hipStream_t;
// an event remembers the stream of its latest record and two counters: records enqueued / records executed (HIPEMU_STREAMS)
typedef struct EmuEvent { double t; struct EmuStream *owner; unsigned long long enq, done; } *hipEvent_t;
typedef struct EmuGraph *hipGraph_t;
typedef struct EmuGraphExec *hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipStreamCaptureModeRelaxed = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
namespace emu { void drain_all(); }
static inline hipError_t hipDeviceSynchronize() { emu::drain_all(); return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "CPU emulation of gfx950"); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; p->totalGlobalMem = (size_t)1 << 34; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { emu::drain_all(); free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
// ---- streams and events (HIPEMU_STREAMS) -------------------------------------------------------------------------------------
// Default ("eager"): every operation runs when it is enqueued, streams are in-order no-ops -- one legal schedule of a correct
// program, and blind to a MISSING cross-stream dependency.  HIPEMU_STREAMS=sidelate | mainlate makes one side of the library's
// two-stream forms (BTS_CVT_OVERLAP: the step's stream + a side stream, ordered by events only) LAZY: its operations queue up and
// run only when something legally forces them -- another stream's hipStreamWaitEvent on one of its records, a synchronising
// call, or a full queue -- i.e. as LATE as the program's own dependencies allow, while the other side runs at once.  "sidelate":
// created streams lag behind the null stream (which the tests step on); "mainlate": the null stream lags behind the created
// ones.  A dependency the program forgot to state (a side-stream kernel that overwrites a buffer the lagging stream has not
// consumed yet, a consumer that does not wait for its producer) then produces a different result, which the digests of
// tests/test_zz_gpu_round5.py::test_cvt_overlap_under_adversarial_stream_schedules catch.  Not a timing model: two legal extremes.
namespace emu {
void stream_op(hipStream_t s, std::function<void()> f);       // a kernel / asynchronous memory operation on stream s
void stream_record(hipStream_t s, hipEvent_t e);
void stream_wait(hipStream_t s, hipEvent_t e);
void stream_sync(hipStream_t s);                               // run everything queued on s (and what it waits for)
void event_sync(hipEvent_t e);
void drain_all();                                              // device idle
hipStream_t stream_create();
void stream_destroy(hipStream_t s);
}
static inline hipError_t hipHostFree(void *p) { emu::drain_all(); free(p); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { emu::drain_all(); memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t = nullptr);
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { emu::drain_all(); memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr);
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st = nullptr) {
    emu::stream_op(st, [=]() { for (size_t r = 0; r < h; ++r) memmove((char *)d + r * dp, (const char *)s + r * sp, w); });
    return hipSuccess;
}
template <class T> static inline hipError_t hipMemcpyFromSymbol(void *d, T *sym, size_t n) { emu::drain_all(); memcpy(d, (const void *)sym, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { emu::stream_sync(s); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = emu::stream_create(); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { emu::stream_destroy(s); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = emu::stream_create(); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { emu::stream_wait(s, e); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(EmuEvent)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { emu::drain_all(); free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr) { emu::stream_record(s, e); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { emu::event_sync(e); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { emu::event_sync(a); emu::event_sync(b); *ms = 0.001f; return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
// hipGraph: while a capture is open, launches / async copies / memsets are RECORDED as closures (arguments frozen by value,
// like kernel nodes) instead of executed; a graph launch runs the recorded list in order.  One capture at a time; forked
// capture streams and the events between them are no-ops here, as every stream is in-order.
struct EmuGraph { std::vector<std::function<void()>> nodes; };
struct EmuGraphExec { std::vector<std::function<void()>> nodes; };
namespace emu {
EmuGraph *&capturing();
void launch_or_record(dim3 grid, dim3 block, size_t dyn_lds, hipStream_t stream, std::function<void()> body);
static inline bool record(std::function<void()> f) { if (!capturing()) return false; capturing()->nodes.push_back(std::move(f)); return true; }
}
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { if (emu::capturing()) return hipErrorInvalidValue; emu::capturing() = new EmuGraph(); return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = emu::capturing(); emu::capturing() = nullptr; return *g ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, void *, void *, size_t) { *e = new EmuGraphExec{g->nodes}; return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s) { emu::stream_op(s, [e]() { for (auto &f : e->nodes) f(); }); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { emu::drain_all(); delete e; return hipSuccess; }

static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t st) { if (!emu::record([=]() { memset(p, v, n); })) emu::stream_op(st, [=]() { memset(p, v, n); }); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void *p, size_t pitch, int v, size_t w, size_t h, hipStream_t st) {
    auto f = [=]() { for (size_t r = 0; r < h; ++r) memset((char *)p + r * pitch, v, w); };
    if (!emu::record(f)) emu::stream_op(st, f);
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st) { if (!emu::record([=]() { memmove(d, s, n); })) emu::stream_op(st, [=]() { memmove(d, s, n); }); return hipSuccess; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch_or_record(dim3(grid), dim3(block), (size_t)(shmem), (hipStream_t)(stream), [=]() { kern(__VA_ARGS__); })
