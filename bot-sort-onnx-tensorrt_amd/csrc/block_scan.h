// block_scan.h -- workgroup-wide exclusive scan / ordered compaction built from wavefront __shfl
// scans (64 lanes, no LDS traffic inside a wave) plus one LDS hop across the <=16 waves.
// All functions must be reached by every thread of the workgroup (blockDim.x multiple of 64).
#include <cstdint>
#pragma once
#include <hip/hip_runtime.h>

namespace bscan {

// Workgroup barrier for hand-offs through LDS only: __syncthreads() also waits for every outstanding global
// store of the wavefront to be acknowledged (a memory round trip, ~1-2 us) -- needed only when global data
// written before the barrier is read by other threads after it.
#if defined(BTS_HOST_EMU)      // (tests/hip_emu: the CPU stand-in has one kind of workgroup barrier)
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// exclusive scan of one int per thread; *total = sum over the workgroup.  smem: >= 17 ints.
__device__ __forceinline__ int exscan(int v, int *smem, int *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    lds_barrier();                       // smem may still be read from a previous call
    if (lane == 63) smem[wid] = incl;
    lds_barrier();
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
    lds_barrier();
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
    lds_barrier();
    return total;
}

// Ordered compaction with a payload, wave-strided: wavefront w owns the contiguous index range
// [w * chunk, (w + 1) * chunk) and walks it 64 indices at a time, lane-strided, so that
//   * every load of the selection step is coalesced and independent of the others (one memory
//     latency for the whole pass instead of one per element),
//   * positions come from ballots (no per-element scan), and the payload `load` fetched travels in
//     registers to `emit` (no second, dependent read).
// load(k, v) -> keep?, may set v;  emit(position, k, v).  Up to 16 rounds per wavefront (L <= 16 * 64 *
// waves, i.e. 16384 at 1024 threads); longer inputs take the chunk-per-thread form above.
// load(k) -> int: first-level fetch of index k (the payload, e.g. the slot a list entry names);
// pred(k, v) -> keep?: may fetch through the payload (second level);  emit(position, k, v).
// The two levels are separate passes over the wavefront's R rounds, so that all first-level loads are in flight
// together and then all second-level ones: two memory latencies for the whole selection, whatever R is.
// (A predicate written `a && b` with loads on both sides would serialise them behind branches: use `&`.)
// k0 / basefn: the compaction may be one CHUNK of a longer one (see Chunk below): indices [k0, L), positions offset by
// basefn(local total), which every thread calls once between the counting and the emitting pass.
template <int R, class Load, class Pred, class Emit, class BaseFn>
__device__ __forceinline__ int compact_ws_r(int k0, int L, int chunk, int *smem, Load load, Pred pred, Emit emit, BaseFn basefn) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int base = k0 + wid * chunk;
    int val[R];
    unsigned long long bal[R];
    int cnt = 0;
    // no branches around the loads (out-of-range lanes re-read index k0)
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = base + j * 64 + lane;
        val[j] = load((j * 64 < chunk && k < L) ? k : k0);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = base + j * 64 + lane;
        const bool inb = j * 64 < chunk && k < L;
        const bool keep = pred(inb ? k : k0, val[j]);
        bal[j] = __ballot(inb && keep);
        cnt += __popcll(bal[j]);
    }
    lds_barrier();                         // smem may still be read from a previous call
    if (lane == 0) smem[wid] = cnt;
    lds_barrier();
    int pos = 0, total = 0;
    for (int w = 0; w < nw; ++w) { const int c = smem[w]; if (w < wid) pos += c; total += c; }
    const int off = basefn(total);
    pos += off;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if ((bal[j] >> lane) & 1ull) emit(pos + __popcll(bal[j] & below), base + j * 64 + lane, val[j]);
        pos += __popcll(bal[j]);
    }
    lds_barrier();
    return off + total;
}

template <class Load, class Pred, class Emit>
__device__ __forceinline__ int compact_ws(int L, int *smem, Load load, Pred pred, Emit emit) {
    const int nw = blockDim.x >> 6;
    const int chunk = ((L + nw - 1) / nw + 63) & ~63, rounds = chunk >> 6;
    if (L <= 0) return 0;
    auto none = [](int) { return 0; };
    if (rounds <= 1) return compact_ws_r<1>(0, L, chunk, smem, load, pred, emit, none);
    if (rounds <= 4) return compact_ws_r<4>(0, L, chunk, smem, load, pred, emit, none);
    if (rounds <= 16) return compact_ws_r<16>(0, L, chunk, smem, load, pred, emit, none);
    return compact(L, smem, [&](int k) { return pred(k, load(k)); }, [&](int p, int k) { emit(p, k, load(k)); });
}

// Two ordered compactions of the same index range in ONE pass (one fetch of the payload, one round of barriers):
// items with predA go to emitA, items with predB to emitB (an item may be in both or neither).  Returns both totals.
template <int R, class Load, class PredA, class PredB, class EmitA, class EmitB>
__device__ __forceinline__ int2 compact2_ws_r(int L, int chunk, int *smem, Load load, PredA predA, PredB predB, EmitA emitA, EmitB emitB) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int base = wid * chunk;
    int val[R];
    unsigned long long ba[R], bb[R];
    int ca = 0, cb = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = base + j * 64 + lane;
        val[j] = load((j * 64 < chunk && k < L) ? k : 0);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = base + j * 64 + lane;
        const bool inb = j * 64 < chunk && k < L;
        ba[j] = __ballot(inb && predA(inb ? k : 0, val[j]));
        bb[j] = __ballot(inb && predB(inb ? k : 0, val[j]));
        ca += __popcll(ba[j]); cb += __popcll(bb[j]);
    }
    lds_barrier();                         // smem may still be read from a previous call
    if (lane == 0) { smem[wid] = ca; smem[16 + wid] = cb; }
    lds_barrier();
    int pa = 0, pb = 0, ta = 0, tb = 0;
    for (int w = 0; w < nw; ++w) { const int x = smem[w], y = smem[16 + w]; if (w < wid) { pa += x; pb += y; } ta += x; tb += y; }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int k = base + j * 64 + lane;
        if ((ba[j] >> lane) & 1ull) emitA(pa + __popcll(ba[j] & below), k, val[j]);
        if ((bb[j] >> lane) & 1ull) emitB(pb + __popcll(bb[j] & below), k, val[j]);
        pa += __popcll(ba[j]); pb += __popcll(bb[j]);
    }
    lds_barrier();
    return make_int2(ta, tb);
}

template <class Load, class PredA, class PredB, class EmitA, class EmitB>
__device__ __forceinline__ int2 compact2_ws(int L, int *smem, Load load, PredA predA, PredB predB, EmitA emitA, EmitB emitB) {
    const int nw = blockDim.x >> 6;
    const int chunk = ((L + nw - 1) / nw + 63) & ~63, rounds = chunk >> 6;
    if (L <= 0) return make_int2(0, 0);
    if (rounds <= 1) return compact2_ws_r<1>(L, chunk, smem, load, predA, predB, emitA, emitB);
    if (rounds <= 4) return compact2_ws_r<4>(L, chunk, smem, load, predA, predB, emitA, emitB);
    if (rounds <= 16) return compact2_ws_r<16>(L, chunk, smem, load, predA, predB, emitA, emitB);
    // very long inputs: two plain passes
    const int a = compact(L, smem, [&](int k) { return predA(k, load(k)); }, [&](int p, int k) { emitA(p, k, load(k)); });
    const int b = compact(L, smem, [&](int k) { return predB(k, load(k)); }, [&](int p, int k) { emitB(p, k, load(k)); });
    return make_int2(a, b);
}

// ---- one ordered compaction shared by the nch consecutive workgroups of a launch ("chunks") -----------------------
// Chunk ch handles the indices [ch * CL, (ch + 1) * CL) and offsets its output by the counts of the chunks before it,
// which each chunk publishes in agg[] (decoupled look-back: a chunk only ever waits for chunks with a LOWER workgroup
// index, which the dispatcher started first -- the same discipline as k_lap's helper workgroups).  agg[] entries carry
// the step's epoch in their high half, so they are never cleared.  A single CU gathers only ~1 list element per clock:
// chunks are how a 10 000-entry list job gets more than one CU's gather rate.
struct Chunk {
    unsigned long long *agg;       // [nch]
    unsigned epoch;                // != 0, different from the previous uses of agg[]
    int ch, nch;
    int32_t *err;                  // word that gets `err_bit` OR-ed in if a wait times out (never hangs)
    int err_bit;
};
constexpr int CHUNK_MAX = 16;
constexpr unsigned CHUNK_FLAG = 0x80000000u;       // a user bit that travels with the count

// Publishes (count | flag) of this chunk; returns the summed counts of the chunks before it.  want_all: waits for
// EVERY chunk instead and also returns the grand total and the OR of all flags.  smem: >= 40 ints.
__device__ __forceinline__ int chunk_publish(const Chunk &c, int count, bool flag, int *smem, bool want_all = false,
                                             int *grand_total = nullptr, bool *any_flag = nullptr) {
    if (c.nch <= 1) {
        if (grand_total) *grand_total = count;
        if (any_flag) *any_flag = flag;
        return 0;
    }
    lds_barrier();                                   // smem may still be read from a previous call
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(&c.agg[c.ch], ((unsigned long long)c.epoch << 32) | (unsigned)count | (flag ? CHUNK_FLAG : 0u),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned v = 0;
        const bool mine = lane != c.ch && (want_all ? lane < c.nch : lane < c.ch);
        if (mine) {
            unsigned long long a = 0;
            long spins = 0;
            do {
                a = __hip_atomic_load(&c.agg[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(a >> 32) == c.epoch) break;
                __builtin_amdgcn_s_sleep(2);
            } while (++spins < (1l << 22));
            if ((unsigned)(a >> 32) == c.epoch) v = (unsigned)a;
            else atomicOr(c.err, c.err_bit);
        }
        if (lane == c.ch) v = (unsigned)count | (flag ? CHUNK_FLAG : 0u);
        int before = (lane < c.ch) ? (int)(v & ~CHUNK_FLAG) : 0, all = (lane < c.nch) ? (int)(v & ~CHUNK_FLAG) : 0;
        unsigned long long fl = __ballot((v & CHUNK_FLAG) != 0);
#pragma unroll
        for (int off = 1; off < CHUNK_MAX; off <<= 1) { before += __shfl_xor(before, off); all += __shfl_xor(all, off); }
        if (lane == 0) { smem[32] = before; smem[33] = all; smem[34] = fl != 0; }
    }
    lds_barrier();
    if (grand_total) *grand_total = smem[33];
    if (any_flag) *any_flag = smem[34] != 0;
    return smem[32];
}

// the index range of chunk c of an L-long job
__device__ __forceinline__ void chunk_range(const Chunk &c, int L, int &lo, int &hi) {
    const int cl = ((L + c.nch - 1) / c.nch + 63) & ~63;
    lo = min(L, c.ch * cl); hi = min(L, lo + cl);
}

// compact_ws over this chunk's share of [0, L): returns the number of items emitted by chunks 0 .. ch (the last chunk's
// return value is the total).  `flag` / want_all / grand_total / any_flag: see chunk_publish.
template <class Load, class Pred, class Emit>
__device__ __forceinline__ int compact_chunk(const Chunk &c, int L, int *smem, Load load, Pred pred, Emit emit, bool flag = false,
                                             bool want_all = false, int *grand_total = nullptr, bool *any_flag = nullptr) {
    int lo, hi;
    chunk_range(c, L, lo, hi);
    const int nw = blockDim.x >> 6, len = hi - lo;
    const int chunk = ((len + nw - 1) / nw + 63) & ~63, rounds = chunk >> 6;
    auto basefn = [&](int total) { return chunk_publish(c, total, flag, smem, want_all, grand_total, any_flag); };
    if (len <= 0) return basefn(0);                     // an empty chunk touches no list entry, but still publishes
    if (rounds <= 1) return compact_ws_r<1>(lo, hi, chunk, smem, load, pred, emit, basefn);
    if (rounds <= 4) return compact_ws_r<4>(lo, hi, chunk, smem, load, pred, emit, basefn);
    if (rounds <= 16) return compact_ws_r<16>(lo, hi, chunk, smem, load, pred, emit, basefn);
    // very long chunks: chunk-per-thread form on the sub-range
    int cnt = 0;
    const int tid = threadIdx.x, nth = blockDim.x, per = (len + nth - 1) / nth;
    const int a = min(hi, lo + tid * per), b = min(hi, a + per);
    for (int k = a; k < b; ++k) cnt += pred(k, load(k)) ? 1 : 0;
    int total;
    int pos = exscan(cnt, smem, &total);
    const int off = basefn(total);
    pos += off;
    for (int k = a; k < b; ++k) { const int v = load(k); if (pred(k, v)) emit(pos++, k, v); }
    lds_barrier();
    return off + total;
}

// for (k = tid; k < L; k += nth) body(k, load(k)) with the loads of B iterations in flight together
template <int B, class Load, class Body>
__device__ __forceinline__ void for_batched(int L, Load load, Body body) {
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int k0 = tid; k0 < L; k0 += B * nth) {
        int v[B];
#pragma unroll
        for (int j = 0; j < B; ++j) { const int k = k0 + j * nth; v[j] = load(k < L ? k : 0); }
#pragma unroll
        for (int j = 0; j < B; ++j) { const int k = k0 + j * nth; if (k < L) body(k, v[j]); }
    }
}

}  // namespace bscan
