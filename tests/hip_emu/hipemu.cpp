// tests/hip_emu/hipemu.cpp -- TEST INFRASTRUCTURE: the scheduler of the CPU stand-in for HIP (hip/hip_runtime.h next to it).
// One workgroup at a time; its threads are ucontext fibres run round-robin until each blocks at a rendezvous:
//   * __syncthreads: released when every thread of the workgroup that has not returned waits at it;
//   * a wavefront collective: lanes of one wavefront waiting at the SAME call site form a group; when every live lane of the
//     wavefront is blocked (at a collective or at the workgroup barrier), the group with the lowest call-site id is served --
//     its members are the "active lanes" of that instruction, exactly like an exec mask; diverged lanes wait their turn.
#include "hip/hip_runtime.h"
#include <ucontext.h>
#include <sys/mman.h>
#include <algorithm>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace emu {

enum State { READY, AT_BARRIER, AT_WAVE, DONE };
struct Thread {
    ucontext_t ctx;
    int tid = 0;
    State st = READY;
    Op op = OP_SHFL;
    int site = 0, arg = 0, arg2 = 0;
    uint64_t value = 0, old = 0, result = 0;
    int bar_val = 0;
    const void *blob = nullptr;      // OP_GATHER: this lane's deposit
};

static constexpr size_t STACK = 256 * 1024;
static char *g_stacks = nullptr;
static size_t g_nstacks = 0;
static std::vector<Thread> g_th;
static ucontext_t g_sched;
static Thread *g_cur = nullptr;
static dim3 g_bid, g_bdim, g_gdim;
static const std::function<void()> *g_body = nullptr;
alignas(64) static char g_dyn_lds[160 * 1024];
static int g_bar_or = 0;
static int g_spins = 0;

Thread *cur() { return g_cur; }
unsigned tid_x() { return (unsigned)g_cur->tid; }
unsigned bid(int a) { return a == 0 ? g_bid.x : a == 1 ? g_bid.y : g_bid.z; }
unsigned bdim(int a) { return a == 0 ? g_bdim.x : a == 1 ? g_bdim.y : g_bdim.z; }
unsigned gdim(int a) { return a == 0 ? g_gdim.x : a == 1 ? g_gdim.y : g_gdim.z; }
char *dynamic_lds() { return g_dyn_lds; }

static void yield_to_scheduler() { swapcontext(&g_cur->ctx, &g_sched); }

void barrier() {
    g_cur->st = AT_BARRIER; g_cur->bar_val = 0;
    yield_to_scheduler();
}
int barrier_or(int v) {
    g_cur->st = AT_BARRIER; g_cur->bar_val = v;
    yield_to_scheduler();
    return g_bar_or;
}
uint64_t collective(Op op, int site, uint64_t value, int arg, int arg2, uint64_t old_value) {
    Thread *t = g_cur;
    t->op = op; t->site = site; t->value = value; t->arg = arg; t->arg2 = arg2; t->old = old_value; t->st = AT_WAVE;
    yield_to_scheduler();
    return t->result;
}

alignas(64) static char g_gather[16][64][GATHER_MAX];      // [wavefront][lane][bytes]
const char *gather(int site, const void *mine, int n) {
    Thread *t = g_cur;
    if (n > GATHER_MAX || t->tid >= 16 * 64) { fprintf(stderr, "hipemu: gather of %d bytes / thread %d\n", n, t->tid); abort(); }
    t->op = OP_GATHER; t->site = site; t->blob = mine; t->arg = n; t->st = AT_WAVE;
    yield_to_scheduler();
    return &g_gather[t->tid >> 6][0][0];
}

// A spin on something another thread has to write.  The spinner first lets the other threads of its workgroup run (the value
// may come from one of them: lane 0 publishes, its neighbours wait); only when a whole pass over the workgroup was nothing
// but spinners -- the value has to come from another WORKGROUP -- is the workgroup dropped, to be re-run later.
static bool g_spun = false;          // the thread that just came back to the scheduler did so from spin()
void spin() {
    g_spun = true;
    swapcontext(&g_cur->ctx, &g_sched);          // (stays READY; if the workgroup is dropped it is never resumed)
}

// ---- LDS-DMA (global_load_lds) ------------------------------------------------------------------------------------------------
// One in-flight list per workgroup.  issue: fetch now, poison the destination; wait(keep): the calling thread's transfers land in
// issue order except its newest `keep`; workgroup barrier: landed transfers retire (visible to everybody); check: see the header.
struct Dma { char *dst; int n, tid; bool landed; char data[16]; };
static std::vector<Dma> g_dma;
static std::multimap<uintptr_t, int> g_dma_slot;               // 16-byte LDS slot -> index into g_dma
static int dma_mode() {                                        // 0 asynchronous (default), 1 sync, 2 dropwaits
    static const int m = [] { const char *e = getenv("HIPEMU_LDS_DMA"); return !e ? 0 : !strcmp(e, "sync") ? 1 : !strcmp(e, "dropwaits") ? 2 : 0; }();
    return m;
}
static int dma_late() {                                        // HIPEMU_LDS_DMA=late:N -- every wait counts N too many (vmcnt(k + N)): fault injection
    static const int n = [] { const char *e = getenv("HIPEMU_LDS_DMA"); return e && !strncmp(e, "late:", 5) ? atoi(e + 5) : 0; }();
    return n;
}
static void dma_index(int i) {
    const uintptr_t a = (uintptr_t)g_dma[i].dst;
    for (uintptr_t sl = a >> 4; sl <= (a + g_dma[i].n - 1) >> 4; ++sl) g_dma_slot.insert({sl, i});
}
void lds_dma_issue(const void *g, void *dst, int n) {
    if (n <= 0 || n > 16) { fprintf(stderr, "hipemu: global_load_lds of %d bytes per lane\n", n); abort(); }
    if (dma_mode() == 1) { memcpy(dst, g, (size_t)n); return; }
    lds_dma_check(dst, n, "global_load_lds into");              // a transfer into bytes another transfer still owns
    Dma d;
    d.dst = (char *)dst; d.n = n; d.tid = g_cur->tid; d.landed = false;
    memcpy(d.data, g, (size_t)n);
    memset(dst, 0x7f, (size_t)n);                               // (float 3.4e38 / int8 127: a consumer that slips through is wrong loudly)
    g_dma.push_back(d);
    dma_index((int)g_dma.size() - 1);
}
void lds_dma_wait(int keep) {
    if (dma_mode() == 2 || g_dma.empty()) return;
    int mine = 0;
    for (const Dma &d : g_dma) if (d.tid == g_cur->tid && !d.landed) ++mine;
    int land = mine - keep - dma_late();
    for (Dma &d : g_dma) {
        if (land <= 0) break;
        if (d.tid == g_cur->tid && !d.landed) { memcpy(d.dst, d.data, (size_t)d.n); d.landed = true; --land; }
    }
}
void lds_dma_check(const void *p, int n, const char *what) {
    if (g_dma_slot.empty()) return;
    const uintptr_t a = (uintptr_t)p;
    for (uintptr_t sl = a >> 4; sl <= (a + n - 1) >> 4; ++sl) {
        auto r = g_dma_slot.equal_range(sl);
        for (auto it = r.first; it != r.second; ++it) {
            const Dma &d = g_dma[it->second];
            if ((uintptr_t)d.dst >= a + n || (uintptr_t)d.dst + d.n <= a) continue;
            if (d.tid == g_cur->tid && d.landed) continue;
            if (!d.landed || (d.tid >> 6) != (g_cur->tid >> 6)) {
                fprintf(stderr, "hipemu: LDS-DMA hazard in block (%u,%u,%u): thread %d: %s %d bytes at %p, which thread %d's global_load_lds %s\n",
                        g_bid.x, g_bid.y, g_bid.z, g_cur->tid, what, n, p, d.tid,
                        d.landed ? "landed in ANOTHER wavefront with no workgroup barrier since" : "has in flight (no s_waitcnt vmcnt of the issuing wavefront yet)");
                abort();
            }
        }
    }
}
static void dma_retire_landed() {                               // at a workgroup barrier's release
    if (g_dma.empty()) return;
    std::vector<Dma> keep;
    for (const Dma &d : g_dma) if (!d.landed) keep.push_back(d);
    g_dma.swap(keep);
    g_dma_slot.clear();
    for (int i = 0; i < (int)g_dma.size(); ++i) dma_index(i);
}

static void trampoline() {
    (*g_body)();
    g_cur->st = DONE;
    swapcontext(&g_cur->ctx, &g_sched);
}

static int dpp_source(int lane, int ctrl, bool &valid) {       // source lane of a DPP control word, for `lane`
    valid = true;
    const int row = lane & ~15, r = lane & 15;
    if (ctrl >= 0x00 && ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);            // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; valid = r + n <= 15; return lane + n; }   // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; valid = r >= n; return lane - n; }        // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row | ((r - n) & 15); }            // row_ror
    if (ctrl == 0x140) return row | (15 - r);                                                                  // row_mirror
    if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));                                                  // row_half_mirror
    if (ctrl == 0x142) { valid = lane >= 16; return (row - 16) | 15; }                                         // row_bcast:15
    if (ctrl == 0x143) { valid = lane >= 32; return 31; }                                                      // row_bcast:31
    fprintf(stderr, "hipemu: DPP control 0x%x not emulated\n", ctrl);
    abort();
}

static void serve_group(int wave_base, int nlanes, const std::vector<int> &members) {
    Thread *th = &g_th[wave_base];
    bool in[64] = {false};
    for (int l : members) in[l] = true;
    const Thread &first = th[members[0]];
    const Op op = first.op;
    uint64_t ballot = 0;
    if (op == OP_BALLOT) for (int l : members) if (th[l].value) ballot |= 1ull << l;
    for (int l : members) {
        Thread &t = th[l];
        switch (op) {
        case OP_SHFL: { const int w = t.arg2, s = (l & ~(w - 1)) | (t.arg & (w - 1)); t.result = (s < nlanes && in[s]) ? th[s].value : t.value; break; }
        case OP_SHFL_XOR: { const int s = l ^ t.arg; t.result = (s < nlanes && in[s] && (s & ~(t.arg2 - 1)) == (l & ~(t.arg2 - 1))) ? th[s].value : t.value; break; }
        case OP_SHFL_UP: { const int w = t.arg2, s = l - t.arg; t.result = (s >= (l & ~(w - 1)) && in[s]) ? th[s].value : t.value; break; }
        case OP_BALLOT: t.result = ballot; break;
        case OP_READLANE: { const int s = t.arg & 63; t.result = (s < nlanes && in[s]) ? th[s].value : 0; break; }
        case OP_READFIRST: t.result = first.value; break;
        case OP_DPP: {
            bool valid;
            const int s = dpp_source(l, t.arg, valid);
            const int row_mask = t.arg2 & 0xf;
            const bool bound = (t.arg2 & 0x100) != 0;
            if (!((row_mask >> (l >> 4)) & 1)) t.result = t.old;
            else if (!valid || s < 0 || s >= 64 || !(s < nlanes && in[s])) t.result = bound ? 0 : t.old;
            else t.result = th[s].value;
            break;
        }
        case OP_WAVE_BARRIER: t.result = 0; break;
        case OP_GATHER: break;
        }
    }
    if (op == OP_GATHER) {               // deposits are copied while every member is still parked at the call site
        char (*stage)[GATHER_MAX] = g_gather[wave_base >> 6];
        memset(stage, 0, sizeof(g_gather[0]));
        for (int l : members) memcpy(stage[l], th[l].blob, (size_t)th[l].arg);
    }
    for (int l : members) th[l].st = READY;
}

static bool run_block(int nthreads) {        // false: the workgroup was deferred (it waits for workgroups that have not run yet)
    g_spins = 0;
    if ((size_t)nthreads > g_nstacks) {
        if (g_stacks) munmap(g_stacks, g_nstacks * STACK);
        g_nstacks = (size_t)nthreads;
        g_stacks = (char *)mmap(nullptr, g_nstacks * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char *)MAP_FAILED) { perror("hipemu: mmap"); abort(); }
    }
    g_dma.clear(); g_dma_slot.clear();      // (transfers still in flight when a workgroup ends are dropped, as on the hardware)
    g_th.assign((size_t)nthreads, Thread());
    for (int t = 0; t < nthreads; ++t) {
        Thread &T = g_th[t];
        T.tid = t; T.st = READY;
        getcontext(&T.ctx);
        T.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * STACK;
        T.ctx.uc_stack.ss_size = STACK;
        T.ctx.uc_link = &g_sched;
        makecontext(&T.ctx, trampoline, 0);
    }
    const int nwaves = (nthreads + 63) / 64;
    // HIPEMU_THREADS=reverse: the threads of a workgroup take their turns in descending order -- a hand-off between
    // wavefronts that lacks its barrier works in one order and breaks in the other
    static const bool rev = getenv("HIPEMU_THREADS") && !strcmp(getenv("HIPEMU_THREADS"), "reverse");
    for (;;) {
        bool progress = false, all_done = true, moved = false, spinners = false;
        for (int tt = 0; tt < nthreads; ++tt) {
            const int t = rev ? nthreads - 1 - tt : tt;
            if (g_th[t].st == READY) {
                g_cur = &g_th[t];
                g_spun = false;
                swapcontext(&g_sched, &g_th[t].ctx);
                progress = true;
                if (g_spun) spinners = true; else moved = true;
            }
            if (g_th[t].st != DONE) all_done = false;
        }
        if (all_done) return true;
        if (spinners && !moved && ++g_spins >= 4) return false;      // only spinners are left runnable: the wait is for another workgroup
        if (moved) g_spins = 0;
        // wavefront collectives
        bool served = false;
        for (int w = 0; w < nwaves; ++w) {
            const int base = w * 64, nl = std::min(64, nthreads - base);
            int best = 0x7fffffff;
            bool blocked = true, any = false;
            for (int l = 0; l < nl; ++l) {
                const Thread &T = g_th[base + l];
                if (T.st == READY) blocked = false;
                if (T.st == AT_WAVE) { any = true; best = std::min(best, T.site); }
            }
            if (!blocked || !any) continue;
            std::vector<int> members;
            for (int l = 0; l < nl; ++l) if (g_th[base + l].st == AT_WAVE && g_th[base + l].site == best) members.push_back(l);
            serve_group(base, nl, members);
            served = true;
        }
        if (served) continue;
        // workgroup barrier
        bool all_bar = true;
        int orv = 0;
        for (int t = 0; t < nthreads; ++t) {
            if (g_th[t].st == DONE) continue;
            if (g_th[t].st != AT_BARRIER) all_bar = false;
            orv |= g_th[t].bar_val;
        }
        if (all_bar) {
            g_bar_or = orv != 0;
            dma_retire_landed();
            for (int t = 0; t < nthreads; ++t) if (g_th[t].st == AT_BARRIER) g_th[t].st = READY;
            continue;
        }
        if (!progress) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): ", g_bid.x, g_bid.y, g_bid.z);
            for (int t = 0; t < nthreads && t < 128; ++t) fprintf(stderr, "%d", (int)g_th[t].st);
            fprintf(stderr, "\n");
            abort();
        }
    }
}

static std::mutex g_launch_mu;      // host threads that drive several streams take turns: one launch at a time

void launch(dim3 grid, dim3 block, size_t dyn_lds, const std::function<void()> &body) {
    std::lock_guard<std::mutex> lock(g_launch_mu);
    if (g_cur) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
    if (dyn_lds > sizeof(g_dyn_lds)) { fprintf(stderr, "hipemu: %zu bytes of dynamic LDS\n", dyn_lds); abort(); }
    // (the hardware refuses an empty grid or block -- hipErrorInvalidConfiguration -- where a loop over zero workgroups would
    //  silently do nothing: a launch whose size comes out of a frame's counts must be guarded by the caller)
    if ((size_t)grid.x * grid.y * grid.z == 0 || (size_t)block.x * block.y * block.z == 0 || (size_t)block.x * block.y * block.z > 1024) {
        fprintf(stderr, "hipemu: invalid launch configuration: grid (%u,%u,%u) block (%u,%u,%u)\n", grid.x, grid.y, grid.z, block.x, block.y, block.z);
        abort();
    }
    g_body = &body;
    g_gdim = grid; g_bdim = block;
    const int nthreads = (int)(block.x * block.y * block.z);
    std::vector<dim3> todo, later;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) todo.push_back(dim3(x, y, z));
    // HIPEMU_ORDER=reverse | shuffle[:seed]: the hardware promises no execution order between workgroups -- results that
    // depend on it (the order entries are appended through atomic counters, who finishes last) must not change
    static const char *order = getenv("HIPEMU_ORDER");
    if (order && !strncmp(order, "reverse", 7)) std::reverse(todo.begin(), todo.end());
    else if (order && !strncmp(order, "shuffle", 7)) {
        static uint64_t rng = order[7] == ':' ? strtoull(order + 8, nullptr, 10) * 2654435761u + 1 : 12345;
        for (size_t i = todo.size(); i > 1; --i) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(todo[i - 1], todo[(size_t)((rng >> 33) % i)]);
        }
    }
    while (!todo.empty()) {
        later.clear();
        for (const dim3 &b : todo) {
            g_bid = b;
            if (!run_block(nthreads)) later.push_back(b);
        }
        if (later.size() == todo.size()) { fprintf(stderr, "hipemu: %zu workgroups wait for each other forever\n", later.size()); abort(); }
        todo = later;
    }
    g_cur = nullptr;
    g_body = nullptr;
}

EmuGraph *&capturing() { static EmuGraph *g = nullptr; return g; }
void launch_or_record(dim3 grid, dim3 block, size_t dyn_lds, hipStream_t stream, std::function<void()> body) {
    if (capturing()) { capturing()->nodes.push_back([=]() { launch(grid, block, dyn_lds, body); }); return; }
    stream_op(stream, [=]() { launch(grid, block, dyn_lds, body); });
}

// ---- streams and events: see hip/hip_runtime.h (HIPEMU_STREAMS) ---------------------------------------------------------------
namespace {
enum { SOP_RUN, SOP_RECORD, SOP_WAIT };
struct SOp { int kind; std::function<void()> f; EmuEvent *ev; unsigned long long seq; EmuStream *from; };
struct StreamState { std::deque<SOp> q; };
std::map<EmuStream *, StreamState> g_streams;          // key nullptr = the null stream
std::recursive_mutex g_stream_mu;
int g_forced_ops = 0, g_queued_ops = 0, g_cross_forces = 0;
int stream_mode() {                                    // 0 eager (default), 1 sidelate, 2 mainlate
    static const int m = [] { const char *e = getenv("HIPEMU_STREAMS"); return !e ? 0 : !strcmp(e, "sidelate") ? 1 : !strcmp(e, "mainlate") ? 2 : 0; }();
    return m;
}
bool lazy(EmuStream *s) { const int m = stream_mode(); return m == 1 ? s != nullptr : m == 2 ? s == nullptr : false; }
constexpr size_t QUEUE_LIMIT = 48;                     // operations a lagging stream may hold (~4 steps of the tracker)

// run the operations queued on s, oldest first, up to and including the record (ev, seq) -- everything if ev is null.  A wait
// whose record is still queued on another stream forces that stream first (a cycle would be a deadlock of the program itself).
void force(EmuStream *s, EmuEvent *ev, unsigned long long seq, int depth) {
    if (depth > 16) { fprintf(stderr, "hipemu: streams wait for each other in a cycle\n"); abort(); }
    std::deque<SOp> &q = g_streams[s].q;
    while (!q.empty()) {
        SOp op = std::move(q.front());
        q.pop_front();
        ++g_forced_ops;
        if (op.kind == SOP_RUN) op.f();
        else if (op.kind == SOP_RECORD) {
            if (op.ev->done < op.seq) op.ev->done = op.seq;
            if (ev && op.ev == ev && op.seq >= seq) return;
        } else if (op.ev->done < op.seq) { ++g_cross_forces; force(op.from, op.ev, op.seq, depth + 1); }
    }
}
}  // namespace

void stream_op(hipStream_t s, std::function<void()> f) {
    if (stream_mode() == 0) { f(); return; }
    std::lock_guard<std::recursive_mutex> lock(g_stream_mu);
    if (!lazy(s)) { f(); return; }                     // (an eager stream never holds anything back: its queue is empty)
    std::deque<SOp> &q = g_streams[s].q;
    q.push_back(SOp{SOP_RUN, std::move(f), nullptr, 0, nullptr});
    ++g_queued_ops;
    while (q.size() > QUEUE_LIMIT) {                   // running the OLDEST operations early is still a legal schedule
        SOp op = std::move(q.front());
        q.pop_front();
        if (op.kind == SOP_RUN) op.f();
        else if (op.kind == SOP_RECORD) { if (op.ev->done < op.seq) op.ev->done = op.seq; }
        else if (op.ev->done < op.seq) force(op.from, op.ev, op.seq, 1);
    }
}
void stream_record(hipStream_t s, hipEvent_t e) {
    if (!e) return;
    if (stream_mode() == 0) { e->owner = s; e->done = ++e->enq; return; }
    std::lock_guard<std::recursive_mutex> lock(g_stream_mu);
    e->owner = s;
    const unsigned long long seq = ++e->enq;
    if (lazy(s)) g_streams[s].q.push_back(SOp{SOP_RECORD, nullptr, e, seq, s}); else e->done = seq;
}
void stream_wait(hipStream_t s, hipEvent_t e) {
    if (!e || stream_mode() == 0) return;
    // HIPEMU_DROP_WAITS=main | side: FAULT INJECTION for the schedule tests -- the waits issued on the null stream / on created
    // streams are ignored, i.e. the program "forgot" those dependencies; the adversarial schedules must then change a result
    // (tests/test_emu_gpu_subset.py: the detector detects)
    static const char *drop = getenv("HIPEMU_DROP_WAITS");
    if (drop && ((!strcmp(drop, "main") && s == nullptr) || (!strcmp(drop, "side") && s != nullptr))) return;
    std::lock_guard<std::recursive_mutex> lock(g_stream_mu);
    const unsigned long long seq = e->enq;             // (HIP: the wait is for the latest record enqueued at this point)
    if (seq == 0 || e->done >= seq || e->owner == s) return;
    if (lazy(s)) g_streams[s].q.push_back(SOp{SOP_WAIT, nullptr, e, seq, e->owner});
    else { ++g_cross_forces; force(e->owner, e, seq, 0); }
}
void stream_sync(hipStream_t s) {
    if (stream_mode() == 0) return;
    std::lock_guard<std::recursive_mutex> lock(g_stream_mu);
    force(s, nullptr, 0, 0);
    drain_all();       // the lag is bounded: once the host has waited for s, the other streams catch up (AFTER s: late, not never --
                       // a stream that never ran would leave the zero-initialised shadows in place, which the design tolerates)
}
void event_sync(hipEvent_t e) {
    if (!e || stream_mode() == 0) return;
    std::lock_guard<std::recursive_mutex> lock(g_stream_mu);
    if (e->done < e->enq) force(e->owner, e, e->enq, 0);
    drain_all();
}
void drain_all() {
    if (stream_mode() == 0) return;
    std::lock_guard<std::recursive_mutex> lock(g_stream_mu);
    for (bool again = true; again;) {
        again = false;
        for (auto &kv : g_streams) if (!kv.second.q.empty()) { force(kv.first, nullptr, 0, 0); again = true; }
    }
}
hipStream_t stream_create() { return (hipStream_t)malloc(8); }
void stream_destroy(hipStream_t s) {
    if (stream_mode() != 0) { std::lock_guard<std::recursive_mutex> lock(g_stream_mu); force(s, nullptr, 0, 0); g_streams.erase(s); }
    free(s);
}

}  // namespace emu

// for the Python prelude (tests/hip_emu/emu_prelude.py): torch.cuda.synchronize() / Tensor.cpu() are "device idle" points too
extern "C" void hipemu_drain_all() { emu::drain_all(); }
extern "C" int hipemu_stream_mode() { return emu::stream_mode(); }
// [0] operations that were queued on a lagging stream, [1] operations run because something forced them, [2] forces that crossed
// streams (a wait on a record still queued elsewhere): a test asserts that the lagging really happened
extern "C" void hipemu_stream_stats(int *out3) { out3[0] = emu::g_queued_ops; out3[1] = emu::g_forced_ops; out3[2] = emu::g_cross_forces; }

