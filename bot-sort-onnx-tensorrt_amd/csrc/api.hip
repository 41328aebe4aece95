// api.hip -- library-level entry points: version, error reporting, contexts, scratch arena.
#include <stdarg.h>
#include <strings.h>

#include "bts_internal.h"

static thread_local char g_err[512] = "";

void bts_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- execution forms: "KEY=VALUE,KEY=VALUE" handed to bts_tracker_create_forms, consulted before the process environment ----
static thread_local const char *g_forms = nullptr;
static thread_local char g_form_val[64];

bts_forms_scope::bts_forms_scope(const char *forms) : prev(g_forms) { g_forms = forms; }
bts_forms_scope::~bts_forms_scope() { g_forms = prev; }

const char *bts_form_get(const char *key) {
    const size_t kl = strlen(key);
    for (const char *p = g_forms; p && *p;) {
        while (*p == ',' || *p == ' ' || *p == ';') ++p;
        const char *e = p;
        while (*e && *e != ',' && *e != ';') ++e;
        if ((size_t)(e - p) >= kl && !strncmp(p, key, kl) && (p[kl] == '=' || p + kl == e)) {
            // KEY=VALUE -> VALUE; a bare KEY counts as "1"; KEY= (empty value) switches a form OFF whatever the environment says
            const char *v = p[kl] == '=' ? p + kl + 1 : "1";
            size_t n = p[kl] == '=' ? (size_t)(e - v) : 1;
            if (n == 0) return nullptr;
            if (n >= sizeof(g_form_val)) n = sizeof(g_form_val) - 1;
            memcpy(g_form_val, v, n);
            g_form_val[n] = 0;
            return g_form_val;
        }
        p = e;
    }
    return getenv(key);
}

// Boolean forms: on for any value except "", "0", "false", "off", "no" (any case) -- so KEY=0 from a caller's dict or an A/B
// script's environment switches a form OFF like KEY= does (`!= nullptr` turned "KEY=0" ON).
bool bts_form_flag(const char *key) {
    const char *v = bts_form_get(key);
    if (!v) return false;
    while (*v == ' ') ++v;
    if (!*v || !strcmp(v, "0") || !strcasecmp(v, "false") || !strcasecmp(v, "off") || !strcasecmp(v, "no")) return false;
    return true;
}

int bts_arena::ensure(size_t bytes) {
    if (bytes <= cap) return BTS_OK;
    // growing replaces the block: everything enqueued on the old one (any stream of this device) must have finished
    if (base) { (void)hipDeviceSynchronize(); (void)hipFree(base); base = nullptr; cap = 0; }
    const size_t want = bts_align(bytes + bytes / 4, 1 << 20);
    if (hipMalloc((void **)&base, want) != hipSuccess) {
        (void)hipGetLastError();
        bts_set_error("device allocation of %zu bytes failed", want);
        base = nullptr;
        return BTS_ENOMEM;
    }
    cap = want;
    return BTS_OK;
}

void bts_arena::release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = 0;
}

extern "C" {

const char *bts_version(void) { return "botsort_hip 0.1.0 (gfx950)"; }
const char *bts_last_error(void) { return g_err; }

const char *bts_status_string(int status) {
    switch (status) {
        case BTS_OK: return "ok";
        case BTS_EINVAL: return "invalid argument";
        case BTS_EHIP: return "HIP runtime error";
        case BTS_ENOMEM: return "out of memory";
        case BTS_ECAPACITY: return "tracker capacity exceeded";
        case BTS_ENODEVICE: return "no usable gfx950 device";
        default: return "unknown status";
    }
}

int bts_context_create(bts_context **out, int device_ordinal) {
    BTS_REQUIRE(out, "out");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        bts_set_error("no HIP device visible");
        return BTS_ENODEVICE;
    }
    BTS_REQUIRE(device_ordinal >= 0 && device_ordinal < count, "device ordinal out of range");
    hipDeviceProp_t prop;
    BTS_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        bts_set_error("device %d is %s; this library carries gfx950 code only", device_ordinal, prop.gcnArchName);
        return BTS_ENODEVICE;
    }
    // (the caller's current device is left alone: every entry point selects ctx->device itself, BTS_ON_DEVICE)
    bts_context *ctx = new (std::nothrow) bts_context();
    if (!ctx) { bts_set_error("host allocation failed"); return BTS_ENOMEM; }
    ctx->device = device_ordinal;
    ctx->n_cu = prop.multiProcessorCount;
    *out = ctx;
    return BTS_OK;
}

int bts_context_destroy(bts_context *ctx) {
    if (!ctx) return BTS_OK;
    BTS_ON_DEVICE(ctx);
    ctx->scratch.release();
    delete ctx;
    return BTS_OK;
}

int bts_pinned_alloc(void **out, size_t bytes) {
    BTS_REQUIRE(out && bytes > 0, "out, bytes");
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        bts_set_error("pinned host allocation of %zu bytes failed", bytes);
        return BTS_ENOMEM;
    }
    return BTS_OK;
}

int bts_pinned_free(void *p) {
    if (p) BTS_HIP(hipHostFree(p));
    return BTS_OK;
}

int bts_context_set_cos_kernel(bts_context *ctx, int which) {
    BTS_REQUIRE(ctx, "ctx");
    BTS_REQUIRE(which == BTS_COS_MFMA_F32 || which == BTS_COS_VALU_F32, "unknown cosine kernel");
    ctx->cos_kernel = which;
    return BTS_OK;
}

}  // extern "C"
