// ctx.hip — context, error reporting and the plain (host-pointer) match_desc.
#include "common.h"

#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void viso_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* viso_last_error(void) { return g_err; }
extern "C" const char* viso_version(void) { return "libviso_hip 0.1 (gfx950, HIP, wave64)"; }

// ---- handle registry ---------------------------------------------------------------------------------------------
// "We never abort across the ABI" has to hold for a caller that gets the teardown order wrong, too: a batch follows its
// context pointer in every call, so a batch that outlives its context used to hand a dead stream to the HIP runtime
// (std::bad_variant_access inside hipStreamSynchronize: an abort).  The library therefore knows its live handles: a context
// keeps the list of the batches created on it; viso_ctx_destroy frees those that are still alive FIRST and leaves a
// tombstone per batch, so that the caller's later viso_batch_destroy is a no-op (VISO_OK) and any other call on such a handle
// -- or on a handle that never existed, or was destroyed twice -- returns VISO_ERR_ARG.  A tombstoned batch keeps its small
// host object (everything it owned is freed) until the caller's own destroy deletes it: its address cannot be handed to
// another batch meanwhile, so the late destroy can never hit somebody else's handle.
static std::mutex g_reg_mu;
static std::unordered_map<const viso_ctx*, std::vector<viso_batch*>> g_ctx_batches;   // the live contexts and their live batches
static std::unordered_map<const viso_batch*, viso_ctx*> g_batch_ctx;                  // the live batches
static std::unordered_set<const viso_batch*> g_batch_tomb;                            // freed by their context, not yet by the caller

bool viso_ctx_live(const viso_ctx* c) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return g_ctx_batches.count(c) != 0;
}
bool viso_batch_live(const viso_batch* b) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return g_batch_ctx.count(b) != 0;
}
bool viso_batch_register(viso_ctx* c, viso_batch* b) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_ctx_batches.find(c);
    if (it == g_ctx_batches.end()) return false;
    g_batch_tomb.erase(b);   // the address of a batch that died with its context, handed out again
    it->second.push_back(b);
    g_batch_ctx[b] = c;
    return true;
}
// 1 = was live, now the caller's to free; 0 = its context freed it already (tombstone consumed); -1 = unknown handle
int viso_batch_unregister(viso_batch* b) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_batch_ctx.find(b);
    if (it == g_batch_ctx.end()) return g_batch_tomb.erase(b) ? 0 : -1;
    auto& v = g_ctx_batches[it->second];
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == b) { v[i] = v.back(); v.pop_back(); break; }
    g_batch_ctx.erase(it);
    return 1;
}

extern "C" viso_ctx* viso_ctx_create(int device, void* stream) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        viso_set_error("no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) { viso_set_error("device %d out of range [0,%d)", device, ndev); return nullptr; }
    if ((e = hipSetDevice(device)) != hipSuccess) { viso_set_error("hipSetDevice: %s", hipGetErrorString(e)); return nullptr; }
    viso_ctx* c = new viso_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->matcher_variant = viso_matcher_default();
    c->row8_force = -1;
    if (const char* e = getenv("VISO_ROW8_SHIFT")) {   // test / A-B aid: a fixed shift of the 8-bit planes for every new context
        const int v = atoi(e);
        if (*e && v >= 0 && v <= 3) c->row8_force = v;
    }
    c->gn_split = 0;   // 0 = the build's default (VISO_GN_SPLIT, solver.hip)
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            viso_set_error("hipStreamCreate: %s", hipGetErrorString(e));
            delete c;
            return nullptr;
        }
        c->own_stream = true;
    }
    // the RANSAC stream has the matcher stream's priority: a high-priority one made the step slower (its 256-register
    // waves push matcher waves aside the moment they are ready; measured 351 k against 360 k frames/s end to end)
    if ((e = hipStreamCreateWithFlags(&c->solver_stream, hipStreamNonBlocking)) != hipSuccess) {
        viso_set_error("hipStreamCreate (solver stream): %s", hipGetErrorString(e));
        if (c->own_stream) hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    { std::lock_guard<std::mutex> lk(g_reg_mu); g_ctx_batches[c]; }
    return c;
}

// Frees everything it can and reports the FIRST HIP error it met (viso_last_error).  Must not be called once the
// HIP runtime is being unloaded (static destructors / atexit handlers that run after it): destroy contexts before
// the process starts exiting (the Python wrapper does so from an atexit hook of its own).
// Batches of the context that are still alive are destroyed first (see the registry above); a handle that is not a live
// context -- destroyed before, or never created -- is VISO_ERR_ARG.
extern "C" int viso_ctx_destroy(viso_ctx* c) {
    if (!c) return VISO_OK;
    std::vector<viso_batch*> orphans;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_ctx_batches.find(c);
        if (it == g_ctx_batches.end()) { viso_set_error("viso_ctx_destroy: not a live context handle"); return VISO_ERR_ARG; }
        orphans.swap(it->second);
        g_ctx_batches.erase(it);
        for (viso_batch* b : orphans) { g_batch_ctx.erase(b); g_batch_tomb.insert(b); }
    }
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    bool batch_err = false;
    for (viso_batch* b : orphans) if (viso_batch_free(b, true) < 0) batch_err = true;   // while the context's streams still exist; their shells stay
    note(hipSetDevice(c->device));
    note(hipStreamSynchronize(c->stream));
    if (c->solver_stream) { note(hipStreamSynchronize(c->solver_stream)); note(hipStreamDestroy(c->solver_stream)); }
    plain_cache_free(c);
    for (int i = 0; i < 24; ++i) if (c->scratch[i]) note(hipFree(c->scratch[i]));
    for (int i = 0; i < 2; ++i) if (c->pin[i]) note(hipHostFree(c->pin[i]));
    if (c->sig_flag) note(hipHostFree(c->sig_flag));
    if (c->sig_ctr) note(hipFree(c->sig_ctr));
    if (c->own_stream) note(hipStreamDestroy(c->stream));
    delete c;
    if (first != hipSuccess) { viso_set_error("viso_ctx_destroy: %s", hipGetErrorString(first)); return VISO_ERR_HIP; }
    if (batch_err) return VISO_ERR_HIP;   // viso_last_error() holds the batch's own message
    return VISO_OK;
}

static viso_ctx* ctx_or_default(viso_ctx* c) {
    if (!c) return viso_default_ctx();
    if (!viso_ctx_live(c)) { viso_set_error("not a live context handle"); return nullptr; }
    return c;
}

static bool matcher_known(int variant) {
#ifdef VISO_DEBUG_VARIANTS
    return variant >= 2 && variant <= 6;
#else
    return variant == 3 || variant == 5 || variant == 6;
#endif
}

// The variant a new context starts with: the build's default, or $VISO_MATCHER when it names a variant of this build (a
// test / A-B aid: the whole suite through another kernel without touching the callers).  Needs no device.
extern "C" int viso_matcher_default(void) {
    const char* e = getenv("VISO_MATCHER");
    if (e && *e) {
        const int v = atoi(e);
        if (matcher_known(v)) return v;
    }
    return VISO_MATCHER_DEFAULT;
}

extern "C" int viso_ctx_set_matcher(viso_ctx* c, int variant) {
    if (c && !viso_ctx_live(c)) { viso_set_error("not a live context handle"); return VISO_ERR_ARG; }
    c = ctx_or_default(c);
    if (!c) return VISO_ERR_HIP;
    if (!matcher_known(variant)) { viso_set_error("viso_ctx_set_matcher: unknown variant %d", variant); return VISO_ERR_ARG; }
    c->matcher_variant = variant;
    return VISO_OK;
}

extern "C" int viso_ctx_set_row8_shift(viso_ctx* c, int shift) {
    if (c && !viso_ctx_live(c)) { viso_set_error("not a live context handle"); return VISO_ERR_ARG; }
    c = ctx_or_default(c);
    if (!c) return VISO_ERR_HIP;
    if (shift < -1 || shift > 3) { viso_set_error("viso_ctx_set_row8_shift: -1 (from the data) or 0..3"); return VISO_ERR_ARG; }
    c->row8_force = shift;
    return VISO_OK;
}

extern "C" int viso_ctx_set_gn_split(viso_ctx* c, int split) {
    if (c && !viso_ctx_live(c)) { viso_set_error("not a live context handle"); return VISO_ERR_ARG; }
    c = ctx_or_default(c);
    if (!c) return VISO_ERR_HIP;
    if (split < 0 || split > 100) { viso_set_error("viso_ctx_set_gn_split: 0 (default) or 1..100"); return VISO_ERR_ARG; }
    c->gn_split = split;
    return VISO_OK;
}

// Variants this build of the library offers (no device needed): fills out[0..cap) and returns how many exist.
extern "C" int viso_matcher_variants(int* out, int cap) {
#ifdef VISO_DEBUG_VARIANTS
    const int v[] = {2, 3, 4, 5, 6};
#else
    const int v[] = {3, 5, 6};
#endif
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; out && i < n && i < cap; ++i) out[i] = v[i];
    return n;
}

extern "C" const char* viso_ctx_matcher_kernel_name(viso_ctx* c) {
    c = ctx_or_default(c);
    return matcher_kernel_name(c ? c->matcher_variant : VISO_MATCHER_DEFAULT);
}

extern "C" void* viso_ctx_stream(viso_ctx* c) { return c && viso_ctx_live(c) ? (void*)c->stream : nullptr; }

extern "C" int viso_ctx_synchronize(viso_ctx* c) {
    if (!c || !viso_ctx_live(c)) { viso_set_error("viso_ctx_synchronize: not a live context handle"); return VISO_ERR_ARG; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->solver_stream) HIP_TRY(hipStreamSynchronize(c->solver_stream));   // RANSAC stages of its batches
    return VISO_OK;
}

int ctx_scratch(viso_ctx* c, int slot, size_t bytes, void** out, bool zero_new) {
    if (bytes < 256) bytes = 256;
    HIP_TRY(hipSetDevice(c->device));
    if (c->scratch_bytes[slot] < bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->scratch[slot]) HIP_TRY(hipFree(c->scratch[slot]));
        c->scratch[slot] = nullptr;
        c->scratch_bytes[slot] = 0;
        size_t want = bytes + bytes / 2;
        HIP_TRY(hipMalloc(&c->scratch[slot], want));
        c->scratch_bytes[slot] = want;
        if (zero_new) HIP_TRY(hipMemsetAsync(c->scratch[slot], 0, want, c->stream));
    }
    *out = c->scratch[slot];
    return VISO_OK;
}

int ctx_pinned(viso_ctx* c, int which, size_t bytes, char** out) {
    if (bytes < 4096) bytes = 4096;
    if (c->pin_bytes[which] < bytes) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));   // nothing in flight reads the old block
        if (c->pin[which]) HIP_TRY(hipHostFree(c->pin[which]));
        c->pin[which] = nullptr;
        c->pin_bytes[which] = 0;
        const size_t want = bytes + bytes / 2;
        HIP_TRY(hipHostMalloc((void**)&c->pin[which], want, hipHostMallocCoherent));   // results are read right behind the call's signal (PlainSignal)
        c->pin_bytes[which] = want;
    }
    *out = c->pin[which];
    return VISO_OK;
}

// The plain family serialises on one lazily created context (the reference is
// single threaded; callers that want concurrency use explicit contexts).
static std::mutex g_mu;
static std::recursive_mutex g_call_mu;
static viso_ctx* g_default = nullptr;

PlainLock::PlainLock() { g_call_mu.lock(); }
PlainLock::~PlainLock() { g_call_mu.unlock(); }

viso_ctx* viso_default_ctx() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_default) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        g_default = viso_ctx_create(dev, nullptr);
    }
    return g_default;
}

// ---- viso_plain_profile: where a plain-family call's time goes ---------------------------------------------------
#include <chrono>
static bool g_prof_on = false;
static viso_plain_times g_prof[VISO_PLAIN_N];
static hipEvent_t g_prof_ev[4];
static bool g_prof_ev_ok = false;
static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

PlainProf::PlainProf(int fn_, hipStream_t s_) : fn(fn_), s(s_), on(g_prof_on), marks(0), t0(0), tw(0), wait(0) {
    if (!on) return;
    if (!g_prof_ev_ok) {
        for (int i = 0; i < 4; ++i) if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) { on = false; return; }
        g_prof_ev_ok = true;
    }
    t0 = now_us();
    if (hipEventRecord(g_prof_ev[0], s) == hipSuccess) marks = 1;
}
void PlainProf::mark(int k) { if (on && hipEventRecord(g_prof_ev[k], s) == hipSuccess) marks |= 1 << k; }
void PlainProf::wait_begin() { if (on) tw = now_us(); }
void PlainProf::wait_end() { if (on) wait += now_us() - tw; }
PlainProf::~PlainProf() {
    if (!on) return;
    viso_plain_times& p = g_prof[fn];
    p.calls += 1;
    p.host_us += now_us() - t0;
    p.wait_us += wait;
    if (marks == 15 && hipEventSynchronize(g_prof_ev[3]) == hipSuccess) {
        float a = 0, b = 0, c = 0;
        if (hipEventElapsedTime(&a, g_prof_ev[0], g_prof_ev[1]) == hipSuccess &&
            hipEventElapsedTime(&b, g_prof_ev[1], g_prof_ev[2]) == hipSuccess &&
            hipEventElapsedTime(&c, g_prof_ev[2], g_prof_ev[3]) == hipSuccess) {
            p.h2d_us += a * 1e3; p.kernel_us += b * 1e3; p.d2h_us += c * 1e3;
        }
    }
}

extern "C" int viso_plain_profile(int enable) {
    PlainLock lk;
    if (enable) memset(g_prof, 0, sizeof(g_prof));
    g_prof_on = enable != 0;
    return VISO_OK;
}
extern "C" int viso_plain_profile_get(int fn, viso_plain_times* out) {
    if (fn < 0 || fn >= VISO_PLAIN_N || !out) { viso_set_error("viso_plain_profile_get: bad argument"); return VISO_ERR_ARG; }
    PlainLock lk;
    *out = g_prof[fn];
    return VISO_OK;
}
extern "C" const char* viso_plain_profile_name(int fn) {
    static const char* names[VISO_PLAIN_N] = {"match_desc", "collect_matches", "triangulate_rectified", "match_circle",
                                              "ransac_minimize_reproj", "minimize_reproj", "get_inliers"};
    return fn >= 0 && fn < VISO_PLAIN_N ? names[fn] : "";
}

