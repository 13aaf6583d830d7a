// ctx.hip — context, error reporting and the plain (host-pointer) match_desc.
#include "common.h"

#include <mutex>
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
    return c;
}

// Frees everything it can and reports the FIRST HIP error it met (viso_last_error).  Must not be called once the
// HIP runtime is being unloaded (static destructors / atexit handlers that run after it): destroy contexts before
// the process starts exiting (the Python wrapper does so from an atexit hook of its own).
extern "C" int viso_ctx_destroy(viso_ctx* c) {
    if (!c) return VISO_OK;
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    note(hipSetDevice(c->device));
    note(hipStreamSynchronize(c->stream));
    if (c->solver_stream) { note(hipStreamSynchronize(c->solver_stream)); note(hipStreamDestroy(c->solver_stream)); }
    for (int i = 0; i < 16; ++i) if (c->scratch[i]) note(hipFree(c->scratch[i]));
    if (c->own_stream) note(hipStreamDestroy(c->stream));
    delete c;
    if (first != hipSuccess) { viso_set_error("viso_ctx_destroy: %s", hipGetErrorString(first)); return VISO_ERR_HIP; }
    return VISO_OK;
}

static viso_ctx* ctx_or_default(viso_ctx* c) { return c ? c : viso_default_ctx(); }

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
    c = ctx_or_default(c);
    if (!c) return VISO_ERR_HIP;
    if (!matcher_known(variant)) { viso_set_error("viso_ctx_set_matcher: unknown variant %d", variant); return VISO_ERR_ARG; }
    c->matcher_variant = variant;
    return VISO_OK;
}

extern "C" int viso_ctx_set_row8_shift(viso_ctx* c, int shift) {
    c = ctx_or_default(c);
    if (!c) return VISO_ERR_HIP;
    if (shift < -1 || shift > 3) { viso_set_error("viso_ctx_set_row8_shift: -1 (from the data) or 0..3"); return VISO_ERR_ARG; }
    c->row8_force = shift;
    return VISO_OK;
}

extern "C" int viso_ctx_set_gn_split(viso_ctx* c, int split) {
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

extern "C" void* viso_ctx_stream(viso_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int viso_ctx_synchronize(viso_ctx* c) {
    if (!c) return VISO_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->solver_stream) HIP_TRY(hipStreamSynchronize(c->solver_stream));   // RANSAC stages of its batches
    return VISO_OK;
}

int ctx_scratch(viso_ctx* c, int slot, size_t bytes, void** out) {
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
    }
    *out = c->scratch[slot];
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

// match_desc, reference src/viso.cpp:669-726.
extern "C" int viso_match_desc(const float* kp1, int n1, const float* kp2, int n2,
                               const float* d1, const float* d2, int dlen,
                               const viso_match_params* mp, int32_t* out_match, int* out_n) {
    if (n1 < 0 || n2 < 0 || dlen <= 0 || !mp || !out_n || mp->max_neighbors <= 0 ||
        (n1 && (!kp1 || !d1 || !out_match)) || (n2 && (!kp2 || !d2))) {
        viso_set_error("viso_match_desc: bad argument (the reference asserts d1.cols==d2.cols, src/viso.cpp:676)");
        return VISO_ERR_ARG;
    }
    *out_n = 0;
    if (n1 == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    hipStream_t s = c->stream;
    float2 *dk1, *dk2; float *df1, *df2; uint16_t *du1, *du2; int2* dres; int *dsorted, *dpos, *dmisc;
    unsigned char* daux;
    MatchProblem* dprob;
    int r;
    const size_t n2a = (size_t)(n2 > 0 ? n2 : 1);
    if (n1 > VISO_SORT_MAX || n2 > VISO_SORT_MAX) {
        viso_set_error("viso_match_desc: more than %d keypoints per image is not supported by this build", VISO_SORT_MAX);
        return VISO_ERR_UNSUPPORTED;
    }
    if ((r = ctx_scratch(c, 0, sizeof(float2) * n1, (void**)&dk1)) < 0) return r;
    if ((r = ctx_scratch(c, 1, sizeof(float2) * n2a, (void**)&dk2)) < 0) return r;
    if ((r = ctx_scratch(c, 2, sizeof(float) * (size_t)n1 * dlen, (void**)&df1)) < 0) return r;
    if ((r = ctx_scratch(c, 3, sizeof(float) * n2a * dlen, (void**)&df2)) < 0) return r;
    if ((r = ctx_scratch(c, 4, sizeof(uint16_t) * (size_t)n1 * VISO_ROW, (void**)&du1)) < 0) return r;
    if ((r = ctx_scratch(c, 5, sizeof(uint16_t) * n2a * VISO_ROW, (void**)&du2)) < 0) return r;
    if ((r = ctx_scratch(c, 6, sizeof(int2) * n1, (void**)&dres)) < 0) return r;
    if ((r = ctx_scratch(c, 7, sizeof(int) * 3 * (size_t)n1, (void**)&dsorted)) < 0) return r;
    if ((r = ctx_scratch(c, 8, sizeof(int) * n1, (void**)&dpos)) < 0) return r;
    if ((r = ctx_scratch(c, 9, sizeof(int) * 16, (void**)&dmisc)) < 0) return r;
    if ((r = ctx_scratch(c, 10, sizeof(MatchProblem) + 2 * sizeof(ImageView), (void**)&dprob)) < 0) return r;
    // per image: skp (8n) + sidx (4n) + rank (4n) + bstart (4(NB+1)) + xinfo (32) + qord (n rounded up to 64) + sums (8n) + rows8 (128n), 16-B aligned pieces
    auto aux_bytes = [](size_t n) { return ((16 * n + 15) / 16) * 16 + ((4 * (VISO_NB + 1) + 32 + 15) / 16) * 16 + ((n + 63) / 64) * 64 + ((8 * n + 15) / 16) * 16 + VISO_ROW8 * n; };
    if ((r = ctx_scratch(c, 11, aux_bytes((size_t)n1) + aux_bytes(n2a), (void**)&daux)) < 0) return r;
    int* dtile;
    if ((r = ctx_scratch(c, 12, sizeof(int) * ((size_t)n1 / 64 + 1), (void**)&dtile)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(dk1, kp1, sizeof(float2) * n1, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(df1, d1, sizeof(float) * (size_t)n1 * dlen, hipMemcpyHostToDevice, s));
    if (n2) {
        HIP_TRY(hipMemcpyAsync(dk2, kp2, sizeof(float2) * n2, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(df2, d2, sizeof(float) * (size_t)n2 * dlen, hipMemcpyHostToDevice, s));
    }
    // dmisc: [0]=n1 [1]=n2 [2]=bad (both images share one flag) [3]=m_cnt [4..5]=scored (u64) [6]=ovf_cnt [7]=bad_any
    int hm[10] = {n1, n2, 0, 0, 0, 0, 0, 0, 0, 0};   // [8] = tiles match_stereo_kernel declines (follows bad_any)
    // the planes' shift (matcher variant 6): one call, no previous run to learn it from: the default, or the forced one
    const int r8s = c->row8_force >= 0 ? c->row8_force : VISO_R8_DEFAULT;
    HIP_TRY(hipMemcpyAsync(dmisc, hm, sizeof(hm), hipMemcpyHostToDevice, s));
    auto view = [&](unsigned char* base, size_t n, const float2* kp, const float* f, const int* np, uint16_t* rows) {
        ImageView v{};
        v.kp = kp; v.frows = f; v.n = np; v.rows = rows; v.bad = dmisc + 2;
        v.skp = (float2*)base;
        v.sidx = (int*)(base + 8 * n);
        v.rank = (int*)(base + 12 * n);
        unsigned char* tail = base + ((16 * n + 15) / 16) * 16;
        v.bstart = (int*)tail;
        v.xinfo = (float*)(tail + 4 * (VISO_NB + 1));
        v.qord = (uint8_t*)(tail + ((4 * (VISO_NB + 1) + 32 + 15) / 16) * 16);
        v.sums = (uint2*)((unsigned char*)v.qord + ((n + 63) / 64) * 64);
        v.rows8 = (uint8_t*)v.sums + ((8 * n + 15) / 16) * 16;
        return v;
    };
    MatchProblem P{};
    P.q = view(daux, (size_t)n1, dk1, df1, dmisc + 0, du1);
    P.t = view(daux + aux_bytes((size_t)n1), n2a, dk2, df2, dmisc + 1, du2);
    P.res = dres; P.sorted = dsorted; P.pos = dpos;
    P.m_cnt = dmisc + 3; P.scored = (unsigned long long*)(dmisc + 4); P.pidx = 0; P.cap = n1;
    P.tile_flag = dtile;
    int2* dovf;
    if ((r = ctx_scratch(c, 13, sizeof(int2) * (size_t)n1, (void**)&dovf)) < 0) return r;
    P.ovf = dovf; P.ovf_cnt = dmisc + 6;
    struct { MatchProblem p; ImageView v[2]; } up;
    up.p = P; up.v[0] = P.q; up.v[1] = P.t;
    HIP_TRY(hipMemcpyAsync(dprob, &up, sizeof(up), hipMemcpyHostToDevice, s));
    const ImageView* dviews = reinterpret_cast<const ImageView*>(dprob + 1);
    const int capmax = n1 > n2 ? n1 : (int)n2a;
    if ((r = launch_sort_kp(s, dviews, 2, capmax)) < 0) return r;
    if (dlen > VISO_ROW) {   // rows do not fit: the one shared flag
        const int one[1] = {1};
        HIP_TRY(hipMemcpyAsync(dmisc + 2, one, sizeof(int), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(dmisc + 7, one, sizeof(int), hipMemcpyHostToDevice, s));
    } else if ((r = launch_pack(s, dviews, 2, capmax, dlen, dmisc + 2, dmisc + 7, pack_extras(c->matcher_variant, dlen), r8s, nullptr)) < 0) return r;
    MatchParamsDev mpd[2];
    fill_match_params(&mpd[0], mp);
    mpd[1] = mpd[0];
    if ((r = launch_match(s, dprob, 1, n1, dlen, mpd, dmisc + 7, c->matcher_variant, dovf, dmisc + 6, r8s)) < 0) return r;
    if ((r = launch_sort(s, dprob, 1, n1)) < 0) return r;
    int m = 0;
    HIP_TRY(hipMemcpyAsync(&m, dmisc + 3, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (m > 0) HIP_TRY(hipMemcpy(out_match, dsorted, sizeof(int) * 3 * (size_t)m, hipMemcpyDeviceToHost));
    *out_n = m;
    return VISO_OK;
}
