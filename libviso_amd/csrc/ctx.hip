// ctx.hip — context, error reporting and the plain (host-pointer) match_desc.
#include "common.h"

#include <mutex>
#include <stdarg.h>
#include <stdio.h>
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
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            viso_set_error("hipStreamCreate: %s", hipGetErrorString(e));
            delete c;
            return nullptr;
        }
        c->own_stream = true;
    }
    return c;
}

extern "C" void viso_ctx_destroy(viso_ctx* c) {
    if (!c) return;
    hipStreamSynchronize(c->stream);
    for (int i = 0; i < 16; ++i) if (c->scratch[i]) hipFree(c->scratch[i]);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void* viso_ctx_stream(viso_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int viso_ctx_synchronize(viso_ctx* c) {
    if (!c) return VISO_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VISO_OK;
}

int ctx_scratch(viso_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes < 256) bytes = 256;
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
    MatchProblem* dprob;
    int r;
    const size_t n2a = (size_t)(n2 > 0 ? n2 : 1);
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
    if ((r = ctx_scratch(c, 10, sizeof(MatchProblem), (void**)&dprob)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(dk1, kp1, sizeof(float2) * n1, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(df1, d1, sizeof(float) * (size_t)n1 * dlen, hipMemcpyHostToDevice, s));
    if (n2) {
        HIP_TRY(hipMemcpyAsync(dk2, kp2, sizeof(float2) * n2, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(df2, d2, sizeof(float) * (size_t)n2 * dlen, hipMemcpyHostToDevice, s));
    }
    // dmisc: [0]=n1 [1]=n2 [2]=bad [3]=m_cnt [4..5]=scored (u64)
    int hm[8] = {n1, n2, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(dmisc, hm, sizeof(hm), hipMemcpyHostToDevice, s));
    if ((r = launch_pack(s, df1, du1, dmisc + 0, 1, n1, dlen, dmisc + 2)) < 0) return r;
    if (n2 && (r = launch_pack(s, df2, du2, dmisc + 1, 1, n2, dlen, dmisc + 2)) < 0) return r;
    MatchProblem P{};
    P.kp1 = dk1; P.kp2 = dk2; P.d1 = du1; P.d2 = du2; P.f1 = df1; P.f2 = df2;
    P.n1p = dmisc + 0; P.n2p = dmisc + 1; P.res = dres; P.sorted = dsorted; P.pos = dpos;
    P.m_cnt = dmisc + 3; P.scored = (unsigned long long*)(dmisc + 4); P.pidx = 0; P.cap = n1;
    HIP_TRY(hipMemcpyAsync(dprob, &P, sizeof(P), hipMemcpyHostToDevice, s));
    MatchParamsDev mpd[2];
    fill_match_params(&mpd[0], mp);
    mpd[1] = mpd[0];
    if ((r = launch_match(s, dprob, 1, n1, n2, dlen, mpd, dmisc + 2)) < 0) return r;
    if ((r = launch_sort(s, dprob, 1, n1)) < 0) return r;
    int m = 0;
    HIP_TRY(hipMemcpyAsync(&m, dmisc + 3, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (m > 0) HIP_TRY(hipMemcpy(out_match, dsorted, sizeof(int) * 3 * (size_t)m, hipMemcpyDeviceToHost));
    *out_n = m;
    return VISO_OK;
}
