// plain.hip — the plain (host-pointer) match_desc and the image cache behind it.
//
// The plain family is what the patched reference loop calls (adapters/libviso_hip.patch): per frame
//     match_desc(kp1, kp2, d1, d2)              src/viso.cpp:1240   both images new
//     match_desc(kp1, kp1_prev, d1, d1_prev)    :1264               d1 seen a moment ago, d1_prev one frame ago
//     match_desc(kp2, kp2_prev, d2, d2_prev)    :1275               the same for the right images
// so of the six (keypoints, descriptors) sets a frame passes in, four are byte for byte what an earlier call already
// brought to the device -- under another address (`d1.copyTo(d1_prev)`, :1213), so a pointer says nothing.  The context
// keeps the last PLAIN_SLOTS images resident (boundary-layout rows, bucket order, packed u16 rows, 8-bit planes) with a
// pinned host shadow of what the caller passed; an image is recognised by COMPARING its bytes with a shadow (memcmp of
// n, keypoints and descriptors: exact, no hashing), which a core does at 40-75 GB/s where the upload runs at 20-45 and
// drags sort_kp_kernel + pack_desc_kernel behind it (tools/h2d_probe.hip, profiles/r05_drop_in.txt).  A call's small
// inputs travel in ONE host-to-device copy from pinned memory, its results in ONE device-to-host copy behind ONE
// synchronize; kernels that cannot have work (the other call kind's, the general path's when both images are known
// to fit the u16 rows) are not launched.
#include "common.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// ---- helper threads for the two large host-side passes of a call ---------------------------------------------------
// A miss copies the caller's 968 KB of descriptors into the slot's pinned shadow, a hit compares them with it.  One core
// does that at 10-12 GB/s when the caller's array comes from DRAM (85 us: more than the upload and the kernels
// together) and at 40-75 GB/s from its cache.  $VISO_PLAIN_THREADS helpers (default 3, 0 = none) take a slice each;
// they spin for a few tens of microseconds after a job -- the next call of the frame is that close -- and sleep on a
// condition variable otherwise.  Used under the PlainLock only.
struct PlainPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<unsigned long long> gen{0};
    std::atomic<int> done{0}, differ{0}, sleepers{0};
    std::atomic<bool> stop{false};
    int op = 0;                       // 0 = copy, 1 = compare
    const char* a = nullptr; char* b = nullptr; size_t bytes = 0;
    int n = 0;

    static void slice(size_t bytes, int parts, int i, size_t* lo, size_t* hi) {
        *lo = (bytes * (size_t)i / (size_t)parts) & ~(size_t)63;
        *hi = i + 1 == parts ? bytes : (bytes * (size_t)(i + 1) / (size_t)parts) & ~(size_t)63;
    }
    void work(int i) {
        size_t lo, hi;
        slice(bytes, n + 1, i, &lo, &hi);
        if (hi <= lo) return;
        if (op == 0) memcpy(b + lo, a + lo, hi - lo);
        else if (memcmp(b + lo, a + lo, hi - lo) != 0) differ.store(1, std::memory_order_relaxed);
    }
    void run(int i) {
        unsigned long long seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed)) {
                __builtin_ia32_pause();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(60)) {
                    std::unique_lock<std::mutex> lk(mu);
                    sleepers.fetch_add(1);
                    cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen || stop.load(); });
                    sleepers.fetch_sub(1);
                    break;
                }
            }
            if (stop.load()) return;
            seen = gen.load(std::memory_order_acquire);
            work(i);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    explicit PlainPool(int n_) : n(n_) {
        for (int i = 0; i < n; ++i) th.emplace_back([this, i] { run(i); });
    }
    ~PlainPool() {
        { std::lock_guard<std::mutex> lk(mu); stop.store(true); }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    // the caller takes the last slice; returns 1 if (op == compare and) the blocks differ
    int go(int op_, const void* src, void* dst, size_t bytes_) {
        op = op_; a = (const char*)src; b = (char*)dst; bytes = bytes_;
        done.store(0, std::memory_order_relaxed); differ.store(0, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(mu); gen.fetch_add(1, std::memory_order_release); }
        if (sleepers.load() > 0) cv.notify_all();
        work(n);
        while (done.load(std::memory_order_acquire) < n) __builtin_ia32_pause();
        return differ.load(std::memory_order_relaxed);
    }
};
static PlainPool* g_pool = nullptr;
static bool g_pool_tried = false;
static struct PlainPoolReaper { ~PlainPoolReaper() { delete g_pool; g_pool = nullptr; } } g_pool_reaper;

static PlainPool* plain_pool() {
    if (!g_pool_tried) {
        g_pool_tried = true;
        int n = 3;
        if (const char* e = getenv("VISO_PLAIN_THREADS")) n = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw - 1) n = hw - 1;
        if (n > 15) n = 15;
        if (n > 0) g_pool = new PlainPool(n);
    }
    return g_pool;
}
#define PLAIN_POOL_MIN (128 * 1024)   // smaller blocks are not worth a hand-over
static void big_copy(void* dst, const void* src, size_t bytes) {
    PlainPool* p = bytes >= PLAIN_POOL_MIN ? plain_pool() : nullptr;
    if (p) p->go(0, src, dst, bytes); else if (bytes) memcpy(dst, src, bytes);
}
static bool big_equal(const void* shadow, const void* user, size_t bytes) {
    if (bytes == 0) return true;
    const size_t head = bytes < 4096 ? bytes : 4096;   // other images differ within the first bytes: no hand-over for them
    if (memcmp(shadow, user, head) != 0) return false;
    if (bytes == head) return true;
    PlainPool* p = bytes >= PLAIN_POOL_MIN ? plain_pool() : nullptr;
    if (p) return p->go(1, (const char*)user + head, (char*)shadow + head, bytes - head) == 0;
    return memcmp((const char*)shadow + head, (const char*)user + head, bytes - head) == 0;
}

#define PLAIN_SLOTS 4
#define PLAIN_HDR 256      // bytes of an image's header block {n, bad}

struct PlainSlot {
    bool valid;
    int n, dlen, extras, r8s;
    int bad_host;                   // ImageView::bad as the host knows it: 0 / 1, -1 = not read back yet
    unsigned long long stamp;       // LRU clock
    char* pin; size_t pin_bytes;    // shadow of the caller's arrays, in upload layout: kp | desc | hdr
    char* dev; size_t dev_bytes;    // kp | desc | hdr | rows | aux
    size_t o_desc, up_bytes;
    ImageView v;                    // device pointers into dev
};
struct PlainCache {
    PlainSlot slot[PLAIN_SLOTS];
    unsigned long long clock;
    long long hits, misses;
    int enabled;
};

// ---- copy kernels: small blocks between pinned host memory and the device without the copy engine ------------------
__global__ __launch_bounds__(256) void plain_blit_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, unsigned head_words,
                                                         const int* __restrict__ n_rows, int row_words, int max_rows) {
    unsigned total = head_words;
    if (n_rows) {
        int n = *n_rows;
        n = n < 0 ? 0 : n > max_rows ? max_rows : n;
        total += (unsigned)n * (unsigned)row_words;
    }
    const unsigned i = blockIdx.x * 1024u + threadIdx.x;   // four words per thread, 256 apart: coalesced
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned j = i + 256u * k;
        if (j < total) dst[j] = src[j];
    }
}

int plain_blit(hipStream_t s, const void* src, void* dst, size_t head_words, const int* n_rows, int row_words, int max_rows) {
    const size_t total = head_words + (n_rows ? (size_t)max_rows * (size_t)row_words : 0);
    if (total == 0) return VISO_OK;
    if (total > 0x7fffffffu) { viso_set_error("plain_blit: block too large"); return VISO_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(plain_blit_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(dst), (unsigned)head_words, n_rows, row_words, max_rows);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

int PlainStage::flush(hipStream_t s) {
    if (off == 0) return VISO_OK;
    if (off > (1u << 20)) { HIP_TRY(hipMemcpyAsync(d, h, off, hipMemcpyHostToDevice, s)); return VISO_OK; }   // large: the copy engine is faster
    return plain_blit(s, h, d, off / 4);
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
static size_t aux_bytes(size_t n) {   // skp 8n + sidx 4n + rank 4n | bstart + xinfo | qord | sums | rows8, 16-B aligned pieces
    return ((16 * n + 15) / 16) * 16 + ((4 * (VISO_NB + 1) + 32 + 15) / 16) * 16 + ((n + 63) / 64) * 64 + ((8 * n + 15) / 16) * 16 + VISO_ROW8 * n;
}

static PlainCache* plain_cache(viso_ctx* c) {
    if (!c->plain) {
        c->plain = (PlainCache*)calloc(1, sizeof(PlainCache));
        if (!c->plain) return nullptr;
        const char* e = getenv("VISO_PLAIN_CACHE");   // 0: every image is uploaded and packed again (A/B and test aid)
        c->plain->enabled = !(e && *e == '0');
    }
    return c->plain;
}

void plain_cache_free(viso_ctx* c) {
    if (!c->plain) return;
    for (int i = 0; i < PLAIN_SLOTS; ++i) {
        if (c->plain->slot[i].pin) (void)hipHostFree(c->plain->slot[i].pin);
        if (c->plain->slot[i].dev) (void)hipFree(c->plain->slot[i].dev);
    }
    free(c->plain);
    c->plain = nullptr;
}

extern "C" int viso_plain_cache(int enable) {
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    PlainCache* pc = plain_cache(c);
    if (!pc) { viso_set_error("viso_plain_cache: out of memory"); return VISO_ERR_NOMEM; }
    pc->enabled = enable != 0;
    for (int i = 0; i < PLAIN_SLOTS; ++i) pc->slot[i].valid = false;
    return VISO_OK;
}

extern "C" int viso_plain_cache_stats(int64_t* hits, int64_t* misses) {
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    PlainCache* pc = plain_cache(c);
    if (hits) *hits = pc ? pc->hits : 0;
    if (misses) *misses = pc ? pc->misses : 0;
    return VISO_OK;
}

// $VISO_PLAIN_TRACE=1: host microseconds of viso_match_desc by phase, summed per call kind (0 = both images resident,
// 1 / 2 = one / two uploaded), printed to stderr by viso_plain_trace_dump() (a measurement aid, tools/dropin_probe.py)
static double g_tr_us[3][6];
static long g_tr_n[3];
static int g_tr_on = -1;
static double g_acq_us[2];
static long g_acq_n;
static double tr_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
extern "C" void viso_plain_trace_dump(void) {
    static const char* ph[6] = {"acquire_q", "acquire_t", "setup+blit_in", "launches", "wait", "copy_out"};
    for (int k = 0; k < 3; ++k) {
        if (!g_tr_n[k]) continue;
        fprintf(stderr, "viso_match_desc, %d image(s) uploaded, %ld calls:", k, g_tr_n[k]);
        for (int j = 0; j < 6; ++j) fprintf(stderr, "  %s %.1f", ph[j], g_tr_us[k][j] / g_tr_n[k]);
        fprintf(stderr, "  (us per call)\n");
    }
    if (g_acq_n) fprintf(stderr, "uploads: %ld, copy into the shadow %.1f us, hipMemcpyAsync call %.1f us\n", g_acq_n, g_acq_us[0] / g_acq_n, g_acq_us[1] / g_acq_n);
    g_acq_us[0] = g_acq_us[1] = 0; g_acq_n = 0;
    memset(g_tr_us, 0, sizeof(g_tr_us)); memset(g_tr_n, 0, sizeof(g_tr_n));
}

// The slot that holds (kp, d) -- found by comparing bytes, or filled now: shadow copy, ONE upload of kp | desc | hdr.
// `keep` is a slot that must not be evicted (the call's other image), -1 for none.
static int plain_acquire(viso_ctx* c, PlainCache* pc, const float* kp, const float* d, int n, int dlen, int extras, int r8s,
                         int keep, bool* hit) {
    const size_t kb = sizeof(float2) * (size_t)n, db = sizeof(float) * (size_t)n * dlen;
    *hit = false;
    if (pc->enabled)
        for (int i = 0; i < PLAIN_SLOTS; ++i) {
            PlainSlot& s = pc->slot[i];
            if (!s.valid || s.n != n || s.dlen != dlen || s.extras != extras || s.r8s != r8s) continue;
            if ((kb && memcmp(s.pin, kp, kb) != 0) || !big_equal(s.pin + s.o_desc, d, db)) continue;
            pc->hits += 1;   // the stamp stays the upload's: slots are recycled oldest UPLOAD first.  (Refreshed on a hit -- LRU --
            *hit = true;     // the loop's own order evicts frame t-1's left image when frame t's arrives: its last use was the
            return i;        // temporal-left call, before the right images' -- and the temporal calls of frame t upload it again.)
        }
    pc->misses += 1;
    int vi = -1;   // an empty slot, else the one uploaded longest ago
    for (int i = 0; i < PLAIN_SLOTS && vi < 0; ++i)
        if (i != keep && !pc->slot[i].valid) vi = i;
    for (int i = 0; i < PLAIN_SLOTS && (vi < 0 || pc->slot[vi].valid); ++i)
        if (i != keep && (vi < 0 || pc->slot[i].stamp < pc->slot[vi].stamp)) vi = i;
    PlainSlot& s = pc->slot[vi];
    s.valid = false;
    const size_t na = (size_t)(n > 0 ? n : 1);
    const size_t o_desc = al256(sizeof(float2) * na), o_hdr = o_desc + al256(sizeof(float) * na * dlen);
    const size_t up = o_hdr + PLAIN_HDR;
    const size_t o_rows = up, o_aux = o_rows + al256(sizeof(uint16_t) * VISO_ROW * na);
    const size_t total = o_aux + al256(aux_bytes(na));
    if (s.pin_bytes < up) {
        if (s.pin) HIP_TRY(hipHostFree(s.pin));
        s.pin = nullptr; s.pin_bytes = 0;
        HIP_TRY(hipHostMalloc((void**)&s.pin, up + up / 4, hipHostMallocDefault));
        s.pin_bytes = up + up / 4;
    }
    if (s.dev_bytes < total) {
        if (s.dev) HIP_TRY(hipFree(s.dev));
        s.dev = nullptr; s.dev_bytes = 0;
        HIP_TRY(hipMalloc((void**)&s.dev, total + total / 4));
        s.dev_bytes = total + total / 4;
    }
    const double ta0 = g_tr_on > 0 ? tr_now() : 0;
    if (kb) memcpy(s.pin, kp, kb);
    big_copy(s.pin + o_desc, d, db);
    const double ta1 = g_tr_on > 0 ? tr_now() : 0;
    int* hdr = reinterpret_cast<int*>(s.pin + o_hdr);
    hdr[0] = n;
    hdr[1] = dlen > VISO_ROW ? 1 : 0;   // rows that do not fit the packed format: the image takes the general path
    HIP_TRY(hipMemcpyAsync(s.dev, s.pin, up, hipMemcpyHostToDevice, c->stream));
    if (g_tr_on > 0) { g_acq_us[0] += ta1 - ta0; g_acq_us[1] += tr_now() - ta1; g_acq_n += 1; }
    s.n = n; s.dlen = dlen; s.extras = extras; s.r8s = r8s;
    s.bad_host = dlen > VISO_ROW ? 1 : -1;
    s.o_desc = o_desc; s.up_bytes = up;
    ImageView v{};
    v.kp = reinterpret_cast<const float2*>(s.dev);
    v.frows = reinterpret_cast<const float*>(s.dev + o_desc);
    v.n = reinterpret_cast<const int*>(s.dev + o_hdr);
    v.bad = reinterpret_cast<int*>(s.dev + o_hdr) + 1;
    v.rows = reinterpret_cast<uint16_t*>(s.dev + o_rows);
    unsigned char* base = reinterpret_cast<unsigned char*>(s.dev + o_aux);
    v.skp = (float2*)base;
    v.sidx = (int*)(base + 8 * na);
    v.rank = (int*)(base + 12 * na);
    unsigned char* tail = base + ((16 * na + 15) / 16) * 16;
    v.bstart = (int*)tail;
    v.xinfo = (float*)(tail + 4 * (VISO_NB + 1));
    v.qord = (uint8_t*)(tail + ((4 * (VISO_NB + 1) + 32 + 15) / 16) * 16);
    v.sums = (uint2*)((unsigned char*)v.qord + ((na + 63) / 64) * 64);
    v.rows8 = (uint8_t*)v.sums + ((8 * na + 15) / 16) * 16;
    s.v = v;
    s.stamp = ++pc->clock;
    s.valid = true;
    return vi;
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
    if (n1 > VISO_SORT_MAX || n2 > VISO_SORT_MAX) {
        viso_set_error("viso_match_desc: more than %d keypoints per image is not supported by this build", VISO_SORT_MAX);
        return VISO_ERR_UNSUPPORTED;
    }
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    HIP_TRY(hipSetDevice(c->device));
    PlainCache* pc = plain_cache(c);
    if (!pc) { viso_set_error("viso_match_desc: out of memory"); return VISO_ERR_NOMEM; }
    hipStream_t s = c->stream;
    PlainProf pp(VISO_PLAIN_MATCH_DESC, s);
    int r;
    const int variant = matcher_effective(c->matcher_variant, dlen);
    const int extras = pack_extras(c->matcher_variant, dlen);
    // the planes' shift (matcher variant 6): one call, no previous run to learn it from: the default, or the forced one
    const int r8s = c->row8_force >= 0 ? c->row8_force : VISO_R8_DEFAULT;
    // ---- the two images: resident already, or uploaded now (the uploads start before anything else is prepared)
    bool hit_q = false, hit_t = false;
    if (g_tr_on < 0) { const char* e = getenv("VISO_PLAIN_TRACE"); g_tr_on = e && *e == '1'; }
    double tt[7] = {0, 0, 0, 0, 0, 0, 0};
    if (g_tr_on) tt[0] = tr_now();
    const int iq = plain_acquire(c, pc, kp1, d1, n1, dlen, extras, r8s, -1, &hit_q);
    if (iq < 0) return iq;
    if (g_tr_on) tt[1] = tr_now();
    const int it = plain_acquire(c, pc, kp2, d2, n2, dlen, extras, r8s, iq, &hit_t);
    if (it < 0) return it;
    if (g_tr_on) tt[2] = tr_now();
    PlainSlot &sq = pc->slot[iq], &st = pc->slot[it];
    // ---- per-call device memory: one block {problem, views | misc | sorted rows}: its head is uploaded, its tail read back
    struct Head { MatchProblem p; ImageView v[2]; ImageView miss[2]; };
    const size_t o_misc = al256(sizeof(Head)), o_sorted = o_misc + 256, blk_bytes = o_sorted + al256(sizeof(int) * 3 * (size_t)n1);
    char *blk, *hin, *hout;
    int2 *dres, *dovf; int *dpos, *dtile;
    if ((r = ctx_scratch(c, PLAIN_SLOT_IN, blk_bytes, (void**)&blk)) < 0) return r;
    if ((r = ctx_pinned(c, 0, o_sorted, &hin)) < 0) return r;
    if ((r = ctx_pinned(c, 1, blk_bytes - o_misc, &hout)) < 0) return r;
    if ((r = ctx_scratch(c, 6, sizeof(int2) * (size_t)n1, (void**)&dres)) < 0) return r;
    if ((r = ctx_scratch(c, 8, sizeof(int) * (size_t)n1, (void**)&dpos)) < 0) return r;
    if ((r = ctx_scratch(c, 12, sizeof(int) * ((size_t)n1 / 64 + 1), (void**)&dtile)) < 0) return r;
    if ((r = ctx_scratch(c, 13, sizeof(int2) * (size_t)n1, (void**)&dovf)) < 0) return r;
    int* dmisc = reinterpret_cast<int*>(blk + o_misc);
    // misc: [3] m_cnt  [4..5] scored (u64)  [6] ovf_cnt  [7] "some image of this call is flagged" (lets the general
    // kernels leave at once)  [8] tiles match_stereo_kernel declines (follows [7]: BatchMatchArgs::bad[1])
    Head* H = reinterpret_cast<Head*>(hin);
    memset(hin + o_misc, 0, 256);
    int* hmisc = reinterpret_cast<int*>(hin + o_misc);
    // a resident image whose flag the host has not seen yet may be flagged: say so, the kernels look at the image's own flag
    if ((hit_q && sq.bad_host != 0) || (hit_t && st.bad_host != 0) || dlen > VISO_ROW) hmisc[7] = 1;
    MatchProblem P{};
    P.q = sq.v; P.t = st.v;
    P.res = dres; P.sorted = reinterpret_cast<int*>(blk + o_sorted); P.pos = dpos;
    P.m_cnt = dmisc + 3; P.scored = reinterpret_cast<unsigned long long*>(dmisc + 4); P.pidx = 0; P.cap = n1;
    P.tile_flag = dtile;
    P.ovf = dovf; P.ovf_cnt = dmisc + 6;
    H->p = P; H->v[0] = P.q; H->v[1] = P.t;
    int n_miss = 0, cap_miss = 1;
    if (!hit_q) { H->miss[n_miss++] = P.q; cap_miss = n1 > cap_miss ? n1 : cap_miss; }
    if (!hit_t && it != iq) { H->miss[n_miss++] = P.t; cap_miss = n2 > cap_miss ? n2 : cap_miss; }
    if ((r = plain_blit(s, hin, blk, o_sorted / 4)) < 0) return r;
    pp.mark(1);
    if (g_tr_on) tt[3] = tr_now();
    const MatchProblem* dprob = reinterpret_cast<const MatchProblem*>(blk);
    const ImageView* dmiss = reinterpret_cast<const ImageView*>(blk + offsetof(Head, miss));
    if (n_miss) {
        if ((r = launch_sort_kp(s, dmiss, n_miss, cap_miss)) < 0) return r;
        if (dlen <= VISO_ROW && (r = launch_pack(s, dmiss, n_miss, cap_miss, dlen, nullptr, dmisc + 7, extras, r8s, nullptr)) < 0) return r;
    }
    MatchParamsDev mpd[2];
    fill_match_params(&mpd[0], mp);
    mpd[1] = mpd[0];
    const int general_possible = !(sq.bad_host == 0 && st.bad_host == 0);
    const int kinds = mpd[0].epi ? VISO_KIND_STEREO : VISO_KIND_TEMPORAL;
    if ((r = launch_match_timed(s, dprob, 1, n1, dlen, mpd, dmisc + 7, nullptr, nullptr, 0, variant, dovf, dmisc + 6, r8s,
                                general_possible, kinds)) < 0) return r;
    if ((r = launch_sort(s, dprob, 1, n1)) < 0) return r;
    pp.mark(2);
    // ---- ONE read-back: a copy kernel writes misc + the rows that exist (the count is on the device) into pinned memory
    if ((r = plain_blit(s, blk + o_misc, hout, 64, dmisc + 3, 3, n1)) < 0) return r;
    if (g_tr_on) tt[4] = tr_now();
    pp.wait_begin();
    HIP_TRY(hipStreamSynchronize(s));
    pp.wait_end();
    if (g_tr_on) tt[5] = tr_now();
    const int* omisc = reinterpret_cast<const int*>(hout);
    const int m = omisc[3];
    if (m < 0 || m > n1) { viso_set_error("viso_match_desc: device returned %d matches for %d queries", m, n1); return VISO_ERR_HIP; }
    if (m > 0) memcpy(out_match, hout + 256, sizeof(int) * 3 * (size_t)m);
    if (omisc[7] == 0) { sq.bad_host = 0; st.bad_host = 0; }   // no image of this call is flagged: both fit the u16 rows
    pp.mark(3);
    if (g_tr_on) {
        tt[6] = tr_now();
        const int k = (hit_q ? 0 : 1) + (hit_t ? 0 : 1);
        for (int j = 0; j < 6; ++j) g_tr_us[k][j] += tt[j + 1] - tt[j];
        g_tr_n[k] += 1;
    }
    *out_n = m;
    return VISO_OK;
}
