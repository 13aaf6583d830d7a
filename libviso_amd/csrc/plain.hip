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
// inputs travel in ONE copy kernel's read of a pinned block, its results into a pinned mirror by copy workgroups that ride in
// the chain's next launch (common.h, OutArgs) and signal the host through a pinned word (PlainSignal) -- no copy engine, no
// synchronize; kernels that cannot have work (the other call kind's, the general path's when both images are known to
// fit the u16 rows, the wide-band stereo kernel after rectified pairs) are not launched.
#include "common.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// ---- helper threads for the two large host-side passes of a call ---------------------------------------------------
// A miss copies the caller's 968 KB of descriptors into the slot's pinned shadow, a hit compares them with it.  One core
// does that at 10-16 GB/s when the caller's array comes from DRAM and at 40-75 GB/s from its cache.  $VISO_PLAIN_THREADS
// helpers (default 0 = none) take a slice each; they spin for a few tens of microseconds after a job and sleep on a
// condition variable otherwise.  OFF by default, measured (tools/dropin_probe.py, 201 frames): with three helpers the
// copy into the shadow is no faster (the helpers sleep between frames, waking them costs what they save) and the CALLER's
// next pass over the same array -- the loop's `d1.copyTo(d1_prev)`, src/viso.cpp:1213 -- goes from 30-40 to 65-130 us per
// frame, because slices of it now sit in other cores' caches: 1290 against 1400 frames/s.  Kept for hosts whose arrays are
// larger.  Used under the PlainLock only.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
struct PlainPool {
    std::vector<std::thread> th;
    pid_t owner = getpid();           // threads do not survive fork(): a child must not wait for helpers it does not have
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
                cpu_relax();
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
        while (done.load(std::memory_order_acquire) < n) cpu_relax();
        return differ.load(std::memory_order_relaxed);
    }
};
static PlainPool* g_pool = nullptr;
static bool g_pool_tried = false;
// joins the helper threads at process exit (no HIP call in there); not in a forked child, which has none to join
static struct PlainPoolReaper { ~PlainPoolReaper() { if (g_pool && g_pool->owner == getpid()) delete g_pool; g_pool = nullptr; } } g_pool_reaper;

static PlainPool* plain_pool() {
    // a process forked after the pool was made (the per-rank launchers fork BEFORE any HIP call, but a host may not) has the
    // object and none of its threads: go() would spin forever.  The child works without helpers; the parent's object is
    // leaked there on purpose (its destructor would join threads that do not exist).  $VISO_PLAIN_THREADS is not meant to be
    // combined with forked ranks.
    if (g_pool && g_pool->owner != getpid()) g_pool = nullptr;
    if (!g_pool_tried) {
        g_pool_tried = true;
        int n = 0;
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
#define PLAIN_DISTRUST_FRAMES 64
#define PLAIN_HDR 256      // bytes of an image's header block {n, bad}

struct PlainSlot {
    bool valid;
    int n, dlen, extras, r8s;
    int bad_host;                   // ImageView::bad as the host knows it: 0 / 1, -1 = not read back yet
    unsigned long long stamp;       // LRU clock
    char* pin; size_t pin_bytes;    // shadow of the caller's arrays: kp | hdr {n, bad, the ImageView} | desc
    char* dev; size_t dev_bytes;    // kp | hdr | rows | aux          (the f32 rows stay in the shadow: see plain_acquire)
    size_t o_desc, o_hdr;
    ImageView v;                    // device pointers into dev (frows: the pinned shadow)
};
// ---- the frame a stereo call opens ----------------------------------------------------------------------------------
// The reference's loop body is a fixed sequence (src/viso.cpp:1240-1313), and after the stereo call of frame t everything
// its next calls will ask for is already determined by data the library holds: the temporal calls match the two images it
// has just been given against the two of the previous stereo call (resident in their slots), collect_matches /
// triangulate_rectified read the stereo matches and the keypoints.  So the stereo call -- the one round trip the frame
// cannot avoid -- also runs those problems, in the same launches (three match_desc problems in one matcher launch, as the
// batch family does), and keeps the results.  A later call is answered from them only if its ARGUMENTS are byte for byte
// what was assumed (the images by their slots -- already a byte comparison --, match lists, x, the parameter structs by
// memcmp): the functions are pure, equal inputs give the result the direct path would compute, and any other input takes
// the direct path.  Nothing is predicted about the DATA, only about which call comes next; a wrong guess costs the
// guessed work, never a result.  $VISO_PLAIN_SPECULATE=0 / viso_plain_speculate(0) switch it off (every call direct).
#define PF_PROBS 3      // 0 = stereo (L, R), 1 = temporal (L, previous L), 2 = temporal (R, previous R)
struct PlainFrame {
    bool valid;
    int L, R;                              // image slots of the stereo call
    unsigned long long sL, sR;             // their upload stamps (a recycled slot invalidates what was computed from it)
    int cap;                               // rows per problem the blocks are laid out for
    char* dev; size_t dev_bytes;
    char* host; size_t host_bytes;         // pinned mirror of the result part [o_misc, o_end)
    size_t o_misc, o_sorted[PF_PROBS], o_x, o_X, o_circ, o_xc, o_Xpc, o_rs, o_end;
    bool have[PF_PROBS], used[PF_PROBS];
    int nq[PF_PROBS], m[PF_PROBS];
    int tq[PF_PROBS], tt[PF_PROBS];
    unsigned long long stq[PF_PROBS], stt[PF_PROBS];
    viso_match_params mp[PF_PROBS];
    bool have_xX, used_x, used_X;          // collect_matches / triangulate_rectified of the stereo matches
    viso_param tri_p;
    // second part of the chain, still running when the stereo call returns: match_circle of the frame's four lists, the
    // gather of src/viso.cpp:1292-1305, ransac_minimize_reproj with the parameters / stream key the loop is expected to pass
    bool have_B, pending_B, used_circ, used_rs;
    int n_circ;                            // rows of the join (host, once B has been waited for)
    viso_param rs_p; uint64_t rs_seed, rs_frame;
    int seqJ, seqB;                        // sequence numbers of the join's copy-out (match_circle waits for this one only; the RANSAC
                                           // stage runs on) and of the RANSAC stage's copy-out
    bool pending_J;
};
struct PlainCache {
    PlainSlot slot[PLAIN_SLOTS];
    unsigned long long clock;
    long long hits, misses;
    int enabled;
    // frames: [cur] the last stereo call's, [cur ^ 1] the one before, [2] scratch of the calls outside a frame
    PlainFrame frame[3];
    int cur;
    int speculate;                         // 0 = every call direct
    bool tm_known, tm_pattern;             // temporal params seen; a temporal call fitted (L, previous L) / (R, previous R)
    viso_match_params tm;
    bool tri_known, x_pattern;             // triangulate's param seen; a collect call fitted the stereo call's outputs
    viso_param tri_p;
    bool circ_pattern;                     // a match_circle call took the frame's own lists
    unsigned long long frame_no, circ_seen_no; int circ_seen_cnt;
    bool rs_known, rs_pattern, rs_delta_stable;   // ransac's param / seed seen; the call fitted; the stream key advances regularly
    viso_param rs_p; uint64_t rs_seed, rs_last_frame, rs_delta;
    int good_streak;                       // launches in a row whose images all fitted the u16 rows
    int distrust;                          // > 0: a flagged image was seen within the last PLAIN_DISTRUST_FRAMES clean launches
    int narrow_streak;                     // stereo launches in a row in which match_stereo_kernel declined no tile (rectified pairs: always)
    long long general_reruns;
    long long spec_served[4], spec_wasted[4];   // [0] temporal match_desc, [1] collect_matches, [2] triangulate_rectified / match_circle, [3] ransac
};

// ---- the completion signal (common.h, PlainSignal; the device side is there too) -------------------------------------------
int plain_signal_next(viso_ctx* c, PlainSignal* out) {
    if (!c->sig_flag) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipHostMalloc((void**)&c->sig_flag, 256, hipHostMallocCoherent));   // fine grained: the host polls it while the kernel runs
        memset(c->sig_flag, 0, 256);
        HIP_TRY(hipMalloc((void**)&c->sig_ctr, 256));
        HIP_TRY(hipMemsetAsync(c->sig_ctr, 0, 256, c->stream));
        c->sig_seq = 0;
    }
    c->sig_seq += 1;
    out->ctr = c->sig_ctr; out->flag = c->sig_flag; out->seq = c->sig_seq;
    return VISO_OK;
}

int plain_signal_wait(viso_ctx* c, hipStream_t s, int seq) {
    static const int off = [] { const char* e = getenv("VISO_PLAIN_SIGNAL"); return e && *e == '0'; }();   // 0: always hipStreamSynchronize (A/B aid)
    if (c->sig_flag && !off) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 1;; ++spin) {
            if ((int)((unsigned)__atomic_load_n(c->sig_flag, __ATOMIC_ACQUIRE) - (unsigned)seq) >= 0) return VISO_OK;
            cpu_relax();
            // a signal that does not come (a failed launch, a faulting kernel): the stream knows
            if ((spin & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    return VISO_OK;
}

// ---- copy kernels: small blocks between pinned host memory and the device without the copy engine ------------------
__global__ __launch_bounds__(256) void plain_blit_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, unsigned head_words,
                                                         const int* __restrict__ n_rows, int row_words, int max_rows, PlainSignal sig) {
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
    plain_signal_done(sig, gridDim.x);
}

int plain_blit(hipStream_t s, const void* src, void* dst, size_t head_words, const int* n_rows, int row_words, int max_rows, const PlainSignal* sig) {
    size_t total = head_words + (n_rows ? (size_t)max_rows * (size_t)row_words : 0);
    if (total == 0 && sig) total = 1;   // a signal needs its kernel (the kernel copies nothing: its own bound is the device count)
    if (total == 0) return VISO_OK;
    PlainSignal g{nullptr, nullptr, 0};
    if (sig) g = *sig;
    if (total > 0x7fffffffu) { viso_set_error("plain_blit: block too large"); return VISO_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(plain_blit_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(dst), (unsigned)head_words, n_rows, row_words, max_rows, g);
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
        e = getenv("VISO_PLAIN_SPECULATE");
        c->plain->speculate = !(e && *e == '0');
    }
    return c->plain;
}

void plain_cache_free(viso_ctx* c) {
    if (!c->plain) return;
    for (int i = 0; i < PLAIN_SLOTS; ++i) {
        if (c->plain->slot[i].pin) (void)hipHostFree(c->plain->slot[i].pin);
        if (c->plain->slot[i].dev) (void)hipFree(c->plain->slot[i].dev);
    }
    for (int i = 0; i < 3; ++i) {
        if (c->plain->frame[i].host) (void)hipHostFree(c->plain->frame[i].host);
        if (c->plain->frame[i].dev) (void)hipFree(c->plain->frame[i].dev);
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
    HIP_TRY(hipStreamSynchronize(c->stream));
    pc->enabled = enable != 0;
    for (int i = 0; i < PLAIN_SLOTS; ++i) pc->slot[i].valid = false;
    for (int i = 0; i < 3; ++i) { pc->frame[i].valid = false; pc->frame[i].pending_B = false; pc->frame[i].pending_J = false; }
    return VISO_OK;
}

extern "C" int viso_plain_speculate(int enable) {
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    PlainCache* pc = plain_cache(c);
    if (!pc) { viso_set_error("viso_plain_speculate: out of memory"); return VISO_ERR_NOMEM; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    pc->speculate = enable != 0;
    for (int i = 0; i < 3; ++i) { pc->frame[i].valid = false; pc->frame[i].pending_B = false; pc->frame[i].pending_J = false; }
    pc->tm_pattern = pc->x_pattern = pc->circ_pattern = pc->rs_pattern = pc->rs_delta_stable = false;
    return VISO_OK;
}

// out[0..3] = calls answered from a frame's results (temporal match_desc, collect_matches, triangulate_rectified, -),
// out[4..7] = results computed ahead that no call asked for
extern "C" int viso_plain_speculate_stats(int64_t out[8]) {
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c || !out) return VISO_ERR_HIP;
    PlainCache* pc = plain_cache(c);
    for (int i = 0; i < 4; ++i) { out[i] = pc ? pc->spec_served[i] : 0; out[4 + i] = pc ? pc->spec_wasted[i] : 0; }
    return VISO_OK;
}

// launches that left the general kernels out and had to be repeated with them (an image that does not fit the u16 rows
// where none was expected)
extern "C" int64_t viso_plain_general_reruns(void) {
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    return c && c->plain ? c->plain->general_reruns : 0;
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
static double g_acq_us[3];
static long g_acq_n, g_acq_calls;
static double g_srv_us[2];    // temporal calls answered from the frame: [0] the two look-ups, [1] the rest
static long g_srv_n;
static double g_wait_us[2];   // waits of the frame's later calls: [0] match_circle for the join, [1] ransac_minimize_reproj for the stage
static long g_wait_n[2];
static double tr_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
extern "C" void viso_plain_trace_dump(void) {
    static const char* ph[6] = {"acquire_q", "acquire_t", "setup+blit_in", "launches", "wait", "copy_out"};
    for (int k = 0; k < 3; ++k) {
        if (!g_tr_n[k]) continue;
        fprintf(stderr, "viso_match_desc, %d image(s) uploaded, %ld calls:", k, g_tr_n[k]);
        for (int j = 0; j < 6; ++j) fprintf(stderr, "  %s %.1f", ph[j], g_tr_us[k][j] / g_tr_n[k]);
        fprintf(stderr, "  (us per call)\n");
    }
    if (g_acq_n) fprintf(stderr, "uploads: %ld, rows into the shadow %.1f us, pack launch %.1f us; look-ups + keypoints + sort_kp launch %.1f us per call\n", g_acq_n,
                         g_acq_us[0] / g_acq_n, g_acq_us[1] / g_acq_n, g_acq_us[2] / (g_acq_calls > 0 ? g_acq_calls : 1));
    if (g_wait_n[0] || g_wait_n[1])
        fprintf(stderr, "waits behind the stereo call: match_circle for the join %.1f us (%ld), ransac_minimize_reproj for the stage %.1f us (%ld)\n",
                g_wait_n[0] ? g_wait_us[0] / g_wait_n[0] : 0.0, g_wait_n[0], g_wait_n[1] ? g_wait_us[1] / g_wait_n[1] : 0.0, g_wait_n[1]);
    if (g_srv_n) fprintf(stderr, "temporal calls answered from the frame: %ld, look-ups (byte comparison of both images) %.1f us, the rest %.1f us\n", g_srv_n, g_srv_us[0] / g_srv_n, g_srv_us[1] / g_srv_n);
    g_srv_us[0] = g_srv_us[1] = 0; g_srv_n = 0;
    memset(g_wait_us, 0, sizeof(g_wait_us)); memset(g_wait_n, 0, sizeof(g_wait_n));
    memset(g_acq_us, 0, sizeof(g_acq_us)); g_acq_n = 0; g_acq_calls = 0;
    memset(g_tr_us, 0, sizeof(g_tr_us)); memset(g_tr_n, 0, sizeof(g_tr_n));
}

// The slot that holds (kp, d), found by comparing bytes; -1: not resident.
static int plain_lookup(PlainCache* pc, const float* kp, const float* d, int n, int dlen, int extras, int r8s) {
    const size_t kb = sizeof(float2) * (size_t)n, db = sizeof(float) * (size_t)n * dlen;
    if (pc->enabled)
        for (int i = 0; i < PLAIN_SLOTS; ++i) {
            PlainSlot& s = pc->slot[i];
            if (!s.valid || s.n != n || s.dlen != dlen || s.extras != extras || s.r8s != r8s) continue;
            if ((kb && memcmp(s.pin, kp, kb) != 0) || !big_equal(s.pin + s.o_desc, d, db)) continue;
            pc->hits += 1;   // the stamp stays the upload's: slots are recycled oldest UPLOAD first.  (Refreshed on a hit -- LRU --
            return i;        // the loop's own order evicts frame t-1's left image when frame t's arrives: its last use was the
        }                    // temporal-left call, before the right images' -- and the temporal calls of frame t upload it again.)
    pc->misses += 1;
    return -1;
}

// An image that is not resident comes to the device in two pieces.  Keypoints + header (16 KB): sort_kp_kernel fetches them
// from the pinned shadow itself (KpImport) -- plain_prepare picks the slot (`keep`, `keep2`: slots that must not be evicted, the
// call's other image; -1 for none), copies the keypoints into the shadow and fills the import.  The f32 descriptor rows
// (968 KB) are read ONCE, by pack_desc_kernel, which turns them into the u16 rows the matchers use: it reads them straight
// from the shadow over PCIe -- no copy-engine transfer (25 us on the wire + 20 us of host time in hipMemcpyAsync per image,
// and every kernel of the call queued behind both) and no second copy of the rows in device memory; plain_finish copies
// them into the shadow and launches that kernel.  A call's order: prepare both images, ONE sort_kp launch for what is new
// (it needs the keypoints only and runs while the host copies rows), then rows + pack launch image by image, so that the
// GPU packs the first image while the host copies the second.  (Only the general path reads f32 rows again -- images
// with non-integer descriptors -- and then from the shadow: slow, correct.)
static int plain_prepare(viso_ctx* c, PlainCache* pc, const float* kp, int n, int dlen, int extras, int r8s, int keep, int keep2, KpImport* imp) {
    const size_t kb = sizeof(float2) * (size_t)n;
    int vi = -1;   // an empty slot, else the one uploaded longest ago
    for (int i = 0; i < PLAIN_SLOTS && vi < 0; ++i)
        if (i != keep && i != keep2 && !pc->slot[i].valid) vi = i;
    for (int i = 0; i < PLAIN_SLOTS && (vi < 0 || pc->slot[vi].valid); ++i)
        if (i != keep && i != keep2 && (vi < 0 || pc->slot[i].stamp < pc->slot[vi].stamp)) vi = i;
    PlainSlot& s = pc->slot[vi];
    s.valid = false;
    const size_t na = (size_t)(n > 0 ? n : 1);
    const size_t o_hdr = al256(sizeof(float2) * na), o_desc = o_hdr + PLAIN_HDR;
    const size_t pin_need = o_desc + al256(sizeof(float) * na * dlen);
    const size_t o_rows = o_desc, o_aux = o_rows + al256(sizeof(uint16_t) * VISO_ROW * na);
    const size_t total = o_aux + al256(aux_bytes(na));
    if (s.pin_bytes < pin_need) {
        HIP_TRY(hipStreamSynchronize(c->stream));   // a kernel of an earlier call may still read the old shadow
        if (s.pin) HIP_TRY(hipHostFree(s.pin));
        s.pin = nullptr; s.pin_bytes = 0;
        HIP_TRY(hipHostMalloc((void**)&s.pin, pin_need + pin_need / 4, hipHostMallocDefault));
        s.pin_bytes = pin_need + pin_need / 4;
    }
    if (s.dev_bytes < total) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (s.dev) HIP_TRY(hipFree(s.dev));
        s.dev = nullptr; s.dev_bytes = 0;
        HIP_TRY(hipMalloc((void**)&s.dev, total + total / 4));
        s.dev_bytes = total + total / 4;
    }
    if (kb) memcpy(s.pin, kp, kb);
    ImageView v{};
    v.kp = reinterpret_cast<const float2*>(s.dev);
    v.frows = reinterpret_cast<const float*>(s.pin + o_desc);
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
    int* hdr = reinterpret_cast<int*>(s.pin + o_hdr);
    hdr[0] = n;
    hdr[1] = dlen > VISO_ROW ? 1 : 0;   // rows that do not fit the packed format: the image takes the general path
    memcpy(s.pin + o_hdr + 64, &v, sizeof(v));
    // sort_kp_kernel gets the image's view in its arguments, writes the header words {n, bad} and leaves a copy of the view on
    // the device where pack_desc_kernel's launch finds it (the copy kernel that used to bring keypoints + header over first is gone)
    *imp = KpImport{};
    imp->src_kp = reinterpret_cast<const float2*>(s.pin); imp->n = n; imp->bad0 = hdr[1];
    imp->view_dst = reinterpret_cast<ImageView*>(s.dev + o_hdr + 64); imp->view = v;
    s.n = n; s.dlen = dlen; s.extras = extras; s.r8s = r8s;
    s.bad_host = dlen > VISO_ROW ? 1 : -1;
    s.o_desc = o_desc; s.o_hdr = o_hdr;
    s.v = v;
    return vi;
}

static int plain_finish(viso_ctx* c, PlainCache* pc, int vi, const float* d, hipStream_t stream) {
    PlainSlot& s = pc->slot[vi];
    const int n = s.n, dlen = s.dlen;
    const double ta0 = g_tr_on > 0 ? tr_now() : 0;
    // (the rows in two parts, the first packed while the second is copied -- halves in round 5: 2 356 against 2 337 frames/s; 5/8 + 3/8
    // of the call's last image in round 6: 2 929 against 2 932, four alternating runs each -- one more launch on the host for a few us
    // less of the GPU's chain: inside the noise both times; not kept)
    big_copy(s.pin + s.o_desc, d, sizeof(float) * (size_t)n * dlen);
    const double ta1 = g_tr_on > 0 ? tr_now() : 0;
    int r;
    const ImageView* dview = reinterpret_cast<const ImageView*>(s.dev + s.o_hdr + 64);
    if (dlen <= VISO_ROW && (r = launch_pack(stream, dview, 1, n > 0 ? n : 1, dlen, nullptr, const_cast<int*>(s.v.bad), s.extras, s.r8s, nullptr)) < 0) return r;
    if (g_tr_on > 0) { g_acq_us[0] += ta1 - ta0; g_acq_us[1] += tr_now() - ta1; g_acq_n += 1; }
    s.stamp = ++pc->clock;
    s.valid = true;
    return VISO_OK;
}

// a copy-out (common.h, OutArgs) as a kernel of its own: where the chain has no next kernel for it to ride in
__global__ __launch_bounds__(256) void plain_out_kernel(OutArgs a) { plain_out_blocks(a, blockIdx.x); }

static bool params_equal(const viso_match_params& a, const viso_match_params& b) {
    return a.enforce_epipolar == b.enforce_epipolar && a.enforce_2nd_best == b.enforce_2nd_best && a.max_neighbors == b.max_neighbors &&
           memcmp(a.F, b.F, sizeof(a.F)) == 0 && memcmp(&a.sampson_thresh, &b.sampson_thresh, 8) == 0 &&
           memcmp(&a.ratio_2nd_best, &b.ratio_2nd_best, 8) == 0 && memcmp(&a.radius, &b.radius, 8) == 0;
}
static bool tri_equal(const viso_param& a, const viso_param& b) {   // the fields triangulate_rectified reads, bit for bit
    return memcmp(&a.base, &b.base, 8) == 0 && memcmp(&a.f, &b.f, 8) == 0 && memcmp(&a.cu, &b.cu, 8) == 0 && memcmp(&a.cv, &b.cv, 8) == 0;
}

struct FrameHead { MatchProblem p[PF_PROBS]; TriItem tri; SolverItem rs; OutArgs outA, outJ; };

// lays a frame's blocks out for `cap` rows per problem (grow only)
static int frame_reserve(viso_ctx* c, PlainFrame& f, int cap) {
    const size_t C = (size_t)(cap > 0 ? cap : 1);
    size_t o = al256(sizeof(FrameHead));
    f.o_misc = o; o += 256;
    for (int p = 0; p < PF_PROBS; ++p) { f.o_sorted[p] = o; o += al256(sizeof(int) * 3 * C); }
    f.o_x = o; o += al256(sizeof(double) * 4 * C);
    f.o_X = o; o += al256(sizeof(double) * 3 * C);
    f.o_circ = o; o += al256(sizeof(int) * 6 * C);
    f.o_xc = o; o += al256(sizeof(double) * 4 * C);
    f.o_Xpc = o; o += al256(sizeof(double) * 3 * C);
    f.o_rs = o; o += 128 + al256(sizeof(int) * C);
    f.o_end = o;
    const size_t scratch = PF_PROBS * (al256(sizeof(int2) * C) + al256(sizeof(int) * C) + al256(sizeof(int) * (C / 64 + 1))) + al256(sizeof(int2) * PF_PROBS * C);
    const size_t total = o + scratch;
    if (f.dev_bytes < total) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (f.dev) HIP_TRY(hipFree(f.dev));
        f.dev = nullptr; f.dev_bytes = 0;
        HIP_TRY(hipMalloc((void**)&f.dev, total + total / 4));
        f.dev_bytes = total + total / 4;
    }
    if (f.host_bytes < f.o_end - f.o_misc) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (f.host) HIP_TRY(hipHostFree(f.host));
        f.host = nullptr; f.host_bytes = 0;
        const size_t want = (f.o_end - f.o_misc) + (f.o_end - f.o_misc) / 4;
        HIP_TRY(hipHostMalloc((void**)&f.host, want, hipHostMallocCoherent));   // read by the host right behind the signal, possibly before the kernel has retired
        f.host_bytes = want;
    }
    f.cap = cap;
    return VISO_OK;
}

static void frame_reset(PlainFrame& f) {
    f.valid = false;
    for (int p = 0; p < PF_PROBS; ++p) { f.have[p] = f.used[p] = false; f.m[p] = 0; }
    f.have_xX = f.used_x = f.used_X = false;
    f.have_B = f.used_circ = f.used_rs = false;
    f.n_circ = 0;
}

// the second part of a frame's chain has finished: its counters are in the mirror
static int frame_wait_J(viso_ctx* c, PlainFrame& f) {   // the join (and the gathered columns) are in the mirror
    if (!f.pending_J) return VISO_OK;
    const double tw0 = g_tr_on > 0 ? tr_now() : 0;
    { const int r_ = plain_signal_wait(c, c->stream, f.seqJ); if (r_ < 0) return r_; }
    if (g_tr_on > 0) { g_wait_us[0] += tr_now() - tw0; g_wait_n[0] += 1; }
    f.pending_J = false;
    const int* om = reinterpret_cast<const int*>(f.host);
    f.n_circ = om[32];
    return VISO_OK;
}
static int frame_wait_B(viso_ctx* c, PlainFrame& f) {
    if (!f.pending_B) return VISO_OK;
    const double tw0 = g_tr_on > 0 ? tr_now() : 0;
    { const int r_ = plain_signal_wait(c, c->stream, f.seqB); if (r_ < 0) return r_; }   // the chain's last kernel: everything before it on the stream is done
    if (g_tr_on > 0) { g_wait_us[1] += tr_now() - tw0; g_wait_n[1] += 1; }
    f.pending_B = false; f.pending_J = false;
    const int* om = reinterpret_cast<const int*>(f.host);
    f.n_circ = om[32];
    return VISO_OK;
}

// what a frame computed ahead and nobody asked for: the pattern it was guessed from no longer holds
static void frame_retire(PlainCache* pc, PlainFrame& f) {
    if (!f.valid) return;
    if ((f.have[1] && !f.used[1]) || (f.have[2] && !f.used[2])) { pc->tm_pattern = false; pc->spec_wasted[0] += 1; }
    if (f.have_xX && !f.used_x) { pc->x_pattern = false; pc->spec_wasted[1] += 1; }
    if (f.have_xX && f.used_x && !f.used_X) pc->spec_wasted[2] += 1;
    if (f.have_B && !f.used_circ) { pc->circ_pattern = false; pc->spec_wasted[2] += 1; }
    if (f.have_B && !f.used_rs) { pc->rs_pattern = false; pc->spec_wasted[3] += 1; }
}

static const int* frame_rows(const PlainFrame& f, int p) { return reinterpret_cast<const int*>(f.host + (f.o_sorted[p] - f.o_misc)); }

#define PLAIN_RERUN 2
static int match_run(viso_ctx* c, PlainCache* pc, PlainProf& pp, int iq, int it, bool hit_q, bool hit_t, int n1, int n2, int dlen,
                     const viso_match_params* mp, int variant, int extras, int r8s, int32_t* out_match, int* out_n,
                     bool force_general, double* tt);

// An error return must not leave work in flight: the kernels queued so far (copy, sort_kp, pack -- possibly on the side
// stream) read the slots' pinned shadows and the context's pinned block, a slot may already be marked valid although its
// pack kernel never ran to the end, and the next call would memcpy over a shadow a queued kernel is still pulling over
// PCIe, or get a byte-comparison hit on half-packed rows.  So: wait for both streams, forget every image and every frame.
// (Errors here are HIP failures; nothing is optimised for them.)
static void plain_quiesce(viso_ctx* c, PlainCache* pc) {
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < PLAIN_SLOTS; ++i) pc->slot[i].valid = false;
    for (int i = 0; i < 3; ++i) { pc->frame[i].valid = false; pc->frame[i].pending_B = false; pc->frame[i].pending_J = false; }
    pc->good_streak = 0;
    pc->narrow_streak = 0;
}

static int match_desc_locked(viso_ctx* c, PlainCache* pc, const float* kp1, int n1, const float* kp2, int n2, const float* d1, const float* d2,
                             int dlen, const viso_match_params* mp, int32_t* out_match, int* out_n);

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
    const int r = match_desc_locked(c, pc, kp1, n1, kp2, n2, d1, d2, dlen, mp, out_match, out_n);
    if (r < 0) plain_quiesce(c, pc);
    return r;
}

static int match_desc_locked(viso_ctx* c, PlainCache* pc, const float* kp1, int n1, const float* kp2, int n2, const float* d1, const float* d2,
                             int dlen, const viso_match_params* mp, int32_t* out_match, int* out_n) {
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
    int iq = plain_lookup(pc, kp1, d1, n1, dlen, extras, r8s);
    int it = plain_lookup(pc, kp2, d2, n2, dlen, extras, r8s);
    hit_q = iq >= 0; hit_t = it >= 0;
    const bool same_image = !hit_q && !hit_t && kp1 == kp2 && d1 == d2 && n1 == n2;   // one upload serves both sides
    {
        KpImport imp[2];
        int n_imp = 0, cap_imp = 1;
        if (!hit_q) {
            if ((iq = plain_prepare(c, pc, kp1, n1, dlen, extras, r8s, it, -1, &imp[n_imp])) < 0) return iq;
            ++n_imp; cap_imp = n1 > cap_imp ? n1 : cap_imp;
        }
        if (same_image) it = iq;
        else if (!hit_t) {
            if ((it = plain_prepare(c, pc, kp2, n2, dlen, extras, r8s, iq, -1, &imp[n_imp])) < 0) return it;
            ++n_imp; cap_imp = n2 > cap_imp ? n2 : cap_imp;
        }
        // the new images' keypoints: ONE launch, a workgroup per image, while the host copies the rows
        if (n_imp && (r = launch_sort_kp(s, nullptr, n_imp, cap_imp, nullptr, 0, nullptr, imp, n_imp)) < 0) return r;
    }
    if (g_tr_on > 0) { g_acq_us[2] += tr_now() - tt[0]; g_acq_calls += 1; }
    if (!hit_q && (r = plain_finish(c, pc, iq, d1, s)) < 0) return r;
    if (g_tr_on) tt[1] = tr_now();
    if (!hit_t && !same_image && (r = plain_finish(c, pc, it, d2, s)) < 0) return r;
    if (g_tr_on) tt[2] = tr_now();
    const bool stereo_call = mp->enforce_epipolar != 0;
    // ---- a temporal call the frame's stereo call has already answered?
    if (!stereo_call) {
        PlainFrame& cur = pc->frame[pc->cur];
        PlainFrame& prv = pc->frame[pc->cur ^ 1];
        if (pc->speculate && cur.valid)
            for (int p = 1; p < PF_PROBS; ++p)
                if (cur.have[p] && cur.tq[p] == iq && cur.tt[p] == it && pc->slot[iq].stamp == cur.stq[p] &&
                    pc->slot[it].stamp == cur.stt[p] && params_equal(*mp, cur.mp[p])) {
                    const int m = cur.m[p];
                    if (m > 0) memcpy(out_match, frame_rows(cur, p), sizeof(int) * 3 * (size_t)m);
                    *out_n = m;
                    cur.used[p] = true;
                    pc->spec_served[0] += 1;
                    if (g_tr_on > 0) { g_srv_us[0] += tt[2] - tt[0]; g_srv_us[1] += tr_now() - tt[2]; g_srv_n += 1; }
                    return VISO_OK;
                }
        // the direct path it is; remember what a temporal call looks like, and whether it is the loop's
        pc->tm = *mp; pc->tm_known = true;
        if (cur.valid && prv.valid && ((iq == cur.L && it == prv.L && pc->slot[it].stamp == prv.sL) ||
                                       (iq == cur.R && it == prv.R && pc->slot[it].stamp == prv.sR))) pc->tm_pattern = true;
    }
    for (int pass = 0; pass < 2; ++pass) {   // second pass: an image nobody expected turned out not to fit the u16 rows
        r = match_run(c, pc, pp, iq, it, hit_q, hit_t, n1, n2, dlen, mp, variant, extras, r8s, out_match, out_n, pass == 1, tt);
        if (r != PLAIN_RERUN) break;
    }
    if (r < 0) return r;
    if (g_tr_on) {
        tt[6] = tr_now();
        const int k = (hit_q ? 0 : 1) + (hit_t ? 0 : 1);
        for (int j = 0; j < 6; ++j) g_tr_us[k][j] += tt[j + 1] - tt[j];
        g_tr_n[k] += 1;
    }
    return VISO_OK;
}

static int match_run(viso_ctx* c, PlainCache* pc, PlainProf& pp, int iq, int it, bool hit_q, bool hit_t, int n1, int n2, int dlen,
                     const viso_match_params* mp, int variant, int extras, int r8s, int32_t* out_match, int* out_n,
                     bool force_general, double* tt) {
    hipStream_t s = c->stream;
    int r;
    (void)n2; (void)hit_q; (void)hit_t;
    const bool stereo_call = mp->enforce_epipolar != 0;
    const int saved_cur = pc->cur;
    const unsigned long long saved_no = pc->frame_no;
    // ---- which frame object takes the call, and which problems ride along
    PlainFrame* f = &pc->frame[2];
    int np = 1;
    bool spec_x = false, spec_B = false;
    if (stereo_call && pc->speculate) {
        if (pc->frame[pc->cur ^ 1].pending_B && (r = frame_wait_B(c, pc->frame[pc->cur ^ 1])) < 0) return r;
        // The frame that leaves is the one before the last stereo call's: every call that could have used its results is over,
        // so what it computed ahead and nobody asked for is known now, whatever becomes of THIS launch (a repeated launch finds
        // the object reset -- not valid -- and counts nothing twice)
        frame_retire(pc, pc->frame[pc->cur ^ 1]);
        pc->frame_no += 1;
        pc->cur ^= 1;
        f = &pc->frame[pc->cur];
        PlainFrame& prv = pc->frame[pc->cur ^ 1];
        frame_reset(*f);
        f->L = iq; f->R = it; f->sL = pc->slot[iq].stamp; f->sR = pc->slot[it].stamp;
        if (pc->tm_known && pc->tm_pattern && prv.valid && iq != it && prv.L != prv.R &&
            pc->slot[prv.L].valid && pc->slot[prv.L].stamp == prv.sL && pc->slot[prv.R].valid && pc->slot[prv.R].stamp == prv.sR &&
            pc->slot[prv.L].dlen == dlen && pc->slot[prv.R].dlen == dlen && pc->slot[prv.L].extras == extras && pc->slot[prv.R].extras == extras &&
            pc->slot[prv.L].r8s == r8s && pc->slot[prv.R].r8s == r8s && prv.L != iq && prv.L != it && prv.R != iq && prv.R != it)
            np = 3;
        spec_x = pc->tri_known && pc->x_pattern;
        spec_B = np == 3 && spec_x && prv.have_xX && prv.have[0] && pc->circ_pattern && pc->rs_known && pc->rs_pattern && pc->rs_delta_stable &&
                 pc->rs_p.ransac_iter >= 1 && pc->rs_p.ransac_iter <= 4096;
    } else {
        frame_reset(*f);
    }
    PlainSlot &sq = pc->slot[iq], &st = pc->slot[it];
    f->have[0] = true; f->nq[0] = n1; f->tq[0] = iq; f->tt[0] = it; f->stq[0] = sq.stamp; f->stt[0] = st.stamp; f->mp[0] = *mp;
    if (np == 3) {
        PlainFrame& prv = pc->frame[pc->cur ^ 1];
        f->have[1] = true; f->nq[1] = sq.n; f->tq[1] = iq; f->tt[1] = prv.L; f->stq[1] = sq.stamp; f->stt[1] = prv.sL; f->mp[1] = pc->tm;
        f->have[2] = true; f->nq[2] = st.n; f->tq[2] = it; f->tt[2] = prv.R; f->stq[2] = st.stamp; f->stt[2] = prv.sR; f->mp[2] = pc->tm;
    }
    int cap = n1;
    for (int p = 1; p < np; ++p) cap = f->nq[p] > cap ? f->nq[p] : cap;
    if ((r = frame_reserve(c, *f, cap)) < 0) return r;
    const size_t C = (size_t)(cap > 0 ? cap : 1);
    // ---- the head of the frame's block: problems, the images to sort and pack, zeroed counters -- ONE copy in
    char* hin;
    if ((r = ctx_pinned(c, 0, f->o_misc + 256, &hin)) < 0) return r;
    FrameHead* H = reinterpret_cast<FrameHead*>(hin);
    memset(hin + f->o_misc, 0, 256);
    int* hmisc = reinterpret_cast<int*>(hin + f->o_misc);
    int* dmisc = reinterpret_cast<int*>(f->dev + f->o_misc);
    // misc: [6] overflow queue length  [7] "some image of this launch is flagged" (lets the general kernels leave at once)
    // [8] tiles match_stereo_kernel declines (BatchMatchArgs::bad[1])  [16 + 4p] matches of problem p, [18 + 4p] its scored pairs (u64)
    char* sc = f->dev + f->o_end;
    int2* dovf = reinterpret_cast<int2*>(sc + PF_PROBS * (al256(sizeof(int2) * C) + al256(sizeof(int) * C) + al256(sizeof(int) * (C / 64 + 1))));
    // The general (float / double) kernels are for images whose descriptors do not fit the u16 rows.  Whether a NEW image
    // does is known only after its pack kernel has run; once a few launches in a row have seen none, new images are expected
    // to fit and the three general kernels are left out of the launch (15 us of the chain).  If the expectation fails
    // -- the images' own flags come back with the results -- sort_matches_kernel has emitted EMPTY lists for the problems
    // concerned (flagged_empty) and the call is repeated with the general kernels: same results, later.
    bool any_bad = false, any_unknown = false;
    for (int p = 0; p < np; ++p) {
        PlainSlot &a = pc->slot[f->tq[p]], &b = pc->slot[f->tt[p]];
        if (a.bad_host == 1 || b.bad_host == 1) any_bad = true;
        if (a.bad_host < 0 || b.bad_host < 0) any_unknown = true;
    }
    // A source that has produced a flagged image lately keeps the general kernels in the launch: a stream with sporadic
    // fractional images would otherwise pay a dropped launch + a full synchronize + a second launch on every such frame
    const bool trust = pc->good_streak >= 2 && !force_general && pc->distrust == 0;
    const bool need_general = any_bad || (any_unknown && !trust) || dlen > VISO_ROW;
    for (int p = 0; p < np; ++p) {
        PlainSlot &a = pc->slot[f->tq[p]], &b = pc->slot[f->tt[p]];
        MatchProblem P{};
        P.q = a.v; P.t = b.v;
        char* ps = sc + p * (al256(sizeof(int2) * C) + al256(sizeof(int) * C) + al256(sizeof(int) * (C / 64 + 1)));
        P.res = reinterpret_cast<int2*>(ps);
        P.pos = reinterpret_cast<int*>(ps + al256(sizeof(int2) * C));
        P.tile_flag = reinterpret_cast<int*>(ps + al256(sizeof(int2) * C) + al256(sizeof(int) * C));
        P.sorted = reinterpret_cast<int*>(f->dev + f->o_sorted[p]);
        P.m_cnt = dmisc + 16 + 4 * p; P.scored = reinterpret_cast<unsigned long long*>(dmisc + 18 + 4 * p);
        P.pidx = p == 0 ? 0 : 1; P.cap = cap;
        P.ovf = dovf; P.ovf_cnt = dmisc + 6;
        H->p[p] = P;
    }
    if (need_general) hmisc[7] = 1;   // the general kernels' early-exit hint; they look at every image's own flag
    if (spec_x) {
        TriItem T{};
        T.kp1 = sq.v.kp; T.kp2 = st.v.kp; T.match = reinterpret_cast<const int*>(f->dev + f->o_sorted[0]); T.m_cnt = dmisc + 16;
        T.x = reinterpret_cast<double*>(f->dev + f->o_x); T.X = reinterpret_cast<double*>(f->dev + f->o_X); T.ld = (int)C;
        H->tri = T;
    }
    int *rs_tab = nullptr, *rs_queue = nullptr, rs_tabn = 0;
    if (spec_B) {
        PlainFrame& prv = pc->frame[pc->cur ^ 1];
        const int iters = pc->rs_p.ransac_iter;
        double* dtrh; int* dhyp; char* drot;
        rs_tabn = sq.n > st.n ? sq.n : st.n;
        if (pc->slot[prv.L].n > rs_tabn) rs_tabn = pc->slot[prv.L].n;
        if ((r = ctx_scratch(c, 16, sizeof(int) * 3 * (size_t)(rs_tabn + 1), (void**)&rs_tab)) < 0) return r;
        if ((r = ctx_scratch(c, 4, sizeof(int) * (4 + 2 * (size_t)iters), (void**)&dhyp)) < 0) return r;
        if ((r = ctx_scratch(c, 5, sizeof(double) * 6 * (size_t)(iters + 1), (void**)&dtrh)) < 0) return r;
        if ((r = ctx_scratch(c, 8, sizeof(int) * (2 + 4 * (size_t)iters + 3), (void**)&rs_queue, true)) < 0) return r;
        if ((r = ctx_scratch(c, 9, viso_rot_bytes(iters), (void**)&drot)) < 0) return r;
        f->rs_p = pc->rs_p; f->rs_seed = pc->rs_seed; f->rs_frame = pc->rs_last_frame + pc->rs_delta;
        SolverItem it{};
        it.X = reinterpret_cast<const double*>(f->dev + f->o_Xpc); it.obs = reinterpret_cast<const double*>(f->dev + f->o_xc);
        it.m_ptr = dmisc + 32; it.ld = (int)C; it.samples = nullptr; it.samp_h = rs_queue + 2 + iters; it.frame = f->rs_frame;
        it.tr_h = dtrh; it.ok_h = dhyp; it.cnt_h = dhyp + iters; it.rot = drot;
        int* rso = reinterpret_cast<int*>(f->dev + f->o_rs);
        it.kept = rso; it.ok = rso + 1; it.n_inl = rso + 2; it.tr = reinterpret_cast<double*>(f->dev + f->o_rs + 64); it.inl = reinterpret_cast<int*>(f->dev + f->o_rs + 128);
        H->rs = it;
        (void)prv;
    }
    // ---- the copy-outs into the frame's pinned mirror (common.h, OutArgs): the lists (and x, X) behind the sort kernel -- riding in
    // the join kernel's launch when the frame has one, a kernel of their own otherwise; the join and the gathered columns behind
    // the join kernel, riding in ransac_coop_kernel's launch.  Each signals: match_circle returns as soon as the join is there,
    // the RANSAC stage runs on behind the caller's gather loop
    int seqA = 0;
    {
        auto region = [&](OutArgs& o, size_t off, const int* cnt, int row_words, int max_rows) {
            OutRegion& R = o.r[o.n_regions++];
            R.src = reinterpret_cast<const uint32_t*>(f->dev + off);
            R.dst = reinterpret_cast<uint32_t*>(f->host + (off - f->o_misc));
            R.cnt = cnt; R.row_words = row_words; R.max_rows = max_rows;
        };
        OutArgs& oa = H->outA;
        oa = OutArgs{};
        region(oa, f->o_misc, nullptr, 40, 1);
        {   // the two images' own flags: words 40, 41 of the mirror
            OutRegion& R0 = oa.r[oa.n_regions++];
            R0.src = reinterpret_cast<const uint32_t*>(sq.v.bad); R0.dst = reinterpret_cast<uint32_t*>(f->host) + 40; R0.cnt = nullptr; R0.row_words = 1; R0.max_rows = 1;
            OutRegion& R1 = oa.r[oa.n_regions++];
            R1.src = reinterpret_cast<const uint32_t*>(st.v.bad); R1.dst = reinterpret_cast<uint32_t*>(f->host) + 41; R1.cnt = nullptr; R1.row_words = 1; R1.max_rows = 1;
        }
        for (int p = 0; p < np; ++p) region(oa, f->o_sorted[p], dmisc + 16 + 4 * p, 3, f->nq[p]);
        if (spec_x) {
            for (int k = 0; k < 4; ++k) region(oa, f->o_x + sizeof(double) * C * k, dmisc + 16, 2, n1);
            for (int k = 0; k < 3; ++k) region(oa, f->o_X + sizeof(double) * C * k, dmisc + 16, 2, n1);
        }
        const size_t tpb = spec_B ? 1024 : 256;   // threads of the workgroups that will do it: four words each
        oa.gx = (int)((3 * C + 4 * tpb - 1) / (4 * tpb));
        if ((r = plain_signal_next(c, &oa.sig)) < 0) return r;
        seqA = oa.sig.seq;
        if (spec_B) {
            OutArgs& oj = H->outJ;
            oj = OutArgs{};
            region(oj, f->o_misc + 128, nullptr, 8, 1);                          // misc[32..39]: the join's row count
            region(oj, f->o_circ, dmisc + 32, 6, cap);
            for (int k = 0; k < 4; ++k) region(oj, f->o_xc + sizeof(double) * C * k, dmisc + 32, 2, cap);
            for (int k = 0; k < 3; ++k) region(oj, f->o_Xpc + sizeof(double) * C * k, dmisc + 32, 2, cap);
            oj.gx = (int)((6 * C + 1023) / 1024);
            if ((r = plain_signal_next(c, &oj.sig)) < 0) return r;
            f->seqJ = oj.sig.seq;
        }
    }
    if ((r = plain_blit(s, hin, f->dev, (f->o_misc + 256) / 4)) < 0) return r;
    pp.mark(1);
    if (g_tr_on) tt[3] = tr_now();
    const MatchProblem* dprob = reinterpret_cast<const MatchProblem*>(f->dev);
    MatchParamsDev mpd[2];
    fill_match_params(&mpd[0], mp);
    if (np == 3) fill_match_params(&mpd[1], &pc->tm); else mpd[1] = mpd[0];
    const int general_possible = need_general ? 1 : 0;
    int kinds = np == 3 ? VISO_KIND_ALL : mpd[0].epi ? VISO_KIND_STEREO : VISO_KIND_TEMPORAL;
    // match_batch_kernel<1> takes the tiles whose epipolar band match_stereo_kernel finds too wide (pairs that are not
    // rectified): after a few stereo launches without such a tile it is left out (5 us of the chain the caller waits for);
    // the number of declined tiles comes back with the results, and a launch that had one without the kernel is repeated
    const bool skip_wide = mpd[0].epi != 0 && pc->narrow_streak >= 2 && !force_general;
    if (skip_wide) kinds |= VISO_KIND_NO_WIDE;
    if ((r = launch_match_timed(s, dprob, np, cap, dlen, mpd, dmisc + 7, nullptr, nullptr, 0, variant, dovf, dmisc + 6, r8s,
                                general_possible, kinds)) < 0) return r;
    if (spec_x) {   // collect_matches / triangulate_rectified of the stereo list ride in the sort kernel (no kernel of their own)
        SolverParamsDev sp;
        fill_solver_params(&sp, &pc->tri_p);
        if ((r = launch_sort(s, dprob, np, cap, need_general ? 0 : 1, reinterpret_cast<const TriItem*>(f->dev + offsetof(FrameHead, tri)), &sp)) < 0) return r;
    } else if ((r = launch_sort(s, dprob, np, cap, need_general ? 0 : 1)) < 0) return r;
    pp.mark(2);
    if (!spec_B) {   // no join kernel to ride in
        hipLaunchKernelGGL(plain_out_kernel, dim3((unsigned)(H->outA.n_regions * H->outA.gx)), dim3(256), 0, s, H->outA);
        HIP_TRY(hipGetLastError());
    }
    pp.mark(3);
    if (spec_B) {   // ---- the second part: it keeps running while the caller goes through collect / triangulate / the temporal calls
        PlainFrame& prv = pc->frame[pc->cur ^ 1];
        const int* pmisc = reinterpret_cast<const int*>(prv.dev + prv.o_misc);
        CircleArgs ca{};
        ca.lr = reinterpret_cast<const int*>(f->dev + f->o_sorted[0]); ca.lrp = reinterpret_cast<const int*>(prv.dev + prv.o_sorted[0]);
        ca.m11 = reinterpret_cast<const int*>(f->dev + f->o_sorted[1]); ca.m22 = reinterpret_cast<const int*>(f->dev + f->o_sorted[2]);
        ca.n_lr_p = dmisc + 16; ca.n_lrp_p = pmisc + 16; ca.n11_p = dmisc + 20; ca.n22_p = dmisc + 24;
        ca.rows = reinterpret_cast<int*>(f->dev + f->o_circ); ca.cap = cap; ca.out_n = dmisc + 32;
        // x_c / Xp_c of :1292-1305 come out of the join kernel's tail (no gather kernel of their own)
        ca.g_x = reinterpret_cast<const double*>(f->dev + f->o_x); ca.g_ldx = (int)C;
        ca.g_Xp = reinterpret_cast<const double*>(prv.dev + prv.o_X); ca.g_ldXp = prv.cap > 0 ? prv.cap : 1;
        ca.g_xc = reinterpret_cast<double*>(f->dev + f->o_xc); ca.g_Xpc = reinterpret_cast<double*>(f->dev + f->o_Xpc); ca.g_ldc = (int)C;
        ca.ride = reinterpret_cast<const OutArgs*>(f->dev + offsetof(FrameHead, outA)); ca.ride_blocks = H->outA.n_regions * H->outA.gx;
        if ((r = launch_circle_table(s, ca, rs_tab, rs_tabn)) < 0) return r;
        SolverParamsDev sp;
        fill_solver_params(&sp, &f->rs_p);
        RefitMirror rm{};   // the pose and the inliers go into the mirror from the refit kernel itself, which signals
        rm.res_src = reinterpret_cast<const uint32_t*>(f->dev + f->o_rs); rm.res_dst = reinterpret_cast<uint32_t*>(f->host + (f->o_rs - f->o_misc)); rm.res_words = 32;
        rm.n_inl = reinterpret_cast<const int*>(f->dev + f->o_rs) + 2;
        rm.inl_src = reinterpret_cast<const uint32_t*>(f->dev + f->o_rs + 128); rm.inl_dst = reinterpret_cast<uint32_t*>(f->host + (f->o_rs - f->o_misc) + 128); rm.max_inl = cap;
        if ((r = plain_signal_next(c, &rm.sig)) < 0) return r;
        f->seqB = rm.sig.seq;
        if ((r = launch_ransac(s, reinterpret_cast<const SolverItem*>(f->dev + offsetof(FrameHead, rs)), 1, f->rs_p.ransac_iter, f->rs_seed, sp,
                               rs_queue, c->gn_split ? c->gn_split : 1, cap, &rm,
                               reinterpret_cast<const OutArgs*>(f->dev + offsetof(FrameHead, outJ)), H->outJ.n_regions * H->outJ.gx)) < 0) return r;
        f->have_B = true; f->pending_B = true; f->pending_J = true;
    }
    if (g_tr_on) tt[4] = tr_now();
    pp.wait_begin();
    if ((r = plain_signal_wait(c, s, seqA)) < 0) return r;   // the first copy-out: whatever was launched behind it runs on
    pp.wait_end();
    if (g_tr_on) tt[5] = tr_now();
    const int* omisc = reinterpret_cast<const int*>(f->host);
    for (int p = 0; p < np; ++p) {
        const int m = omisc[16 + 4 * p];
        if (m < 0 || m > f->nq[p]) { frame_reset(*f); viso_set_error("viso_match_desc: device returned %d matches for %d queries", m, f->nq[p]); return VISO_ERR_HIP; }
        f->m[p] = m;
    }
    if (mpd[0].epi != 0) {   // tiles the stereo kernel left to the wide-band kernel (misc[8])
        if (omisc[8] != 0) {
            pc->narrow_streak = 0;
            if (skip_wide) {   // ... which was not in the launch: everything it produced is dropped, the call repeated with it
                HIP_TRY(hipStreamSynchronize(s));
                f->pending_B = false; f->pending_J = false;
                frame_reset(*f);
                pc->cur = saved_cur; pc->frame_no = saved_no;
                pc->general_reruns += 1;
                return PLAIN_RERUN;
            }
        } else {
            pc->narrow_streak += 1;
        }
    }
    sq.bad_host = omisc[40] != 0; st.bad_host = omisc[41] != 0;   // the images' own flags, as their pack kernels left them
    if (sq.bad_host || st.bad_host) {
        pc->good_streak = 0;
        pc->distrust = PLAIN_DISTRUST_FRAMES;
        if (!need_general) {   // unexpected: the launch had no kernel for them.  Everything it produced is dropped, the call repeated
            HIP_TRY(hipStreamSynchronize(s));
            f->pending_B = false; f->pending_J = false;
            frame_reset(*f);
            pc->cur = saved_cur; pc->frame_no = saved_no;
            pc->general_reruns += 1;
            return PLAIN_RERUN;
        }
    } else {
        pc->good_streak += 1;
        if (pc->distrust > 0) pc->distrust -= 1;
    }
    const int m = f->m[0];
    if (m > 0) memcpy(out_match, frame_rows(*f, 0), sizeof(int) * 3 * (size_t)m);
    f->valid = stereo_call && pc->speculate;
    f->have_xX = spec_x; f->tri_p = pc->tri_p;
    *out_n = m;
    return VISO_OK;
}

// ---- the frame's other calls (circle.hip asks before it goes to the device) ----------------------------------------
// collect_matches(kp1, kp2, match) of the stereo call's own output: x is there already.  1 = served, 0 = not.
int plain_try_collect(viso_ctx* c, const float* kp1, int n1, const float* kp2, int n2, const int32_t* match, int n, double* x) {
    PlainCache* pc = c->plain;
    if (!pc || !pc->speculate) return 0;
    PlainFrame& f = pc->frame[pc->cur];
    if (!f.valid || !f.have[0] || f.m[0] != n || n <= 0) return 0;
    const PlainSlot &a = pc->slot[f.L], &b = pc->slot[f.R];
    if (!a.valid || !b.valid || a.stamp != f.sL || b.stamp != f.sR || a.n != n1 || b.n != n2) return 0;
    if (memcmp(a.pin, kp1, sizeof(float2) * (size_t)n1) != 0 || memcmp(b.pin, kp2, sizeof(float2) * (size_t)n2) != 0) return 0;
    if (memcmp(frame_rows(f, 0), match, sizeof(int) * 3 * (size_t)n) != 0) return 0;
    // the call is the loop's: worth computing ahead from the next frame on
    pc->x_pattern = true;
    if (!f.have_xX) return 0;
    const size_t C = (size_t)(f.cap > 0 ? f.cap : 1);
    const double* hx = reinterpret_cast<const double*>(f.host + (f.o_x - f.o_misc));
    for (int k = 0; k < 4; ++k) memcpy(x + (size_t)k * n, hx + C * k, sizeof(double) * (size_t)n);
    f.used_x = true;
    pc->spec_served[1] += 1;
    return 1;
}

// triangulate_rectified(x, param) of that x with the parameters of the last call: X is there already.
int plain_try_triangulate(viso_ctx* c, const double* x, int m, const viso_param* p, double* X) {
    PlainCache* pc = c->plain;
    if (!pc) return 0;
    pc->tri_p = *p; pc->tri_known = true;
    if (!pc->speculate) return 0;
    PlainFrame& f = pc->frame[pc->cur];
    if (!f.valid || !f.have_xX || f.m[0] != m || m <= 0 || !tri_equal(*p, f.tri_p)) return 0;
    const size_t C = (size_t)(f.cap > 0 ? f.cap : 1);
    const double* hx = reinterpret_cast<const double*>(f.host + (f.o_x - f.o_misc));
    for (int k = 0; k < 4; ++k)
        if (memcmp(hx + C * k, x + (size_t)k * m, sizeof(double) * (size_t)m) != 0) return 0;
    const double* hX = reinterpret_cast<const double*>(f.host + (f.o_X - f.o_misc));
    for (int k = 0; k < 3; ++k) memcpy(X + (size_t)k * m, hX + C * k, sizeof(double) * (size_t)m);
    f.used_X = true;
    pc->spec_served[2] += 1;
    return 1;
}

static bool rows_equal(const int* a, const int32_t* b, int n) { return n == 0 || memcmp(a, b, sizeof(int) * 3 * (size_t)n) == 0; }

// match_circle(match_lr, match_lr_prev, match11, match22) of the frame's own four lists: joined already.
int plain_try_circle(viso_ctx* c, const int32_t* lr, int n_lr, const int32_t* lr_prev, int n_lrp, const int32_t* m11, int n11,
                     const int32_t* m22, int n22, int32_t* circ, int32_t* pcl, int cap, int* out_n, int* ret) {
    PlainCache* pc = c->plain;
    if (!pc || !pc->speculate) return 0;
    PlainFrame& f = pc->frame[pc->cur];
    PlainFrame& prv = pc->frame[pc->cur ^ 1];
    if (!f.valid || !prv.valid || !f.have[0] || !prv.have[0] || f.m[0] != n_lr || prv.m[0] != n_lrp) return 0;
    if (!rows_equal(frame_rows(f, 0), lr, n_lr) || !rows_equal(frame_rows(prv, 0), lr_prev, n_lrp)) return 0;
    const bool full = f.have[1] && f.have[2] && f.m[1] == n11 && f.m[2] == n22 && rows_equal(frame_rows(f, 1), m11, n11) && rows_equal(frame_rows(f, 2), m22, n22);
    if (f.have[1] && f.have[2] && !full) return 0;
    pc->circ_pattern = true;   // the loop's call: the stereo lists of this frame and the last
    if (!f.have_B || !full) return 0;
    if (frame_wait_J(c, f) < 0) return 0;
    const int cnt = f.n_circ;
    if (cnt < 0 || cnt > f.cap) return 0;
    const int w = cnt < cap ? cnt : cap;
    const int* rows = reinterpret_cast<const int*>(f.host + (f.o_circ - f.o_misc));
    for (int i = 0; i < w; ++i) {
        circ[4 * i + 0] = rows[6 * i + 0]; circ[4 * i + 1] = rows[6 * i + 1]; circ[4 * i + 2] = rows[6 * i + 2]; circ[4 * i + 3] = rows[6 * i + 3];
        pcl[2 * i + 0] = rows[6 * i + 4]; pcl[2 * i + 1] = rows[6 * i + 5];
    }
    *out_n = cnt;
    f.used_circ = true;
    pc->circ_seen_no = pc->frame_no; pc->circ_seen_cnt = cnt;
    pc->spec_served[2] += 1;
    *ret = VISO_OK;
    if (cnt > cap) { viso_set_error("viso_match_circle: %d rows needed, cap %d", cnt, cap); *ret = VISO_ERR_ARG; }
    return 1;
}

// match_circle took the direct path with `cnt` rows: remember it (the RANSAC call of the same frame is recognised by it)
void plain_note_circle(viso_ctx* c, int cnt) {
    PlainCache* pc = c->plain;
    if (!pc) return;
    pc->circ_seen_no = pc->frame_no; pc->circ_seen_cnt = cnt;
}

static bool rs_param_equal(const viso_param& a, const viso_param& b) {   // every field the solver reads, bit for bit
    return tri_equal(a, b) && a.ransac_iter == b.ransac_iter && memcmp(&a.inlier_threshold, &b.inlier_threshold, 8) == 0 &&
           memcmp(&a.thresh, &b.thresh, 8) == 0;
}

// ransac_minimize_reproj on the gathered columns of that join, with the parameters and the stream key that were expected.
int plain_try_ransac(viso_ctx* c, const double* X, const double* obs, int m, double best_tr[6], int32_t* best_inl, int* n_inl,
                     const viso_param* p, const int32_t* samples, uint64_t seed, uint64_t frame, int* ret) {
    PlainCache* pc = c->plain;
    if (!pc) return 0;
    PlainFrame& f = pc->frame[pc->cur];
    bool served = false;
    if (pc->speculate && !samples && f.valid && f.have_B && rs_param_equal(*p, f.rs_p) && seed == f.rs_seed && frame == f.rs_frame &&
        frame_wait_B(c, f) >= 0 && f.n_circ == m && m >= 3 && m <= f.cap) {
        const size_t C = (size_t)(f.cap > 0 ? f.cap : 1);
        const double* hXp = reinterpret_cast<const double*>(f.host + (f.o_Xpc - f.o_misc));
        const double* hxc = reinterpret_cast<const double*>(f.host + (f.o_xc - f.o_misc));
        bool same = true;
        for (int k = 0; k < 3 && same; ++k) same = memcmp(hXp + C * k, X + (size_t)k * m, sizeof(double) * (size_t)m) == 0;
        for (int k = 0; k < 4 && same; ++k) same = memcmp(hxc + C * k, obs + (size_t)k * m, sizeof(double) * (size_t)m) == 0;
        const int* rso = reinterpret_cast<const int*>(f.host + (f.o_rs - f.o_misc));
        if (same && rso[2] >= 0 && rso[2] <= m) {
            // computed ahead without knowing the caller's best_tr: where the reference would leave it alone (no hypothesis with
            // support, src/viso.cpp:1564-1568) the stage wrote none and says so in rso[0]
            if (!rso[0]) memcpy(best_tr, f.host + (f.o_rs - f.o_misc) + 64, sizeof(double) * 6);
            *n_inl = rso[2];
            if (rso[2] > 0) memcpy(best_inl, f.host + (f.o_rs - f.o_misc) + 128, sizeof(int) * (size_t)rso[2]);
            *ret = rso[1] ? 1 : 0;
            f.used_rs = true;
            pc->spec_served[3] += 1;
            served = true;
        }
    }
    if (!samples) {   // what the next frame's call is expected to look like
        const uint64_t d = frame - pc->rs_last_frame;
        pc->rs_delta_stable = pc->rs_known && d == pc->rs_delta;
        pc->rs_delta = d; pc->rs_last_frame = frame; pc->rs_seed = seed; pc->rs_p = *p; pc->rs_known = true;
        if (served || (pc->circ_seen_no == pc->frame_no && pc->circ_seen_cnt == m)) pc->rs_pattern = true;
    }
    return served ? 1 : 0;
}
