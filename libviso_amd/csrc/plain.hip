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

#include <stdlib.h>
#include <string.h>

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
            if ((kb && memcmp(s.pin, kp, kb) != 0) || (db && memcmp(s.pin + s.o_desc, d, db) != 0)) continue;
            s.stamp = ++pc->clock;
            pc->hits += 1;
            *hit = true;
            return i;
        }
    pc->misses += 1;
    int vi = -1;   // an empty slot, else the least recently used one
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
    if (kb) memcpy(s.pin, kp, kb);
    if (db) memcpy(s.pin + o_desc, d, db);
    int* hdr = reinterpret_cast<int*>(s.pin + o_hdr);
    hdr[0] = n;
    hdr[1] = dlen > VISO_ROW ? 1 : 0;   // rows that do not fit the packed format: the image takes the general path
    HIP_TRY(hipMemcpyAsync(s.dev, s.pin, up, hipMemcpyHostToDevice, c->stream));
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
    const int iq = plain_acquire(c, pc, kp1, d1, n1, dlen, extras, r8s, -1, &hit_q);
    if (iq < 0) return iq;
    const int it = plain_acquire(c, pc, kp2, d2, n2, dlen, extras, r8s, iq, &hit_t);
    if (it < 0) return it;
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
    HIP_TRY(hipMemcpyAsync(blk, hin, o_sorted, hipMemcpyHostToDevice, s));
    pp.mark(1);
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
    // ---- ONE read-back: misc + the worst-case rows (12 B x n1: microseconds of PCIe), then the rows that count
    HIP_TRY(hipMemcpyAsync(hout, blk + o_misc, blk_bytes - o_misc, hipMemcpyDeviceToHost, s));
    pp.wait_begin();
    HIP_TRY(hipStreamSynchronize(s));
    pp.wait_end();
    const int* omisc = reinterpret_cast<const int*>(hout);
    const int m = omisc[3];
    if (m < 0 || m > n1) { viso_set_error("viso_match_desc: device returned %d matches for %d queries", m, n1); return VISO_ERR_HIP; }
    if (m > 0) memcpy(out_match, hout + 256, sizeof(int) * 3 * (size_t)m);
    if (omisc[7] == 0) { sq.bad_host = 0; st.bad_host = 0; }   // no image of this call is flagged: both fit the u16 rows
    pp.mark(3);
    *out_n = m;
    return VISO_OK;
}
