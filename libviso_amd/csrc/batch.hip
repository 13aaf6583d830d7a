// batch.hip — device-resident frame batches: the loop body of
// sequence_odometry (reference src/viso.cpp:1205-1327) for many frames per
// launch, every stage on one HIP stream with no host round trip in between.
//
// HBM layout (all allocations made once, at viso_batch_create):
//   kp      [nf][2][cap] float2          boundary layout (x,y)
//   desc    [nf][2][cap][dlen] float     boundary layout (reference Mat N x 121 CV_32F)
//   packed  [nf][2][cap][128] u16        biased rows the matcher reads (256 B, 16-B aligned)
//   res     [3][nf][cap] int2            per query (target | -1, SAD)
//   sorted  [3][nf][cap][3] int          match lists in (dist,i1) order; pos = inverse
//   x_c [nf][4][cap], Xp_c [nf][3][cap]   double, SoA rows like cv::Mat(4,M): the solver's inputs, written by the circle join
// `which` = 0 stereo L->R of frame t, 1 temporal left (t vs t-1), 2 temporal right.
#include "common.h"

#include <string.h>
#include <vector>

#define VISO_NPIN_SLOTS 4

struct viso_batch {
    viso_ctx* ctx;
    int nf, cap, dlen, iters;
    int n_probs;               // padded problem count (multiple of 24)
    float2* kp; float* desc; int* n; uint16_t* packed; uint8_t* packed8; uint2* sums;
    // the 8-bit planes' shift (VISO_R8_*, csrc/common.h): device counters, their pinned landing place, the event behind the copy
    int* r8cnt = nullptr; int* r8pin = nullptr; hipEvent_t r8ev = nullptr; bool r8pending = false; int r8shift = VISO_R8_DEFAULT; int r8last = VISO_R8_DEFAULT; unsigned r8runs = 0; int* bad_img; int* bad_any; int* zero;
    float2* skp; int *sidx, *rank, *bstart; float* xinfo; uint8_t* qord;   // column-bucket view of every image
    uint8_t* images; int img_rows, img_cols;                // optional: [nf][2][rows][cols] uint8 (image-in mode)
    float* h_resp; float2* h_tmp_kp; float* h_tmp_resp; int* h_cnt; size_t h_slots;   // Harris detector scratch
    void* h_part = nullptr; size_t h_part_bytes = 0;                                   // ... of the strip kernel (harris_strip_bytes)
    ImageView* views;                                       // [nf*2] (+1 empty)
    MatchProblem* probs;
    int2* ovf_q;                 // the launch's overflow queue: up to one entry per query of the batch
    int* tile_flag; int tiles;   // [3][nf][tiles] per-64-query-tile scratch of the stereo kernels
    int2* res; int* sorted; int* pos; int* m_cnt; int* ovf_cnt; unsigned long long* scored; size_t zeroed_bytes;
    double *x_c, *Xp_c;          // the solver's inputs: gathered + triangulated by the circle join
    JoinItem* join; SolverItem* sitems;
    int *circ, *pcl, *mc;
    double* tr_h; int *ok_h, *cnt_h, *hq; char* rot;   // hq: list of undecided hypotheses (launch_ransac)
    int* samp_h;                            // [nf][iters][3] sample triples of the run (ransac_hyp_kernel)
    // the *_async uploads stage the caller's (pageable, possibly temporary) n array through a small pinned ring:
    // slot k is reusable once the copy that read it has passed (n_pin_ev[k])
    int* n_pin; hipEvent_t n_pin_ev[VISO_NPIN_SLOTS]; bool n_pin_used[VISO_NPIN_SLOTS]; int n_pin_next;
    double* tr; int *ok, *n_inl, *inl;   // tr, ok, n_inl: ONE device block (tr first), mirrored in pinned memory by every run's last kernel
    unsigned char* pose_pin = nullptr;   // [n_frames] x (6 doubles) | [n_frames] ok | [n_frames] n_inl: what viso_batch_get_poses reads
    size_t pose_bytes = 0;
    MatchParamsDev mp[2];
    SolverParamsDev sp;
    unsigned long long seed, first_frame;
    bool params_set;
    bool timing;
    bool desc_i16;             // the descriptor buffer holds int16 rows (viso_batch_upload_i16*), not the f32 boundary layout
    std::vector<signed char> desc_family;   // per frame: 0 = never uploaded, 1 = f32 rows, 2 = int16 rows (the two must not mix in a run)
    // The RANSAC stage of run k (latency bound: a few hundred waves on serial fp64 chains for ~1 ms) runs on a
    // stream of its own (the context's second stream), so that the matcher of run k+1 — which touches none of its
    // buffers — fills the GPU beside it: stream (matcher, triangulation, circle join) --ev_join--> solver_stream (RANSAC) --ev_ransac--> the next
    // run's circle join (which rewrites the RANSAC inputs).
    hipStream_t solver_stream;
    hipEvent_t ev_join, ev_ransac;
    bool ransac_pending;
    // matcher-kernel timing: event pairs of the runs not yet read back (bounded: the oldest pair is folded into
    // the running sum and reused once VISO_EVENT_POOL pairs are outstanding)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t ev_next;            // ring position of the oldest outstanding pair
    double ev_ms_sum; int ev_n;
    // viso_batch_stamp: time stamps of a run (0 = before its uploads, 1 = after them, 2 = behind its last kernel)
    hipEvent_t ev_stamp[3];
    bool stamps;
};
#define VISO_EVENT_POOL 64

// A handle the library does not know -- null, destroyed, or taken along by viso_ctx_destroy of its context (ctx.hip keeps the
// registry): every entry point answers VISO_ERR_ARG instead of following a freed pointer.
static inline bool dead(const viso_batch* b) { return !b || !viso_batch_live(b); }

static int enter(viso_batch* b) {   // every entry point that allocates, copies or launches
    HIP_TRY(hipSetDevice(b->ctx->device));
    return VISO_OK;
}

static int batch_sync(viso_batch* b) {   // everything the batch has in flight: matcher stream, then its RANSAC stream
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    if (b->solver_stream) HIP_TRY(hipStreamSynchronize(b->solver_stream));
    return VISO_OK;
}

static inline int prob_slot(int which, int t) { return (t / 8) * 24 + which * 8 + (t % 8); }

template <class T>
static int dalloc(T** p, size_t count) {
    *p = nullptr;
    HIP_TRY(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
    return VISO_OK;
}

static void free_solver_bufs(viso_batch* b) {
    if (b->tr_h) hipFree(b->tr_h);
    if (b->ok_h) hipFree(b->ok_h);
    if (b->cnt_h) hipFree(b->cnt_h);
    if (b->rot) hipFree(b->rot);
    if (b->hq) hipFree(b->hq);
    if (b->samp_h) hipFree(b->samp_h);
    b->tr_h = nullptr; b->ok_h = b->cnt_h = b->hq = nullptr; b->samp_h = nullptr; b->rot = nullptr;
}

// Frees everything it can; the first HIP error met is recorded (viso_last_error) and returned.  Like
// viso_ctx_destroy it must run before the HIP runtime starts unloading (not from static destructors).
extern "C" int viso_batch_destroy(viso_batch* b) {
    if (!b) return VISO_OK;
    // the registry decides: a live handle is ours to free; one that viso_ctx_destroy of its context already freed is a no-op
    // (the context left a tombstone for it); anything else -- a second destroy, a pointer that never was a batch -- is an
    // argument error, never a dereference
    const int st = viso_batch_unregister(b);
    if (st == 0) { delete b; return VISO_OK; }   // the shell its context left behind (viso_batch_free(b, true)): nothing else to free
    if (st < 0) { viso_set_error("viso_batch_destroy: not a live batch handle"); return VISO_ERR_ARG; }
    return viso_batch_free(b, false);
}

// Frees a batch that has left the registry (viso_batch_destroy, or viso_ctx_destroy for the batches still alive on it).
// keep_shell (viso_ctx_destroy): everything the batch owns goes, the small host object itself stays allocated until the caller's
// own viso_batch_destroy -- a freed address could be handed to ANOTHER batch (of another thread) meanwhile, and the late
// destroy this library promises to tolerate would then hit that one.
int viso_batch_free(viso_batch* b, bool keep_shell) {
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    note(hipSetDevice(b->ctx->device));
    note(hipStreamSynchronize(b->ctx->stream));
    if (b->solver_stream) note(hipStreamSynchronize(b->solver_stream));   // the context's: not destroyed here
    if (b->ev_join) note(hipEventDestroy(b->ev_join));
    if (b->ev_ransac) note(hipEventDestroy(b->ev_ransac));
    for (int k = 0; k < 3; ++k) if (b->ev_stamp[k]) note(hipEventDestroy(b->ev_stamp[k]));
    for (auto& e : b->events) { note(hipEventDestroy(e.first)); note(hipEventDestroy(e.second)); }
    for (int k = 0; k < VISO_NPIN_SLOTS; ++k) if (b->n_pin_ev[k]) note(hipEventDestroy(b->n_pin_ev[k]));
    if (b->n_pin) note(hipHostFree(b->n_pin));
    if (b->r8pin) note(hipHostFree(b->r8pin));
    if (b->pose_pin) note(hipHostFree(b->pose_pin));
    if (b->r8ev) note(hipEventDestroy(b->r8ev));
    void* ptrs[] = {b->h_part, b->h_resp, b->h_tmp_kp, b->h_tmp_resp, b->h_cnt, b->images, b->skp, b->sidx, b->rank, b->bstart, b->xinfo, b->views,
                    b->kp, b->desc, b->n, b->packed, b->packed8, b->r8cnt, b->sums, b->zero, b->probs, b->res, b->sorted,
                    b->pos, b->m_cnt, b->scored, b->x_c, b->Xp_c, b->join,
                    b->sitems, b->circ, b->pcl, b->mc, b->tr /* + ok, n_inl */, b->inl, b->tr_h, b->ok_h, b->cnt_h, b->hq, b->samp_h, b->rot, b->tile_flag, b->qord, b->ovf_q};
    for (void* p : ptrs) if (p) note(hipFree(p));
    if (keep_shell) { b->ctx = nullptr; b->events.clear(); b->desc_family.clear(); b->desc_family.shrink_to_fit(); }
    else delete b;
    if (first != hipSuccess) { viso_set_error("viso_batch_destroy: %s", hipGetErrorString(first)); return VISO_ERR_HIP; }
    return VISO_OK;
}

static int build_items(viso_batch* b) {
    const int nf = b->nf, cap = b->cap;
    const size_t kpi = (size_t)cap, dsi = (size_t)cap * VISO_ROW, dfi = (size_t)cap * b->dlen;
    // image views: index t*2+side, plus one empty view (n -> 0) for padding problems
    std::vector<ImageView> V((size_t)nf * 2 + 1);
    for (int t = 0; t < nf; ++t)
        for (int side = 0; side < 2; ++side) {
            const size_t i = (size_t)t * 2 + side;
            ImageView& v = V[i];
            v.kp = b->kp + i * kpi; v.frows = b->desc + i * dfi; v.n = b->n + i;
            v.skp = b->skp + i * kpi; v.sidx = b->sidx + i * kpi; v.rank = b->rank + i * kpi;
            v.bstart = b->bstart + i * (VISO_NB + 1); v.xinfo = b->xinfo + i * 8;
            v.rows = b->packed + i * dsi;
            v.sums = b->sums + i * kpi;
            v.rows8 = b->packed8 + i * kpi * VISO_ROW8;
            v.qord = b->qord + i * (((size_t)cap + 63) & ~(size_t)63);
            v.bad = b->bad_img + i;
        }
    {
        ImageView& e = V[(size_t)nf * 2];
        e = V[0];
        e.n = b->zero;
        e.bad = b->zero + 5;
    }
    HIP_TRY(hipMemcpy(b->views, V.data(), sizeof(ImageView) * V.size(), hipMemcpyHostToDevice));
    std::vector<MatchProblem> P((size_t)b->n_probs);
    for (auto& p : P) { memset(&p, 0, sizeof(p)); p.tile_flag = b->tile_flag; p.q = V[(size_t)nf * 2]; p.t = V[(size_t)nf * 2]; p.m_cnt = b->zero + 1; p.scored = (unsigned long long*)(b->zero + 2); p.ovf = b->ovf_q; p.ovf_cnt = b->ovf_cnt; p.res = b->res; p.sorted = b->sorted; p.pos = b->pos; }
    auto img_kp = [&](int t, int side) { return b->kp + ((size_t)t * 2 + side) * kpi; };
    for (int t = 0; t < nf; ++t) {
        for (int which = 0; which < 3; ++which) {
            if (which > 0 && t == 0) continue;   // first frame has no predecessor (:1256-1260)
            MatchProblem& p = P[(size_t)prob_slot(which, t)];
            int qs, qt, ts, tt;                  // query side/frame, target side/frame
            if (which == 0) { qs = 0; qt = t; ts = 1; tt = t; }          // match_desc(kp1,kp2,...) :1240
            else if (which == 1) { qs = 0; qt = t; ts = 0; tt = t - 1; } // (kp1,kp1_prev) :1264
            else { qs = 1; qt = t; ts = 1; tt = t - 1; }                 // (kp2,kp2_prev) :1275
            p.q = V[(size_t)qt * 2 + qs];
            p.t = V[(size_t)tt * 2 + ts];
            const size_t o = (size_t)which * nf + t;
            p.res = b->res + o * cap; p.sorted = b->sorted + o * cap * 3; p.pos = b->pos + o * cap;
            p.m_cnt = b->m_cnt + o; p.scored = b->scored + o;
            p.ovf = b->ovf_q; p.ovf_cnt = b->ovf_cnt;   // one queue and one counter for the whole launch
            p.tile_flag = b->tile_flag + o * b->tiles;
            p.pidx = which == 0 ? 0 : 1; p.cap = cap;
        }
    }
    HIP_TRY(hipMemcpy(b->probs, P.data(), sizeof(MatchProblem) * P.size(), hipMemcpyHostToDevice));
    if (nf > 1) {
        std::vector<JoinItem> J((size_t)nf - 1);
        for (int t = 1; t < nf; ++t) {
            JoinItem& j = J[(size_t)t - 1];
            j.lr = b->sorted + ((size_t)0 * nf + t) * cap * 3; j.lr_cnt = b->m_cnt + t;
            j.res11 = b->res + ((size_t)1 * nf + t) * cap;
            j.res22 = b->res + ((size_t)2 * nf + t) * cap;
            j.pos_lrp = b->pos + ((size_t)0 * nf + (t - 1)) * cap;
            j.res_lrp = b->res + ((size_t)0 * nf + (t - 1)) * cap;
            j.kp1 = img_kp(t, 0); j.kp2 = img_kp(t, 1); j.kp1p = img_kp(t - 1, 0); j.kp2p = img_kp(t - 1, 1);
            j.circ = b->circ + (size_t)t * cap * 4; j.pcl = b->pcl + (size_t)t * cap * 2; j.mc = b->mc + t;
            j.x_c = b->x_c + (size_t)t * 4 * cap; j.Xp_c = b->Xp_c + (size_t)t * 3 * cap; j.ldc = cap;
        }
        HIP_TRY(hipMemcpy(b->join, J.data(), sizeof(JoinItem) * J.size(), hipMemcpyHostToDevice));
    }
    return VISO_OK;
}

static int build_solver_items(viso_batch* b) {
    const int nf = b->nf, cap = b->cap, iters = b->iters;
    if (nf <= 1) return VISO_OK;
    std::vector<SolverItem> S((size_t)nf - 1);
    for (int t = 1; t < nf; ++t) {
        SolverItem& s = S[(size_t)t - 1];
        memset(&s, 0, sizeof(s));
        s.X = b->Xp_c + (size_t)t * 3 * cap; s.obs = b->x_c + (size_t)t * 4 * cap;
        s.m_ptr = b->mc + t; s.ld = cap; s.samples = nullptr; s.samp_h = b->samp_h + (size_t)t * iters * 3;
        s.frame = b->first_frame + (unsigned long long)t;
        s.tr_h = b->tr_h + (size_t)t * iters * 6; s.ok_h = b->ok_h + (size_t)t * iters; s.cnt_h = b->cnt_h + (size_t)t * iters;
        s.rot = b->rot + (size_t)t * viso_rot_bytes(iters);
        s.tr = b->tr + (size_t)t * 6; s.ok = b->ok + t; s.n_inl = b->n_inl + t; s.inl = b->inl + (size_t)t * cap;
    }
    HIP_TRY(hipMemcpy(b->sitems, S.data(), sizeof(SolverItem) * S.size(), hipMemcpyHostToDevice));
    return VISO_OK;
}

extern "C" viso_batch* viso_batch_create(viso_ctx* ctx, int n_frames, int cap, int dlen) {
    if (!ctx || n_frames <= 0 || cap <= 0 || dlen <= 0) { viso_set_error("viso_batch_create: bad argument"); return nullptr; }
    if (!viso_ctx_live(ctx)) { viso_set_error("viso_batch_create: not a live context handle"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { viso_set_error("hipSetDevice failed"); return nullptr; }
    viso_batch* b = new viso_batch();
    b->ctx = ctx; b->nf = n_frames; b->cap = cap; b->dlen = dlen; b->iters = 0;
    b->n_probs = ((n_frames + 7) / 8) * 24;
    b->params_set = false; b->timing = false; b->desc_i16 = false;
    b->desc_family.assign((size_t)n_frames, 0);
    b->ev_next = 0; b->ev_ms_sum = 0; b->ev_n = 0;
    b->solver_stream = nullptr; b->ev_join = nullptr; b->ev_ransac = nullptr; b->ransac_pending = false;
    b->ev_stamp[0] = b->ev_stamp[1] = b->ev_stamp[2] = nullptr; b->stamps = false;
    if (ctx->solver_stream) {
        if (hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b->ev_ransac, hipEventDisableTiming) != hipSuccess) {
            viso_set_error("viso_batch_create: event creation failed");
            if (b->ev_join) hipEventDestroy(b->ev_join);
            if (b->ev_ransac) hipEventDestroy(b->ev_ransac);
            delete b;
            return nullptr;
        }
        b->solver_stream = ctx->solver_stream;
    }
    b->n_pin = nullptr; b->n_pin_next = 0;
    for (int k = 0; k < VISO_NPIN_SLOTS; ++k) { b->n_pin_ev[k] = nullptr; b->n_pin_used[k] = false; }
    b->images = nullptr; b->img_rows = b->img_cols = 0;
    b->h_resp = nullptr; b->h_tmp_kp = nullptr; b->h_tmp_resp = nullptr; b->h_cnt = nullptr; b->h_slots = 0;
    b->tr_h = nullptr; b->ok_h = b->cnt_h = b->hq = nullptr; b->samp_h = nullptr; b->rot = nullptr;
    const size_t nf = (size_t)n_frames, c = (size_t)cap;
    int r = VISO_OK;
    auto A = [&](int rr) { if (r >= 0 && rr < 0) r = rr; };
    A(dalloc(&b->kp, nf * 2 * c)); A(dalloc(&b->desc, nf * 2 * c * dlen)); A(dalloc(&b->n, nf * 2));
    A(dalloc(&b->packed, nf * 2 * c * VISO_ROW)); A(dalloc(&b->packed8, nf * 2 * c * VISO_ROW8)); A(dalloc(&b->r8cnt, 4)); A(dalloc(&b->sums, nf * 2 * c)); A(dalloc(&b->zero, 8));
    A(dalloc(&b->probs, (size_t)b->n_probs));
    A(dalloc(&b->skp, nf * 2 * c)); A(dalloc(&b->sidx, nf * 2 * c)); A(dalloc(&b->rank, nf * 2 * c));
    A(dalloc(&b->bstart, nf * 2 * (VISO_NB + 1))); A(dalloc(&b->xinfo, nf * 2 * 8)); A(dalloc(&b->views, nf * 2 + 1));
    A(dalloc(&b->qord, nf * 2 * ((c + 63) & ~(size_t)63)));
    A(dalloc(&b->res, 3 * nf * c)); A(dalloc(&b->sorted, 3 * nf * c * 3)); A(dalloc(&b->pos, 3 * nf * c));
    A(dalloc(&b->m_cnt, 3 * nf));
    A(dalloc(&b->ovf_q, 3 * nf * c));
    b->tiles = (cap + 63) / 64;
    A(dalloc(&b->tile_flag, 3 * nf * (size_t)b->tiles));
    // per-run counters zeroed by ONE memset: scored[3nf] (u64) | ovf_cnt[3nf] (int) | bad_img[2nf] (int) | bad_any (int)
    b->zeroed_bytes = 3 * nf * sizeof(unsigned long long) + (3 * nf + 2 * nf + 4) * sizeof(int);
    A(dalloc(&b->scored, b->zeroed_bytes / sizeof(unsigned long long) + 1));
    b->ovf_cnt = r >= 0 ? reinterpret_cast<int*>(b->scored + 3 * nf) : nullptr;
    b->bad_img = r >= 0 ? b->ovf_cnt + 3 * nf : nullptr;
    b->bad_any = r >= 0 ? b->bad_img + 2 * nf : nullptr;
    A(dalloc(&b->x_c, nf * 4 * c)); A(dalloc(&b->Xp_c, nf * 3 * c));
    A(dalloc(&b->join, nf)); A(dalloc(&b->sitems, nf));
    A(dalloc(&b->circ, nf * c * 4)); A(dalloc(&b->pcl, nf * c * 2)); A(dalloc(&b->mc, nf));
    // poses, flags and inlier counts in one block: one blit into the pinned mirror per run (three blocking copies of a
    // few bytes were 50 us of a one-pair batch's 0.42 ms)
    b->pose_bytes = nf * (6 * sizeof(double) + 2 * sizeof(int));
    { unsigned char* blk = nullptr; A(dalloc(&blk, b->pose_bytes)); b->tr = reinterpret_cast<double*>(blk); }
    b->ok = r >= 0 ? reinterpret_cast<int*>(b->tr + nf * 6) : nullptr;
    b->n_inl = r >= 0 ? b->ok + nf : nullptr;
    A(dalloc(&b->inl, nf * c));
    if (r >= 0 && hipHostMalloc((void**)&b->pose_pin, b->pose_bytes, hipHostMallocDefault) != hipSuccess) {
        b->pose_pin = nullptr;
        viso_set_error("viso_batch_create: hipHostMalloc of the pose mirror failed");
        r = VISO_ERR_NOMEM;
    }
    if (r >= 0) memset(b->pose_pin, 0, b->pose_bytes);
    if (r >= 0 && (hipHostMalloc((void**)&b->r8pin, sizeof(int) * 4, hipHostMallocDefault) != hipSuccess ||
                   hipEventCreateWithFlags(&b->r8ev, hipEventDisableTiming) != hipSuccess)) {
        viso_set_error("viso_batch_create: pinned buffer / event for the planes' statistics failed");
        r = VISO_ERR_HIP;
    }
    if (r >= 0 && hipHostMalloc((void**)&b->n_pin, sizeof(int) * VISO_NPIN_SLOTS * 2 * nf, hipHostMallocDefault) != hipSuccess) {
        b->n_pin = nullptr;
        viso_set_error("viso_batch_create: hipHostMalloc of the n staging ring failed");
        r = VISO_ERR_NOMEM;
    }
    for (int k = 0; r >= 0 && k < VISO_NPIN_SLOTS; ++k)
        if (hipEventCreateWithFlags(&b->n_pin_ev[k], hipEventDisableTiming) != hipSuccess) { b->n_pin_ev[k] = nullptr; r = VISO_ERR_HIP; }
    if (r < 0) { viso_batch_free(b, false); return nullptr; }
    bool ok = hipMemset(b->zero, 0, 8 * sizeof(int)) == hipSuccess &&
              hipMemset(b->n, 0, nf * 2 * sizeof(int)) == hipSuccess &&
              hipMemset(b->m_cnt, 0, 3 * nf * sizeof(int)) == hipSuccess &&
              hipMemset(b->mc, 0, nf * sizeof(int)) == hipSuccess &&
              hipMemset(b->tr, 0, b->pose_bytes) == hipSuccess &&
              hipMemset(b->scored, 0, b->zeroed_bytes) == hipSuccess;
    if (!ok || build_items(b) < 0) { viso_set_error("viso_batch_create: device initialisation failed"); viso_batch_free(b, false); return nullptr; }
    if (!viso_batch_register(ctx, b)) { viso_set_error("viso_batch_create: the context was destroyed meanwhile"); viso_batch_free(b, false); return nullptr; }
    return b;
}

extern "C" int viso_batch_upload(viso_batch* b, int f0, int nf, const float* kp, const float* desc,
                                 const int32_t* n) {
    if (dead(b) || f0 < 0 || nf < 0 || f0 + nf > b->nf || (nf && (!kp || !desc || !n))) { viso_set_error("viso_batch_upload: bad argument"); return VISO_ERR_ARG; }
    for (int i = 0; i < 2 * nf; ++i)
        if (n[i] < 0 || n[i] > b->cap) { viso_set_error("viso_batch_upload: n[%d]=%d exceeds cap %d", i, n[i], b->cap); return VISO_ERR_ARG; }
    if (nf == 0) return VISO_OK;
    int r;
    if ((r = enter(b)) < 0) return r;
    b->desc_i16 = false;
    for (int t = f0; t < f0 + nf; ++t) b->desc_family[(size_t)t] = 1;
    // the batch's kernels run on a non-blocking stream: order the copies behind them, then wait (synchronous call)
    hipStream_t s = b->ctx->stream;
    const size_t c = (size_t)b->cap;
    HIP_TRY(hipMemcpyAsync(b->kp + (size_t)f0 * 2 * c, kp, sizeof(float2) * (size_t)nf * 2 * c, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->desc + (size_t)f0 * 2 * c * b->dlen, desc, sizeof(float) * (size_t)nf * 2 * c * b->dlen, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->n + (size_t)f0 * 2, n, sizeof(int) * (size_t)nf * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return VISO_OK;
}

// Pinned host memory for the *_async uploads (hipHostMalloc): copies from it run as DMA on the context's stream.
extern "C" void* viso_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { viso_set_error("viso_host_alloc: hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
extern "C" int viso_host_free(void* p) {
    if (p) HIP_TRY(hipHostFree(p));
    return VISO_OK;
}

// n[2*nf] of the caller -> the next slot of the batch's pinned ring -> device, on stream s.  The caller's array is
// not read after this returns.
static int stage_n_async(viso_batch* b, int f0, int nf, const int32_t* n, hipStream_t s) {
    const int k = b->n_pin_next;
    b->n_pin_next = (k + 1) % VISO_NPIN_SLOTS;
    if (b->n_pin_used[k]) HIP_TRY(hipEventSynchronize(b->n_pin_ev[k]));   // the copy that read this slot VISO_NPIN_SLOTS uploads ago
    int* slot = b->n_pin + (size_t)k * 2 * b->nf;
    memcpy(slot, n, sizeof(int) * (size_t)nf * 2);
    HIP_TRY(hipMemcpyAsync(b->n + (size_t)f0 * 2, slot, sizeof(int) * (size_t)nf * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(b->n_pin_ev[k], s));
    b->n_pin_used[k] = true;
    return VISO_OK;
}

// viso_batch_upload without the wait: the three copies are enqueued on the context's stream (behind the batch's
// previous run, in front of the next one) and the call returns.  The host buffers must stay untouched until the
// stream has passed the copies (viso_ctx_synchronize, or any result getter of a later run); with buffers from
// viso_host_alloc the copies are true DMA and overlap the kernels of other contexts.  `n` is validated now and
// copied through a small pinned slot of the batch, so the caller's n array need not be pinned.
extern "C" int viso_batch_upload_async(viso_batch* b, int f0, int nf, const float* kp, const float* desc,
                                       const int32_t* n) {
    if (dead(b) || f0 < 0 || nf < 0 || f0 + nf > b->nf || (nf && (!kp || !desc || !n))) { viso_set_error("viso_batch_upload_async: bad argument"); return VISO_ERR_ARG; }
    for (int i = 0; i < 2 * nf; ++i)
        if (n[i] < 0 || n[i] > b->cap) { viso_set_error("viso_batch_upload_async: n[%d]=%d exceeds cap %d", i, n[i], b->cap); return VISO_ERR_ARG; }
    if (nf == 0) return VISO_OK;
    int r;
    if ((r = enter(b)) < 0) return r;
    b->desc_i16 = false;
    for (int t = f0; t < f0 + nf; ++t) b->desc_family[(size_t)t] = 1;
    hipStream_t s = b->ctx->stream;
    const size_t c = (size_t)b->cap;
    HIP_TRY(hipMemcpyAsync(b->kp + (size_t)f0 * 2 * c, kp, sizeof(float2) * (size_t)nf * 2 * c, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->desc + (size_t)f0 * 2 * c * b->dlen, desc, sizeof(float) * (size_t)nf * 2 * c * b->dlen, hipMemcpyHostToDevice, s));
    return stage_n_async(b, f0, nf, n, s);
}

// The descriptors as int16 (N x dlen, tightly packed): the lossless encoding of the reference's Sobel windows (integers
// in [-1020, 1020], src/viso.cpp:1004-1024) at half the bytes of the CV_32F boundary layout.  They live in the same
// device buffer as the f32 rows (reinterpreted), so all frames of a batch must come through ONE of the two families;
// the last upload decides which pack kernel the next run uses.  sync != 0: wait for the copies.
static int upload_i16_impl(viso_batch* b, int f0, int nf, const float* kp, const int16_t* desc16, const int32_t* n, bool sync,
                           const char* who) {
    if (dead(b) || f0 < 0 || nf < 0 || f0 + nf > b->nf || (nf && (!kp || !desc16 || !n))) { viso_set_error("%s: bad argument", who); return VISO_ERR_ARG; }
    if (b->dlen > VISO_ROW) { viso_set_error("%s: int16 descriptors need dlen <= %d", who, VISO_ROW); return VISO_ERR_UNSUPPORTED; }
    for (int i = 0; i < 2 * nf; ++i)
        if (n[i] < 0 || n[i] > b->cap) { viso_set_error("%s: n[%d]=%d exceeds cap %d", who, i, n[i], b->cap); return VISO_ERR_ARG; }
    if (nf == 0) return VISO_OK;
    int r;
    if ((r = enter(b)) < 0) return r;
    b->desc_i16 = true;
    for (int t = f0; t < f0 + nf; ++t) b->desc_family[(size_t)t] = 2;
    hipStream_t s = b->ctx->stream;
    const size_t c = (size_t)b->cap;
    int16_t* d16 = reinterpret_cast<int16_t*>(b->desc);
    HIP_TRY(hipMemcpyAsync(b->kp + (size_t)f0 * 2 * c, kp, sizeof(float2) * (size_t)nf * 2 * c, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d16 + (size_t)f0 * 2 * c * b->dlen, desc16, sizeof(int16_t) * (size_t)nf * 2 * c * b->dlen, hipMemcpyHostToDevice, s));
    if ((r = stage_n_async(b, f0, nf, n, s)) < 0) return r;
    if (sync) HIP_TRY(hipStreamSynchronize(s));
    return VISO_OK;
}
extern "C" int viso_batch_upload_i16(viso_batch* b, int f0, int nf, const float* kp, const int16_t* desc16, const int32_t* n) {
    return upload_i16_impl(b, f0, nf, kp, desc16, n, true, "viso_batch_upload_i16");
}
extern "C" int viso_batch_upload_i16_async(viso_batch* b, int f0, int nf, const float* kp, const int16_t* desc16, const int32_t* n) {
    return upload_i16_impl(b, f0, nf, kp, desc16, n, false, "viso_batch_upload_i16_async");
}

extern "C" int viso_batch_upload_images_async(viso_batch* b, int f0, int nf, const uint8_t* images, int rows, int cols,
                                              const float* kp, const int32_t* n) {
    if (dead(b) || f0 < 0 || nf < 0 || f0 + nf > b->nf || rows <= 0 || cols <= 0 || (nf && !images) || ((kp == nullptr) != (n == nullptr))) {
        viso_set_error("viso_batch_upload_images_async: bad argument");
        return VISO_ERR_ARG;
    }
    if (!b->images || rows != b->img_rows || cols != b->img_cols) {
        viso_set_error("viso_batch_upload_images_async: image buffers not allocated for %d x %d (call viso_batch_upload_images once first)", rows, cols);
        return VISO_ERR_ARG;
    }
    for (int i = 0; n && i < 2 * nf; ++i)
        if (n[i] < 0 || n[i] > b->cap) { viso_set_error("viso_batch_upload_images_async: n[%d]=%d exceeds cap %d", i, n[i], b->cap); return VISO_ERR_ARG; }
    if (nf == 0) return VISO_OK;
    int r;
    if ((r = enter(b)) < 0) return r;
    hipStream_t s = b->ctx->stream;
    const size_t per = (size_t)rows * cols, c = (size_t)b->cap;
    HIP_TRY(hipMemcpyAsync(b->images + (size_t)f0 * 2 * per, images, per * 2 * (size_t)nf, hipMemcpyHostToDevice, s));
    if (kp) {
        HIP_TRY(hipMemcpyAsync(b->kp + (size_t)f0 * 2 * c, kp, sizeof(float2) * (size_t)nf * 2 * c, hipMemcpyHostToDevice, s));
        return stage_n_async(b, f0, nf, n, s);
    }
    return VISO_OK;
}

// Diagnostics: the shift of the 8-bit planes the last run used (matcher variant 6; see VISO_R8_* in csrc/common.h).
extern "C" int viso_batch_get_row8_shift(viso_batch* b, int* shift) {
    if (dead(b) || !shift) { viso_set_error("viso_batch_get_row8_shift: bad argument"); return VISO_ERR_ARG; }
    *shift = b->r8last;
    return VISO_OK;
}

extern "C" int viso_batch_device_ptrs(viso_batch* b, void** kp, void** desc, void** n) {
    if (dead(b)) return VISO_ERR_ARG;
    if (kp) *kp = b->kp;
    if (desc) *desc = b->desc;
    if (n) *n = b->n;
    return VISO_OK;
}

extern "C" int viso_batch_set_params(viso_batch* b, const viso_match_params* stereo,
                                     const viso_match_params* temporal, const viso_param* p,
                                     uint64_t seed, uint64_t first_frame_index) {
    if (dead(b) || !stereo || !temporal || !p || p->ransac_iter < 0 || stereo->max_neighbors <= 0 || temporal->max_neighbors <= 0) {
        viso_set_error("viso_batch_set_params: bad argument");
        return VISO_ERR_ARG;
    }
    int r0;
    if ((r0 = enter(b)) < 0) return r0;
    // kernels of a run still in flight read the solver items rewritten below (null-stream copies do not order
    // against the context's non-blocking stream)
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    fill_match_params(&b->mp[0], stereo);
    fill_match_params(&b->mp[1], temporal);
    fill_solver_params(&b->sp, p);
    b->seed = seed; b->first_frame = first_frame_index;
    if (p->ransac_iter != b->iters || !b->tr_h) {
        free_solver_bufs(b);
        b->iters = p->ransac_iter;
        const size_t k = (size_t)b->nf * (size_t)(b->iters > 0 ? b->iters : 1);
        int r;
        if ((r = dalloc(&b->tr_h, k * 6)) < 0 || (r = dalloc(&b->ok_h, k)) < 0 || (r = dalloc(&b->cnt_h, k)) < 0 ||
            (r = dalloc(&b->hq, k + 2)) < 0 || (r = dalloc(&b->samp_h, k * 3)) < 0 ||
            (r = dalloc(&b->rot, (size_t)b->nf * viso_rot_bytes(b->iters))) < 0) return r;
        // frame 0 has no solve: its rows are never written, and viso_batch_get_hypotheses hands them out with the rest
        HIP_TRY(hipMemset(b->tr_h, 0, sizeof(double) * 6 * k));
        HIP_TRY(hipMemset(b->ok_h, 0, sizeof(int) * k));
        HIP_TRY(hipMemset(b->cnt_h, 0, sizeof(int) * k));
        HIP_TRY(hipMemset(b->hq, 0, sizeof(int) * 2));   // the list of undecided hypotheses starts empty; every chain leaves it empty
    }
    int r = build_solver_items(b);
    if (r < 0) return r;
    b->params_set = true;
    return VISO_OK;
}

extern "C" int viso_batch_kernel_timing(viso_batch* b, int enable) {
    if (dead(b)) return VISO_ERR_ARG;
    b->timing = enable != 0;
    return VISO_OK;
}

static int run_matcher_impl(viso_batch* b, bool from_images) {
    if (dead(b) || !b->params_set) { viso_set_error("viso_batch_run: parameters not set"); return VISO_ERR_ARG; }
    if (from_images && (!b->images || b->dlen != VISO_DESC_LEN)) {
        viso_set_error("viso_batch_run_images: no images uploaded (or descriptor length is not 121)");
        return VISO_ERR_ARG;
    }
    if (!from_images) {   // f32 rows and int16 rows share one device buffer: a run over frames of both families would
                          // reinterpret one of them (garbage matches, no error) -- refuse it
        bool f32 = false, i16 = false;
        for (signed char f : b->desc_family) { f32 = f32 || f == 1; i16 = i16 || f == 2; }
        if (f32 && i16) {
            viso_set_error("viso_batch_run: frames of this batch were uploaded through both viso_batch_upload (f32 rows) and "
                           "viso_batch_upload_i16 (int16 rows); upload all frames through one family");
            return VISO_ERR_ARG;
        }
    }
    int r;
    if ((r = enter(b)) < 0) return r;
    hipStream_t s = b->ctx->stream;
    const int with_sums = pack_extras(b->ctx->matcher_variant, b->dlen);   // block sums / 8-bit planes: what the selected temporal kernel reads
    // the run's counters (scored, ovf_cnt, bad_img, bad_any) are zeroed by the first kernel of the run, not by a memset
    // the shift of this run's 8-bit planes (variant 6): forced, or the smallest that clamps at most one element pair in 256
    // of the sample the last counting run took (frames of a sequence look alike; a resident batch sees its own data
    // again).  Counting runs: the first, then every VISO_R8_EVERY-th; their counts come back asynchronously (no wait: a
    // copy that has not landed yet is looked at by a later run).  Any shift gives the same results: this is speed only
    int* r8cnt = nullptr;
    if (with_sums & VISO_PACK_ROWS8) {
        if (b->r8pending && hipEventQuery(b->r8ev) == hipSuccess) {
            b->r8pending = false;
            const unsigned long long rows = (unsigned)b->r8pin[VISO_R8_ROWS], lim = rows * 64ull / 256ull;
            if (rows) b->r8shift = (unsigned)b->r8pin[VISO_R8_C128] <= lim ? 0 : (unsigned)b->r8pin[VISO_R8_C128 + 1] <= lim ? 1 : (unsigned)b->r8pin[VISO_R8_C128 + 2] <= lim ? 2 : 3;
        }
        if (!b->r8pending && b->ctx->row8_force < 0 && b->r8runs % VISO_R8_EVERY == 0) r8cnt = b->r8cnt;
        ++b->r8runs;
        b->r8last = b->ctx->row8_force >= 0 ? b->ctx->row8_force : b->r8shift;
    }
    const int r8s = b->r8last;
    if ((r = launch_sort_kp(s, b->views, b->nf * 2, b->cap, reinterpret_cast<uint32_t*>(b->scored), (int)(b->zeroed_bytes / 4), r8cnt)) < 0) return r;
    if (from_images) {   // Sobel windows straight into packed rows (never bad: integers in [-1020,1020])
        if ((r = launch_extract_pack(s, b->views, b->nf * 2, b->cap, b->images, b->img_rows, b->img_cols, with_sums, r8s, r8cnt)) < 0) return r;
    } else {
        if (b->desc_i16) r = launch_pack_i16(s, b->views, b->nf * 2, b->cap, b->dlen, reinterpret_cast<const int16_t*>(b->desc), with_sums, r8s, r8cnt);
        else r = launch_pack(s, b->views, b->nf * 2, b->cap, b->dlen, b->bad_img, b->bad_any, with_sums, r8s, r8cnt);
        if (r < 0) return r;
    }
    if (r8cnt) {   // the sample's counts on their way to the host
        HIP_TRY(hipMemcpyAsync(b->r8pin, b->r8cnt, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(b->r8ev, s));
        b->r8pending = true;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (b->timing) {
        if (b->events.size() < VISO_EVENT_POOL) {
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            b->events.push_back({e0, e1});
        } else {   // pool full: fold the oldest pair into the running sum (waits for that run) and reuse it
            auto& e = b->events[b->ev_next];
            float ms = 0;
            HIP_TRY(hipEventSynchronize(e.second));
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { b->ev_ms_sum += ms; ++b->ev_n; }
            e0 = e.first; e1 = e.second;
            b->ev_next = (b->ev_next + 1) % VISO_EVENT_POOL;
        }
    }
    if ((r = launch_match_timed(s, b->probs, b->n_probs, b->cap, b->dlen, b->mp, b->bad_any, e0, e1, 1, b->ctx->matcher_variant, b->ovf_q, b->ovf_cnt, r8s, from_images ? 0 : 1)) < 0) return r;
    if ((r = launch_sort(s, b->probs, b->n_probs, b->cap)) < 0) return r;
    if (b->stamps) HIP_TRY(hipEventRecord(b->ev_stamp[2], s));   // re-recorded behind the solver by run_rest
    return VISO_OK;
}

// Time stamps of a run: which = 0 before the run's uploads, 1 after them (both on the context's stream); the end of
// the run is stamped by viso_batch_run* itself once stamping is on (i.e. after the first viso_batch_stamp call).
extern "C" int viso_batch_stamp(viso_batch* b, int which) {
    if (dead(b) || which < 0 || which > 1) { viso_set_error("viso_batch_stamp: bad argument"); return VISO_ERR_ARG; }
    int r;
    if ((r = enter(b)) < 0) return r;
    for (int k = 0; k < 3; ++k)
        if (!b->ev_stamp[k]) HIP_TRY(hipEventCreate(&b->ev_stamp[k]));
    if (!b->stamps) {   // all three recorded once, so that viso_batch_stamp_ms never meets an unrecorded event
        for (int k = 0; k < 3; ++k) HIP_TRY(hipEventRecord(b->ev_stamp[k], b->ctx->stream));
        b->stamps = true;
    }
    HIP_TRY(hipEventRecord(b->ev_stamp[which], b->ctx->stream));
    return VISO_OK;
}

// Waits for the batch; ms[0] = stamp 0 -> stamp 1 (the uploads), ms[1] = stamp 1 -> behind the run's last kernel.
extern "C" int viso_batch_stamp_ms(viso_batch* b, double ms[2]) {
    if (dead(b) || !ms || !b->stamps) { viso_set_error("viso_batch_stamp_ms: no stamps taken"); return VISO_ERR_ARG; }
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    float a = 0, c = 0;
    HIP_TRY(hipEventElapsedTime(&a, b->ev_stamp[0], b->ev_stamp[1]));
    HIP_TRY(hipEventElapsedTime(&c, b->ev_stamp[1], b->ev_stamp[2]));
    ms[0] = a; ms[1] = c;
    return VISO_OK;
}

extern "C" int viso_batch_run_matcher(viso_batch* b) { return run_matcher_impl(b, false); }

static int run_rest(viso_batch* b);

extern "C" int viso_batch_run(viso_batch* b) {
    int r = run_matcher_impl(b, false);
    if (r < 0) return r;
    return run_rest(b);
}

extern "C" int viso_batch_run_images(viso_batch* b, int matcher_only) {
    int r = run_matcher_impl(b, true);
    if (r < 0 || matcher_only) return r;
    return run_rest(b);
}

extern "C" int viso_batch_upload_images(viso_batch* b, int f0, int nf, const uint8_t* images, int rows, int cols,
                                        const float* kp, const int32_t* n) {
    if (dead(b) || f0 < 0 || nf < 0 || f0 + nf > b->nf || rows <= 0 || cols <= 0 || (nf && !images) || ((kp == nullptr) != (n == nullptr))) {
        viso_set_error("viso_batch_upload_images: bad argument");
        return VISO_ERR_ARG;
    }
    for (int i = 0; n && i < 2 * nf; ++i)
        if (n[i] < 0 || n[i] > b->cap) { viso_set_error("viso_batch_upload_images: n[%d]=%d exceeds cap %d", i, n[i], b->cap); return VISO_ERR_ARG; }
    int r0;
    if ((r0 = enter(b)) < 0) return r0;
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));   // a run in flight may still read what the (null-stream) copies below rewrite
    if (b->images && (rows != b->img_rows || cols != b->img_cols)) {
        HIP_TRY(hipFree(b->images));
        b->images = nullptr;
    }
    const size_t per = (size_t)rows * cols;
    if (!b->images) {
        HIP_TRY(hipMalloc((void**)&b->images, per * 2 * (size_t)b->nf));
        b->img_rows = rows; b->img_cols = cols;
    }
    if (nf == 0) return VISO_OK;
    const size_t c = (size_t)b->cap;
    HIP_TRY(hipMemcpy(b->images + (size_t)f0 * 2 * per, images, per * 2 * (size_t)nf, hipMemcpyHostToDevice));
    if (kp) {   // keypoints may instead come from viso_batch_detect
        HIP_TRY(hipMemcpy(b->kp + (size_t)f0 * 2 * c, kp, sizeof(float2) * (size_t)nf * 2 * c, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->n + (size_t)f0 * 2, n, sizeof(int) * (size_t)nf * 2, hipMemcpyHostToDevice));
    }
    return VISO_OK;
}

// HarrisBinnedFeatureDetector on every uploaded image (src/viso.cpp:1226-1227): fills the batch's
// keypoint arrays and counts on the device; viso_batch_run_images then extracts descriptors there too.
extern "C" int viso_batch_detect(viso_batch* b, int n_features, int nbinx, int nbiny, double k) {
    if (dead(b) || !b->images) { viso_set_error("viso_batch_detect: no images uploaded"); return VISO_ERR_ARG; }
    if (n_features < 0 || nbinx <= 0 || nbiny <= 0 || b->img_cols / nbinx <= 0 || b->img_rows / nbiny <= 0 ||
        (long long)nbinx * nbiny > 16384) {
        viso_set_error("viso_batch_detect: bad bin geometry");
        return VISO_ERR_ARG;
    }
    const int nbins = nbinx * nbiny, per = n_features / nbins;
    if ((long long)nbins * per > b->cap) { viso_set_error("viso_batch_detect: %d features exceed the batch capacity %d", nbins * per, b->cap); return VISO_ERR_ARG; }
    const int n_img = b->nf * 2;
    int r0;
    if ((r0 = enter(b)) < 0) return r0;
    hipStream_t s = b->ctx->stream;
    const bool fused = harris_fused_lds(b->img_rows, b->img_cols, nbinx, nbiny, per) != 0;   // no response image then
    if (!fused && !b->h_resp) HIP_TRY(hipMalloc((void**)&b->h_resp, sizeof(float) * (size_t)n_img * b->img_rows * b->img_cols));
    const size_t slots = (size_t)nbins * (per > 0 ? per : 1);
    if (slots > b->h_slots) {
        HIP_TRY(hipStreamSynchronize(s));
        if (b->h_tmp_kp) HIP_TRY(hipFree(b->h_tmp_kp));
        if (b->h_tmp_resp) HIP_TRY(hipFree(b->h_tmp_resp));
        if (b->h_cnt) HIP_TRY(hipFree(b->h_cnt));
        b->h_tmp_kp = nullptr; b->h_tmp_resp = nullptr; b->h_cnt = nullptr;
        HIP_TRY(hipMalloc((void**)&b->h_tmp_kp, sizeof(float2) * slots * n_img));
        HIP_TRY(hipMalloc((void**)&b->h_tmp_resp, sizeof(float) * slots * n_img));
        HIP_TRY(hipMalloc((void**)&b->h_cnt, sizeof(int) * (size_t)16384 * n_img));
        b->h_slots = slots;
    }
    int r;
    if (per == 0) { HIP_TRY(hipMemsetAsync(b->n, 0, sizeof(int) * (size_t)n_img, s)); return VISO_OK; }
    if (fused) {
        const size_t pb = harris_strip_bytes(n_img, b->img_rows, b->img_cols, nbinx, nbiny, per);   // 0: the wave-per-bin kernel
        if (pb > b->h_part_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            if (b->h_part) HIP_TRY(hipFree(b->h_part));
            b->h_part = nullptr; b->h_part_bytes = 0;
            HIP_TRY(hipMalloc(&b->h_part, pb));
            b->h_part_bytes = pb;
        }
        return launch_harris_detect(s, b->images, n_img, b->img_rows, b->img_cols, n_features, nbinx, nbiny, k, b->h_tmp_kp,
                                    b->h_tmp_resp, b->h_cnt, b->kp, nullptr, b->n, b->cap, (size_t)b->cap, pb ? b->h_part : nullptr);
    }
    if ((r = launch_harris_response(s, b->images, n_img, b->img_rows, b->img_cols, k, b->h_resp)) < 0) return r;
    return launch_harris_bins(s, b->h_resp, n_img, b->img_rows, b->img_cols, n_features, nbinx, nbiny, b->h_tmp_kp,
                              b->h_tmp_resp, b->h_cnt, b->kp, nullptr, b->n, b->cap, (size_t)b->cap);
}

// Keypoints of frame t, image side (after viso_batch_detect or an upload).
extern "C" int viso_batch_get_keypoints(viso_batch* b, int t, int side, float* kp, int* n_out) {
    if (dead(b) || t < 0 || t >= b->nf || side < 0 || side > 1 || !n_out) { viso_set_error("viso_batch_get_keypoints: bad argument"); return VISO_ERR_ARG; }
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    int n = 0;
    HIP_TRY(hipMemcpy(&n, b->n + (size_t)t * 2 + side, sizeof(int), hipMemcpyDeviceToHost));
    if (n > 0 && kp) HIP_TRY(hipMemcpy(kp, b->kp + ((size_t)t * 2 + side) * b->cap, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost));
    *n_out = n;
    return VISO_OK;
}

static int run_rest(viso_batch* b) {
    int r;
    if ((r = enter(b)) < 0) return r;
    hipStream_t s = b->ctx->stream;
    hipStream_t ss = b->solver_stream ? b->solver_stream : s;
    // the circle join rewrites what the previous run's RANSAC reads (x_c, Xp_c, mc)
    if (ss != s && b->ransac_pending) HIP_TRY(hipStreamWaitEvent(s, b->ev_ransac, 0));
    if (b->nf > 1) {
        if ((r = launch_circle_join(s, b->join, b->nf - 1, b->sp)) < 0) return r;         // :1245-1247 (the rows it joins), :1282, 1292-1305
        if (ss != s) {
            HIP_TRY(hipEventRecord(b->ev_join, s));
            HIP_TRY(hipStreamWaitEvent(ss, b->ev_join, 0));
        }
    }
    // vector<double> tr(6,0), :1312: ransac_refit_kernel writes the zeros itself where no solve succeeds
    if (b->nf > 1) {
        if ((r = launch_ransac(ss, b->sitems, b->nf - 1, b->iters, b->seed, b->sp, b->hq, b->ctx->gn_split, b->cap)) < 0) return r;   // :1313
        // the run's poses, flags and inlier counts into the pinned mirror (the kernel writes over PCIe; viso_batch_get_poses
        // waits for the streams and reads host memory)
        if ((r = plain_blit(ss, b->tr, b->pose_pin, b->pose_bytes / 4)) < 0) return r;
    }
    if (ss != s) {
        HIP_TRY(hipEventRecord(b->ev_ransac, ss));
        b->ransac_pending = true;
    }
    if (b->stamps) HIP_TRY(hipEventRecord(b->ev_stamp[2], ss));
    return VISO_OK;
}

extern "C" int viso_batch_kernel_ms(viso_batch* b, double* matcher_ms_avg, int* n_launches) {
    if (dead(b)) return VISO_ERR_ARG;
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    double tot = b->ev_ms_sum;
    int n = b->ev_n;
    for (auto& e : b->events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { tot += ms; ++n; }   // folded pairs were re-recorded since
        hipEventDestroy(e.first); hipEventDestroy(e.second);
    }
    b->events.clear();
    b->ev_next = 0; b->ev_ms_sum = 0; b->ev_n = 0;
    if (matcher_ms_avg) *matcher_ms_avg = n ? tot / n : 0.0;
    if (n_launches) *n_launches = n;
    return VISO_OK;
}

static bool slot_ok(viso_batch* b, int which, int t) { return !dead(b) && which >= 0 && which < 3 && t >= 0 && t < b->nf; }

extern "C" int viso_batch_get_matches(viso_batch* b, int which, int t, int32_t* out_match, int* out_n) {
    if (!slot_ok(b, which, t) || !out_n) { viso_set_error("viso_batch_get_matches: bad argument"); return VISO_ERR_ARG; }
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    const size_t o = (size_t)which * b->nf + t;
    int m = 0;
    HIP_TRY(hipMemcpy(&m, b->m_cnt + o, sizeof(int), hipMemcpyDeviceToHost));
    if (m > 0 && out_match) HIP_TRY(hipMemcpy(out_match, b->sorted + o * b->cap * 3, sizeof(int) * 3 * (size_t)m, hipMemcpyDeviceToHost));
    *out_n = m;
    return VISO_OK;
}

extern "C" int viso_batch_get_circle(viso_batch* b, int t, int32_t* circ, int32_t* pcl, int* out_n) {
    if (!slot_ok(b, 0, t) || !out_n) { viso_set_error("viso_batch_get_circle: bad argument"); return VISO_ERR_ARG; }
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    int m = 0;
    HIP_TRY(hipMemcpy(&m, b->mc + t, sizeof(int), hipMemcpyDeviceToHost));
    if (m > 0 && circ) HIP_TRY(hipMemcpy(circ, b->circ + (size_t)t * b->cap * 4, sizeof(int) * 4 * (size_t)m, hipMemcpyDeviceToHost));
    if (m > 0 && pcl) HIP_TRY(hipMemcpy(pcl, b->pcl + (size_t)t * b->cap * 2, sizeof(int) * 2 * (size_t)m, hipMemcpyDeviceToHost));
    *out_n = m;
    return VISO_OK;
}

extern "C" int viso_batch_get_pose(viso_batch* b, int t, double tr[6], int* ok, int32_t* inliers, int* n_inl) {
    if (!slot_ok(b, 0, t)) { viso_set_error("viso_batch_get_pose: bad argument"); return VISO_ERR_ARG; }
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    int o = 0, n = 0;
    const size_t nf = (size_t)b->nf;
    if (tr) memcpy(tr, b->pose_pin + sizeof(double) * 6 * (size_t)t, sizeof(double) * 6);
    memcpy(&o, b->pose_pin + sizeof(double) * 6 * nf + sizeof(int) * (size_t)t, sizeof(int));
    memcpy(&n, b->pose_pin + sizeof(double) * 6 * nf + sizeof(int) * (nf + (size_t)t), sizeof(int));
    if (n > 0 && inliers) HIP_TRY(hipMemcpy(inliers, b->inl + (size_t)t * b->cap, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    if (ok) *ok = o;
    if (n_inl) *n_inl = n;
    return VISO_OK;
}

extern "C" int viso_batch_get_poses(viso_batch* b, double* tr, int32_t* ok, int32_t* n_inl) {
    if (dead(b)) return VISO_ERR_ARG;
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    // the pinned mirror of the device block (run_rest's last kernel; zeros before the first run, like the device's)
    const size_t nf = (size_t)b->nf;
    if (tr) memcpy(tr, b->pose_pin, sizeof(double) * 6 * nf);
    if (ok) memcpy(ok, b->pose_pin + sizeof(double) * 6 * nf, sizeof(int) * nf);
    if (n_inl) memcpy(n_inl, b->pose_pin + sizeof(double) * 6 * nf + sizeof(int) * nf, sizeof(int) * nf);
    return VISO_OK;
}

// The per-hypothesis state of the last run's RANSAC stage (test / diagnostics): tr_h [n_frames][iters][6], ok_h and
// cnt_h [n_frames][iters] (frame 0 unused), *n_undecided = hypotheses the lane-per-hypothesis kernel handed on.
extern "C" int viso_batch_get_hypotheses2(viso_batch* b, int iters_capacity, double* tr_h, int32_t* ok_h, int32_t* cnt_h,
                                          int32_t* n_undecided) {
    if (dead(b) || iters_capacity < (b->iters > 0 ? b->iters : 1)) {
        viso_set_error("viso_batch_get_hypotheses2: arrays hold %d hypotheses per frame, the batch has %d", iters_capacity, b ? b->iters : 0);
        return VISO_ERR_ARG;
    }
    const int iters = b->iters > 0 ? b->iters : 1;
    if (iters_capacity == iters) return viso_batch_get_hypotheses(b, tr_h, ok_h, cnt_h, n_undecided);
    // the caller's rows are longer than the batch's: frame by frame, at the caller's stride
    if (!b->tr_h) { viso_set_error("viso_batch_get_hypotheses2: no run yet"); return VISO_ERR_ARG; }
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    const size_t nf = (size_t)b->nf;
    if (tr_h) HIP_TRY(hipMemcpy2D(tr_h, sizeof(double) * 6 * (size_t)iters_capacity, b->tr_h, sizeof(double) * 6 * (size_t)iters,
                                  sizeof(double) * 6 * (size_t)iters, nf, hipMemcpyDeviceToHost));
    if (ok_h) HIP_TRY(hipMemcpy2D(ok_h, sizeof(int) * (size_t)iters_capacity, b->ok_h, sizeof(int) * (size_t)iters, sizeof(int) * (size_t)iters, nf,
                                  hipMemcpyDeviceToHost));
    if (cnt_h) HIP_TRY(hipMemcpy2D(cnt_h, sizeof(int) * (size_t)iters_capacity, b->cnt_h, sizeof(int) * (size_t)iters, sizeof(int) * (size_t)iters, nf,
                                   hipMemcpyDeviceToHost));
    if (n_undecided) HIP_TRY(hipMemcpy(n_undecided, b->hq + 1, sizeof(int), hipMemcpyDeviceToHost));   // [1]: the last chain's count (solver.hip)
    return VISO_OK;
}

extern "C" int viso_batch_get_hypotheses(viso_batch* b, double* tr_h, int32_t* ok_h, int32_t* cnt_h, int32_t* n_undecided) {
    if (dead(b) || !b->tr_h) { viso_set_error("viso_batch_get_hypotheses: no run yet"); return VISO_ERR_ARG; }
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    const size_t k = (size_t)b->nf * (size_t)(b->iters > 0 ? b->iters : 1);
    if (tr_h) HIP_TRY(hipMemcpy(tr_h, b->tr_h, sizeof(double) * 6 * k, hipMemcpyDeviceToHost));
    if (ok_h) HIP_TRY(hipMemcpy(ok_h, b->ok_h, sizeof(int) * k, hipMemcpyDeviceToHost));
    if (cnt_h) HIP_TRY(hipMemcpy(cnt_h, b->cnt_h, sizeof(int) * k, hipMemcpyDeviceToHost));
    if (n_undecided) HIP_TRY(hipMemcpy(n_undecided, b->hq + 1, sizeof(int), hipMemcpyDeviceToHost));   // [1]: the last chain's count (solver.hip)
    return VISO_OK;
}

// Which images the pack kernel flagged in the last run (descriptor values that are not integers in
// [-32768, 32767]): the problems reading them took the general (double) kernel, all others the u16 kernels.
extern "C" int viso_batch_get_general_path_flags(viso_batch* b, int32_t* flags) {
    if (dead(b) || !flags) return VISO_ERR_ARG;
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    HIP_TRY(hipMemcpy(flags, b->bad_img, sizeof(int) * 2 * (size_t)b->nf, hipMemcpyDeviceToHost));
    return VISO_OK;
}

// How many queries of the last run the tile kernels handed to match_overflow_kernel (more than K in-radius
// candidates, a candidate list that outgrew its LDS slot, an exact tie of the minimum): the data-dependent slow path.
extern "C" int viso_batch_get_overflow_count(viso_batch* b, int32_t* n) {
    if (dead(b) || !n) return VISO_ERR_ARG;
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    HIP_TRY(hipMemcpy(n, b->ovf_cnt, sizeof(int), hipMemcpyDeviceToHost));
    return VISO_OK;
}

extern "C" int viso_batch_get_counters(viso_batch* b, int64_t* scored, int64_t* m_out) {
    if (dead(b)) return VISO_ERR_ARG;
    { const int rs_ = batch_sync(b); if (rs_ < 0) return rs_; }
    const size_t k = 3 * (size_t)b->nf;
    if (scored) HIP_TRY(hipMemcpy(scored, b->scored, sizeof(int64_t) * k, hipMemcpyDeviceToHost));
    if (m_out) {
        std::vector<int> m(k);
        HIP_TRY(hipMemcpy(m.data(), b->m_cnt, sizeof(int) * k, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < k; ++i) m_out[i] = m[i];
    }
    return VISO_OK;
}
