// match_stereo.hip — the stereo call (epipolar gate on, src/viso.cpp:1240), ONE LANE PER QUERY.
//
// With the Sampson gate on, ~3 of a query's ~50 in-radius targets are ever scored.  match_batch_kernel<1> still
// paid a wave-wide scan of the whole +-radius column window per query and a wave-wide fp64 gate pass per round:
// ~200 vector instructions per query for 3 SADs.  Here a wave owns a tile of 64 x-adjacent queries, one per lane:
//
//   band    the tile's epipolar band (match_dev.h, epipolar_band): targets with |dy| > band are rejected by the
//           gate whatever their x, so they are never looked at; sqrt(2 thresh) for a rectified pair;
//   index   the tile's column window of the target image (keypoints only: 8 B each), bucket-sorted by y inside
//           the wave (LDS histogram with returning atomics, DPP scan): 64 buckets;
//   walk    every lane walks the few entries of the buckets its band touches: L1 radius + Q1 cut + band test,
//           then the exact gate (sampson_dev, the reference's arithmetic bit for bit) on what is left;
//   score   the survivors of all 64 queries form one flat pair list, scored 8 lanes per pair (two 16-B row gathers
//           for the target, two for the query, 8 x v_sad_u16, 3 DPP adds), SADs to LDS;
//   reduce  every lane folds its own few SADs into (min, second min with multiplicity, argmin, tie).
//
// The K cap (max_neighbors, applied by the reference BEFORE the gate, :181/:692) needs the number of in-radius
// targets: the y index gives an upper bound in O(1) (entries with |dy| <= radius); only if that bound exceeds K
// (dense clusters) is the exact count taken.  Irregular queries (more than K in radius, an exact SAD tie, more
// than ST_SLOTS survivors) go to match_overflow_kernel, tiles whose band is wide or unbounded (pairs that are not
// rectified) to match_batch_kernel<1>, which walks the full radius: the tile flag tells it which.
#include "common.h"
#include "match_dev.h"

#define ST_THREADS 256
#define ST_WAVES 4
#define ST_QPW 64            // queries per wave = per tile
#define ST_WCAP 384          // window keypoints per chunk (6 per lane; a uniform tile of 2000-keypoint images sees ~320)
#define ST_NBY 64            // y buckets
#define ST_SLOTS 8           // candidates per query and chunk that may reach the scorer
#define ST_PCAP (ST_QPW * ST_SLOTS)
#ifndef ST_NP
#define ST_NP 2              // passes of 8 pairs in flight
#endif
#define ST_PAD (8 * ST_NP)   // the scoring pipeline runs up to ST_NP passes of 8 pairs past the end
#define ST_BAND_MAX 6.0f     // wider bands: match_batch_kernel<1>

struct StWaveLds {
    float2 ykp[ST_WCAP];                 // window keypoints in y-bucket order
    uint16_t ypos[ST_WCAP];              // their window positions
    int hist[ST_NBY + 1];                // bucket counts, then bucket starts (hist[ST_NBY] = number of entries)
    uint16_t slot[ST_SLOTS][ST_QPW];     // per query: y-order indices of its candidates
    uint32_t flat[ST_PCAP + ST_PAD];     // query << 16 | window position; the scorer leaves SAD << 9 | window position in the
                                         // pair's own slot (SAD < 2^23, position < ST_WCAP <= 512): no second array
};
static_assert(ST_WCAP <= 512, "window positions share a word with the SAD");

template <int CTRL>
__device__ __forceinline__ uint32_t st_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// (wave-wide scans and extrema: viso_wave_scan / viso_wave_fext, common.h -- DPP operands, not ds_bpermute round trips: a tile is a
// latency chain and the kernel is latency bound)
__device__ __forceinline__ int st_scan_incl(int v, int) { return (int)viso_wave_scan((uint32_t)v); }

__device__ __forceinline__ int st_bucket(float y, float y0, float scale) {
    if (y != y) return ST_NBY - 1;
    const float f = floorf((y - y0) * scale);
    if (f != f) return 0;                       // inf * 0: never (int)NaN
    return f <= 0.f ? 0 : (f >= (float)(ST_NBY - 1) ? ST_NBY - 1 : (int)f);
}

#ifndef ST_CLK
#define ST_CLK(I) do {} while (0)   // (match_frame.hip, debug builds: time stamps of the first tile's phases)
#endif
// (match_frame.hip includes this file with ST_KERNEL_SIG / ST_BLOCK defined: see match_union8.hip)
#ifndef ST_KERNEL_SIG
#define ST_KERNEL_SIG __global__ __launch_bounds__(ST_THREADS) void match_stereo_kernel(BatchMatchArgs a)
#define ST_BLOCK blockIdx.x
#endif
ST_KERNEL_SIG {
    __shared__ __attribute__((aligned(16))) StWaveLds s_w[ST_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int prob, tile;
    {
        const int b = ST_BLOCK;
        const int xcd = b & 7, slot = b >> 3;
        const int g = slot / a.bpp;                 // a.bpp = blocks (of ST_WAVES tiles) per problem
        prob = ((g / a.gc) * a.gs + a.gf + g % a.gc) * 8 + xcd;
        tile = (slot % a.bpp) * ST_WAVES + wave;
        if (prob >= a.n_probs) return;
    }
    ST_CLK(0);
    const MatchProblem P = a.probs[prob];
    const MatchParamsDev& mp = a.mp[P.pidx];
    if (mp.epi == 0) return;                        // temporal problems: the other kernels
    // Everything that depends on the problem alone is asked for TOGETHER, ahead of the branches that need only part of it: a
    // tile is a chain of dependent loads (problem -> counts -> query -> bucket starts -> window), the kernel is latency bound,
    // and loads the compiler may not move across an early return each cost a round trip of their own (eight of them before
    // the window arrived; five now).  (xinfo, rank[0] exist for any image, also one without keypoints: arrays of >= 1 entry.)
    const int badq = *P.q.bad, badt = *P.t.bad;
    const int n1 = *P.q.n, n2 = *P.t.n;
    const float tx0 = P.t.xinfo[0], tscale = P.t.xinfo[1];
    const int trank0 = P.t.rank[0];
    if ((badq | badt) != 0) return;                 // non-integer descriptors: the general kernel does this problem
    const int q0 = tile * ST_QPW;
    if (q0 >= n1) return;
    const int q1 = min(q0 + ST_QPW, n1);
    StWaveLds& L = s_w[wave];
    // ---- the lane's query
    const int j = q0 + lane;
    const bool live = j < q1;
    const float2 qv = live ? P.q.skp[j] : make_float2(__builtin_nanf(""), __builtin_nanf(""));
    const int orig = live ? P.q.sidx[j] : -1;
    float2 kp0 = make_float2(0.f, 0.f);             // the target image's first keypoint (Q1), with the queries: it waits for rank[0] only
    if (n2 > 0) kp0 = P.t.skp[trank0];
    const float radius = mp.radius;
    const int K = mp.K;
    // ---- tile: x range -> window, y range -> band
    const float xa = viso_wave_fext<false>(qv.x), xb = viso_wave_fext<true>(qv.x), ya = viso_wave_fext<false>(qv.y), yb = viso_wave_fext<true>(qv.y);
    const bool ynan = __any(live && qv.y != qv.y);
    float band = __builtin_huge_valf();
    if (!ynan) band = epipolar_band(mp.F, mp.sampson_thresh, xa, xb, ya, yb, radius);
    band = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(band)));
    const bool mine = band <= ST_BAND_MAX;
    if (lane == 0) {
        P.tile_flag[tile] = mine ? 0 : 1;   // 1: match_batch_kernel<1> walks the full radius for this tile
        if (!mine) atomicAdd(&a.bad[1], 1);
    }
    if (!mine) return;
    int lo = 0, W = 0;
    if (n2 > 0 && xa == xa && radius >= 0.f) {
        const float slack = (fabsf(xa) + fabsf(xb) + fabsf(radius)) * 1e-6f + 1e-6f;
        lo = P.t.bstart[bucket_of(xa - radius - slack, tx0, tscale)];
        W = P.t.bstart[bucket_of(xb + radius + slack, tx0, tscale) + 1] - lo;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    W = __builtin_amdgcn_readfirstlane(W);
    // Q1 (src/viso.cpp:693): (d <= radius && d < d0cut) as one unsigned compare of the bits of d (see match_union.hip)
    uint32_t thr = __float_as_uint(radius) + 1u;
    if (n2 > 0) {
        const float d0 = l1_kp(qv.x, qv.y, kp0);
        if (d0 <= radius) thr = __float_as_uint(d0);
    }
    if (!live) thr = 0u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* grow_t;
    typedef const __attribute__((address_space(1))) char* gbytes_t;
    const gbytes_t trows = (gbytes_t)reinterpret_cast<const char*>(P.t.rows);
    const gbytes_t qrows = (gbytes_t)reinterpret_cast<const char*>(P.q.rows) + (size_t)q0 * (VISO_ROW * 2);
    const int g8 = lane >> 3, sub = lane & 7;
    // per-query state, carried over the chunks
    uint32_t d1 = 0xffffffffu, d2 = 0xffffffffu, bw = 0, tie = 0;
    int cnt_ub = 0, nscored = 0;

    ST_CLK(1);
    for (int cb = 0; cb < W; cb += ST_WCAP) {
        const int cw = min(W - cb, ST_WCAP);
        // ---- y index of the chunk: bucket sort inside the wave
        float2 e_kp[ST_WCAP / 64];
        int e_b[ST_WCAP / 64], e_r[ST_WCAP / 64];
        float y0 = __builtin_huge_valf(), y1 = -__builtin_huge_valf();
#pragma unroll
        for (int i = 0; i < ST_WCAP / 64; ++i) {
            const int w = lane + 64 * i;
            e_kp[i] = w < cw ? P.t.skp[lo + cb + w] : make_float2(__builtin_nanf(""), __builtin_nanf(""));
            y0 = fminf(y0, e_kp[i].y); y1 = fmaxf(y1, e_kp[i].y);
        }
        y0 = viso_wave_fext<false>(y0); y1 = viso_wave_fext<true>(y1);
        float yscale = 0.f;
        if (y1 > y0) yscale = (float)ST_NBY / (y1 - y0);
        if (!(yscale > 0.f) || !(yscale < 3.0e38f)) yscale = 0.f;
        if (!(y0 == y0) || !(fabsf(y0) < 3.0e38f)) { y0 = 0.f; yscale = 0.f; }
        L.hist[lane] = 0;
        if (lane == 0) L.hist[ST_NBY] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < ST_WCAP / 64; ++i) {
            e_b[i] = st_bucket(e_kp[i].y, y0, yscale);
            e_r[i] = 0;
            if (lane + 64 * i < cw) e_r[i] = atomicAdd(&L.hist[e_b[i]], 1);   // returning LDS atomic: rank inside the bucket
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int h = L.hist[lane];
            const int incl = st_scan_incl(h, lane);
            __builtin_amdgcn_wave_barrier();
            L.hist[lane] = incl - h;           // bucket start
            if (lane == 63) L.hist[ST_NBY] = incl;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < ST_WCAP / 64; ++i) {
            if (lane + 64 * i < cw) {
                const int p = L.hist[e_b[i]] + e_r[i];
                L.ykp[p] = e_kp[i];
                L.ypos[p] = (uint16_t)(lane + 64 * i);
            }
        }
        __builtin_amdgcn_wave_barrier();
        ST_CLK(2);
        // ---- walk: the buckets the lane's band touches.  A lane keeps ST_SLOTS candidates per pass; where keypoints
        // are dense and a band holds more, the walk / gate / score / reduce sequence below repeats for the next
        // ST_SLOTS of them (`skip`): sparse data makes one pass
        // the bucket map is monotone in y; `ys` covers the rounding of the float differences the tests below take
        const float ys = (fabsf(qv.y) + fabsf(radius)) * 1e-6f + 1e-6f;
        const int s0 = L.hist[st_bucket(qv.y - band - ys, y0, yscale)];
        const int s1 = L.hist[st_bucket(qv.y + band + ys, y0, yscale) + 1];
        // upper bound of the in-radius count (K cap): entries with |dy| <= radius
        cnt_ub += L.hist[st_bucket(qv.y + radius + ys, y0, yscale) + 1] - L.hist[st_bucket(qv.y - radius - ys, y0, yscale)];
        for (int skip = 0;; skip += ST_SLOTS) {
        int n = 0, seen = 0;
        for (int i = s0; __any(i < s1); ++i) {
            if (i < s1) {
                const float2 t = L.ykp[i];
                const float dx = qv.x - t.x, dy = qv.y - t.y;
                const bool in = __float_as_uint(fabsf(dx) + fabsf(dy)) < thr && fabsf(dy) <= band;
                if (in) {
                    if (seen >= skip && n < ST_SLOTS) { L.slot[n][lane] = (uint16_t)i; ++n; }
                    ++seen;
                }
            }
        }
        ST_CLK(3);
        // ---- the exact gate on what is left (src/viso.cpp:695-701), compaction in place
        int n2g = 0;
        // two candidates per step: their fp64 chains (a division each) interleave
        for (int s = 0; __any(s < n); s += 2) {
            const bool v0 = s < n, v1 = s + 1 < n;
            const int i0 = v0 ? L.slot[s][lane] : 0, i1 = v1 ? L.slot[s + 1][lane] : 0;
            const float2 t0 = L.ykp[i0], t1 = L.ykp[i1];
            const double sd0 = sampson_dev(mp.F, qv.x, qv.y, t0.x, t0.y), sd1 = sampson_dev(mp.F, qv.x, qv.y, t1.x, t1.y);
            if (v0 && isfinite(sd0) && !(sd0 > mp.sampson_thresh)) { L.slot[n2g][lane] = (uint16_t)i0; ++n2g; }
            if (v1 && isfinite(sd1) && !(sd1 > mp.sampson_thresh)) { L.slot[n2g][lane] = (uint16_t)i1; ++n2g; }
        }
        ST_CLK(4);
        // ---- flat pair list of the tile
        const int incl = st_scan_incl(n2g, lane);
        const int base = incl - n2g;
        const int ntot = __builtin_amdgcn_readlane(incl, 63);
        for (int s = 0; s < n2g; ++s) L.flat[base + s] = ((uint32_t)lane << 16) | (uint32_t)L.ypos[L.slot[s][lane]];
        __builtin_amdgcn_wave_barrier();
        if (ntot > 0 && lane < ST_PAD) L.flat[ntot + lane] = L.flat[ntot - 1];
        __builtin_amdgcn_wave_barrier();
        // ---- score: 8 lanes per pair, 8 pairs per pass, two passes in flight
        {
            const int npass = (ntot + 7) >> 3;
            const gbytes_t wrows = trows + (size_t)(lo + cb) * (VISO_ROW * 2);
            u32x4 t0[ST_NP], t1[ST_NP], u0[ST_NP], u1[ST_NP];
            int dst[ST_NP];
            uint32_t wpos[ST_NP];
#define ST_ISSUE(SLOT, T)                                                                                  \
            do {                                                                                           \
                const int gi_ = (T) * 8 + g8;                                                              \
                const uint32_t e_ = L.flat[gi_];                                                           \
                dst[SLOT] = gi_;                                                                           \
                wpos[SLOT] = e_ & 0x1ffu;                                                                  \
                const grow_t tr_ = (grow_t)(wrows + (((e_ & 0xffffu) << 8) | (uint32_t)(sub << 4)));       \
                const grow_t qr_ = (grow_t)(qrows + (((e_ >> 16) << 8) | (uint32_t)(sub << 4)));           \
                t0[SLOT] = tr_[0]; t1[SLOT] = tr_[8];                                                      \
                u0[SLOT] = qr_[0]; u1[SLOT] = qr_[8];                                                      \
            } while (0)
#define ST_REDUCE(SLOT)                                                                                    \
            do {                                                                                           \
                uint32_t s_ = __builtin_amdgcn_sad_u16(t0[SLOT].x, u0[SLOT].x, 0u);                        \
                s_ = __builtin_amdgcn_sad_u16(t0[SLOT].y, u0[SLOT].y, s_);                                 \
                s_ = __builtin_amdgcn_sad_u16(t0[SLOT].z, u0[SLOT].z, s_);                                 \
                s_ = __builtin_amdgcn_sad_u16(t0[SLOT].w, u0[SLOT].w, s_);                                 \
                s_ = __builtin_amdgcn_sad_u16(t1[SLOT].x, u1[SLOT].x, s_);                                 \
                s_ = __builtin_amdgcn_sad_u16(t1[SLOT].y, u1[SLOT].y, s_);                                 \
                s_ = __builtin_amdgcn_sad_u16(t1[SLOT].z, u1[SLOT].z, s_);                                 \
                s_ = __builtin_amdgcn_sad_u16(t1[SLOT].w, u1[SLOT].w, s_);                                 \
                s_ += st_dpp<0xB1>(s_);                                                                    \
                s_ += st_dpp<0x4E>(s_);                                                                    \
                s_ += st_dpp<0x141>(s_);                                                                   \
                if (sub == 0) L.flat[dst[SLOT]] = (s_ << 9) | wpos[SLOT];   /* slots past ntot are scratch */ \
            } while (0)
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < ST_NP; ++p) ST_ISSUE(p, p);
            }
            int t = 0;
            for (; t + ST_NP < npass; t += ST_NP) {
#pragma unroll
                for (int p = 0; p < ST_NP; ++p) {
                    ST_REDUCE(p);
                    __builtin_amdgcn_sched_barrier(0);
                    ST_ISSUE(p, t + p + ST_NP);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < ST_NP; ++p) ST_REDUCE(p);
            }
#undef ST_REDUCE
#undef ST_ISSUE
        }
        __builtin_amdgcn_wave_barrier();
        ST_CLK(5);
        // ---- reduce: the lane's own SADs into its running order statistics
        for (int s = 0; s < n2g; ++s) {
            const uint32_t f = L.flat[base + s];
            const uint32_t v = f >> 9;
            const uint32_t w = (f & 0x1ffu) + (uint32_t)cb;
            const bool lt = v < d1, eq = v == d1;
            d2 = (v <= d1) ? d1 : min(d2, v);
            bw = lt ? w : bw;
            tie = lt ? 0u : (eq ? 1u : tie);
            d1 = min(d1, v);
        }
        nscored += n2g;
        __builtin_amdgcn_wave_barrier();
        if (!__any(seen > skip + ST_SLOTS)) break;
        }   // skip
    }
    ST_CLK(6);
    // ---- K cap: exact in-radius count where the bound does not settle it (dense clusters only)
    int cnt = cnt_ub;
    {
        unsigned long long need = __ballot(live && cnt_ub > K);
        while (need) {
            const int ql = __ffsll((long long)need) - 1;
            need &= need - 1;
            const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv.x), ql));
            const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv.y), ql));
            const uint32_t th = (uint32_t)__builtin_amdgcn_readlane((int)thr, ql);
            int c = 0;
            for (int base = 0; base < W; base += VISO_WAVE) {
                const int w = base + lane;
                bool in = false;
                if (w < W) in = __float_as_uint(l1_kp(qx, qy, P.t.skp[lo + w])) < th;
                c += __popcll(__ballot(in));
            }
            if (lane == ql) cnt = c;
        }
    }
    ST_CLK(7);
    // ---- results
    unsigned long long scored = 0;
    if (live) {
        const bool none = d1 == 0xffffffffu;
        if (cnt > K || (!none && tie)) {
            // more than K in radius / exact tie of the minimum (largest-key rule): overflow kernel
            P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, j);
        } else {
            bool accept = !none;
            int idx = -1;
            if (accept) {
                idx = P.t.sidx[lo + (int)bw];
                if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                    const double bd2 = d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)d2;
                    accept = (double)d1 < bd2 * mp.ratio;
                }
            }
            P.res[orig] = make_int2(accept ? idx : -1, (int)d1);
            scored = (unsigned long long)nscored;
        }
    }
    scored = viso_wave_sum63(scored);
    if (lane == 63 && scored) atomicAdd(P.scored, scored);
}

#ifndef ST_NO_LAUNCHER
int launch_match_stereo(hipStream_t s, const BatchMatchArgs& a64, int cap_max) {
    BatchMatchArgs a = a64;
    const int tiles = (cap_max + ST_QPW - 1) / ST_QPW;
    a.bpp = (tiles + ST_WAVES - 1) / ST_WAVES;
    const int groups = (a.n_probs + 7) / 8;
    long long blocks = (long long)groups * 8 * a.bpp;
    if (a.gs == 3) blocks = (long long)((groups + 2) / 3) * a.gc * 8 * a.bpp;
    if (blocks > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(match_stereo_kernel, dim3((unsigned)blocks), dim3(ST_THREADS), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_stereo_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
#endif   // ST_NO_LAUNCHER
