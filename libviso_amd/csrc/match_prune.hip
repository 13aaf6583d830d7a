// match_prune.hip — temporal (no epipolar gate) matcher: exact candidate pruning + cell-granular scoring.
//
// match_desc (reference src/viso.cpp:692-722) needs, per query, the smallest SAD over its gate-passing candidates, the
// candidate that attains it, whether it is attained twice, and ONE bit of the second smallest: whether
// best_d1 < best_d2 * ratio (:715).  The second smallest SAD itself is never output.  So a candidate t may be left
// unscored whenever a lower bound b(t) <= SAD(q, t) proves that t is neither the minimum, nor tied with it, nor able to
// fail that test.  With d* = SAD of any one scored candidate of q (d* >= best_d1):
//     b(t) > d*                              t is not the minimum and does not tie with it
//     (double)d* < (double)b(t) * ratio      fl(SAD(t) * ratio) >= fl(b(t) * ratio) > d* >= best_d1: t alone passes :715
// (the second line only when the ratio test is on).  Both are monotone in b, so they collapse into one integer
// threshold bmin(d*) per query: t is scored iff b(t) < bmin.  The bound is the L1 distance of the rows' four 32-element
// block sums (ImageView::sums, 8 B per keypoint, written by the pack kernels): 2 v_sad_u16 per (query, candidate)
// cell.  d* comes from the candidate with the smallest bound (the true match for ~3 of 4 queries that have one);
// after it, a query with a good match keeps 1-2 of its ~52 candidates, a query without one keeps all of them:
// ~15 SADs per query instead of 52 (synthetic 1241x376 frames; what real images keep depends on how distinctive the
// windows are, never on correctness).  `scored` still counts the gate-passing candidates (the reference's C).
//
//   tile     64 queries of one problem = one 256-thread workgroup; the +-radius column window of the target image:
//            keypoints bucket-sorted by y in LDS, block sums by window position in LDS           (as match_union.hip)
//   round    eight y-adjacent queries per wave, two rounds per wave
//   phase 1  scan of the y buckets the round's diamonds touch -> union list of rows with 8-bit membership masks,
//            per-query in-radius counts (K cap)                                                   (as match_union.hip)
//   phase 1b lane per union row: the eight bounds; per query the member row with the smallest one (transposing wave min)
//   phase 1c the eight probes are scored in one pass (lane group k: query k against its probe row) -> d*, bmin
//   phase 1d lane per union row: cells (query k, row) with member && (bound < bmin || probe), appended row-major to
//            the round's cell list (same-row cells are adjacent: their loads coalesce in the texture unit)
//   phase 2  rolling pipeline over the CELLS, 8 lanes per cell: the row's 2 x 16 B per lane from the XCD's L2, the
//            query's from the wave's LDS copy, 8 v_sad_u16, 3 DPP adds -> every lane of the group has the SAD; the
//            lane whose position in the group equals the cell's query folds it into its packed-key tracker
//            (key = SAD << 9 | cell index; m2 = med3, m1 = min)
//   phase 3  merge the 8 groups per query, ratio test in double, store; ties of the minimum, more than K in-radius
//            candidates and lists that outgrow their LDS slots -> match_overflow_kernel (literal rules)
// Same results as every other matcher kernel (tests/test_gpu_parity.py, test_gpu_fuzz.py, test_gpu_fullsize.py).
#include "common.h"
#include "match_dev.h"

#define MP_THREADS 256
#define MP_WAVES 4
#define MP_QPB 64          // queries per tile
#define MP_G 8             // queries per round
#ifndef MP_QSTRIDE
#define MP_QSTRIDE 64      // dwords per staged query row
#endif
#ifndef MP_WPE
#define MP_WPE 4           // lower bound for the register allocator (forcing 6 makes it spill; left alone it lands at 77 VGPRs = 6 waves)
#endif
#ifndef MP_NP
#define MP_NP 2            // passes in flight
#endif
#define MP_UCAP 448        // union rows per round (a list position fits the 9 low bits of the probe key)
#define MP_CCAP 480        // cells per round (a cell index fits the 9 low bits of a tracker key)
#define MP_CPAD (8 * MP_NP + 16)   // cell-list padding: the pipeline runs up to MP_NP passes of 8 cells past the end
#define MP_LCAP 576        // the wave's list buffer: union rows at [0, nu), then the cells at [nu, nu + ccnt) + padding
#define MP_KPCAP 512       // window keypoints staged in LDS
#define MP_NBY 64          // y buckets of the staged window

template <int CTRL>
__device__ __forceinline__ uint32_t mp_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// bits of |qx - tx| + |qy - ty| (cvflann::L1 order, see l1_kp); abs as source modifiers of the add
__device__ __forceinline__ uint32_t mp_l1_bits(float qx, float qy, float2 t) {
    const float dx = qx - t.x, dy = qy - t.y;
    float d;
    asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(d) : "v"(dx), "v"(dy));
    return __float_as_uint(d);
}

// packed-key order statistics: see match_union.hip (MuTrack)
struct MpTrack { uint32_t m1, m2; };
__device__ __forceinline__ void mp_update(MpTrack& t, uint32_t key) {
    uint32_t med;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(t.m1), "v"(t.m2), "v"(key));
    t.m2 = med;
    t.m1 = min(t.m1, key);
}
__device__ __forceinline__ void mp_merge(MpTrack& a, const MpTrack& b) {
    const uint32_t hi = max(a.m1, b.m1);
    a.m2 = min(hi, min(a.m2, b.m2));
    a.m1 = min(a.m1, b.m1);
}

__device__ __forceinline__ int mp_ybucket(float y, float y0, float scale) {   // monotone in y
    if (y != y) return MP_NBY - 1;
    const float f = floorf((y - y0) * scale);
    if (f != f) return 0;                       // inf * 0: never (int)NaN
    return f <= 0.f ? 0 : (f >= (float)(MP_NBY - 1) ? MP_NBY - 1 : (int)f);
}

// A VALID pruning threshold: every bound b >= the returned value satisfies b > dstar and, when the ratio test is on,
// (double)dstar < (double)b * ratio — the reference's own expression (:715) with b in place of best_d2 (monotone in b,
// so checking it at the returned value covers everything above).  The guess comes from float arithmetic with a margin,
// the check is the double expression itself; a guess that fails it (never seen) means "prune nothing".  Bounds never
// exceed 4 * 65535, so 0xffffffff prunes nothing.  rinv = 1 / ratio as a float (any value: only a guess).
__device__ __forceinline__ uint32_t mp_bmin(uint32_t dstar, int second, double ratio, float rinv) {
    if (!second) return dstar + 1u;
    const float g = (float)dstar * rinv * 1.000001f + 2.0f;
    if (!(g < 1.0e6f)) return 0xffffffffu;                  // also NaN (ratio <= 0, NaN): nothing passes :715 on its own
    const uint32_t b = max((uint32_t)g, dstar + 1u);
    return ((double)dstar < (double)b * ratio) ? b : 0xffffffffu;
}

__global__ __attribute__((amdgpu_waves_per_eu(MP_WPE, 8))) __launch_bounds__(MP_THREADS) void match_prune_kernel(BatchMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_list[MP_WAVES][MP_LCAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_qrow[MP_WAVES][MP_G][MP_QSTRIDE];
    __shared__ float2 s_ykp[MP_KPCAP];       // staged window keypoints in y-bucket order
    __shared__ uint16_t s_ypos[MP_KPCAP];    // their window positions
    __shared__ uint2 s_sum[MP_KPCAP];        // block sums by window position
    __shared__ int s_ys[MP_NBY + 1];         // bucket counts, then bucket starts
    __shared__ float s_xr[2];
    __shared__ int s_next;                   // next unclaimed round of the tile
    int prob, qblk;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        const int g = slot / a.bpp;
        prob = ((g / a.gc) * a.gs + a.gf + g % a.gc) * 8 + xcd;
        qblk = slot % a.bpp;
        if (prob >= a.n_probs) return;
    }
    const MatchProblem P = a.probs[prob];
    if ((*P.q.bad | *P.t.bad) != 0) return;   // non-integer descriptors: the general kernel does this problem
    const int n1 = *P.q.n, n2 = *P.t.n;
    const int q0 = qblk * MP_QPB;
    if (q0 >= n1) return;
    const int q1 = min(q0 + MP_QPB, n1);
    const MatchParamsDev& mp = a.mp[P.pidx];
    if (mp.epi != 0) return;   // stereo problems: match_stereo_kernel / match_batch_kernel<1>
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // ---- tile: x range (window) of its queries; their y order (round composition) comes from sort_kp_kernel (ImageView::qord)
    if (wave == 0) {
        const bool live = q0 + lane < q1;
        const float qx = live ? P.q.skp[q0 + lane].x : __builtin_nanf("");
        float mn = qx, mx = qx;
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, m));
            mx = fmaxf(mx, __shfl_xor(mx, m));
        }
        if (lane == 0) { s_xr[0] = mn; s_xr[1] = mx; }
    }
    if (threadIdx.x <= MP_NBY) s_ys[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_next = MP_WAVES;
    __syncthreads();
    int lo = 0, W = 0;
    {
        const float xa = s_xr[0], xb = s_xr[1];
        const float r = mp.radius;
        if (n2 > 0 && xa == xa && r >= 0.f) {
            const float slack = (fabsf(xa) + fabsf(xb) + fabsf(r)) * 1e-6f + 1e-6f;
            const float x0 = P.t.xinfo[0], scale = P.t.xinfo[1];
            lo = P.t.bstart[bucket_of(xa - r - slack, x0, scale)];
            W = P.t.bstart[bucket_of(xb + r + slack, x0, scale) + 1] - lo;
        }
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    W = __builtin_amdgcn_readfirstlane(W);
    const int wcap = min(W, MP_KPCAP);
    const int wpad = (wcap + 127) & ~127;   // NaN padded: the scan needs no bounds test
    // ---- y index of the staged window: bucket sort (histogram with returning LDS atomics, scan, scatter) over the
    // target image's y range, so that a round only scans the buckets its diamonds can touch
    float ty0 = P.t.xinfo[2];
    float yscale = 0.f;
    {
        const float ty1 = P.t.xinfo[3];
        if (ty1 > ty0) yscale = (float)MP_NBY / (ty1 - ty0);
        if (!(yscale > 0.f) || !(yscale < 3.0e38f)) yscale = 0.f;
    }
    static_assert(MP_KPCAP <= 2 * MP_THREADS, "two window entries per thread");
    float2 e_kp[2];
    int e_b[2], e_r[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = threadIdx.x + i * MP_THREADS;
        e_b[i] = 0; e_r[i] = 0;
        if (w < wcap) {
            e_kp[i] = P.t.skp[lo + w];
            s_sum[w] = P.t.sums[lo + w];
            e_b[i] = mp_ybucket(e_kp[i].y, ty0, yscale);
            e_r[i] = atomicAdd(&s_ys[e_b[i]], 1);
        }
    }
    float2 kp0 = make_float2(0.f, 0.f);
    const bool has0 = n2 > 0;
    if (has0) kp0 = P.t.skp[P.t.rank[0]];
    __syncthreads();
    if (wave == 0) {
        const int h = s_ys[lane];
        int incl = h;
#pragma unroll
        for (int d = 1; d < VISO_WAVE; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        s_ys[lane] = incl - h;
        if (lane == VISO_WAVE - 1) s_ys[MP_NBY] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = threadIdx.x + i * MP_THREADS;
        if (w < wcap) {
            const int p = s_ys[e_b[i]] + e_r[i];
            s_ykp[p] = e_kp[i];
            s_ypos[p] = (uint16_t)w;
        } else if (w < wpad) {
            s_ykp[w] = make_float2(__builtin_nanf(""), __builtin_nanf(""));   // entries [wcap, wpad): never in radius
            s_ypos[w] = 0;
        }
    }
    __syncthreads();
    const float radius = mp.radius;
    const int K = mp.K;
    const float rinv = mp.ratio > 0.0 ? 1.0f / (float)mp.ratio : __builtin_nanf("");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* grow_t;
    typedef const __attribute__((address_space(1))) char* gbytes_t;
    // window base (scalar) + 32-bit byte offset per lane: (window position << 8) | (sub << 4)
    const gbytes_t wrows = (gbytes_t)reinterpret_cast<const char*>(P.t.rows) + (size_t)lo * (VISO_ROW * 2);
    uint32_t* ul = s_list[wave];
    const int sub = lane & 7;
    const int g8 = lane >> 3;
    const int half = lane >> 5;          // phase 1: lanes 0..31 test queries 0..3 of the round, lanes 32..63 queries 4..7
    unsigned long long scored = 0;
    constexpr int ROUNDS = MP_QPB / MP_G;   // 8 rounds of 8 y-adjacent queries per tile: a wave starts with round `wave`,
                                            // then claims the next free one (rounds differ a lot in their number of cells:
                                            // a fixed split leaves the waves of a workgroup waiting for the slowest)

    // query data one round ahead: lane l carries local index / keypoint / original index / block sums of query (l & 7)
    int pli;
    float2 pq;
    int po;
    uint2 psum;
#define MP_PREFETCH(R)                                                                                    \
    do {                                                                                                  \
        const int base_ = (R) * MP_G;                                                                     \
        pli = (int)P.q.qord[q0 + base_ + (lane & (MP_G - 1))];                                            \
        const int j_ = q0 + pli;                                                                          \
        const int jc_ = min(j_, q1 - 1);                                                                  \
        pq = P.q.skp[jc_];                                                                                \
        psum = P.q.sums[jc_];                                                                             \
        po = j_ < q1 ? P.q.sidx[jc_] : -1;                                                                \
    } while (0)
    auto claim = [&]() {   // the next free round of the tile (wave uniform)
        int nx = 0;
        if (lane == 0) nx = atomicAdd(&s_next, 1);
        return __builtin_amdgcn_readfirstlane(nx);
    };
    for (int r = wave; r < ROUNDS; r = claim()) {
        MP_PREFETCH(r);
        // ---------------- round setup.  Lane l holds query (l & 7); phase 1 wants, per lane, the four queries of its
        // half: query 4 * half + i sits in lane 36 * half + i
        const int my_orig = po, my_j = q0 + pli;   // lanes 0..7: the round's queries, for phase 3
        if (!__any(po >= 0)) continue;   // wave uniform
        // d = |dx| + |dy| is +0, positive or NaN: its bit pattern orders like the value and NaNs are above +inf,
        // so (d <= radius && d < d0cut) is one unsigned compare against bits(d0) (target 0 in radius: Q1,
        // src/viso.cpp:693) or bits(radius) + 1.  Dead slots (past the tile) get 0: nothing passes.
        uint32_t tq = __float_as_uint(radius) + 1u;
        if (has0) {
            const float d0 = l1_kp(pq.x, pq.y, kp0);
            if (d0 <= radius) tq = __float_as_uint(d0);
        }
        if (po < 0) tq = 0u;
        float qx[4], qy[4];
        uint32_t thr[4];
        float ymn = pq.y, ymx = pq.y;   // y extent of the round's queries (all lanes hold one of them)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xa_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.x), i));
            const float xb_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.x), 4 + i));
            const float ya_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.y), i));
            const float yb_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.y), 4 + i));
            const uint32_t ta_ = (uint32_t)__builtin_amdgcn_readlane((int)tq, i);
            const uint32_t tb_ = (uint32_t)__builtin_amdgcn_readlane((int)tq, 4 + i);
            qx[i] = half ? xb_ : xa_;
            qy[i] = half ? yb_ : ya_;
            thr[i] = half ? tb_ : ta_;
            ymn = fminf(ymn, fminf(ya_, yb_));
            ymx = fmaxf(ymx, fmaxf(ya_, yb_));
        }
        // the eight queries' block sums as scalars
        uint32_t qsx[MP_G], qsy[MP_G];
#pragma unroll
        for (int k = 0; k < MP_G; ++k) {
            qsx[k] = (uint32_t)__builtin_amdgcn_readlane((int)psum.x, k);
            qsy[k] = (uint32_t)__builtin_amdgcn_readlane((int)psum.y, k);
        }
        // the eight query rows: one word per lane and row from global memory (the loads land during the scan), then LDS
        uint32_t qw[MP_G];
#pragma unroll
        for (int k = 0; k < MP_G; ++k) {
            const int jk = q0 + __builtin_amdgcn_readlane(pli, k);
            qw[k] = ((const __attribute__((address_space(1))) uint32_t*)reinterpret_cast<const uint32_t*>(P.q.rows))[(size_t)min(jk, q1 - 1) * (VISO_ROW / 2) + lane];
        }
        // ---------------- phase 1: one scan over the y buckets the round's diamonds touch, 32 targets per step: both
        // halves read the same 32 entries, each tests its four queries -> 8-bit membership mask (bit 7 - k = query k),
        // targets with a non-zero mask go to the union list.  entry = mask << 24 | window position << 8
        int ucnt = 0;
        {
            const float ys = (fabsf(ymn) + fabsf(ymx) + fabsf(radius)) * 1e-6f + 1e-6f;   // covers the rounding of dy in the test
            int sc0 = s_ys[mp_ybucket(ymn - radius - ys, ty0, yscale)] & ~63;   // steps of 64 stay inside the NaN padded array
            int sc1 = s_ys[mp_ybucket(ymx + radius + ys, ty0, yscale) + 1];
            sc0 = __builtin_amdgcn_readfirstlane(sc0);
            sc1 = __builtin_amdgcn_readfirstlane(sc1);
            const int l31 = lane & 31;
            for (int base = sc0; base < sc1; base += 64) {
                // two steps of 32 targets in flight
                const float2 ta = s_ykp[base + l31], tb = s_ykp[base + 32 + l31];
                const uint32_t pa = s_ypos[base + l31], pb = s_ypos[base + 32 + l31];
                uint32_t ma = 0, mb = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ina = mp_l1_bits(qx[i], qy[i], ta) < thr[i];
                    const bool inb = mp_l1_bits(qx[i], qy[i], tb) < thr[i];
                    ma = ma + ma + (ina ? 1u : 0u);
                    mb = mb + mb + (inb ? 1u : 0u);
                }
                // lanes 0..31: own nibble = queries 0..3 (high nibble of the byte), partner's = queries 4..7
                const uint32_t oa = (uint32_t)__shfl_xor((int)ma, 32), ob = (uint32_t)__shfl_xor((int)mb, 32);
                const uint32_t m8a = half ? 0u : ((ma << 4) | oa), m8b = half ? 0u : ((mb << 4) | ob);
                const uint32_t ua = (uint32_t)__ballot(m8a != 0), ub = (uint32_t)__ballot(m8b != 0);
                const int ca = __popc(ua);
                if (m8a) ul[min(ucnt + (int)__builtin_amdgcn_mbcnt_lo(ua, 0u), MP_UCAP - 1)] = (m8a << 24) | (pa << 8);
                if (m8b) ul[min(ucnt + ca + (int)__builtin_amdgcn_mbcnt_lo(ub, 0u), MP_UCAP - 1)] = (m8b << 24) | (pb << 8);
                ucnt += ca + __popc(ub);
            }
            for (int base = wcap; base < W; base += 32) {   // windows wider than MP_KPCAP (dense data only)
                const int w = base + l31;
                float2 t2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
                if (w < W) t2 = P.t.skp[lo + w];
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) m = m + m + ((mp_l1_bits(qx[i], qy[i], t2) < thr[i]) ? 1u : 0u);
                const uint32_t o = (uint32_t)__shfl_xor((int)m, 32);
                const uint32_t m8 = half ? 0u : ((m << 4) | o);
                const uint32_t u = (uint32_t)__ballot(m8 != 0);
                if (m8) ul[min(ucnt + (int)__builtin_amdgcn_mbcnt_lo(u, 0u), MP_UCAP - 1)] = (m8 << 24) | ((uint32_t)w << 8);
                ucnt += __popc(u);
            }
        }
        bool list_ovf = ucnt > MP_UCAP || W > 0xffff;
        const int nu = list_ovf ? 0 : ucnt;
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 1b: lane per union row: in-radius candidates per query (K cap) as scalar bit counts,
        // the eight bounds of the row, and per query the lane's smallest (bound << 9 | list position) over its member rows
        int cnt[MP_G];
        uint32_t pmin[MP_G];
#pragma unroll
        for (int k = 0; k < MP_G; ++k) { cnt[k] = 0; pmin[k] = 0xffffffffu; }
        for (int b = 0; b < nu; b += VISO_WAVE) {
            const int li = b + lane;
            const uint32_t e = li < nu ? ul[li] : 0u;
            const uint32_t w = (e >> 8) & 0xffffu;
            uint2 ts = s_sum[min(w, (uint32_t)(MP_KPCAP - 1))];
            if (w >= (uint32_t)wcap && e) { const unsigned long long v_ = ((const __attribute__((address_space(1))) unsigned long long*)reinterpret_cast<const unsigned long long*>(P.t.sums))[lo + w]; ts = make_uint2((uint32_t)v_, (uint32_t)(v_ >> 32)); }   // wide windows only
#pragma unroll
            for (int k = 0; k < MP_G; ++k) {
                const bool member = ((e >> (31 - k)) & 1u) != 0;
                cnt[k] += __popcll(__ballot(member));
                uint32_t bd = __builtin_amdgcn_sad_u16(ts.x, qsx[k], 0u);
                bd = __builtin_amdgcn_sad_u16(ts.y, qsy[k], bd);
                pmin[k] = min(pmin[k], member ? ((bd << 9) | (uint32_t)li) : 0xffffffffu);
            }
        }
        int my_cnt = 0;
#pragma unroll
        for (int k = 0; k < MP_G; ++k) if (lane == k) my_cnt = cnt[k];
        // wave minimum per query by a transposing reduction (as match_union.hip's sums): three exchange steps inside the
        // 8-lane group halve the values a lane carries, three more across the groups; lane (g, j) ends with the probe
        // of query myq(j)
        uint32_t pkey;
        {
            const bool sel0 = ((lane ^ (lane >> 2)) & 1) != 0, sel1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0, sel2 = ((lane >> 2) & 1) != 0;
#define MP_M1(A, B) ({ const uint32_t k_ = sel0 ? (B) : (A); const uint32_t g_ = sel0 ? (A) : (B); min(k_, mp_dpp<0xB1>(g_)); })    /* lane ^ 1 */
#define MP_M2(A, B) ({ const uint32_t k_ = sel1 ? (B) : (A); const uint32_t g_ = sel1 ? (A) : (B); min(k_, mp_dpp<0x4E>(g_)); })    /* lane ^ 2 */
#define MP_M4(A, B) ({ const uint32_t k_ = sel2 ? (B) : (A); const uint32_t g_ = sel2 ? (A) : (B); min(k_, mp_dpp<0x141>(g_)); })   /* 7 - lane */
            const uint32_t a0 = MP_M1(pmin[0], pmin[1]), a1 = MP_M1(pmin[2], pmin[3]), a2 = MP_M1(pmin[4], pmin[5]), a3 = MP_M1(pmin[6], pmin[7]);
            const uint32_t c0 = MP_M2(a0, a1), c1 = MP_M2(a2, a3);
            uint32_t m = MP_M4(c0, c1);
#undef MP_M4
#undef MP_M2
#undef MP_M1
            m = min(m, (uint32_t)__shfl_xor((int)m, 8));
            m = min(m, (uint32_t)__shfl_xor((int)m, 16));
            m = min(m, (uint32_t)__shfl_xor((int)m, 32));
            // lane j (< 8) holds query myq(j) = 0 1 2 3 7 6 5 4: group k fetches query k's
            const int srcl = g8 < 4 ? g8 : 11 - g8;
            pkey = (uint32_t)__shfl((int)m, srcl);
        }
#pragma unroll
        for (int k = 0; k < MP_G; ++k) s_qrow[wave][k][lane] = qw[k];
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 1c: the eight probes, one pass: lane group k scores query k against its probe row
#define MP_SADROW(QK, R0, R1)                                                                              \
        ({                                                                                                 \
            const u32x4 qa_ = *reinterpret_cast<const u32x4*>(&s_qrow[wave][QK][sub * 4]);                 \
            const u32x4 qb_ = *reinterpret_cast<const u32x4*>(&s_qrow[wave][QK][sub * 4 + 32]);            \
            uint32_t s_ = __builtin_amdgcn_sad_u16((R0).x, qa_.x, 0u);                                     \
            s_ = __builtin_amdgcn_sad_u16((R0).y, qa_.y, s_);                                              \
            s_ = __builtin_amdgcn_sad_u16((R0).z, qa_.z, s_);                                              \
            s_ = __builtin_amdgcn_sad_u16((R0).w, qa_.w, s_);                                              \
            s_ = __builtin_amdgcn_sad_u16((R1).x, qb_.x, s_);                                              \
            s_ = __builtin_amdgcn_sad_u16((R1).y, qb_.y, s_);                                              \
            s_ = __builtin_amdgcn_sad_u16((R1).z, qb_.z, s_);                                              \
            s_ = __builtin_amdgcn_sad_u16((R1).w, qb_.w, s_);                                              \
            s_ += mp_dpp<0xB1>(s_);    /* lane ^ 1 */                                                      \
            s_ += mp_dpp<0x4E>(s_);    /* lane ^ 2 */                                                      \
            s_ += mp_dpp<0x141>(s_);   /* 7 - lane: every lane of the group of 8 holds the row's SAD */    \
            s_;                                                                                            \
        })
        const uint32_t plpos = pkey & 511u;               // group k: query k's probe (pkey = bound << 9 | list position)
        uint32_t bmin = 0u;                               // no probe (no candidate): nothing to keep
        if (nu > 0) {
            const uint32_t pe = ul[pkey == 0xffffffffu ? 0u : plpos];
            const grow_t prow = (grow_t)(wrows + ((pe & 0x00ffffffu) | (uint32_t)(sub << 4)));
            const u32x4 p0 = prow[0], p1 = prow[8];
            const uint32_t dstar = MP_SADROW(g8, p0, p1);
            if (pkey != 0xffffffffu) bmin = mp_bmin(dstar, mp.second, mp.ratio, rinv);
        }
        // -bmin per query, as a VECTOR register each (a v_sad_u16 takes one scalar operand: the query's sums); bmin is
        // clamped to 2^31 - 1 so that the sign of (bound - bmin) is the comparison.  The probe itself always stays:
        // its bound <= d* < bmin.
        uint32_t nbm[MP_G];
#pragma unroll
        for (int k = 0; k < MP_G; ++k) {
            const uint32_t bk = min((uint32_t)__builtin_amdgcn_readlane((int)bmin, 8 * k), 0x7fffffffu);
            uint32_t v = 0u - bk;
            asm volatile("" : "+v"(v));
            nbm[k] = v;
        }
        // ---------------- phase 1d: cells, row-major: (query k, row) with member && bound < bmin; the cell list follows
        // the union list in the wave's buffer
        uint32_t* cells = ul + nu;
        const int ccap = min(MP_CCAP, MP_LCAP - MP_CPAD - nu);
        int ccnt = 0;
        for (int b = 0; b < nu; b += VISO_WAVE) {
            const int li = b + lane;
            const uint32_t e = li < nu ? ul[li] : 0u;
            const uint32_t w = (e >> 8) & 0xffffu;
            uint2 ts = s_sum[min(w, (uint32_t)(MP_KPCAP - 1))];
            if (w >= (uint32_t)wcap && e) { const unsigned long long v_ = ((const __attribute__((address_space(1))) unsigned long long*)reinterpret_cast<const unsigned long long*>(P.t.sums))[lo + w]; ts = make_uint2((uint32_t)v_, (uint32_t)(v_ >> 32)); }
            uint32_t lt = 0;                                  // bit 7 - k: bound(k) < bmin(k)
#pragma unroll
            for (int k = 0; k < MP_G; ++k) {
                uint32_t t_ = __builtin_amdgcn_sad_u16(ts.x, qsx[k], nbm[k]);     // bound - bmin (mod 2^32)
                t_ = __builtin_amdgcn_sad_u16(ts.y, qsy[k], t_);
                lt = lt + lt + (t_ >> 31);
            }
            uint32_t keep = (e >> 24) & lt;                   // bit 7 - k: cell (query k, this row)
            // exclusive prefix of the per-lane cell counts (0..8): bit planes of the count through ballots
            const int nk = __popc(keep);
            int off = ccnt;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long bal = __ballot(((nk >> j) & 1) != 0);
                off += (int)mbcnt(bal) << j;
                ccnt += __popcll(bal) << j;
            }
            const uint32_t wb = w << 8;
            while (keep) {                                   // the lane's cells: a few turns
                const uint32_t k = 7u - ((uint32_t)__ffs((int)keep) - 1u);
                if (off < ccap) cells[off] = (k << 29) | wb;
                ++off;
                keep &= keep - 1u;
            }
        }
        if (ccnt > ccap) { list_ovf = true; ccnt = 0; }
        __builtin_amdgcn_wave_barrier();
        // padding behind the list: copies of the last cell (loaded and scored, never counted)
        if (ccnt > 0 && lane < MP_CPAD) cells[ccnt + lane] = cells[ccnt - 1];
        static_assert(MP_CPAD <= 64, "one lane per padding entry");
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2: rolling pipeline over the cells, 8 lanes per cell; the lane whose position in the group
        // is the cell's query folds the SAD into its tracker: lane (g, k) ends with group g's share of query k
        MpTrack tr;
        tr.m1 = 0xffffffffu; tr.m2 = 0xffffffffu;
        {
            const int npass = (ccnt + 7) >> 3;
            u32x4 r0[MP_NP], r1[MP_NP];
            uint32_t ent[MP_NP];
#define MP_ISSUE(SLOT, T)                                                                                  \
            do {                                                                                           \
                ent[SLOT] = cells[(T) * 8 + g8];                                                           \
                const grow_t row_ = (grow_t)(wrows + ((ent[SLOT] & 0x00ffffffu) | (uint32_t)(sub << 4)));  \
                r0[SLOT] = row_[0];                                                                        \
                r1[SLOT] = row_[8];                                                                        \
            } while (0)
#define MP_REDUCE(SLOT, T)                                                                                 \
            do {                                                                                           \
                const uint32_t k_ = ent[SLOT] >> 29;                                                       \
                const uint32_t s_sad = MP_SADROW(k_, r0[SLOT], r1[SLOT]);                                  \
                const int c_ = (T) * 8 + g8;                                                               \
                const bool mine_ = k_ == (uint32_t)sub && c_ < ccnt;                                       \
                mp_update(tr, mine_ ? ((s_sad << 9) | (uint32_t)c_) : 0xffffffffu);                        \
            } while (0)
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < MP_NP; ++p) MP_ISSUE(p, p);
            }
            int t = 0;
            for (; t + MP_NP < npass; t += MP_NP) {
#pragma unroll
                for (int p = 0; p < MP_NP; ++p) {
                    MP_REDUCE(p, t + p);
                    __builtin_amdgcn_sched_barrier(0);   // keep the refill of this slot here (see match_union.hip)
                    MP_ISSUE(p, t + p + MP_NP);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < MP_NP; ++p) MP_REDUCE(p, t + p);
            }
#undef MP_REDUCE
#undef MP_ISSUE
        }
        // ---------------- phase 3: merge the 8 lane groups (lane (g, k) tracks query k); lanes 0..7 = queries 0..7;
        // fetch the original target index, ratio test, store
#pragma unroll
        for (int m = 8; m < VISO_WAVE; m <<= 1) {
            MpTrack o;
            o.m1 = (uint32_t)__shfl_xor((int)tr.m1, m);
            o.m2 = (uint32_t)__shfl_xor((int)tr.m2, m);
            mp_merge(tr, o);
        }
        if (lane < MP_G && my_orig >= 0) {
            const bool none = tr.m1 == 0xffffffffu;
            const uint32_t d1 = tr.m1 >> 9;
            const bool tie = !none && tr.m2 != 0xffffffffu && (tr.m2 >> 9) == d1;
            if (list_ovf || my_cnt > K || tie) {
                // more than K candidates / lists too long / exact tie of the minimum (largest-key rule): overflow kernel
                P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, my_j);
            } else {
                bool accept = !none;
                int idx = -1;
                if (accept) {
                    const int w = (int)((cells[tr.m1 & 511u] >> 8) & 0xffffu);   // window position of the winner
                    idx = P.t.sidx[lo + w];
                    if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3).  m2 is the second smallest SAD
                        // among the SCORED candidates; every unscored one passes this test on its own (header)
                        const double bd2 = tr.m2 == 0xffffffffu ? 1.7976931348623157e308 : (double)(tr.m2 >> 9);
                        accept = (double)d1 < bd2 * mp.ratio;
                    }
                }
                P.res[my_orig] = make_int2(accept ? idx : -1, none ? -1 : (int)d1);
                scored += (unsigned long long)my_cnt;
            }
        }
        __builtin_amdgcn_wave_barrier();   // the wave's lists are rewritten by its next round
    }
#undef MP_SADROW
    // scored pairs of the tile's queries whose result stands (lanes 0..7 of every wave hold partial sums)
#pragma unroll
    for (int m = 1; m < MP_G; m <<= 1) scored += (unsigned long long)__shfl_xor((long long)scored, m);
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
}

int launch_match_prune_temporal(hipStream_t s, const BatchMatchArgs& a, long long blocks) {
    hipLaunchKernelGGL(match_prune_kernel, dim3((unsigned)blocks), dim3(MP_THREADS), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_prune_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
