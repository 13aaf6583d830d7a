// match_union8.hip — temporal matcher, matcher variant 6: match_union_kernel's tile / round / scan, but the round's
// union list is ranked on the rows' 8-BIT PLANES and only the two best candidates of a query are scored exactly.
//
// Why that is still match_desc (src/viso.cpp:669-726), bit for bit.  A query's outputs are the smallest SAD d1, its
// candidate, and ONE bit of the second smallest d2: d1 < d2 * ratio (:715).  With h(v) = clamp((v + (128 << s)) >> s, 0, 255),
// s = the run's shift (ImageView::rows8; floor and clamp are monotone and 1-Lipschitz in units of 2^s), every element obeys
// 2^s |h(a) - h(b)| - (2^s - 1) <= |a - b|, so L(t) = (SAD8(t) << s) - 128 (2^s - 1) <= SAD(t) for ANY descriptors of up to
// 128 elements and ANY s in 0..3 (s = 3 covers the whole range of a 3x3 Sobel of uint8 with a slack of 896; the host picks s
// per run from the magnitudes the pack kernels sampled, common.h VISO_R8_*: speed only, never a result).  The kernel keeps, per query,
// the THREE smallest SAD8 keys of the union pass; scores the first two exactly (u16 rows, v_sad_u16): e1, e2, d1' = min,
// d2' = max; and lets L3 = L(third key) — a lower bound of EVERY unscored candidate's SAD, because L is monotone in the
// key — decide whether the unscored ones can matter:
//   accept case, d1' < fl(d2' ratio):  if L3 > d1' and d1' < fl(L3 ratio), no unscored t is the minimum or ties it
//       (SAD(t) >= L3 > d1'), and none can fail :715 as the second best (fl(SAD(t) ratio) >= fl(L3 ratio) > d1': rounding
//       of a product with a non-negative factor is monotone) — the result is (argmin, d1'), accepted;
//   reject case, d1' >= fl(d2' ratio):  if L3 >= fl(d2' ratio), the true d1 = min(d1', unscored) >= fl(d2' ratio) >=
//       fl(true d2 ratio) (true d2 <= d2') — rejected whatever the unscored candidates are;
//   anything else (the third key too close, e1 == e2, fewer checks when :713 is off) goes to match_overflow_kernel like
//   a tied minimum always did.  On the bench's frames that is 0.2 % of the queries; on data whose SADs are packed within
//   ~1800 of each other it is most of them — slower, never different.
// A pass over 64 (row, query) cells is 32 v_sad_u8 instead of 64 v_sad_u16, the row gather is 128 instead of 256 B, and
// the exact scoring is two passes of 8 cells per round (lane group g = query g).  The SAD of a REJECTED query is not
// produced (res.y = d1', an upper bound): nothing reads it (sort_matches_kernel and the join read accepted rows only).
//
// Tile, round composition and scan are match_union.hip's (see there).
#include "common.h"
#include "match_dev.h"
#include <stddef.h>
#include <stdlib.h>

#define MU_THREADS 256
#define MU_WAVES 4
#define MU_QPB 64          // queries per tile
#define MU_G 8             // queries per round
#define MU_UCAP 448        // union list entries per round (+ MU_PAD < 512: a list position fits the 9 low bits of a tracker key)
#define MU_PAD 32          // list padding: the pipeline runs up to 4 passes of 8 rows past the end
#ifndef MU_NP
#define MU_NP 2            // passes in flight (3 measured slower even without spills)
#endif
#ifndef MU_RNP
#define MU_RNP 2           // rescue: passes of eight survivors in flight
#endif
#define MU_S8ROWS 176      // list positions whose SAD8 (in units of 128: one byte) is kept per query for the rescue (with this much
                           // LDS 7 workgroups per CU still fit; the bench's lists hold 94 rows on average)
#define MU_KPCAP 512       // window keypoints staged in LDS
#define MU_NBY 64          // y buckets of the staged window

template <int CTRL>
__device__ __forceinline__ uint32_t mu_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// quad broadcast (every lane has a source: no old value to keep, bound_ctrl spares the compiler its initialisation)
template <int CTRL>
__device__ __forceinline__ uint32_t mu_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// fminf / fmaxf without the canonicalising v_max x, x the compiler puts in front (v_min / v_max return the other operand
// for a NaN, like fminf / fmaxf)
__device__ __forceinline__ float mu_fmin(float a, float b) { float r; asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float mu_fmax(float a, float b) { float r; asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// bits of |qx - tx| + |qy - ty| (cvflann::L1 order, see l1_kp); abs as source modifiers of the add
__device__ __forceinline__ uint32_t mu_l1_bits(float qx, float qy, float2 t) {
    const float dx = qx - t.x, dy = qy - t.y;
    float d;
    asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(d) : "v"(dx), "v"(dy));
    return __float_as_uint(d);
}

// Running order statistics of one query's 8-bit-plane SADs as three packed keys, key = SAD8 << 9 | position in the round's
// union list (< 512; SAD8 <= 121 * 255): m1 <= m2 <= m3 = the three smallest keys (0xffffffff = none).  A query sees every
// list position at most once, so keys are distinct.
struct MuTrack { uint32_t m1, m2, m3; };

__device__ __forceinline__ uint32_t mu_med3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void mu_update(MuTrack& t, uint32_t key) {
    t.m3 = mu_med3(t.m2, t.m3, key);   // m2 <= m3: the median of (m2, m3, key) is the new third smallest
    t.m2 = mu_med3(t.m1, t.m2, key);
    t.m1 = min(t.m1, key);
}

// the three smallest of two sorted triples: k-th smallest of a merge = min over i + j = k of max(a_i, b_j)
__device__ __forceinline__ void mu_merge(MuTrack& a, const MuTrack& b) {
    const uint32_t c3 = min(min(a.m3, b.m3), min(max(a.m2, b.m1), max(a.m1, b.m2)));
    const uint32_t c2 = min(max(a.m1, b.m1), min(a.m2, b.m2));
    a.m1 = min(a.m1, b.m1);
    a.m2 = c2;
    a.m3 = c3;
}

// y bucket of the staged window: monotone in y, total (NaN -> 0, +-inf saturate): v_cvt_i32_f32 truncates, saturates and
// turns NaN into 0 (the C conversion would be undefined there), the clamp makes truncation and floor the same thing
__device__ __forceinline__ int mu_ybucket(float y, float y0, float scale) {
    const float f = (y - y0) * scale;
    int b;
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(b) : "v"(f));
    return min(max(b, 0), MU_NBY - 1);
}

// The shift of this run's planes (BatchMatchArgs::r8s), read from the kernel-argument segment WHERE it is used, through a
// laundered pointer: as a value the compiler preloads at the kernel's entry it is one more scalar register live across
// the whole kernel, and this kernel already spills scalar registers into vector lanes (19 of them; 26 with that one,
// their reloads inside the loops: 0.530 -> 0.542 ms)
__device__ __forceinline__ int mu_r8s() {
    const int* kp = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp[offsetof(BatchMatchArgs, r8s) / sizeof(int)];
}

// the lane that carries query k of a round (lanes 0..31: queries 0..3, lanes 32..63: queries 4..7, repeated every four
// lanes, so that a quad broadcast hands every lane the four queries its half tests in phase 1)
__device__ __forceinline__ constexpr int mu_qlane(int k) { return (k & 3) + 32 * (k >> 2); }

// (match_frame.hip includes this file with MU_KERNEL_SIG / MU_BLOCK defined: the same body as a device function that takes its
// block number as an argument -- one launch for a single frame's stereo and temporal problems.  Here: the kernel.)
#ifndef MU_CLK
#define MU_CLK(I) do {} while (0)   // (match_frame.hip, debug builds: time stamps of one tile's phases)
#endif
#ifndef MU_KERNEL_SIG
#define MU_KERNEL_SIG __global__ __attribute__((amdgpu_waves_per_eu(7, 8))) __launch_bounds__(MU_THREADS) void match_union8_kernel(BatchMatchArgs a)
#define MU_BLOCK blockIdx.x
#endif
MU_KERNEL_SIG {
    __shared__ __attribute__((aligned(16))) uint32_t s_ul[MU_WAVES][MU_UCAP + MU_PAD];
    __shared__ __attribute__((aligned(16))) uint32_t s_qrow[MU_WAVES][MU_G][32];   // the round's query rows, 8-bit planes
    __shared__ uint8_t s_s8[MU_WAVES][MU_G][MU_S8ROWS];   // SAD8 >> 7 of (query, list position) of the round: what the rescue selects from
    __shared__ float2 s_ykp[MU_KPCAP];       // staged window keypoints in y-bucket order
    __shared__ uint16_t s_ypos[MU_KPCAP];    // their window positions
    __shared__ int s_ys[MU_NBY + 1];         // bucket counts, then bucket starts
    __shared__ float s_xr[2];
    int prob, qblk;
    {
        const int b = MU_BLOCK;
        const int xcd = b & 7, slot = b >> 3;
        const int g = slot / a.bpp;
        prob = ((g / a.gc) * a.gs + a.gf + g % a.gc) * 8 + xcd;
        qblk = slot % a.bpp;
        if (prob >= a.n_probs) return;
    }
    MU_CLK(0);
    const MatchProblem P = a.probs[prob];
    if ((*P.q.bad | *P.t.bad) != 0) return;   // non-integer descriptors: the general kernel does this problem
    const int n1 = *P.q.n, n2 = *P.t.n;
    const int q0 = qblk * MU_QPB;
    if (q0 >= n1) return;
    const int q1 = min(q0 + MU_QPB, n1);
    const MatchParamsDev& mp = a.mp[P.pidx];
    if (mp.epi != 0) return;   // stereo problems: match_batch_kernel<1>
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // ---- tile: x range (window) of its queries; their y order (round composition) comes from sort_kp_kernel (ImageView::qord)
    if (wave == 0) {
        const bool live = q0 + lane < q1;
        const float qx = live ? P.q.skp[q0 + lane].x : __builtin_nanf("");
        const float mn = viso_wave_fext<false>(qx), mx = viso_wave_fext<true>(qx);
        if (lane == 0) { s_xr[0] = mn; s_xr[1] = mx; }
    }
    if (threadIdx.x <= MU_NBY) s_ys[threadIdx.x] = 0;
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
    const int wcap = min(W, MU_KPCAP);
    const int wpad = (wcap + 127) & ~127;   // NaN padded: the scan needs no bounds test
    // ---- y index of the staged window: bucket sort (histogram with returning LDS atomics, scan, scatter) over the
    // target image's y range, so that a round only scans the buckets its four diamonds can touch
    float ty0 = P.t.xinfo[2];
    float yscale = 0.f;
    {
        const float ty1 = P.t.xinfo[3];
        if (ty1 > ty0) yscale = (float)MU_NBY / (ty1 - ty0);
        if (!(yscale > 0.f) || !(yscale < 3.0e38f)) yscale = 0.f;
    }
    // wave-uniform floats belong in scalar registers (the vector ALU takes one as an operand): four VGPRs less
    ty0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ty0)));
    yscale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(yscale)));
    static_assert(MU_KPCAP <= 2 * MU_THREADS, "two window entries per thread");
    float2 e_kp[2];
    int e_b[2], e_r[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = threadIdx.x + i * MU_THREADS;
        e_b[i] = 0; e_r[i] = 0;
        if (w < wcap) {
            e_kp[i] = P.t.skp[lo + w];
            e_b[i] = mu_ybucket(e_kp[i].y, ty0, yscale);
            e_r[i] = atomicAdd(&s_ys[e_b[i]], 1);
        }
    }
    float2 kp0 = make_float2(0.f, 0.f);
    const bool has0 = n2 > 0;
    if (has0) kp0 = P.t.skp[P.t.rank[0]];
    kp0.x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(kp0.x)));
    kp0.y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(kp0.y)));
    __syncthreads();
    if (wave == 0) {
        const int h = s_ys[lane];
        const int incl = (int)viso_wave_scan((uint32_t)h);
        s_ys[lane] = incl - h;
        if (lane == VISO_WAVE - 1) s_ys[MU_NBY] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = threadIdx.x + i * MU_THREADS;
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
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* grow_t;
    typedef const __attribute__((address_space(1))) char* gbytes_t;
    // window bases (scalar) + 32-bit byte offset per lane: 8-bit planes (window position << 7) | (sub << 4), u16 rows twice that
    const gbytes_t wrows8 = (gbytes_t)reinterpret_cast<const char*>(P.t.rows8) + (size_t)lo * VISO_ROW8;
    const gbytes_t wrows = (gbytes_t)reinterpret_cast<const char*>(P.t.rows) + (size_t)lo * (VISO_ROW * 2);
    const gbytes_t qrows = (gbytes_t)reinterpret_cast<const char*>(P.q.rows);
    const gbytes_t qrows8 = (gbytes_t)reinterpret_cast<const char*>(P.q.rows8);
    const gbytes_t qord_b = (gbytes_t)reinterpret_cast<const char*>(P.q.qord), qskp_b = (gbytes_t)reinterpret_cast<const char*>(P.q.skp),
                   qsidx_b = (gbytes_t)reinterpret_cast<const char*>(P.q.sidx);
    uint32_t* ul = s_ul[wave];
    const int sub = lane & 7;
    const int half = lane >> 5;          // phase 1: lanes 0..31 test queries 0..3 of the round, lanes 32..63 queries 4..7
    unsigned long long scored = 0;
    constexpr int ROUNDS = MU_QPB / (MU_WAVES * MU_G);   // 2 rounds of 8 queries per wave
    // which of the round's eight queries the lane tracks after the reduction of phase 2: query bit 1 = lane bit 2 (the first
    // exchange step, lane ^ 4, keeps the pair (0,1 | 4,5) or (2,3 | 6,7)), query bit 2 = lane bit 1 (lane ^ 2 keeps the
    // low or the high four), query bit 0 = lane bit 0 (which HALF of the packed pair the lane extracts at the end)
    const bool sel1 = ((lane >> 1) & 1) != 0;
    const int myq = (lane & 1) + 2 * ((lane >> 2) & 1) + (sel1 ? 4 : 0);
    const uint32_t hoff = (uint32_t)(lane & 1) << 4;   // bit offset of the lane's half in a packed pair of totals
    const int msh = 31 - myq;   // membership bit of query myq in a list entry (bit 7 - k of the mask byte)

    // query data one round ahead: lane l carries local index / keypoint / original index of query (l & 3) + 4 * half
    // of the round (mu_qlane)
    const int qslot = (lane & 3) + 4 * half;
    int pli;
    float2 pq;
    int po;
#define MU_PREFETCH(R)                                                                                    \
    do {   /* scalar bases + 32-bit offsets: no 64-bit address arithmetic per lane */                    \
        const int base_ = wave * (MU_QPB / MU_WAVES) + (R) * MU_G;                                        \
        pli = (int)*(const __attribute__((address_space(1))) uint8_t*)(qord_b + (uint32_t)(q0 + base_ + qslot)); \
        const int j_ = q0 + pli;                                                                          \
        const int jc_ = min(j_, q1 - 1);                                                                  \
        {                                                                                                 \
            typedef float f32x2_ __attribute__((ext_vector_type(2)));                                     \
            const f32x2_ k_ = *(const __attribute__((address_space(1))) f32x2_*)(qskp_b + (uint32_t)jc_ * 8u); \
            pq = make_float2(k_.x, k_.y);                                                                 \
        }                                                                                                 \
        po = *(const __attribute__((address_space(1))) int*)(qsidx_b + (uint32_t)jc_ * 4u);             \
        po = j_ < q1 ? po : -1;                                                                           \
    } while (0)
    MU_CLK(1);
    MU_PREFETCH(0);

#pragma unroll   // both rounds inline: left rolled (the compiler's choice once the body grew) the loop spills 22 registers
    for (int r = 0; r < ROUNDS; ++r) {
        // ---------------- round setup.  Lane l holds query (l & 3) + 4 * half: the four queries its half tests in phase 1
        // are the four lanes of its quad (DPP quad broadcasts, no scalar traffic)
        // the round's queries for the exact scoring and the store: lane group g (lanes 8g..8g+7) takes query g
        const int gq_orig = __shfl(po, mu_qlane(lane >> 3)), gq_j = q0 + __shfl(pli, mu_qlane(lane >> 3));
        if (!__any(po >= 0)) { if (r + 1 < ROUNDS) MU_PREFETCH(r + 1); continue; }   // wave uniform
        // d = |dx| + |dy| is +0, positive or NaN (sign bit clear: the add sees |dx| and |dy|): its bit pattern orders like
        // the value and NaNs are above +inf, so (d <= radius && d < d0cut) is one unsigned compare against bits(d0)
        // (target 0 in radius: Q1, src/viso.cpp:693) or bits(radius) + 1 — and, both sides being below 2^31, the SIGN of
        // (thr - 1) - bits(d): CLEAR for a member — masks are kept inverted (bit 7 - k set = query k is NOT a member), which
        // lets phase 2 turn a non-member's tracker key into "none" with an OR.  Dead slots (past the tile) get 0: nothing passes.
        uint32_t tq = __float_as_uint(radius) + 1u;
        if (has0) {
            const float d0 = l1_kp(pq.x, pq.y, kp0);
            if (d0 <= radius) tq = __float_as_uint(d0);
        }
        if (po < 0) tq = 0u;
        tq -= 1u;   // dead slots: 0xffffffff - bits(d) has its sign set
        float qx[4], qy[4];
        uint32_t thr[4];
        qx[0] = __uint_as_float(mu_bcast<0x00>(__float_as_uint(pq.x))); qy[0] = __uint_as_float(mu_bcast<0x00>(__float_as_uint(pq.y))); thr[0] = mu_bcast<0x00>(tq);
        qx[1] = __uint_as_float(mu_bcast<0x55>(__float_as_uint(pq.x))); qy[1] = __uint_as_float(mu_bcast<0x55>(__float_as_uint(pq.y))); thr[1] = mu_bcast<0x55>(tq);
        qx[2] = __uint_as_float(mu_bcast<0xAA>(__float_as_uint(pq.x))); qy[2] = __uint_as_float(mu_bcast<0xAA>(__float_as_uint(pq.y))); thr[2] = mu_bcast<0xAA>(tq);
        qx[3] = __uint_as_float(mu_bcast<0xFF>(__float_as_uint(pq.x))); qy[3] = __uint_as_float(mu_bcast<0xFF>(__float_as_uint(pq.y))); thr[3] = mu_bcast<0xFF>(tq);
        // y extent of the four queries of the lane's half (the two halves' scan ranges are joined as scalars below)
        const float ymn = mu_fmin(mu_fmin(qy[0], qy[1]), mu_fmin(qy[2], qy[3]));
        const float ymx = mu_fmax(mu_fmax(qy[0], qy[1]), mu_fmax(qy[2], qy[3]));
        // the eight query rows' 8-bit planes (32 dwords each): one word per lane and PAIR of rows from global memory (lanes
        // 0..31 query k, lanes 32..63 query k + 4: the row of quad lane k of the lane's own half, so its byte offset is ONE
        // quad broadcast of the lane's own; the loads land during the scan), then LDS
        uint32_t qw[MU_G / 2];
        {
            const uint32_t own = (uint32_t)min(q0 + pli, q1 - 1) * (uint32_t)VISO_ROW8;   // the row of the lane's own query
            const uint32_t l4 = (uint32_t)((lane & 31) << 2);
            qw[0] = *(const __attribute__((address_space(1))) uint32_t*)(qrows8 + (mu_bcast<0x00>(own) + l4));   // scalar base + 32-bit offset
            qw[1] = *(const __attribute__((address_space(1))) uint32_t*)(qrows8 + (mu_bcast<0x55>(own) + l4));
            qw[2] = *(const __attribute__((address_space(1))) uint32_t*)(qrows8 + (mu_bcast<0xAA>(own) + l4));
            qw[3] = *(const __attribute__((address_space(1))) uint32_t*)(qrows8 + (mu_bcast<0xFF>(own) + l4));
        }
        if (r + 1 < ROUNDS) MU_PREFETCH(r + 1);
        // ---------------- phase 1: one scan over the y buckets the round's diamonds touch, 32 targets per step: both
        // halves read the same 32 entries, each tests its four queries (sign of bits(d) - thr shifted into a 4-bit mask:
        // sub + alignbit, no condition code), one v_permlane32_swap joins the halves' nibbles into the 8-bit membership
        // mask (bit 7 - k = query k) in every lane; targets with a non-zero mask go to the union list (lanes 0..31
        // write).  entry = inverted mask << 24 | window position << 7 (the row's offset in the 8-bit planes).  ncnt accumulates the set bits = (query, target)
        // pairs tested and NOT in radius (the same in both halves)
        int ucnt = 0;
        uint32_t ncnt = 0, ntest = 0;
#define MU_TEST4(T)                                                                                       \
        ({                                                                                                \
            uint32_t m_ = 0;                                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
                m_ = __builtin_amdgcn_alignbit(m_, thr[i] - mu_l1_bits(qx[i], qy[i], (T)), 31);           \
            /* swap(A, B): A's lanes 32..63 <-> B's lanes 0..31; with A = B = m_: r_[0] = the low half's nibble everywhere, */ \
            /* r_[1] = the high half's */                                                                 \
            const auto r_ = __builtin_amdgcn_permlane32_swap(m_, m_, false, false);                       \
            (uint32_t)((r_[0] << 4) | r_[1]);                                                             \
        })
        {
            const float ys = (fabsf(ymn) + fabsf(ymx) + fabsf(radius)) * 1e-6f + 1e-6f;   // covers the rounding of dy in the test
            const int h0 = s_ys[mu_ybucket(ymn - radius - ys, ty0, yscale)];
            const int h1 = s_ys[mu_ybucket(ymx + radius + ys, ty0, yscale) + 1];
            const int sc0 = min(__builtin_amdgcn_readlane(h0, 0), __builtin_amdgcn_readlane(h0, 32)) & ~63;   // steps of 64 stay inside the NaN padded array
            const int sc1 = max(__builtin_amdgcn_readlane(h1, 0), __builtin_amdgcn_readlane(h1, 32));
            const int l31 = lane & 31;
            if (sc1 > sc0) ntest = (uint32_t)((sc1 - sc0 + 63) >> 6) * 16u;   // two steps of eight tests per lane and iteration
            for (int base = sc0; base < sc1; base += 64) {
                // two steps of 32 targets in flight
                const float2 ta = s_ykp[base + l31], tb = s_ykp[base + 32 + l31];
                const uint32_t pa = s_ypos[base + l31], pb = s_ypos[base + 32 + l31];
                const uint32_t m8a = MU_TEST4(ta), m8b = MU_TEST4(tb);
                ncnt += (uint32_t)__popc(m8a) + (uint32_t)__popc(m8b);
                const uint32_t ua = (uint32_t)__ballot(m8a != 0xffu), ub = (uint32_t)__ballot(m8b != 0xffu);
                const int ca = __popc(ua);
                if (m8a != 0xffu && half == 0) ul[min(ucnt + (int)__builtin_amdgcn_mbcnt_lo(ua, 0u), MU_UCAP - 1)] = (m8a << 24) | (pa << 7);
                if (m8b != 0xffu && half == 0) ul[min(ucnt + ca + (int)__builtin_amdgcn_mbcnt_lo(ub, 0u), MU_UCAP - 1)] = (m8b << 24) | (pb << 7);
                ucnt += ca + __popc(ub);
            }
            for (int base = wcap; base < W; base += 32) {   // windows wider than MU_KPCAP (dense data only)
                const int w = base + l31;
                float2 t2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
                if (w < W) t2 = P.t.skp[lo + w];
                const uint32_t m8 = MU_TEST4(t2);
                ncnt += (uint32_t)__popc(m8);
                ntest += 8u;
                const uint32_t u = (uint32_t)__ballot(m8 != 0xffu);
                if (m8 != 0xffu && half == 0) ul[min(ucnt + (int)__builtin_amdgcn_mbcnt_lo(u, 0u), MU_UCAP - 1)] = (m8 << 24) | ((uint32_t)w << 7);
                ucnt += __popc(u);
            }
        }
#undef MU_TEST4
        const bool list_ovf = ucnt > MU_UCAP;
        const int nu = __builtin_amdgcn_readfirstlane(list_ovf ? 0 : ucnt);   // wave uniform: the pipeline's loop control stays scalar
        __builtin_amdgcn_wave_barrier();
        // padding behind the list: copies of the last entry with nobody's membership (scored, never counted)
        if (nu > 0 && lane < MU_PAD) ul[nu + lane] = ul[nu - 1] | 0xff000000u;
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < MU_G / 2; ++k) s_qrow[wave][k + 4 * half][lane & 31] = qw[k];
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2: rolling pipeline over the union list on the 8-BIT PLANES: load a row's plane once (one
        // dwordx4 per lane, 8 lanes per row), 4 x v_sad_u8 against each of the eight query planes (staged per wave in LDS),
        // transposing reduction (every lane of the 8-lane group ends with the SAD8 of ITS query), three-key tracker
        MuTrack tr;
        tr.m1 = 0xffffffffu; tr.m2 = 0xffffffffu; tr.m3 = 0xffffffffu;
        {
            const int npass = (nu + 7) >> 3;
            // the lane group's list position, derived HERE: as a loop invariant of the round loop its LDS address stays
            // live across phase 1
            int g8 = lane;
            asm volatile("" : "+v"(g8));
            g8 >>= 3;
            u32x4 r0[MU_NP];
            uint32_t un[MU_NP];   // the pass's list position for the tracker key, all ones where the lane's query is not a member
            // LDS byte address of the lane's SAD8 slot of pass 0 (query myq, list position g8)
            const uint32_t s8w = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)&s_s8[wave][myq][g8];
            uint32_t s8a = s8w;
#define MU_ISSUE(SLOT, T)                                                                                  \
            do {                                                                                           \
                const uint32_t ent_ = ul[(T) * 8 + g8];                                                    \
                un[SLOT] = (uint32_t)((T) * 8 + g8) | (uint32_t)__builtin_amdgcn_sbfe((int)ent_, (uint32_t)msh, 1u); \
                const grow_t row_ = (grow_t)(wrows8 + ((ent_ & 0x00ffffffu) | (uint32_t)(sub << 4)));      \
                r0[SLOT] = row_[0];                                                                        \
            } while (0)
            // SAD8 of queries 2J (low half) and 2J + 1 (high half, v_sad_hi_u8: (sad << 16) + accumulator) against the lane's
            // 16 bytes of the row, in ONE register: a lane's share is <= 16 * 255 and a row's total <= 128 * 255 < 2^15, so
            // the halves never carry into each other on the way through the reduction
#define MU_SAD2(J, SLOT)                                                                                   \
            ({                                                                                             \
                const u32x4 qa_ = *reinterpret_cast<const u32x4*>(&s_qrow[wave][2 * (J)][sub * 4]);     \
                const u32x4 qb_ = *reinterpret_cast<const u32x4*>(&s_qrow[wave][2 * (J) + 1][sub * 4]); \
                uint32_t s_ = __builtin_amdgcn_sad_u8(r0[SLOT].x, qa_.x, 0u);                              \
                s_ = __builtin_amdgcn_sad_u8(r0[SLOT].y, qa_.y, s_);                                       \
                s_ = __builtin_amdgcn_sad_u8(r0[SLOT].z, qa_.z, s_);                                       \
                s_ = __builtin_amdgcn_sad_u8(r0[SLOT].w, qa_.w, s_);                                       \
                s_ = __builtin_amdgcn_sad_hi_u8(r0[SLOT].x, qb_.x, s_);                                    \
                s_ = __builtin_amdgcn_sad_hi_u8(r0[SLOT].y, qb_.y, s_);                                    \
                s_ = __builtin_amdgcn_sad_hi_u8(r0[SLOT].z, qb_.z, s_);                                    \
                s_ = __builtin_amdgcn_sad_hi_u8(r0[SLOT].w, qb_.w, s_);                                    \
                s_;                                                                                        \
            })
#define MU_X2(A, B) ({ uint32_t k_ = sel1 ? (B) : (A); const uint32_t g_ = sel1 ? (A) : (B); k_ += mu_dpp<0x4E>(g_); k_; })   /* lane ^ 2 */
            // first exchange step, lane ^ 4, on both pairs of registers at once: a lane's bit 2 is its DPP BANK, so "keep A
            // and add the partner's A" / "keep B and add the partner's B" are two bank-masked v_add_u32_dpp (banks 0, 2 take
            // A + A[lane + 4], banks 1, 3 take B + B[lane - 4]) instead of two selects and an add.  One asm block behind an
            // s_nop 1: the compiler's hazard recognizer does not see DPP reads inside inline asm (a VGPR written by the
            // previous two VALU instructions must not be a DPP source); inside the block every source is older than that
#define MU_X4x2(S0, S1, S2, S3)                                                                            \
            asm("s_nop 1\n\t"                                                                              \
                "v_add_u32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                         \
                "v_add_u32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                         \
                "v_add_u32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"                         \
                "v_add_u32_dpp %1, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa"                              \
                : "+v"(S0), "+v"(S2) : "v"(S1), "v"(S3))
#define MU_REDUCE(SLOT, T)                                                                                \
            do {                                                                                           \
                uint32_t p0_ = MU_SAD2(0, SLOT), p1_ = MU_SAD2(1, SLOT), p2_ = MU_SAD2(2, SLOT), p3_ = MU_SAD2(3, SLOT); \
                MU_X4x2(p0_, p1_, p2_, p3_);   /* p0_: queries (0,1) or (2,3) by lane bit 2; p2_: (4,5) or (6,7) */ \
                uint32_t c_ = MU_X2(p0_, p2_);                                                             \
                c_ += mu_dpp<0xB1>(c_);        /* lane ^ 1: both lanes hold the pair's two totals */       \
                const uint32_t m_ = __builtin_amdgcn_ubfe(c_, hoff, 16u);                                  \
                /* members and non-members alike; lists longer than the store pile up in its last row (the rescue then leaves   */ \
                /* them to the overflow kernel).  As asm: a C store into LDS between the pipeline's LDS reads made the       */ \
                /* register allocator spill 26 registers                                                                      */ \
                asm volatile("ds_write_b8 %0, %1 offset:%2" : : "v"(s8a), "v"(m_ >> 7), "n"((SLOT) * 8));   /* SAD8 < 2^15 */ \
                mu_update(tr, (m_ << 9) | un[SLOT]);                                                       \
            } while (0)
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < MU_NP; ++p) MU_ISSUE(p, p);
            }
            int t = 0;
            for (; t + MU_NP < npass; t += MU_NP) {   // steady state: no branch between reduce and refill
#pragma unroll
                for (int p = 0; p < MU_NP; ++p) {
                    MU_REDUCE(p, t + p);
                    __builtin_amdgcn_sched_barrier(0);   // keep the refill of this slot HERE (see match_union.hip)
                    MU_ISSUE(p, t + p + MU_NP);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the SAD8 slot of the next MU_NP passes: a running address (an index min(pass, ..) of the loop counter made the
                // compiler spill 24 registers), advanced once per iteration; lists longer than the store pile up in its last rows
                s8a = min(s8a + 8u * MU_NP, s8w + (MU_S8ROWS / 8 - MU_NP) * 8u);
            }
            if (npass > 0) {   // the last passes (no pass of padding is scored)
#pragma unroll
                for (int p = 0; p < MU_NP; ++p)
                    if (t + p < npass) MU_REDUCE(p, t + p);
            }
#undef MU_REDUCE
#undef MU_X4x2
#undef MU_X2
#undef MU_SAD2
#undef MU_ISSUE
        }
        const int r8s = mu_r8s();   // asked for here: the merges and the exact scoring's loads cover the scalar load's latency
        // ---------------- phase 3: merge the 8 lane groups (lanes with equal position in the group track the same
        // query): lane ^ 8 by a rotation within the row of 16 (DPP), lane ^ 16 by ds_swizzle, lane ^ 32 by
        // v_permlane32_swap; every lane then holds the round's three best keys of query myq(lane)
        {
            MuTrack o;
            o.m1 = mu_dpp<0x128>(tr.m1); o.m2 = mu_dpp<0x128>(tr.m2); o.m3 = mu_dpp<0x128>(tr.m3);
            mu_merge(tr, o);
            o.m1 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)tr.m1, 0x401F);
            o.m2 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)tr.m2, 0x401F);
            o.m3 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)tr.m3, 0x401F);
            mu_merge(tr, o);
            const auto h1 = __builtin_amdgcn_permlane32_swap(tr.m1, tr.m1, false, false);
            const auto h2 = __builtin_amdgcn_permlane32_swap(tr.m2, tr.m2, false, false);
            const auto h3 = __builtin_amdgcn_permlane32_swap(tr.m3, tr.m3, false, false);
            tr.m1 = h1[0]; tr.m2 = h2[0]; tr.m3 = h3[0];
            o.m1 = h1[1]; o.m2 = h2[1]; o.m3 = h3[1];
            mu_merge(tr, o);
        }
        {   // lane group g takes query g: its keys sit in the group's lane with myq == g (lane bits (0, 2, 1) = query bits (0, 1, 2))
            const int k = lane >> 3;
            const int src = (lane & 0x38) | (k & 1) | ((k & 2) << 1) | ((k & 4) >> 1);
            tr.m1 = (uint32_t)__shfl((int)tr.m1, src);
            tr.m2 = (uint32_t)__shfl((int)tr.m2, src);
            tr.m3 = (uint32_t)__shfl((int)tr.m3, src);
        }
        // ---------------- exact scoring: the query's two best candidates by SAD8 against its u16 row, 8 lanes per pair
        // (both candidates and the query row in flight together), then the verdict (header of this file)
        const bool none = tr.m1 == 0xffffffffu;
        const bool has2 = tr.m2 != 0xffffffffu, has3 = tr.m3 != 0xffffffffu;
        uint32_t dA = 0, dB = 0;
        const uint32_t eA = ul[none ? 0u : (tr.m1 & 511u)], eB = ul[has2 ? (tr.m2 & 511u) : 0u];
        if (__any(!none)) {   // wave uniform
            const grow_t ra = (grow_t)(wrows + (((eA & 0x00ffffffu) << 1) | (uint32_t)(sub << 4)));
            const grow_t rb = (grow_t)(wrows + (((eB & 0x00ffffffu) << 1) | (uint32_t)(sub << 4)));
            const grow_t rq = (grow_t)(qrows + ((uint32_t)min(gq_j, q1 - 1) * (uint32_t)(VISO_ROW * 2) + (uint32_t)(sub << 4)));   // scalar base + 32-bit offset
            const u32x4 a0 = ra[0], a1 = ra[8], b0 = rb[0], b1 = rb[8], x0 = rq[0], x1 = rq[8];
#define MU_SAD16(R0, R1)                                                                                   \
            ({                                                                                             \
                uint32_t s_ = __builtin_amdgcn_sad_u16((R0).x, x0.x, 0u);                                  \
                s_ = __builtin_amdgcn_sad_u16((R0).y, x0.y, s_);                                           \
                s_ = __builtin_amdgcn_sad_u16((R0).z, x0.z, s_);                                           \
                s_ = __builtin_amdgcn_sad_u16((R0).w, x0.w, s_);                                           \
                s_ = __builtin_amdgcn_sad_u16((R1).x, x1.x, s_);                                           \
                s_ = __builtin_amdgcn_sad_u16((R1).y, x1.y, s_);                                           \
                s_ = __builtin_amdgcn_sad_u16((R1).z, x1.z, s_);                                           \
                s_ = __builtin_amdgcn_sad_u16((R1).w, x1.w, s_);                                           \
                s_ += mu_dpp<0xB1>(s_);                                          /* lane ^ 1 */            \
                s_ += mu_dpp<0x4E>(s_);                                          /* lane ^ 2 */            \
                s_ += (uint32_t)__builtin_amdgcn_ds_swizzle((int)s_, 0x101F);    /* lane ^ 4 */            \
                s_;                                                                                        \
            })
            dA = MU_SAD16(a0, a1);
            dB = MU_SAD16(b0, b1);
#undef MU_SAD16
        }
        // verdict of the group's query (the same in its eight lanes); the two exact candidates as keys SAD << 9 | list position
        const bool mine = sub == 0 && gq_orig >= 0;   // lanes 8k of live queries
        uint32_t kA = none ? 0xffffffffu : ((dA << 9) | (tr.m1 & 511u)), kB = has2 ? ((dB << 9) | (tr.m2 & 511u)) : 0xffffffffu;
        bool accept = !none, irregular = false, rescue = false;
        if (!none && has2) {
            const uint32_t d1 = min(dA, dB), d2 = max(dA, dB);
            const int L3 = (int)((tr.m3 >> 9) << r8s) - VISO_ROW8_SLACK(r8s);   // <= SAD of every unscored candidate (has3)
            if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                const double lim = (double)d2 * mp.ratio;
                accept = (double)d1 < lim;
                if (has3) rescue = !(accept ? (L3 > (int)d1 && (double)d1 < (double)L3 * mp.ratio) : ((double)L3 >= lim));
            } else if (has3) {
                rescue = !(L3 > (int)d1);
            }
            // exact tie of the two: the overflow kernel applies the largest-key rule (or finds a smaller third)
            if (dA == dB) { irregular = true; rescue = false; }
        } else if (!none && mp.second) {
            accept = (double)dA < 1.7976931348623157e308 * mp.ratio;   // one candidate: best_d2 keeps its initial value
        }
        // ---------------- rescue: the two exact SADs and the third key's bound do not settle the query (0.3 % of the bench's
        // queries; many more where SADs lie within the bound's slack of each other: low-contrast descriptors, repeated
        // texture).  Phase 2 left every cell's SAD8 (in units of 128) in LDS: the WAVE picks the query's members whose bound
        // L fails the very test the third key failed (the two it has already scored excepted), scores them exactly — their
        // list positions compacted into LDS, eight lanes per row and two passes in flight, the query's u16 row read from LDS, (min, second min,
        // argmin) as two packed keys SAD << 9 | position — and takes the verdict again over those and the first two.  What
        // the new verdict needs of the unscored ones follows from the test they passed (the minimum can only have fallen,
        // and with it both thresholds), unless a REJECT turned into an accept: then the selection runs once more with the
        // accept's test (scored cells are marked).  Rounds whose list is longer than the SAD8 store (dense keypoints) leave
        // such queries to match_overflow_kernel, as every query of this kind would otherwise: 36 us of that kernel per step
        // for the bench's few, against ~1 us here
        {
            asm volatile("" ::: "memory");   // phase 2 wrote s_s8 through asm: nothing below may be moved above it
            unsigned long long rm = __ballot(rescue && sub == 0);   // bit 8k: query k
            const int g8 = lane >> 3;
            uint16_t* ml = reinterpret_cast<uint16_t*>(&s_qrow[wave][0][0]);   // survivors' list positions (the planes' staging area is free by now)
            uint32_t* qst = &s_qrow[wave][4][0];                                // the query's u16 row: 64 dwords
            static_assert(MU_S8ROWS <= 256, "the survivor list holds every position that may survive");
            if (nu > MU_S8ROWS) {   // wave uniform: not every cell's SAD8 was kept (dense data): overflow kernel
                rm = 0;
                if (rescue) { irregular = true; rescue = false; }
            }
#define MU_UPD2(KEY) do { const uint32_t k_ = (KEY); m2 = mu_med3(m1, m2, k_); m1 = min(m1, k_); } while (0)
            while (rm) {   // wave uniform
                const int bit = __builtin_ctzll(rm);
                rm &= rm - 1;
                const int k = bit >> 3;
                const int jq = min(__builtin_amdgcn_readlane(gq_j, bit), q1 - 1);
                const uint32_t ka = (uint32_t)__builtin_amdgcn_readlane((int)kA, bit), kb = (uint32_t)__builtin_amdgcn_readlane((int)kB, bit);
                const uint32_t pa = ka & 511u, pb = kb & 511u;
                const int d1k = (int)(min(ka, kb) >> 9);
                const double limk = (double)(max(ka, kb) >> 9) * mp.ratio;
                const bool acck = !mp.second || (double)d1k < limk;   // the verdict of the two smallest exact SADs so far, as above
                const uint32_t msk = 31u - (uint32_t)k;
                const int r8q = mu_r8s();
                // the query's u16 row: requested now, stored to LDS behind the selection (its latency under the selection's)
                const uint32_t qv = *(const __attribute__((address_space(1))) uint32_t*)(qrows + ((uint32_t)jq * (uint32_t)(VISO_ROW * 2) + (uint32_t)(lane << 2)));
                __builtin_amdgcn_wave_barrier();
                int nm = 0;
                for (int b = 0; b < nu; b += VISO_WAVE) {   // nu <= MU_S8ROWS: three steps at most
                    const uint32_t i = (uint32_t)(b + lane);
                    bool sel = false;
                    if (i < (uint32_t)nu) {
                        const uint32_t e = ul[i];
                        const uint32_t q8 = s_s8[wave][k][i];   // floor(SAD8 / 128), or 255: scored exactly already
                        const int Lt = (int)(q8 << (7 + r8q)) - VISO_ROW8_SLACK(r8q);   // <= (SAD8 << s) - slack <= SAD
                        // can the candidate be ignored?  accept: it is not the minimum, does not tie it and passes :715 as the
                        // second best; reject: it is above the limit the minimum must stay below
                        const bool clear = q8 == 255u || (!mp.second ? Lt > d1k : (acck ? (Lt > d1k && (double)d1k < (double)Lt * mp.ratio) : ((double)Lt >= limk)));
                        sel = ((e >> msk) & 1u) == 0 && !clear && i != pa && i != pb;
                        if (sel) s_s8[wave][k][i] = 255;   // scored below: never again
                    }
                    const unsigned long long bal = __ballot(sel);
                    if (sel) ml[nm + mbcnt(bal)] = (uint16_t)i;
                    nm += __popcll(bal);
                }
                // EIGHT lanes per survivor, eight survivors per pass, MU_RNP passes in flight; a lane takes its two 16-byte
                // pieces of the row against the query's row in LDS (re-read per pass: kept in registers it costs the 8 the
                // kernel does not have)
                __builtin_amdgcn_wave_barrier();
                qst[lane] = qv;
                if (nm > 0 && lane < 8 * MU_RNP) ml[nm + lane] = ml[nm - 1];   // padding: scored, key masked out below
                __builtin_amdgcn_wave_barrier();
                uint32_t m1 = 0xffffffffu, m2 = 0xffffffffu;
                if (nm > 0) {
                    for (int t = 0; t * 8 < nm; t += MU_RNP) {
                        uint32_t pos[MU_RNP];
                        u32x4 a0[MU_RNP], a1[MU_RNP];
#pragma unroll
                        for (int p = 0; p < MU_RNP; ++p) {
                            pos[p] = ml[(t + p) * 8 + g8];
                            const grow_t ra = (grow_t)(wrows + (((ul[pos[p]] & 0x00ffffffu) << 1) | (uint32_t)(sub << 4)));
                            a0[p] = ra[0]; a1[p] = ra[8];
                        }
#pragma unroll
                        for (int p = 0; p < MU_RNP; ++p) {
                            int o_ = sub * 4;
                            asm volatile("" : "+v"(o_));   // not loop invariant: the query row is re-read
                            const u32x4 x0_ = *reinterpret_cast<const u32x4*>(&qst[o_]);
                            uint32_t sa = __builtin_amdgcn_sad_u16(a0[p].x, x0_.x, 0u);
                            sa = __builtin_amdgcn_sad_u16(a0[p].y, x0_.y, sa);
                            sa = __builtin_amdgcn_sad_u16(a0[p].z, x0_.z, sa);
                            sa = __builtin_amdgcn_sad_u16(a0[p].w, x0_.w, sa);
                            const u32x4 x1_ = *reinterpret_cast<const u32x4*>(&qst[o_ + 32]);
                            sa = __builtin_amdgcn_sad_u16(a1[p].x, x1_.x, sa);
                            sa = __builtin_amdgcn_sad_u16(a1[p].y, x1_.y, sa);
                            sa = __builtin_amdgcn_sad_u16(a1[p].z, x1_.z, sa);
                            sa = __builtin_amdgcn_sad_u16(a1[p].w, x1_.w, sa);
                            sa += mu_dpp<0xB1>(sa);                                          // lane ^ 1
                            sa += mu_dpp<0x4E>(sa);                                          // lane ^ 2
                            sa += (uint32_t)__builtin_amdgcn_ds_swizzle((int)sa, 0x101F);    // lane ^ 4
                            // one key per survivor (the group's first lane), none past the list's end
                            MU_UPD2(((t + p) * 8 + g8 < nm && sub == 0) ? ((sa << 9) | pos[p]) : 0xffffffffu);
                        }
                    }
                    // (min, second min) over the groups' first lanes (the only ones with keys; lane 8k reads the result): keys are distinct
#define MU_MRG2(O1, O2) do { const uint32_t o1_ = (O1), o2_ = (O2); m2 = min(max(m1, o1_), min(m2, o2_)); m1 = min(m1, o1_); } while (0)
                    MU_MRG2(mu_dpp<0x128>(m1), mu_dpp<0x128>(m2));                                                       // lane ^ 8
                    MU_MRG2((uint32_t)__builtin_amdgcn_ds_swizzle((int)m1, 0x401F), (uint32_t)__builtin_amdgcn_ds_swizzle((int)m2, 0x401F));   // lane ^ 16
                    {
                        const auto h1 = __builtin_amdgcn_permlane32_swap(m1, m1, false, false);
                        const auto h2 = __builtin_amdgcn_permlane32_swap(m2, m2, false, false);
                        m1 = h1[0]; m2 = h2[0];
                        MU_MRG2(h1[1], h2[1]);
                    }
#undef MU_MRG2
                }
                // the group's query: the survivors' keys and the two it had (keys are distinct), the verdict again
                bool again = false;
                if (g8 == k) {
                    MU_UPD2(kA);
                    MU_UPD2(kB);
                    kA = m1; kB = m2;
                    accept = !mp.second || (double)(m1 >> 9) < (double)(m2 >> 9) * mp.ratio;
                    // a reject that became an accept: the unscored candidates were only shown to be above the reject's limit —
                    // once more, with the accept's test (what is scored is marked: the second time decides)
                    again = accept && !acck;
                }
                if (__any(again)) rm |= 1ull << bit;   // wave uniform
            }
#undef MU_UPD2
            // exact tie of the minimum (largest-key rule): overflow kernel
            if (rescue && (kA >> 9) == (kB >> 9)) irregular = true;
        }
        const uint32_t kwin = min(kA, kB);
        const uint32_t d1 = kwin >> 9;
        const uint32_t ewin = ul[none ? 0u : (kwin & 511u)];   // list entry of the winner
        // in-radius candidates per query (K cap, and what a query that leaves for the overflow kernel must not count):
        // no query can have more than the list holds, so they are only needed when the list is longer than K or a
        // query leaves — bit counts over the list then; the lanes of group k keep query k's
        const bool slow = nu > K || __any(mine && irregular);   // wave uniform
        int my_cnt = 0;
        if (slow) {
            for (int b = 0; b < nu; b += VISO_WAVE) {
                const uint32_t e = (b + lane) < nu ? ul[b + lane] : 0xffffffffu;
#pragma unroll
                for (int k = 0; k < MU_G; ++k) {
                    const int c = __popcll(__ballot(((e >> (31 - k)) & 1u) == 0));
                    if ((lane >> 3) == k) my_cnt += c;
                }
            }
        } else if (!list_ovf && half == 0) {
            scored += ntest - ncnt;   // every result of the round stands: all its cells count (lanes 0..31 hold partial sums)
        }
        if (mine) {
            if (list_ovf || my_cnt > K || irregular) {
                // more than K candidates / union too long / exact tie of the minimum: overflow kernel
                P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, gq_j);
            } else {
                int idx = -1;
                if (accept) {
                    const int w = (int)((ewin & 0x00ffffffu) >> 7);   // window position of the winner
                    idx = P.t.sidx[lo + w];
                }
                P.res[gq_orig] = make_int2(idx, none ? -1 : (int)d1);
                if (slow) scored += (unsigned long long)my_cnt;
            }
        }
        MU_CLK(2 + r);
    }
    // scored pairs of the tile's queries whose result stands (partial sums in every lane)
    scored = viso_wave_sum63(scored);
    if (lane == 63 && scored) atomicAdd(P.scored, scored);
}

#ifndef MU_NO_LAUNCHER
int launch_match_union8_temporal(hipStream_t s, const BatchMatchArgs& a, long long blocks) {
    size_t pad = 0;
#ifdef VISO_DEBUG_VARIANTS   // experiment ($VISO_EXP_U8_LDS_PAD bytes of unused dynamic LDS): 6 instead of 7 workgroups per CU, so that another
    // batch's pack kernel finds LDS and wave slots beside this kernel (HISTORY.md round 6)
    { static const int p = [] { const char* e = getenv("VISO_EXP_U8_LDS_PAD"); return e ? atoi(e) : 0; }(); if (p > 0 && p < 32768) pad = (size_t)p; }
#endif
    hipLaunchKernelGGL(match_union8_kernel, dim3((unsigned)blocks), dim3(MU_THREADS), pad, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_union8_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
#endif   // MU_NO_LAUNCHER
