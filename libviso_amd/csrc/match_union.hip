// match_union.hip — temporal (no epipolar gate) matcher: one row load scored against EIGHT y-adjacent queries.
//
// The dominant kernel (97 % of the scored pairs).  It is bound by vector-ALU issue, not by memory: VALUBusy 1.0 with
// 7 waves per SIMD; HBM traffic is the compulsory u16 rows (x 1.05), the row gathers are served by the
// XCD's L2.  What it spends per useful SAD is what matters, so everything is arranged to share work between queries
// and to keep the instructions around the SADs few and free of condition codes:
//
//   tile    64 consecutive (column-bucket order) queries of one problem = one 256-thread workgroup; the +-radius column
//           window of the target image (two loads from the bucket index), its KEYPOINTS (8 B each, not its rows)
//           bucket-sorted by y in LDS
//   round   eight y-adjacent queries (ImageView::qord) per wave, two rounds per wave: their L1 diamonds overlap so much
//           that the union of the candidate sets is ~94 rows for 8 x 51.5 candidates.  Lane l carries query
//           (l & 3) + 4 * (l >> 5): a DPP quad broadcast hands every lane the four queries its half tests
//   phase 1 one scan of the y buckets the round's diamonds touch, 32 targets per step with both lane halves on the same
//           targets (lanes 0..31 test queries 0..3, lanes 32..63 queries 4..7).  Q1 cut and radius are ONE threshold on
//           the bit pattern of |dx| + |dy|; the verdict is the SIGN of (thr - 1) - bits(d), shifted into a 4-bit mask
//           by v_alignbit (no v_cmp, no select); one v_permlane32_swap joins the halves' nibbles; targets with a member
//           are appended (ballot + mbcnt) to the round's union list in LDS.  Masks are kept INVERTED (set = not a member).
//           Per-query in-radius counts (K cap) are only taken when the list is longer than K or a minimum is tied
//   phase 2 rolling pipeline over the union list, 8 lanes per row, 2 passes in flight: two global_load_dwordx4 per lane,
//           8 x v_sad_u16 against each of the eight query rows (staged per wave in LDS), a transposing reduction (three
//           exchange steps, each halving the partial sums a lane carries: every lane of the 8-lane group ends with the
//           total of ITS query; the first step, on four pairs, is two bank-masked v_add_u32_dpp per pair — a lane's
//           bit 2 is its DPP bank — instead of two selects and an add) and a packed-key tracker (key = SAD << 9 | list
//           position, or all ones for a non-member by an OR prepared when the row was requested; m2 = med3, m1 = min)
//           — no SAD ever goes to memory
//   phase 3 merge the 8 partial trackers per query across the lane groups (DPP row rotation, ds_swizzle, permlane
//           swap), ratio test in double, store
//
// Same results as the other matcher kernels.  Irregular rounds (a query with more than K in-radius candidates, a union
// list that does not fit, an exact tie of the minimum) go to match_overflow_kernel.  match_prune.hip (variant 5) puts
// exact candidate pruning in front of a cell-granular phase 2; it keeps the tile / scan code this kernel had before
// its instruction diet and is the slower of the two.
#include "common.h"
#include "match_dev.h"

#define MU_THREADS 256
#define MU_WAVES 4
#define MU_QPB 64          // queries per tile
#define MU_G 8             // queries per round
#define MU_UCAP 448        // union list entries per round (+ MU_PAD < 512: a list position fits the 9 low bits of a tracker key)
#define MU_PAD 32          // list padding: the pipeline runs up to 4 passes of 8 rows past the end
#ifndef MU_NP
#define MU_NP 2            // passes in flight (3 measured slower even without spills)
#endif
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

// Running order statistics of one query's SADs as two packed keys, key = SAD << 9 | position in the round's union
// list (< 512; SAD < 2^23 for int16-valued descriptors: 121 * 65535): m1 = smallest key, m2 = second smallest
// (0xffffffff = none).  A query sees every list position at most once, so keys are distinct and
//   min SAD = m1 >> 9,  second smallest SAD counting multiplicity = m2 >> 9,  exact tie of the minimum <=> equal SAD fields.
struct MuTrack { uint32_t m1, m2; };

__device__ __forceinline__ void mu_update(MuTrack& t, uint32_t key) {
    uint32_t med;   // m1 <= m2: the median of (m1, m2, key) is the new second smallest
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(t.m1), "v"(t.m2), "v"(key));
    t.m2 = med;
    t.m1 = min(t.m1, key);
}

__device__ __forceinline__ void mu_merge(MuTrack& a, const MuTrack& b) {
    const uint32_t hi = max(a.m1, b.m1);
    a.m2 = min(hi, min(a.m2, b.m2));
    a.m1 = min(a.m1, b.m1);
}

// y bucket of the staged window: monotone in y, total (NaN -> 0, +-inf saturate): v_cvt_i32_f32 truncates, saturates and
// turns NaN into 0 (the C conversion would be undefined there), the clamp makes truncation and floor the same thing
__device__ __forceinline__ int mu_ybucket(float y, float y0, float scale) {
    const float f = (y - y0) * scale;
    int b;
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(b) : "v"(f));
    return min(max(b, 0), MU_NBY - 1);
}

// the lane that carries query k of a round (lanes 0..31: queries 0..3, lanes 32..63: queries 4..7, repeated every four
// lanes, so that a quad broadcast hands every lane the four queries its half tests in phase 1)
__device__ __forceinline__ constexpr int mu_qlane(int k) { return (k & 3) + 32 * (k >> 2); }

__global__ __attribute__((amdgpu_waves_per_eu(7, 8))) __launch_bounds__(MU_THREADS) void match_union_kernel(BatchMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_ul[MU_WAVES][MU_UCAP + MU_PAD];
    __shared__ __attribute__((aligned(16))) uint32_t s_qrow[MU_WAVES][MU_G][64];
    __shared__ float2 s_ykp[MU_KPCAP];       // staged window keypoints in y-bucket order
    __shared__ uint16_t s_ypos[MU_KPCAP];    // their window positions
    __shared__ int s_ys[MU_NBY + 1];         // bucket counts, then bucket starts
    __shared__ float s_xr[2];
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
        float mn = qx, mx = qx;
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, m));
            mx = fmaxf(mx, __shfl_xor(mx, m));
        }
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
    // window base (scalar) + 32-bit byte offset per lane: (window position << 8) | (sub << 4)
    const gbytes_t wrows = (gbytes_t)reinterpret_cast<const char*>(P.t.rows) + (size_t)lo * (VISO_ROW * 2);
    uint32_t* ul = s_ul[wave];
    const int sub = lane & 7;
    const int half = lane >> 5;          // phase 1: lanes 0..31 test queries 0..3 of the round, lanes 32..63 queries 4..7
    unsigned long long scored = 0;
    constexpr int ROUNDS = MU_QPB / (MU_WAVES * MU_G);   // 2 rounds of 8 queries per wave
    // which of the round's eight queries the lane tracks after the transposing reduction (phase 2): the partners of
    // the three exchange steps (lane ^ 4, lane ^ 2, lane ^ 1 within the 8-lane group) differ in exactly one of these
    const bool sel1 = ((lane >> 1) & 1) != 0, sel0 = (lane & 1) != 0;
    const int myq = ((lane >> 2) & 1) + (sel1 ? 2 : 0) + (sel0 ? 4 : 0);
    const int msh = 31 - myq;   // membership bit of query myq in a list entry (bit 7 - k of the mask byte)

    // query data one round ahead: lane l carries local index / keypoint / original index of query (l & 3) + 4 * half
    // of the round (mu_qlane)
    const int qslot = (lane & 3) + 4 * half;
    int pli;
    float2 pq;
    int po;
#define MU_PREFETCH(R)                                                                                    \
    do {                                                                                                  \
        const int base_ = wave * (MU_QPB / MU_WAVES) + (R) * MU_G;                                        \
        pli = (int)P.q.qord[q0 + base_ + qslot];                                                          \
        const int j_ = q0 + pli;                                                                          \
        const int jc_ = min(j_, q1 - 1);                                                                  \
        pq = P.q.skp[jc_];                                                                                \
        po = j_ < q1 ? P.q.sidx[jc_] : -1;                                                                \
    } while (0)
    MU_PREFETCH(0);

    for (int r = 0; r < ROUNDS; ++r) {
        // ---------------- round setup.  Lane l holds query (l & 3) + 4 * half: the four queries its half tests in phase 1
        // are the four lanes of its quad (DPP quad broadcasts, no scalar traffic)
        const int my_orig = po, my_j = q0 + pli;   // lanes mu_qlane(k): the round's queries, for phase 3
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
        // the eight query rows: one word per lane and row from global memory (the loads land during the scan), then LDS
        uint32_t qw[MU_G];
#pragma unroll
        for (int k = 0; k < MU_G; ++k) {
            const int jk = q0 + __builtin_amdgcn_readlane(pli, mu_qlane(k));
            qw[k] = ((const __attribute__((address_space(1))) uint32_t*)reinterpret_cast<const uint32_t*>(P.q.rows))[(size_t)min(jk, q1 - 1) * (VISO_ROW / 2) + lane];
        }
        if (r + 1 < ROUNDS) MU_PREFETCH(r + 1);
        // ---------------- phase 1: one scan over the y buckets the round's diamonds touch, 32 targets per step: both
        // halves read the same 32 entries, each tests its four queries (sign of bits(d) - thr shifted into a 4-bit mask:
        // sub + alignbit, no condition code), one v_permlane32_swap joins the halves' nibbles into the 8-bit membership
        // mask (bit 7 - k = query k) in every lane; targets with a non-zero mask go to the union list (lanes 0..31
        // write).  entry = inverted mask << 24 | window position << 8.  ncnt accumulates the set bits = (query, target)
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
                if (m8a != 0xffu && half == 0) ul[min(ucnt + (int)__builtin_amdgcn_mbcnt_lo(ua, 0u), MU_UCAP - 1)] = (m8a << 24) | (pa << 8);
                if (m8b != 0xffu && half == 0) ul[min(ucnt + ca + (int)__builtin_amdgcn_mbcnt_lo(ub, 0u), MU_UCAP - 1)] = (m8b << 24) | (pb << 8);
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
                if (m8 != 0xffu && half == 0) ul[min(ucnt + (int)__builtin_amdgcn_mbcnt_lo(u, 0u), MU_UCAP - 1)] = (m8 << 24) | ((uint32_t)w << 8);
                ucnt += __popc(u);
            }
        }
#undef MU_TEST4
        const bool list_ovf = ucnt > MU_UCAP;
        const int nu = list_ovf ? 0 : ucnt;
        __builtin_amdgcn_wave_barrier();
        // padding behind the list: copies of the last entry with nobody's membership (scored, never counted)
        if (nu > 0 && lane < MU_PAD) ul[nu + lane] = ul[nu - 1] | 0xff000000u;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < MU_G; ++k) s_qrow[wave][k][lane] = qw[k];
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2: rolling pipeline over the union list: load a row once, SAD against the eight query
        // rows (staged per wave in LDS), transposing reduction (8 partial sums per lane -> every lane of the 8-lane
        // group holds the total of ITS query: 14 selects + 7 DPP adds), packed-key tracker (2 instructions)
        MuTrack tr;
        tr.m1 = 0xffffffffu; tr.m2 = 0xffffffffu;
        {
            const int npass = (nu + 7) >> 3;
            // the lane group's list position, derived HERE: as a loop invariant of the round loop its LDS address stays
            // live across phase 1 and is the one register too many (a spill: scratch memory for every wave of the launch)
            int g8 = lane;
            asm volatile("" : "+v"(g8));
            g8 >>= 3;
            u32x4 r0[MU_NP], r1[MU_NP];
            uint32_t un[MU_NP];   // the pass's list position for the tracker key, all ones where the lane's query is not a member
#define MU_ISSUE(SLOT, T)                                                                                  \
            do {                                                                                           \
                const uint32_t ent_ = ul[(T) * 8 + g8];                                                    \
                un[SLOT] = (uint32_t)((T) * 8 + g8) | (uint32_t)__builtin_amdgcn_sbfe((int)ent_, (uint32_t)msh, 1u); \
                const grow_t row_ = (grow_t)(wrows + ((ent_ & 0x00ffffffu) | (uint32_t)(sub << 4)));       \
                r0[SLOT] = row_[0];                                                                        \
                r1[SLOT] = row_[8];                                                                        \
            } while (0)
#define MU_SAD(K, SLOT)                                                                                    \
            ({                                                                                             \
                const u32x4 qa_ = *reinterpret_cast<const u32x4*>(&s_qrow[wave][K][sub * 4]);           \
                const u32x4 qb_ = *reinterpret_cast<const u32x4*>(&s_qrow[wave][K][sub * 4 + 32]);      \
                uint32_t s_ = __builtin_amdgcn_sad_u16(r0[SLOT].x, qa_.x, 0u);                             \
                s_ = __builtin_amdgcn_sad_u16(r0[SLOT].y, qa_.y, s_);                                      \
                s_ = __builtin_amdgcn_sad_u16(r0[SLOT].z, qa_.z, s_);                                      \
                s_ = __builtin_amdgcn_sad_u16(r0[SLOT].w, qa_.w, s_);                                      \
                s_ = __builtin_amdgcn_sad_u16(r1[SLOT].x, qb_.x, s_);                                      \
                s_ = __builtin_amdgcn_sad_u16(r1[SLOT].y, qb_.y, s_);                                      \
                s_ = __builtin_amdgcn_sad_u16(r1[SLOT].z, qb_.z, s_);                                      \
                s_ = __builtin_amdgcn_sad_u16(r1[SLOT].w, qb_.w, s_);                                      \
                s_;                                                                                        \
            })
#define MU_X1(A, B) ({ uint32_t k_ = sel0 ? (B) : (A); const uint32_t g_ = sel0 ? (A) : (B); k_ += mu_dpp<0xB1>(g_); k_; })   /* lane ^ 1 */
#define MU_X2(A, B) ({ uint32_t k_ = sel1 ? (B) : (A); const uint32_t g_ = sel1 ? (A) : (B); k_ += mu_dpp<0x4E>(g_); k_; })   /* lane ^ 2 */
            // first exchange step, lane ^ 4, on all four pairs at once: a lane's bit 2 is its DPP BANK, so "keep A and add
            // the partner's A" / "keep B and add the partner's B" are two bank-masked v_add_u32_dpp instead of two selects
            // and an add: banks 0, 2 take A + A[lane + 4] (row_shl:4), banks 1, 3 take B + B[lane - 4] (row_shr:4), in
            // place.  One asm block: the compiler's hazard recognizer does not see DPP reads inside inline asm (a VGPR
            // written by the previous two VALU instructions must not be a DPP source), hence the leading s_nop 1; inside
            // the block every source was written at least four instructions earlier.
#define MU_X4x4(S0, S1, S2, S3, S4, S5, S6, S7)                                                            \
            asm("s_nop 1\n\t"                                                                              \
                "v_add_u32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                         \
                "v_add_u32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                         \
                "v_add_u32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                         \
                "v_add_u32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                         \
                "v_add_u32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"                         \
                "v_add_u32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"                         \
                "v_add_u32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"                         \
                "v_add_u32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa"                              \
                : "+v"(S0), "+v"(S2), "+v"(S4), "+v"(S6) : "v"(S1), "v"(S3), "v"(S5), "v"(S7))
#define MU_REDUCE(SLOT)                                                                                   \
            do {                                                                                           \
                uint32_t s0_ = MU_SAD(0, SLOT), s1_ = MU_SAD(1, SLOT), s2_ = MU_SAD(2, SLOT), s3_ = MU_SAD(3, SLOT); \
                __builtin_amdgcn_sched_barrier(0);   /* two halves: all sixteen query-row reads in flight at once cost 64 VGPRs */ \
                uint32_t s4_ = MU_SAD(4, SLOT), s5_ = MU_SAD(5, SLOT), s6_ = MU_SAD(6, SLOT), s7_ = MU_SAD(7, SLOT); \
                MU_X4x4(s0_, s1_, s2_, s3_, s4_, s5_, s6_, s7_);   /* s0_ s2_ s4_ s6_: queries (0|1) (2|3) (4|5) (6|7) by bit 2 */ \
                const uint32_t c0_ = MU_X2(s0_, s2_), c1_ = MU_X2(s4_, s6_);                               \
                const uint32_t m_ = MU_X1(c0_, c1_);                                                       \
                mu_update(tr, (m_ << 9) | un[SLOT]);                                                       \
            } while (0)
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < MU_NP; ++p) MU_ISSUE(p, p);
            }
            int t = 0;
            for (; t + MU_NP < npass; t += MU_NP) {   // steady state: no branch between reduce and refill (the hardware
#pragma unroll                                        // counts outstanding loads; a branch would make the compiler drain them)
                for (int p = 0; p < MU_NP; ++p) {
                    MU_REDUCE(p);
                    // keep the refill of this slot HERE: left alone the scheduler sinks all refills to the end of the
                    // loop body, where the next iteration waits for them at once (no load is in flight during a reduce)
                    __builtin_amdgcn_sched_barrier(0);
                    MU_ISSUE(p, t + p + MU_NP);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (npass > 0) {   // the last one or two passes (an odd count would otherwise score a pass of padding)
                MU_REDUCE(0);
                if (MU_NP > 1 && t + 1 < npass) MU_REDUCE(MU_NP > 1 ? 1 : 0);
            }
#undef MU_REDUCE
#undef MU_X4x4
#undef MU_X2
#undef MU_X1
#undef MU_SAD
#undef MU_ISSUE
        }
        // ---------------- phase 3: merge the 8 lane groups (lanes with equal position in the group track the same
        // query); lane l (< 8) then holds query myq(l): 0 4 2 6 1 5 3 7 -> bring query k to the lane that carries its
        // data (mu_qlane); fetch the original target index, ratio test, store
        {   // lane ^ 8: rotation by 8 within the row of 16 (DPP); lane ^ 16: ds_swizzle; lane ^ 32: v_permlane32_swap hands
            // every lane both halves' values
            MuTrack o;
            o.m1 = mu_dpp<0x128>(tr.m1);
            o.m2 = mu_dpp<0x128>(tr.m2);
            mu_merge(tr, o);
            o.m1 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)tr.m1, 0x401F);
            o.m2 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)tr.m2, 0x401F);
            mu_merge(tr, o);
            const auto h1 = __builtin_amdgcn_permlane32_swap(tr.m1, tr.m1, false, false);
            const auto h2 = __builtin_amdgcn_permlane32_swap(tr.m2, tr.m2, false, false);
            tr.m1 = h1[0]; tr.m2 = h2[0];
            o.m1 = h1[1]; o.m2 = h2[1];
            mu_merge(tr, o);
        }
        {
            const int k = qslot;   // the lane of query k after the reduction: bits (k & 1, k >> 1 & 1, k >> 2) -> lane bits (2, 1, 0)
            const int src = ((k & 1) << 2) | (k & 2) | (k >> 2);
            tr.m1 = (uint32_t)__shfl((int)tr.m1, src);
            tr.m2 = (uint32_t)__shfl((int)tr.m2, src);
        }
        const bool mine = (lane & 28) == 0 && my_orig >= 0;   // lanes mu_qlane(k) of live queries
        const bool none = tr.m1 == 0xffffffffu;
        const uint32_t d1 = tr.m1 >> 9;
        const bool tie = !none && tr.m2 != 0xffffffffu && (tr.m2 >> 9) == d1;
        // in-radius candidates per query (K cap, and what a query that leaves for the overflow kernel must not count):
        // no query can have more than the list holds, so they are only needed when the list is longer than K or a
        // minimum is tied — bit counts over the list then; lane mu_qlane(k) keeps query k's
        const bool slow = nu > K || __any(mine && tie);   // wave uniform
        int my_cnt = 0;
        if (slow) {
            for (int b = 0; b < nu; b += VISO_WAVE) {
                const uint32_t e = (b + lane) < nu ? ul[b + lane] : 0xffffffffu;
#pragma unroll
                for (int k = 0; k < MU_G; ++k) {
                    const int c = __popcll(__ballot(((e >> (31 - k)) & 1u) == 0));
                    if (lane == mu_qlane(k)) my_cnt += c;
                }
            }
        } else if (!list_ovf && half == 0) {
            scored += ntest - ncnt;   // every result of the round stands: all its cells count (lanes 0..31 hold partial sums)
        }
        if (mine) {
            if (list_ovf || my_cnt > K || tie) {
                // more than K candidates / union too long / exact tie of the minimum (largest-key rule): overflow kernel
                P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, my_j);
            } else {
                bool accept = !none;
                int idx = -1;
                if (accept) {
                    const int w = (int)((ul[tr.m1 & 511u] >> 8) & 0xffffu);   // window position of the winner
                    idx = P.t.sidx[lo + w];
                    if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                        const double bd2 = tr.m2 == 0xffffffffu ? 1.7976931348623157e308 : (double)(tr.m2 >> 9);
                        accept = (double)d1 < bd2 * mp.ratio;
                    }
                }
                P.res[my_orig] = make_int2(accept ? idx : -1, none ? -1 : (int)d1);
                if (slow) scored += (unsigned long long)my_cnt;
            }
        }
    }
    // scored pairs of the tile's queries whose result stands (partial sums in every lane)
#pragma unroll
    for (int m = 1; m < VISO_WAVE; m <<= 1) scored += (unsigned long long)__shfl_xor((long long)scored, m);
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
}

// stereo problems keep match_batch_kernel<1> (the fp64 Sampson gate leaves ~3 pairs per query: nothing to share)
int launch_match_union_temporal(hipStream_t s, const BatchMatchArgs& a, long long blocks) {
    hipLaunchKernelGGL(match_union_kernel, dim3((unsigned)blocks), dim3(MU_THREADS), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_union_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
