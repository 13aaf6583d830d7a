// match_union.hip — temporal (no epipolar gate) matcher, "one row load, four queries".
//
// The matcher is bound by the texture-address unit: every candidate pair costs two 16-B gathers per lane
// (match_batch.hip: TA busy ~80 %).  Queries that are close in the image share most of their candidates, so
// this kernel scores a loaded target row against FOUR queries at once:
//
//   tile    64 x-adjacent queries (one workgroup, 4 waves) and their target window in LDS, as match_batch.hip;
//           the tile's queries are additionally ranked by y, and every wave works on rounds of four
//           y-adjacent queries (their L1 diamonds overlap by ~2/3: the union of the four candidate sets is
//           about a third of their sum)
//   phase 1 one scan of the window for the round: per target a 4-bit membership mask (which of the four
//           queries has it in radius, Q1 cut included); targets with a non-zero mask go to the union list
//   phase 2 rolling pipeline over the union list, 8 lanes per row: load the row once, SAD against the four
//           query rows (staged per wave in LDS: 8 x ds_read_b128 per pass cost less than the 32 registers that
//           would hold them — 6 instead of 5 waves per SIMD), lanes sub = 0..3 of each 8-lane group keep a
//           running (min, second min with multiplicity, argmin, tie) for "their" query — no SAD goes to memory
//   phase 3 merge the 8 partial trackers per query across the lane groups, ratio test, store
//
// Same results as the other matcher kernels.  Irregular rounds (a query with more than K in-radius
// candidates, a union list that does not fit, an exact tie of the minimum) go to match_overflow_kernel.
#include "common.h"
#include "match_dev.h"

#define MU_THREADS 256
#define MU_WAVES 4
#define MU_QPB 64          // queries per tile
#define MU_G 4             // queries per round
#define MU_UCAP 256        // union list entries per round
#define MU_PAD 32          // list padding: the pipeline runs up to 4 passes of 8 rows past the end
#define MU_NP 2            // passes in flight (3 measured slower even without spills)
#define MU_KPCAP 512       // window keypoints staged in LDS

template <int CTRL>
__device__ __forceinline__ uint32_t mu_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// bits of |qx - tx| + |qy - ty| (cvflann::L1 order, see l1_kp); abs as source modifiers of the add
__device__ __forceinline__ uint32_t mu_l1_bits(float qx, float qy, float2 t) {
    const float dx = qx - t.x, dy = qy - t.y;
    float d;
    asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(d) : "v"(dx), "v"(dy));
    return __float_as_uint(d);
}

// running order statistics of one query's SADs: d1 = min, d2 = second smallest counting multiplicity,
// w = window position of the (first seen) minimum, tie = the minimum was seen more than once
struct MuTrack { uint32_t d1, d2, w, tie; };

__device__ __forceinline__ void mu_update(MuTrack& t, uint32_t s, uint32_t w) {
    const bool lt = s < t.d1, eq = s == t.d1;
    const uint32_t m2 = min(t.d2, s);
    t.d2 = (s <= t.d1) ? t.d1 : m2;
    t.w = lt ? w : t.w;
    t.tie = lt ? 0u : (eq ? 1u : t.tie);
    t.d1 = min(t.d1, s);
}

__device__ __forceinline__ void mu_merge(MuTrack& a, const MuTrack& b) {
    const bool lt = b.d1 < a.d1, eq = b.d1 == a.d1;
    // second smallest of the union: the smaller minimum's d2 against the larger minimum
    const uint32_t d2 = eq ? a.d1 : (lt ? min(b.d2, a.d1) : min(a.d2, b.d1));
    a.tie = eq ? 1u : (lt ? b.tie : a.tie);
    a.w = lt ? b.w : a.w;
    a.d1 = min(a.d1, b.d1);
    a.d2 = d2;
}

__global__ __attribute__((amdgpu_waves_per_eu(6, 8))) __launch_bounds__(MU_THREADS) void match_union_kernel(BatchMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_ul[MU_WAVES][MU_UCAP + MU_PAD];
    __shared__ __attribute__((aligned(16))) uint32_t s_qrow[MU_WAVES][MU_G][64];
    __shared__ float2 s_kp[MU_KPCAP];
    __shared__ int s_idx[MU_KPCAP];
    __shared__ int s_qord[MU_QPB];
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
    // ---- tile: x range (window) and y ranks (round composition) of its queries
    if (wave == 0) {
        const bool live = q0 + lane < q1;
        const float2 qv = live ? P.q.skp[q0 + lane] : make_float2(__builtin_nanf(""), __builtin_nanf(""));
        float mn = qv.x, mx = qv.x;
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, m));
            mx = fmaxf(mx, __shfl_xor(mx, m));
        }
        if (lane == 0) { s_xr[0] = mn; s_xr[1] = mx; }
        // rank by (y, lane): any total order gives a valid permutation, this one puts neighbours in y together
        const uint32_t yb = __float_as_uint(qv.y);
        const uint32_t key = live ? (yb ^ ((yb >> 31) ? 0xffffffffu : 0x80000000u)) : 0xffffffffu;
        int rank = 0;
        for (int m = 0; m < VISO_WAVE; ++m) {
            const uint32_t km = (uint32_t)__builtin_amdgcn_readlane((int)key, m);
            rank += (km < key || (km == key && m < lane)) ? 1 : 0;
        }
        s_qord[rank] = lane;
    }
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
    for (int w = threadIdx.x; w < wpad; w += MU_THREADS) {
        float2 t2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
        if (w < wcap) { t2 = P.t.skp[lo + w]; s_idx[w] = P.t.sidx[lo + w]; }
        s_kp[w] = t2;
    }
    float2 kp0 = make_float2(0.f, 0.f);
    const bool has0 = n2 > 0;
    if (has0) kp0 = P.t.skp[P.t.rank[0]];
    __syncthreads();
    const float radius = mp.radius;
    const int K = mp.K;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* grow_t;
    typedef const __attribute__((address_space(1))) char* gbytes_t;
    // window base (scalar) + 32-bit byte offset per lane: (window position << 8) | (sub << 4)
    const gbytes_t wrows = (gbytes_t)reinterpret_cast<const char*>(P.t.rows) + (size_t)lo * (VISO_ROW * 2);
    uint32_t* ul = s_ul[wave];
    const int g8 = lane >> 3, sub = lane & 7;
    unsigned long long scored = 0;
    constexpr int ROUNDS = MU_QPB / (MU_WAVES * MU_G);   // 4 rounds per wave

    // query data one round ahead: lane l carries local index / keypoint / original index of query (l & 3) of the round
    int pli;
    float2 pq;
    int po;
#define MU_PREFETCH(R)                                                                                    \
    do {                                                                                                  \
        const int base_ = wave * (MU_QPB / MU_WAVES) + (R) * MU_G;                                        \
        pli = s_qord[base_ + (lane & (MU_G - 1))];                                                        \
        const int j_ = q0 + pli;                                                                          \
        const int jc_ = min(j_, q1 - 1);                                                                  \
        pq = P.q.skp[jc_];                                                                                \
        po = j_ < q1 ? P.q.sidx[jc_] : -1;                                                                \
    } while (0)
    MU_PREFETCH(0);

    for (int r = 0; r < ROUNDS; ++r) {
        // ---------------- round setup: scalars of the four queries, their rows into registers
        float2 qk[MU_G];
        int orig[MU_G], jq[MU_G], cnt[MU_G];
        uint32_t thr[MU_G];
        bool any_live = false;
#pragma unroll
        for (int k = 0; k < MU_G; ++k) {
            qk[k].x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.x), k));
            qk[k].y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.y), k));
            orig[k] = __builtin_amdgcn_readlane(po, k);
            jq[k] = q0 + __builtin_amdgcn_readlane(pli, k);
            cnt[k] = 0;
            // d = |dx| + |dy| is +0, positive or NaN: its bit pattern orders like the value and NaNs are above +inf,
            // so (d <= radius && d < d0cut) is one unsigned compare against bits(d0) (target 0 in radius: Q1,
            // src/viso.cpp:693) or bits(radius) + 1.  Dead slots (past the tile) get 0: nothing passes.
            uint32_t t = __float_as_uint(radius) + 1u;
            if (has0) {
                const float d0 = l1_kp(qk[k].x, qk[k].y, kp0);
                if (d0 <= radius) t = __float_as_uint(d0);
            }
            thr[k] = orig[k] >= 0 ? t : 0u;
            any_live = any_live || orig[k] >= 0;
        }
        if (r + 1 < ROUNDS) MU_PREFETCH(r + 1);
        if (!any_live) continue;   // wave uniform
        // query rows: one word per lane and row from global memory (the loads land during the scan), then LDS
        uint32_t qw[MU_G];
#pragma unroll
        for (int k = 0; k < MU_G; ++k)
            qw[k] = ((const __attribute__((address_space(1))) uint32_t*)reinterpret_cast<const uint32_t*>(P.q.rows))[(size_t)min(jq[k], q1 - 1) * (VISO_ROW / 2) + lane];
        // ---------------- phase 1: one scan, membership masks, union list.  entry = mask << 28 | position << 8
        // (mask bit 3 - k = query k)
        int ucnt = 0;
        for (int base = 0; base < wpad; base += 2 * VISO_WAVE) {
            const float2 ta = s_kp[base + lane], tb = s_kp[base + VISO_WAVE + lane];
            uint32_t ma = 0, mb = 0;
#pragma unroll
            for (int k = 0; k < MU_G; ++k) {
                const bool ina = mu_l1_bits(qk[k].x, qk[k].y, ta) < thr[k];
                const bool inb = mu_l1_bits(qk[k].x, qk[k].y, tb) < thr[k];
                cnt[k] += __popcll(__ballot(ina)) + __popcll(__ballot(inb));
                ma = ma + ma + (ina ? 1u : 0u);
                mb = mb + mb + (inb ? 1u : 0u);
            }
            const unsigned long long ua = __ballot(ma != 0), ub = __ballot(mb != 0);
            const int ca = __popcll(ua);
            const uint32_t ea = (uint32_t)(base + lane) << 8;
            if (ma) ul[min(ucnt + mbcnt(ua), MU_UCAP - 1)] = (ma << 28) | ea;
            if (mb) ul[min(ucnt + ca + mbcnt(ub), MU_UCAP - 1)] = (mb << 28) | (ea + (VISO_WAVE << 8));
            ucnt += ca + __popcll(ub);
        }
        for (int base = wcap; base < W; base += VISO_WAVE) {   // windows wider than MU_KPCAP (dense data only)
            const int w = base + lane;
            float2 t2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
            if (w < W) t2 = P.t.skp[lo + w];
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < MU_G; ++k) {
                const bool in = mu_l1_bits(qk[k].x, qk[k].y, t2) < thr[k];
                cnt[k] += __popcll(__ballot(in));
                m = m + m + (in ? 1u : 0u);
            }
            const unsigned long long u = __ballot(m != 0);
            if (m) ul[min(ucnt + mbcnt(u), MU_UCAP - 1)] = (m << 28) | ((uint32_t)w << 8);
            ucnt += __popcll(u);
        }
        int flags = 0;   // bit k: query k is left to the overflow kernel
#pragma unroll
        for (int k = 0; k < MU_G; ++k)
            if (orig[k] >= 0 && (cnt[k] > K || ucnt > MU_UCAP)) flags |= 1 << k;
        const int nu = ucnt > MU_UCAP ? 0 : ucnt;
        // padding behind the list: copies of the last entry with an empty mask (scored, never counted)
        __builtin_amdgcn_wave_barrier();
        if (nu > 0 && lane < MU_PAD) ul[nu + lane] = ul[nu - 1] & 0x0fffffffu;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < MU_G; ++k) s_qrow[wave][k][lane] = qw[k];
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2: rolling pipeline over the union list
        const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
        const int msh = 31 - (lane & 3);   // membership bit of the query this lane tracks (lanes sub and sub + 4 both track query sub & 3)
        MuTrack tr;
        tr.d1 = 0xffffffffu; tr.d2 = 0xffffffffu; tr.w = 0; tr.tie = 0;
        {
            const int npass = (nu + 7) >> 3;
            u32x4 r0[MU_NP], r1[MU_NP];
            uint32_t ent[MU_NP];
#define MU_ISSUE(SLOT, T)                                                                                  \
            do {                                                                                           \
                ent[SLOT] = ul[(T) * 8 + g8];                                                              \
                const grow_t row_ = (grow_t)(wrows + ((ent[SLOT] & 0x0fffffffu) | (uint32_t)(sub << 4)));  \
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
            // four partial SADs per lane, reduced over the 8 lanes of the group together (their DPP steps
            // interleave), then lane sub = k takes query k's total and its membership bit and updates its tracker
#define MU_REDUCE(SLOT)                                                                                   \
            do {                                                                                           \
                uint32_t s0_ = MU_SAD(0, SLOT), s1_ = MU_SAD(1, SLOT), s2_ = MU_SAD(2, SLOT), s3_ = MU_SAD(3, SLOT); \
                s0_ += mu_dpp<0xB1>(s0_); s1_ += mu_dpp<0xB1>(s1_); s2_ += mu_dpp<0xB1>(s2_); s3_ += mu_dpp<0xB1>(s3_); \
                s0_ += mu_dpp<0x4E>(s0_); s1_ += mu_dpp<0x4E>(s1_); s2_ += mu_dpp<0x4E>(s2_); s3_ += mu_dpp<0x4E>(s3_); \
                s0_ += mu_dpp<0x141>(s0_); s1_ += mu_dpp<0x141>(s1_); s2_ += mu_dpp<0x141>(s2_); s3_ += mu_dpp<0x141>(s3_); \
                const uint32_t lo_ = lb0 ? s1_ : s0_, hi_ = lb0 ? s3_ : s2_;                               \
                const uint32_t mine_ = lb1 ? hi_ : lo_;   /* lanes sub and sub + 4 both track query sub & 3 */ \
                const bool member_ = ((ent[SLOT] >> msh) & 1u) != 0;                                       \
                mu_update(tr, member_ ? mine_ : 0xffffffffu, (ent[SLOT] >> 8) & 0xfffffu);                 \
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
                    MU_ISSUE(p, t + p + MU_NP);
                }
            }
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < MU_NP; ++p) MU_REDUCE(p);
            }
#undef MU_REDUCE
#undef MU_SAD
#undef MU_ISSUE
        }
        // ---------------- phase 3: merge the 8 lane groups (lanes with equal sub), lane k of group 0 ends up with
        // query k; fetch the original target index, ratio test, store
#pragma unroll
        for (int m = 8; m < VISO_WAVE; m <<= 1) {
            MuTrack o;
            o.d1 = (uint32_t)__shfl_xor((int)tr.d1, m);
            o.d2 = (uint32_t)__shfl_xor((int)tr.d2, m);
            o.w = (uint32_t)__shfl_xor((int)tr.w, m);
            o.tie = (uint32_t)__shfl_xor((int)tr.tie, m);
            mu_merge(tr, o);
        }
        // lane k (< 4) now holds query k
        {
            int my_orig = -1, my_j = 0, my_cnt = 0;
            bool my_flag = false;
#pragma unroll
            for (int k = 0; k < MU_G; ++k)
                if (lane == k) { my_orig = orig[k]; my_j = jq[k]; my_cnt = cnt[k]; my_flag = (flags >> k) & 1; }
            if (lane < MU_G && my_orig >= 0) {
                const bool none = tr.d1 == 0xffffffffu;
                if (my_flag || (!none && tr.tie)) {
                    // more than K candidates / union too long / exact tie of the minimum (largest-key rule): overflow kernel
                    P.ovf[atomicAdd(P.ovf_cnt, 1)] = my_j;
                } else {
                    bool accept = !none;
                    int idx = -1;
                    if (accept) {
                        if ((int)tr.w < wcap) idx = s_idx[tr.w]; else idx = P.t.sidx[lo + (int)tr.w];
                        if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                            const double bd2 = tr.d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)tr.d2;
                            accept = (double)tr.d1 < bd2 * mp.ratio;
                        }
                    }
                    P.res[my_orig] = make_int2(accept ? idx : -1, (int)tr.d1);
                    scored += (unsigned long long)my_cnt;
                }
            }
        }
    }
    // scored pairs of the tile's queries whose result stands (lanes 0..3 of every wave hold partial sums)
#pragma unroll
    for (int m = 1; m < MU_G; m <<= 1) scored += (unsigned long long)__shfl_xor((long long)scored, m);
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
}

// stereo problems keep match_batch_kernel<1> (the fp64 Sampson gate leaves ~3 pairs per query: nothing to share)
int launch_match_union_temporal(hipStream_t s, const BatchMatchArgs& a, long long blocks) {
    hipLaunchKernelGGL(match_union_kernel, dim3((unsigned)blocks), dim3(MU_THREADS), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_union_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
