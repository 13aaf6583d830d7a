// match_strip.hip — temporal (no epipolar gate) matcher with the target window's descriptor rows RESIDENT IN LDS.
//
// match_union_kernel (match_union.hip) gathers every candidate row from the XCD's L2, 8 lanes x 2 x 16 B per row:
// its scoring phase sits on the chip's L2 row-gather rate (~16-18 TB/s, MI355X_MICROARCH.md "Indexed rows").  The
// rows a tile of x-adjacent queries can ever need are the CONTIGUOUS range [lo, lo + W) of the x-sorted target
// image (its +-radius column window), ~390 rows = 100 KB for 128 queries at 2000 keypoints/image: they fit the
// CU's 160 KB of LDS.  So this kernel
//
//   tile    128 x-adjacent queries = two sub-blocks of 64, each ranked by y (rounds of four y-adjacent queries share
//           ~2/3 of their candidates); one workgroup of 16 waves per tile, ONE workgroup per CU;
//   stage   copies the window's rows global -> LDS once, as a linear asynchronous copy (global_load_lds_dwordx4,
//           1 KiB per wave-instruction, no VGPRs), 3x less L2 traffic than the gathers and none of it indexed;
//           windows wider than the LDS budget (dense keypoints) are processed in chunks of MS_WCAP rows with the
//           per-query order statistics carried in LDS;
//   phase 1 per round one scan of the chunk's keypoints: 4-bit membership masks, union list (as match_union_kernel);
//   phase 2 rolling pipeline over the union list, 8 lanes per row, rows read from LDS with ds_read_b128 (lanes with
//           bit 4 set read the two 128-B halves of a row in the opposite order, which makes every 16-lane service
//           group of the instruction hit 64 distinct banks), the four query rows held in registers, 8 x v_sad_u16 per
//           query, a transposing 3-step DPP reduction (10 instructions for the four sums), running
//           (min, second min with multiplicity, argmin, tie) per query in registers — no SAD ever goes to memory;
//   phase 3 merge across lane groups and chunks, ratio test, store.
//
// Same results as the other matcher kernels (bit-exact, parity tests).  Irregular queries (more than K in-radius
// candidates, a union list that does not fit, an exact tie of the minimum) go to match_overflow_kernel.
#include "common.h"
#include "match_dev.h"

#include <stdlib.h>

#define MS_THREADS 1024
#define MS_WAVES 16
#define MS_QPB 128         // queries per tile
#define MS_SUB 64          // queries per y-ranked sub-block
#define MS_G 4             // queries per round
#define MS_ROUNDS (MS_QPB / (MS_WAVES * MS_G))   // 2 rounds per wave
#define MS_WCAP 480        // window rows resident in LDS per chunk (120 KB)
#define MS_KPCAP 512       // keypoint slots (MS_WCAP padded to a multiple of 128 with NaNs)
#define MS_UCAP 256        // union list entries per round
#define MS_PAD 32          // list padding: the pipeline runs passes of 8 rows past the end
#define MS_NP 2            // passes in flight

// LDS carve (bytes); the rows sit at offset 0 so that (window position << 8) IS the LDS address of a row
#define MS_OFF_ROWS 0
#define MS_OFF_KP (MS_OFF_ROWS + MS_WCAP * 256)
#define MS_OFF_UL (MS_OFF_KP + MS_KPCAP * 8)
#define MS_OFF_QORD (MS_OFF_UL + MS_WAVES * (MS_UCAP + MS_PAD) * 4)
#define MS_OFF_STATE (MS_OFF_QORD + MS_QPB * 4)          // uint4 per query: d1, d2, w (window position), tie | force << 1
#define MS_OFF_CNT (MS_OFF_STATE + MS_QPB * 16)          // in-radius candidates per query, summed over chunks
#define MS_OFF_XR (MS_OFF_CNT + MS_QPB * 4)
#define MS_LDS_BYTES (MS_OFF_XR + 16)

template <int CTRL>
__device__ __forceinline__ uint32_t ms_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t ms_l1_bits(float qx, float qy, float2 t) {
    const float dx = qx - t.x, dy = qy - t.y;
    float d;
    asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(d) : "v"(dx), "v"(dy));
    return __float_as_uint(d);
}

// running order statistics of one query's SADs (see match_union.hip)
struct MsTrack { uint32_t d1, d2, w, tie; };

__device__ __forceinline__ void ms_update(MsTrack& t, uint32_t s, uint32_t w) {
    const bool lt = s < t.d1, eq = s == t.d1;
    const uint32_t m2 = min(t.d2, s);
    t.d2 = (s <= t.d1) ? t.d1 : m2;
    t.w = lt ? w : t.w;
    t.tie = lt ? 0u : (eq ? 1u : t.tie);
    t.d1 = min(t.d1, s);
}

__device__ __forceinline__ void ms_merge(MsTrack& a, const MsTrack& b) {
    const bool lt = b.d1 < a.d1, eq = b.d1 == a.d1;
    const uint32_t d2 = eq ? a.d1 : (lt ? min(b.d2, a.d1) : min(a.d2, b.d1));
    a.tie = eq ? 1u : (lt ? b.tie : a.tie);
    a.w = lt ? b.w : a.w;
    a.d1 = min(a.d1, b.d1);
    a.d2 = d2;
}

__global__ __launch_bounds__(MS_THREADS) void match_strip_kernel(BatchMatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const s_rows = smem + MS_OFF_ROWS;
    float2* const s_kp = reinterpret_cast<float2*>(smem + MS_OFF_KP);
    int* const s_qord = reinterpret_cast<int*>(smem + MS_OFF_QORD);
    uint4* const s_state = reinterpret_cast<uint4*>(smem + MS_OFF_STATE);
    int* const s_cnt = reinterpret_cast<int*>(smem + MS_OFF_CNT);
    float* const s_xr = reinterpret_cast<float*>(smem + MS_OFF_XR);
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
    const int q0 = qblk * MS_QPB;
    if (q0 >= n1) return;
    const int q1 = min(q0 + MS_QPB, n1);
    const MatchParamsDev& mp = a.mp[P.pidx];
    if (mp.epi != 0) return;   // stereo problems: match_batch_kernel<1>
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // ---- tile: x range (window) of its queries, y ranks inside each 64-query sub-block (round composition)
    if (wave < MS_QPB / MS_SUB) {
        const int jj = q0 + wave * MS_SUB + lane;
        const bool live = jj < q1;
        const float2 qv = live ? P.q.skp[jj] : make_float2(__builtin_nanf(""), __builtin_nanf(""));
        float mn = qv.x, mx = qv.x;
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, m));
            mx = fmaxf(mx, __shfl_xor(mx, m));
        }
        if (lane == 0) { s_xr[2 * wave] = mn; s_xr[2 * wave + 1] = mx; }
        // rank by (y, lane): any total order gives a valid permutation, this one puts neighbours in y together
        const uint32_t yb = __float_as_uint(qv.y);
        const uint32_t key = live ? (yb ^ ((yb >> 31) ? 0xffffffffu : 0x80000000u)) : 0xffffffffu;
        int rank = 0;
        for (int m = 0; m < VISO_WAVE; ++m) {
            const uint32_t km = (uint32_t)__builtin_amdgcn_readlane((int)key, m);
            rank += (km < key || (km == key && m < lane)) ? 1 : 0;
        }
        s_qord[wave * MS_SUB + rank] = wave * MS_SUB + lane;
    }
    if (tid < MS_QPB) {
        s_state[tid] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
        s_cnt[tid] = 0;
    }
    __syncthreads();
    int lo = 0, W = 0;
    {
        const float xa = fminf(s_xr[0], s_xr[2]), xb = fmaxf(s_xr[1], s_xr[3]);   // fmin/fmax ignore the NaN of an empty sub-block
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
    float2 kp0 = make_float2(0.f, 0.f);
    const bool has0 = n2 > 0;
    if (has0) kp0 = P.t.skp[P.t.rank[0]];
    const float radius = mp.radius;
    const int K = mp.K;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* grow_t;
    typedef const __attribute__((address_space(1))) char* gbytes_t;
    typedef __attribute__((address_space(3))) void* lds_t;
    uint32_t* const ul = reinterpret_cast<uint32_t*>(smem + MS_OFF_UL) + wave * (MS_UCAP + MS_PAD);
    const int g8 = lane >> 3, sub = lane & 7;
    // the lane's two 16-B chunks of a 256-B row, in the order it reads them: lanes with bit 4 set take the upper
    // 128-B half first (see the header: conflict-free ds_read_b128)
    const uint32_t sw = (lane >> 4) & 1u;
    const uint32_t off_a = (uint32_t)(sub + 8 * sw) << 4, off_b = (uint32_t)(sub + 8 * (1 - sw)) << 4;
    // which of the round's four queries this lane ends up tracking (transposing reduction below): lanes 4..7 of an
    // 8-lane group mirror lanes 3..0
    const bool sel0 = ((lane ^ (lane >> 2)) & 1) != 0, sel1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0;
    const int myq = (sel0 ? 1 : 0) + (sel1 ? 2 : 0);
    const int msh = 31 - myq;   // membership bit of query myq in a list entry (bit 3 - k of the mask nibble)
    unsigned long long scored = 0;
    const int sblk = wave / (MS_WAVES / (MS_QPB / MS_SUB));            // sub-block of this wave
    const int swave = wave % (MS_WAVES / (MS_QPB / MS_SUB));           // its index among the sub-block's waves

    for (int cb = 0; cb == 0 || cb < W; cb += MS_WCAP) {
        const int cw = min(max(W - cb, 0), MS_WCAP);
        const int cwpad = (cw + 127) & ~127;   // NaN padded: the scan needs no bounds test
        const bool last = cb + MS_WCAP >= W;
        __syncthreads();   // every wave is done with the previous chunk's rows
        // ---- stage the chunk: rows by asynchronous linear copy (1 KiB = 4 rows per wave-instruction), keypoints by hand
        {
            const gbytes_t src = (gbytes_t)reinterpret_cast<const char*>(P.t.rows) + (size_t)(lo + cb) * (VISO_ROW * 2);
            const int nbytes = cw * (VISO_ROW * 2);
            const int npieces = (nbytes + 1023) >> 10;
            if (!(a._pad & 2)) for (int p = wave; p < npieces; p += MS_WAVES) {
                const int o = min(p * 1024 + lane * 16, nbytes - 16);   // the last piece may be partial: clamp the source
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o),
                                                 (lds_t)(s_rows + p * 1024), 16, 0, 0);
            }
            for (int w = tid; w < cwpad; w += MS_THREADS) {
                float2 t2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
                if (w < cw) t2 = P.t.skp[lo + cb + w];
                s_kp[w] = t2;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int r = 0; r < MS_ROUNDS; ++r) {
            // ---------------- round setup: scalars of the four queries
            float2 qk[MS_G];
            int orig[MS_G], jq[MS_G], cnt[MS_G], pl[MS_G];
            uint32_t thr[MS_G];
            bool any_live = false;
            {
                const int base_ = sblk * MS_SUB + (swave * MS_ROUNDS + r) * MS_G;
                const int pli = s_qord[base_ + (lane & (MS_G - 1))];
                const int j_ = q0 + pli;
                const int jc_ = min(j_, q1 - 1);
                const float2 pq = P.q.skp[jc_];
                const int po = j_ < q1 ? P.q.sidx[jc_] : -1;
#pragma unroll
                for (int k = 0; k < MS_G; ++k) {
                    qk[k].x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.x), k));
                    qk[k].y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.y), k));
                    orig[k] = __builtin_amdgcn_readlane(po, k);
                    pl[k] = __builtin_amdgcn_readlane(pli, k);
                    jq[k] = q0 + pl[k];
                    cnt[k] = 0;
                    // (d <= radius && d < d0cut) as ONE unsigned compare of the bits of d = |dx| + |dy| (see match_union.hip)
                    uint32_t t = __float_as_uint(radius) + 1u;
                    if (has0) {
                        const float d0 = l1_kp(qk[k].x, qk[k].y, kp0);
                        if (d0 <= radius) t = __float_as_uint(d0);
                    }
                    thr[k] = orig[k] >= 0 ? t : 0u;
                    any_live = any_live || orig[k] >= 0;
                }
            }
            if (!any_live) continue;   // wave uniform
            // query rows into registers: this lane's two chunks of each (the loads land during the scan)
            u32x4 qa[MS_G], qb[MS_G];
#pragma unroll
            for (int k = 0; k < MS_G; ++k) {
                const gbytes_t qrow = (gbytes_t)reinterpret_cast<const char*>(P.q.rows) + (size_t)min(jq[k], q1 - 1) * (VISO_ROW * 2);
                qa[k] = *(grow_t)(qrow + off_a);
                qb[k] = *(grow_t)(qrow + off_b);
            }
            // ---------------- phase 1: one scan, membership masks, union list.  entry = mask << 28 | position << 8
            int ucnt = 0;
            if (!(a._pad & 4)) for (int base = 0; base < cwpad; base += 2 * VISO_WAVE) {
                const float2 ta = s_kp[base + lane], tb = s_kp[base + VISO_WAVE + lane];
                uint32_t ma = 0, mb = 0;
#pragma unroll
                for (int k = 0; k < MS_G; ++k) {
                    const bool ina = ms_l1_bits(qk[k].x, qk[k].y, ta) < thr[k];
                    const bool inb = ms_l1_bits(qk[k].x, qk[k].y, tb) < thr[k];
                    cnt[k] += __popcll(__ballot(ina)) + __popcll(__ballot(inb));
                    ma = ma + ma + (ina ? 1u : 0u);
                    mb = mb + mb + (inb ? 1u : 0u);
                }
                const unsigned long long ua = __ballot(ma != 0), ub = __ballot(mb != 0);
                const int ca = __popcll(ua);
                const uint32_t ea = (uint32_t)(base + lane) << 8;
                if (ma) ul[min(ucnt + mbcnt(ua), MS_UCAP - 1)] = (ma << 28) | ea;
                if (mb) ul[min(ucnt + ca + mbcnt(ub), MS_UCAP - 1)] = (mb << 28) | (ea + (VISO_WAVE << 8));
                ucnt += ca + __popcll(ub);
            }
            const bool list_ovf = ucnt > MS_UCAP;
            const int nu = (list_ovf || (a._pad & 1)) ? 0 : ucnt;
            __builtin_amdgcn_wave_barrier();
            if (nu > 0 && lane < MS_PAD) ul[nu + lane] = ul[nu - 1] & 0x0fffffffu;   // padding: scored, never counted
            __builtin_amdgcn_wave_barrier();
            // ---------------- phase 2: rolling pipeline over the union list, rows from LDS
            MsTrack tr;
            tr.d1 = 0xffffffffu; tr.d2 = 0xffffffffu; tr.w = 0; tr.tie = 0;
            {
                const int npass = (nu + 7) >> 3;
                u32x4 r0[MS_NP], r1[MS_NP];
                uint32_t ent[MS_NP];
#define MS_ISSUE(SLOT, T)                                                                                  \
                do {                                                                                       \
                    ent[SLOT] = ul[(T) * 8 + g8];                                                          \
                    const uint32_t ro_ = ent[SLOT] & 0x0fffff00u;                                          \
                    r0[SLOT] = *reinterpret_cast<const u32x4*>(s_rows + (ro_ | off_a));                    \
                    r1[SLOT] = *reinterpret_cast<const u32x4*>(s_rows + (ro_ | off_b));                    \
                } while (0)
#define MS_SAD(K, SLOT)                                                                                    \
                ({                                                                                         \
                    uint32_t s_ = __builtin_amdgcn_sad_u16(r0[SLOT].x, qa[K].x, 0u);                       \
                    s_ = __builtin_amdgcn_sad_u16(r0[SLOT].y, qa[K].y, s_);                                \
                    s_ = __builtin_amdgcn_sad_u16(r0[SLOT].z, qa[K].z, s_);                                \
                    s_ = __builtin_amdgcn_sad_u16(r0[SLOT].w, qa[K].w, s_);                                \
                    s_ = __builtin_amdgcn_sad_u16(r1[SLOT].x, qb[K].x, s_);                                \
                    s_ = __builtin_amdgcn_sad_u16(r1[SLOT].y, qb[K].y, s_);                                \
                    s_ = __builtin_amdgcn_sad_u16(r1[SLOT].z, qb[K].z, s_);                                \
                    s_ = __builtin_amdgcn_sad_u16(r1[SLOT].w, qb[K].w, s_);                                \
                    s_;                                                                                    \
                })
                // four partial SADs per lane -> every lane of the 8-lane group holds the total of query `myq`:
                // a transposing reduction (each step halves the number of values a lane carries)
#define MS_REDUCE(SLOT)                                                                                    \
                do {                                                                                       \
                    const uint32_t s0_ = MS_SAD(0, SLOT), s1_ = MS_SAD(1, SLOT), s2_ = MS_SAD(2, SLOT), s3_ = MS_SAD(3, SLOT); \
                    uint32_t a01_ = sel0 ? s1_ : s0_, a23_ = sel0 ? s3_ : s2_;                             \
                    const uint32_t b01_ = sel0 ? s0_ : s1_, b23_ = sel0 ? s2_ : s3_;                       \
                    a01_ += ms_dpp<0xB1>(b01_);   /* quad_perm 1,0,3,2 */                                  \
                    a23_ += ms_dpp<0xB1>(b23_);                                                            \
                    uint32_t m_ = sel1 ? a23_ : a01_;                                                      \
                    const uint32_t o_ = sel1 ? a01_ : a23_;                                                \
                    m_ += ms_dpp<0x4E>(o_);       /* quad_perm 2,3,0,1 */                                  \
                    m_ += ms_dpp<0x141>(m_);      /* row_half_mirror: lane i <-> 7 - i track the same query */ \
                    const bool member_ = ((ent[SLOT] >> msh) & 1u) != 0;                                   \
                    ms_update(tr, member_ ? m_ : 0xffffffffu, (ent[SLOT] >> 8) & 0xfffffu);                \
                } while (0)
                if (npass > 0) {
#pragma unroll
                    for (int p = 0; p < MS_NP; ++p) MS_ISSUE(p, p);
                }
                int t = 0;
                for (; t + MS_NP < npass; t += MS_NP) {
#pragma unroll
                    for (int p = 0; p < MS_NP; ++p) {
                        MS_REDUCE(p);
                        MS_ISSUE(p, t + p + MS_NP);
                    }
                }
                if (npass > 0) {
#pragma unroll
                    for (int p = 0; p < MS_NP; ++p) MS_REDUCE(p);
                }
#undef MS_REDUCE
#undef MS_SAD
#undef MS_ISSUE
            }
            // ---------------- phase 3: merge the 8 lane groups (lanes with equal position in the group track the same
            // query); lanes 0..3 end up with queries 0..3 of the round
#pragma unroll
            for (int m = 8; m < VISO_WAVE; m <<= 1) {
                MsTrack o;
                o.d1 = (uint32_t)__shfl_xor((int)tr.d1, m);
                o.d2 = (uint32_t)__shfl_xor((int)tr.d2, m);
                o.w = (uint32_t)__shfl_xor((int)tr.w, m);
                o.tie = (uint32_t)__shfl_xor((int)tr.tie, m);
                ms_merge(tr, o);
            }
            {
                int my_orig = -1, my_j = 0, my_cnt = 0, my_pl = 0;
#pragma unroll
                for (int k = 0; k < MS_G; ++k)
                    if (lane == k) { my_orig = orig[k]; my_j = jq[k]; my_cnt = cnt[k]; my_pl = pl[k]; }
                if (lane < MS_G && my_orig >= 0) {
                    tr.w += (uint32_t)cb;   // chunk-local -> window position
                    tr.tie &= (tr.d1 != 0xffffffffu) ? 1u : 0u;
                    if (list_ovf) tr.tie |= 2u;   // union list too long: the overflow kernel redoes this query
                    if (cb > 0 || !last) {        // several chunks: fold into the state carried in LDS
                        const uint4 sv = s_state[my_pl];
                        MsTrack st;
                        st.d1 = sv.x; st.d2 = sv.y; st.w = sv.z; st.tie = sv.w & 1u;
                        const uint32_t force = (sv.w | tr.tie) & 2u;
                        tr.tie &= 1u;
                        ms_merge(st, tr);
                        st.tie &= (st.d1 != 0xffffffffu) ? 1u : 0u;
                        tr = st;
                        tr.tie |= force;
                        my_cnt += s_cnt[my_pl];
                        if (!last) {
                            s_state[my_pl] = make_uint4(tr.d1, tr.d2, tr.w, tr.tie);
                            s_cnt[my_pl] = my_cnt;
                        }
                    }
                    if (last) {
                        const bool none = tr.d1 == 0xffffffffu;
                        if (my_cnt > K || tr.tie != 0) {
                            // more than K candidates / union too long / exact tie of the minimum (largest-key rule): overflow kernel
                            P.ovf[atomicAdd(P.ovf_cnt, 1)] = my_j;
                        } else {
                            bool accept = !none;
                            int idx = -1;
                            if (accept) {
                                idx = P.t.sidx[lo + (int)tr.w];
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
        }
    }
    // scored pairs of the tile's queries whose result stands (lanes 0..3 of every wave hold partial sums)
#pragma unroll
    for (int m = 1; m < MS_G; m <<= 1) scored += (unsigned long long)__shfl_xor((long long)scored, m);
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
}

int launch_match_strip_temporal(hipStream_t s, const BatchMatchArgs& a64, int cap_max) {
    // same problem enumeration as the 64-query kernels, tiles of MS_QPB queries
    BatchMatchArgs a = a64;
    a.bpp = (cap_max + MS_QPB - 1) / MS_QPB;
    {   // timing experiments only (results are wrong with any bit set): 1 = no scoring, 2 = no row staging, 4 = no scan
        const char* e = getenv("VISO_STRIP_DEBUG");
        a._pad = e ? atoi(e) : 0;
    }
    const int groups = (a.n_probs + 7) / 8;
    long long blocks = (long long)groups * 8 * a.bpp;
    if (a.gs == 3) blocks = (long long)((groups + 2) / 3) * a.gc * 8 * a.bpp;
    if (blocks > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    static unsigned long long attr_set = 0;   // bit d: done for device d (the attribute is per function AND device)
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_set >> dev) & 1ull)) {
        HIP_TRY(hipFuncSetAttribute((const void*)match_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MS_LDS_BYTES));
        if (dev < 64) attr_set |= 1ull << dev;
    }
    hipLaunchKernelGGL(match_strip_kernel, dim3((unsigned)blocks), dim3(MS_THREADS), MS_LDS_BYTES, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_strip_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
