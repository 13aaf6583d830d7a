// match_batch.hip — u16 matcher in rounds of FOUR queries per wave: the stereo instantiation <1> is the default
// for the stereo call of every frame (match_union.hip takes the temporal calls); <0> is the previous temporal
// kernel, still selectable.  Same tile / window data path as match_kernel<false, EPI> (match.hip).
//
//   phase 1  one scan of the tile's target window (LDS, NaN padded) for the round's four queries: per-query
//            candidate segments in LDS; (stereo) fp64 Sampson gate, one candidate per lane, in-place compaction
//   phase 1b the segments are closed up into one contiguous list, entry = query << 30 | row byte offset, padded so
//            that neither the last pass nor the passes the pipeline runs ahead need a bounds test
//   phase 2  rolling pipeline over ALL pairs of the four queries: 8 lanes per pair, two 16-B global loads of the
//            target row + two LDS reads of the staged query row, 8 x v_sad_u16, 3 DPP adds, SAD -> LDS; two
//            passes reduced together so that their DPP wait states interleave
//   phase 3  per query: wave-wide min / second min (with multiplicity) / count over its SAD segment — or, when
//            all four lists hold at most 16 pairs (the stereo call), one 16-lane row per query, all four at once
//
// Same results as the other kernels.  Irregular queries (more than K or 128 in-radius candidates, an exact SAD
// tie) go to match_overflow_kernel.
#include "common.h"
#include <stdlib.h>
#include "match_dev.h"

#define MB_THREADS 256
#define MB_WAVES 4
#define MB_QPB 64          // queries per tile
#define MB_G 4             // queries batched per wave round
#define MB_SEG 128         // pair-list entries per query (more in-radius candidates: overflow kernel)
#define MB_PAD 32          // list padding: 4 passes of 8 pairs (the deepest pipeline) past the last pair
#define MB_KPCAP 512       // window keypoints staged in LDS


__device__ __forceinline__ uint32_t mb_wave_min(uint32_t v) {
    const int ident = -1;
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x140, 0xf, 0xf, false));  // row_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// bits of |qx - tx| + |qy - ty| (cvflann::L1 order, see l1_kp): the absolute values ride on the add as source
// modifiers — left to the compiler the two differences are packed and the abs becomes two v_and
__device__ __forceinline__ uint32_t mb_l1_bits(float qx, float qy, float2 t) {
    const float dx = qx - t.x, dy = qy - t.y;
    float d;
    asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(d) : "v"(dx), "v"(dy));
    return __float_as_uint(d);
}

template <int CTRL>
__device__ __forceinline__ uint32_t mb_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

template <int EPI>
__global__ __attribute__((amdgpu_waves_per_eu(5, 8))) __launch_bounds__(MB_THREADS) void match_batch_kernel(BatchMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_pairs[MB_WAVES][MB_G * MB_SEG + MB_PAD];   // + the passes the pipeline may run ahead
    __shared__ __attribute__((aligned(16))) uint32_t s_sads[MB_WAVES][MB_G * MB_SEG + MB_PAD];
    __shared__ __attribute__((aligned(16))) uint32_t s_qrow[MB_WAVES][MB_G][64];
    __shared__ float2 s_kp[MB_KPCAP];
    __shared__ int s_idx[MB_KPCAP];
    __shared__ float s_xr[4];
    if (EPI && a.bad[1] == 0) return;   // match_stereo_kernel has done every stereo tile (rectified pairs: always)
    // The grid walks the (problem, tile) slots: for the stereo instantiation it is a SMALL grid (the kernel is normally
    // idle — rectified pairs decline no tile — and thousands of workgroups that leave at once cost ~19 us per step)
    for (int vb = blockIdx.x; vb < a.vblocks; vb += gridDim.x) {
    if (vb != (int)blockIdx.x) __syncthreads();   // the previous tile's LDS is still being read by other waves
    int prob, qblk;
    {
        const int b = vb;
        const int xcd = b & 7, slot = b >> 3;
        const int g = slot / a.bpp;
        prob = ((g / a.gc) * a.gs + a.gf + g % a.gc) * 8 + xcd;
        qblk = slot % a.bpp;
        if (prob >= a.n_probs) continue;
    }
    const MatchProblem P = a.probs[prob];
    if ((*P.q.bad | *P.t.bad) != 0) continue;   // non-integer descriptors: the general kernel does this problem
    const int n1 = *P.q.n, n2 = *P.t.n;
    const int q0 = qblk * MB_QPB;
    if (q0 >= n1) continue;
    const int q1 = min(q0 + MB_QPB, n1);
    const MatchParamsDev& mp = a.mp[P.pidx];
    if ((mp.epi != 0) != (EPI != 0)) continue;
    if (EPI && P.tile_flag[qblk] == 0) continue;   // match_stereo_kernel (narrow epipolar band) has done this tile
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // ---- tile window (identical to match_kernel)
    if (wave == 0) {
        const float2 qv = (q0 + lane < q1) ? P.q.skp[q0 + lane] : make_float2(__builtin_nanf(""), __builtin_nanf(""));
        float mn = qv.x, mx = qv.x;
        float yn = qv.y, yx = qv.y;
        bool ynan = qv.y != qv.y && (q0 + lane < q1);
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, m));
            mx = fmaxf(mx, __shfl_xor(mx, m));
            if (EPI) {
                yn = fminf(yn, __shfl_xor(yn, m));
                yx = fmaxf(yx, __shfl_xor(yx, m));
            }
        }
        if (lane == 0) { s_xr[0] = mn; s_xr[1] = mx; }
        if (EPI) {
            // epipolar band of the tile (match_dev.h): targets with |dy| > band cannot pass the Sampson gate
            const bool anynan = __any(ynan);
            if (lane == 0) s_xr[2] = anynan ? __builtin_huge_valf() : epipolar_band(mp.F, mp.sampson_thresh, mn, mx, yn, yx, mp.radius);
        }
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
    lo = __builtin_amdgcn_readfirstlane(lo);   // wave uniform by construction; tell the compiler
    W = __builtin_amdgcn_readfirstlane(W);
    const int wcap = min(W, MB_KPCAP);
    // staged window, padded with NaN keypoints (never in radius) to a multiple of 128 so the scan needs no bounds test
    const int wpad = (wcap + 127) & ~127;
    for (int w = threadIdx.x; w < wpad; w += MB_THREADS) {
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
    const float band = EPI ? s_xr[2] : 0.f;
    // the row gathers are the hot loads: pin their address space (global_load, not flat_load)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* grow_t;
    typedef const __attribute__((address_space(1))) char* gbytes_t;
    // window base (scalar) + 32-bit byte offset per lane: (window position << 8) | (sub << 4)
    const gbytes_t wrows = (gbytes_t)reinterpret_cast<const char*>(P.t.rows) + (size_t)lo * (VISO_ROW * 2);
    uint32_t* pairs = s_pairs[wave];
    uint32_t* sads = s_sads[wave];
    const int g8 = lane >> 3, sub = lane & 7;
    unsigned long long scored = 0;

    // query data of the next round, loaded one round ahead (the loads land during phases 2 and 3): lane l
    // carries keypoint and original index of query (l & 3) of the round, every lane one word of each row
    float2 pq;
    int po;
    uint32_t prow[MB_G];
#define MB_PREFETCH(JG)                                                                                   \
    do {                                                                                                  \
        const int j_ = (JG) + (lane & (MB_G - 1)) * MB_WAVES;                                             \
        const int jc_ = min(j_, q1 - 1);                                                                  \
        pq = P.q.skp[jc_];                                                                                \
        po = j_ < q1 ? P.q.sidx[jc_] : -1;                                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < MB_G; ++k_)                                               \
            prow[k_] = reinterpret_cast<const uint32_t*>(P.q.rows + (size_t)min((JG) + k_ * MB_WAVES, q1 - 1) * VISO_ROW)[lane]; \
    } while (0)
    MB_PREFETCH(q0 + wave);

    for (int jg = q0 + wave; jg < q1; jg += MB_WAVES * MB_G) {
        // ---------------- phase 1: one scan of the window for the round's MB_G queries
        float2 qk[MB_G];      // wave uniform (scalar registers)
        int orig[MB_G], cnt[MB_G], seg_n[MB_G];
        int qn[MB_G];   // queued (EPI: in radius AND inside the epipolar band); cnt = all in radius (the K cap counts those)
        uint32_t thr[MB_G];
#pragma unroll
        for (int k = 0; k < MB_G; ++k) {
            qk[k].x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.x), k));
            qk[k].y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.y), k));
            orig[k] = __builtin_amdgcn_readlane(po, k);
            s_qrow[wave][k][lane] = prow[k];
            cnt[k] = 0;
            qn[k] = 0;
            // d = |dx| + |dy| is +0, positive or NaN, so its bit pattern orders like the value and every NaN is
            // above +inf: (d <= radius && d < d0cut) is one unsigned compare against bits(d0) (target 0 in
            // radius: Q1, src/viso.cpp:693) or bits(radius) + 1.  Dead slots (past the tile) get 0: nothing passes.
            uint32_t t = __float_as_uint(radius) + 1u;
            if (has0) {
                const float d0 = l1_kp(qk[k].x, qk[k].y, kp0);
                if (d0 <= radius) t = __float_as_uint(d0);
            }
            thr[k] = orig[k] >= 0 ? t : 0u;
        }
        if (jg + MB_WAVES * MB_G < q1) MB_PREFETCH(jg + MB_WAVES * MB_G);
        for (int base = 0; base < wpad; base += 2 * VISO_WAVE) {   // LDS-resident (NaN padded) part of the window
            const float2 ta = s_kp[base + lane], tb = s_kp[base + VISO_WAVE + lane];
            const uint32_t ea = (uint32_t)(base + lane), eb = ea + VISO_WAVE;
#pragma unroll
            for (int k = 0; k < MB_G; ++k) {
                bool ina = mb_l1_bits(qk[k].x, qk[k].y, ta) < thr[k];
                bool inb = mb_l1_bits(qk[k].x, qk[k].y, tb) < thr[k];
                if (EPI) {
                    cnt[k] += __popcll(__ballot(ina)) + __popcll(__ballot(inb));
                    ina = ina && fabsf(qk[k].y - ta.y) <= band;
                    inb = inb && fabsf(qk[k].y - tb.y) <= band;
                }
                const unsigned long long ma = __ballot(ina), mb = __ballot(inb);
                const int ca = __popcll(ma);
                // a segment holds MB_SEG entries; a query that needs more is flagged below, its clamped writes are ignored
                uint32_t* dst = pairs + k * MB_SEG;
                if (ina) dst[min(qn[k] + mbcnt(ma), MB_SEG - 1)] = ea;
                if (inb) dst[min(qn[k] + ca + mbcnt(mb), MB_SEG - 1)] = eb;
                qn[k] += ca + __popcll(mb);
                if (!EPI) cnt[k] = qn[k];
            }
        }
        if (W > wcap) {   // windows wider than MB_KPCAP (dense data only): the rest from global memory
#pragma unroll
            for (int k = 0; k < MB_G; ++k) {
                for (int base = wcap; base < W; base += VISO_WAVE) {
                    const int w = base + lane;
                    bool in = false;
                    float2 t2 = make_float2(0.f, 0.f);
                    if (w < W) { t2 = P.t.skp[lo + w]; in = __float_as_uint(l1_kp(qk[k].x, qk[k].y, t2)) < thr[k]; }
                    if (EPI) {
                        cnt[k] += __popcll(__ballot(in));
                        in = in && fabsf(qk[k].y - t2.y) <= band;
                    }
                    const unsigned long long m = __ballot(in);
                    if (in) pairs[k * MB_SEG + min(qn[k] + mbcnt(m), MB_SEG - 1)] = (uint32_t)w;
                    qn[k] += __popcll(m);
                    if (!EPI) cnt[k] = qn[k];
                }
            }
        }
        int flags = 0, npass = 0;
#pragma unroll
        for (int k = 0; k < MB_G; ++k) {
            const bool fits = cnt[k] <= K && qn[k] <= MB_SEG;
            if (!fits && orig[k] >= 0) flags |= 1 << k;   // left to the overflow kernel
            seg_n[k] = fits ? qn[k] : 0;
        }
        // ---------------- phase 1b: close the gaps — segments 1..3 move down behind segment 0 (ascending, each
        // 64-entry chunk read before it is written, destination never above the source) and get their query tag, so
        // that the round's pairs are one contiguous list: entry = query << 30 | window position << 8
        static_assert(MB_G == 4, "segment bookkeeping below is written out for 4 segments");
        int c1 = seg_n[0], c2 = c1 + seg_n[1], c3 = c2 + seg_n[2], ntot = c3 + seg_n[3];
        {
            const int cs[MB_G] = {0, c1, c2, c3};
#pragma unroll
            for (int k = 0; k < MB_G; ++k) {
                for (int b = 0; b < seg_n[k]; b += VISO_WAVE) {
                    const int i = b + lane;
                    uint32_t e = 0;
                    if (i < seg_n[k]) e = pairs[k * MB_SEG + i];
                    __builtin_amdgcn_wave_barrier();
                    if (i < seg_n[k]) pairs[cs[k] + i] = (e << 8) | ((uint32_t)k << 30);   // byte offset of the row | query
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (EPI && ntot > 0) {
            // Sampson gate (src/viso.cpp:695-701) over the candidates of ALL FOUR queries at once, one candidate per
            // lane (the epipolar band leaves a handful per query, so a round usually is one pass), in-place compaction
            // that keeps the list ordered by query.  sampson_dev is the reference's arithmetic, bit for bit.
            int wr = 0;
            seg_n[0] = seg_n[1] = seg_n[2] = seg_n[3] = 0;
            for (int b = 0; b < ntot; b += VISO_WAVE) {
                const int i = b + lane;
                bool pass = false;
                uint32_t e = 0;
                if (i < ntot) {
                    e = pairs[i];
                    const uint32_t kq = e >> 30, w = (e & 0x3fffffffu) >> 8;
                    float2 t2;
                    if ((int)w < wcap) t2 = s_kp[w]; else t2 = P.t.skp[lo + (int)w];
                    const float qx = kq == 0 ? qk[0].x : kq == 1 ? qk[1].x : kq == 2 ? qk[2].x : qk[3].x;
                    const float qy = kq == 0 ? qk[0].y : kq == 1 ? qk[1].y : kq == 2 ? qk[2].y : qk[3].y;
                    const double sd = sampson_dev(mp.F, qx, qy, t2.x, t2.y);
                    pass = isfinite(sd) && !(sd > mp.sampson_thresh);
                }
                const unsigned long long m = __ballot(pass);
                const int pos = wr + mbcnt(m);
                __builtin_amdgcn_wave_barrier();
                if (pass) pairs[pos] = e;   // pos <= i: in place
                wr += __popcll(m);
#pragma unroll
                for (int k = 0; k < MB_G; ++k) seg_n[k] += __popcll(__ballot(pass && (e >> 30) == (uint32_t)k));
            }
            c1 = seg_n[0]; c2 = c1 + seg_n[1]; c3 = c2 + seg_n[2]; ntot = c3 + seg_n[3];
            __builtin_amdgcn_wave_barrier();
        }
        // padding behind the list (copies of the last pair): the pipeline runs up to MB_NP - 1 passes past the
        // end and the last pass may be partial — those lanes re-score the last pair into scratch slots
        if (ntot > 0 && lane < MB_PAD) pairs[ntot + lane] = pairs[ntot - 1];
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2: score every pair of the round, 8 lanes per pair, 8 consecutive pairs per pass
        {
            npass = (ntot + 7) >> 3;
            // rolling pipeline: the row loads of MB_NP passes are in flight while two passes are reduced
            constexpr int MB_NP = EPI ? 2 : 4;
            static_assert(MB_NP * 8 <= MB_PAD, "list padding must cover the pipeline depth");   // the stereo kernel scores ~3 pairs per query and needs its registers for fp64
            u32x4 r0[MB_NP], r1[MB_NP];
            int dst[MB_NP], qoff[MB_NP];
#define MB_ISSUE(SLOT, T)                                                                                  \
            do {                                                                                           \
                const int gi_ = (T) * 8 + g8;                                                              \
                const uint32_t e_ = pairs[gi_];                                                            \
                dst[SLOT] = gi_;                                                                           \
                qoff[SLOT] = (int)(e_ >> 30) * 64 + sub * 4;                                               \
                const grow_t row_ = (grow_t)(wrows + ((e_ & 0x3fffffffu) | (uint32_t)(sub << 4)));         \
                r0[SLOT] = row_[0];                                                                        \
                r1[SLOT] = row_[8];                                                                        \
            } while (0)
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < MB_NP; ++p) MB_ISSUE(p, p);
            }
            // two passes reduced together: their DPP steps interleave and fill each other's wait states
#define MB_REDUCE2(SA, SB)                                                                                \
            do {                                                                                           \
                const uint32_t* qa_ = &s_qrow[wave][0][0] + qoff[SA];                                      \
                const uint32_t* qb_ = &s_qrow[wave][0][0] + qoff[SB];                                      \
                const uint4 a0_ = *reinterpret_cast<const uint4*>(qa_);                                    \
                const uint4 a1_ = *reinterpret_cast<const uint4*>(qa_ + 32);                               \
                const uint4 b0_ = *reinterpret_cast<const uint4*>(qb_);                                    \
                const uint4 b1_ = *reinterpret_cast<const uint4*>(qb_ + 32);                               \
                uint32_t sa_ = __builtin_amdgcn_sad_u16(r0[SA].x, a0_.x, 0u);                              \
                uint32_t sb_ = __builtin_amdgcn_sad_u16(r0[SB].x, b0_.x, 0u);                              \
                sa_ = __builtin_amdgcn_sad_u16(r0[SA].y, a0_.y, sa_);                                      \
                sb_ = __builtin_amdgcn_sad_u16(r0[SB].y, b0_.y, sb_);                                      \
                sa_ = __builtin_amdgcn_sad_u16(r0[SA].z, a0_.z, sa_);                                      \
                sb_ = __builtin_amdgcn_sad_u16(r0[SB].z, b0_.z, sb_);                                      \
                sa_ = __builtin_amdgcn_sad_u16(r0[SA].w, a0_.w, sa_);                                      \
                sb_ = __builtin_amdgcn_sad_u16(r0[SB].w, b0_.w, sb_);                                      \
                sa_ = __builtin_amdgcn_sad_u16(r1[SA].x, a1_.x, sa_);                                      \
                sb_ = __builtin_amdgcn_sad_u16(r1[SB].x, b1_.x, sb_);                                      \
                sa_ = __builtin_amdgcn_sad_u16(r1[SA].y, a1_.y, sa_);                                      \
                sb_ = __builtin_amdgcn_sad_u16(r1[SB].y, b1_.y, sb_);                                      \
                sa_ = __builtin_amdgcn_sad_u16(r1[SA].z, a1_.z, sa_);                                      \
                sb_ = __builtin_amdgcn_sad_u16(r1[SB].z, b1_.z, sb_);                                      \
                sa_ = __builtin_amdgcn_sad_u16(r1[SA].w, a1_.w, sa_);                                      \
                sb_ = __builtin_amdgcn_sad_u16(r1[SB].w, b1_.w, sb_);                                      \
                sa_ += mb_dpp<0xB1>(sa_);                                                                  \
                sb_ += mb_dpp<0xB1>(sb_);                                                                  \
                sa_ += mb_dpp<0x4E>(sa_);                                                                  \
                sb_ += mb_dpp<0x4E>(sb_);                                                                  \
                sa_ += mb_dpp<0x141>(sa_);                                                                 \
                sb_ += mb_dpp<0x141>(sb_);                                                                 \
                if (sub == 0) { sads[dst[SA]] = sa_; sads[dst[SB]] = sb_; }   /* slots past ntot are scratch */ \
            } while (0)
            int t = 0;
            for (; t + MB_NP < npass; t += MB_NP) {   // steady state: reduce two passes, refill their slots (no branch:
#pragma unroll                                        // the hardware counts outstanding loads, a branch would drain them)
                for (int p = 0; p < MB_NP; p += 2) {
                    MB_REDUCE2(p, p + 1);
                    MB_ISSUE(p, t + p + MB_NP);
                    MB_ISSUE(p + 1, t + p + 1 + MB_NP);
                }
            }
            if (npass > 0) {
#pragma unroll
                for (int p = 0; p < MB_NP; p += 2) MB_REDUCE2(p, p + 1);
            }
#undef MB_REDUCE2
#undef MB_ISSUE
        }
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 3: per query, min / second min (with multiplicity) / argmin / tie
        const int nmax = max(max(seg_n[0], seg_n[1]), max(seg_n[2], seg_n[3]));
        if (EPI && nmax <= 16) {
            // short lists (the stereo call: ~3 pairs survive the gate): row r of 16 lanes reduces query r, all four at once
            const int row = lane >> 4, i16 = lane & 15;
            const int n = row == 0 ? seg_n[0] : row == 1 ? seg_n[1] : row == 2 ? seg_n[2] : seg_n[3];
            const int st = row == 0 ? 0 : row == 1 ? c1 : row == 2 ? c2 : c3;
            const int my_orig = row == 0 ? orig[0] : row == 1 ? orig[1] : row == 2 ? orig[2] : orig[3];
            const bool valid = i16 < n;
            const uint32_t sv = valid ? sads[st + i16] : 0xffffffffu;
            auto row_min = [](uint32_t v) {
                const int ident = -1;
                v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm 1,0,3,2
                v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm 2,3,0,1
                v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
                v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
                return v;
            };
            const uint32_t m1 = row_min(sv);
            const bool eq = valid && sv == m1;
            const unsigned long long em = __ballot(eq);
            const uint32_t emr = (uint32_t)(em >> (16 * row)) & 0xffffu;
            const int c = __popc(emr);
            const uint32_t m2 = row_min(eq ? 0xffffffffu : sv);
            const uint32_t r_d2 = c > 1 ? m1 : m2;
            if (i16 == 0 && my_orig >= 0) {
                const int j = jg + row * MB_WAVES;
                if (((flags >> row) & 1) || (m1 != 0xffffffffu && c > 1)) {
                    P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, j);   // K cap / exact tie: overflow kernel
                } else {
                    bool accept = m1 != 0xffffffffu;
                    int idx = -1;
                    if (accept) {
                        const uint32_t r_w = (pairs[st + (__ffs((int)emr) - 1)] & 0x3fffffffu) >> 8;
                        if ((int)r_w < wcap) idx = s_idx[r_w]; else idx = P.t.sidx[lo + (int)r_w];
                        if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                            const double bd2 = r_d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)r_d2;
                            accept = (double)m1 < bd2 * mp.ratio;
                        }
                    }
                    P.res[my_orig] = make_int2(accept ? idx : -1, (int)m1);
                    scored += (unsigned long long)n;
                }
            }
        } else {
    #pragma unroll
            for (int k = 0; k < MB_G; ++k) {
                if (orig[k] < 0) continue;
                const int j = jg + k * MB_WAVES;
                if (flags & (1 << k)) {
                    if (lane == 0) P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, j);
                    continue;
                }
                const int n = seg_n[k], st = k == 0 ? 0 : k == 1 ? c1 : k == 2 ? c2 : c3;
                uint32_t r_d1 = 0xffffffffu, r_d2 = 0xffffffffu, r_w = 0, r_tie = 0;
                for (int b = 0; b < n; b += VISO_WAVE) {
                    const bool valid = (b + lane) < n;
                    const uint32_t s = valid ? sads[st + b + lane] : 0xffffffffu;
                    const uint32_t m1 = mb_wave_min(s);
                    const bool eq = valid && s == m1;
                    const unsigned long long em = __ballot(eq);
                    const int c = __popcll(em);
                    const uint32_t m2 = mb_wave_min(eq ? 0xffffffffu : s);
                    const uint32_t wfirst = (pairs[st + b + (__ffsll((long long)em) - 1)] & 0x3fffffffu) >> 8;
                    const uint32_t o2 = c > 1 ? m1 : m2;
                    if (m1 < r_d1) { r_d2 = min(r_d1, o2); r_d1 = m1; r_w = wfirst; r_tie = c > 1; }
                    else if (m1 == r_d1) { r_d2 = r_d1; r_tie = 1; }
                    else r_d2 = min(r_d2, m1);
                }
                if (lane == 0) {
                    if (r_tie) {
                        P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, j);   // exact tie: the largest-key rule is applied by the overflow kernel
                    } else {
                        bool accept = r_d1 != 0xffffffffu;
                        int idx = -1;
                        if (accept) {
                            if ((int)r_w < wcap) idx = s_idx[r_w]; else idx = P.t.sidx[lo + (int)r_w];
                            if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                                const double bd2 = r_d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)r_d2;
                                accept = (double)r_d1 < bd2 * mp.ratio;
                            }
                        }
                        P.res[orig[k]] = make_int2(accept ? idx : -1, (int)r_d1);
                        scored += (unsigned long long)n;
                    }
                }
            }

        }
    }
    // per-lane partial sums (lane 0 of the generic path, lanes 0/16/32/48 of the short-list path)
    scored += (unsigned long long)__shfl_xor((long long)scored, 16);
    scored += (unsigned long long)__shfl_xor((long long)scored, 32);
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
    }   // walk over the slots
}

int launch_match_batch(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max,
                       const MatchParamsDev mp[2], int* bad, int layout, hipEvent_t e_mid, int variant, int r8s, int kinds) {
    BatchMatchArgs a;
    a.probs = probs_dev;
    a.n_probs = n_probs;
    a.bpp = (cap_max + MB_QPB - 1) / MB_QPB;
    a.gs = 1; a.gf = 0; a.gc = 1; a.vblocks = 0;
    a.bad = bad;
    a.r8s = r8s; a._pad8 = 0;
    a.mp[0] = mp[0];
    a.mp[1] = mp[1];
    const int groups = (n_probs + 7) / 8;
    long long bt = (long long)groups * 8 * a.bpp, bs = bt;
    BatchMatchArgs at = a, as = a;
    if (layout == 1) {
        const int g3 = (groups + 2) / 3;
        at.gs = 3; at.gf = 1; at.gc = 2; bt = (long long)g3 * 2 * 8 * a.bpp;
        as.gs = 3; as.gf = 0; as.gc = 1; bs = (long long)g3 * 8 * a.bpp;
    }
    if (bt > 0x7fffffffLL || bs > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    // one frame's problems (the plain family's stereo call with the temporal problems riding along): stereo and temporal
    // workgroups in ONE launch, side by side (match_frame.hip).  $VISO_FRAME_KERNEL=0: the two kernels (A/B and test aid)
    if (variant == 6 && (kinds & VISO_KIND_ALL) == VISO_KIND_ALL && bt <= VISO_FRAME_MAX_BLOCKS && !e_mid) {
        static const int off = [] { const char* e = getenv("VISO_FRAME_KERNEL"); return e && *e == '0'; }();
        if (!off) {
            const int r = launch_match_frame(s, at, as, bt, cap_max);
            if (r < 0) return r;
            if (kinds & VISO_KIND_NO_WIDE) return VISO_OK;
            as.vblocks = (int)bs;
            hipLaunchKernelGGL((match_batch_kernel<1>), dim3((unsigned)(bs < 1024 ? bs : 1024)), dim3(MB_THREADS), 0, s, as);
            HIP_TRY(hipGetLastError());
            return VISO_OK;
        }
    }
    if (!(kinds & VISO_KIND_TEMPORAL)) { /* no temporal problem in this launch */ } else
#ifdef VISO_DEBUG_VARIANTS
    if (variant == 2) {
        at.vblocks = (int)bt;
        hipLaunchKernelGGL((match_batch_kernel<0>), dim3((unsigned)bt), dim3(MB_THREADS), 0, s, at);
        HIP_TRY(hipGetLastError());
    } else
#endif
#ifdef VISO_DEBUG_VARIANTS
    if (variant == 4) {
        const int r = launch_match_strip_temporal(s, at, cap_max);
        if (r < 0) return r;
    } else
#endif
    if (variant == 5) {
        const int r = launch_match_prune_temporal(s, at, bt);
        if (r < 0) return r;
    } else if (variant == 6) {
#ifdef VISO_DEBUG_VARIANTS
        // experiment ($VISO_EXP_HEAVY_TOKEN=1, HISTORY.md round 6): the VALU-bound temporal kernels of ALL batches of the process
        // run one after the other (a device-wide event chain), so that what runs beside one is another batch's pack / stereo /
        // sort kernels, never a second copy of itself
        static const int token = [] { const char* e = getenv("VISO_EXP_HEAVY_TOKEN"); return e ? atoi(e) : 0; }();
        static hipEvent_t heavy_ev = nullptr;
        if (token) {
            if (!heavy_ev) HIP_TRY(hipEventCreateWithFlags(&heavy_ev, hipEventDisableTiming));
            HIP_TRY(hipStreamWaitEvent(s, heavy_ev, 0));
        }
#endif
        const int r = launch_match_union8_temporal(s, at, bt);
        if (r < 0) return r;
#ifdef VISO_DEBUG_VARIANTS
        if (token) HIP_TRY(hipEventRecord(heavy_ev, s));
#endif
    } else {
        const int r = launch_match_union_temporal(s, at, bt);
        if (r < 0) return r;
    }
    if (e_mid) HIP_TRY(hipEventRecord(e_mid, s));
    if (!(kinds & VISO_KIND_STEREO)) return VISO_OK;
    {   // stereo problems: lane-per-query kernel for tiles with a narrow epipolar band, the kernel below for the rest
        const int r = launch_match_stereo(s, as, cap_max);
        if (r < 0) return r;
    }
    if (kinds & VISO_KIND_NO_WIDE) return VISO_OK;
    as.vblocks = (int)bs;
    hipLaunchKernelGGL((match_batch_kernel<1>), dim3((unsigned)(bs < 1024 ? bs : 1024)), dim3(MB_THREADS), 0, s, as);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

