// match.hip — descriptor-window SAD matcher for gfx950 (MI355X).
//
// Replaces match_desc + radiusSearch + sampsonDistance of the reference
// (src/viso.cpp:669-726, 170-203, 655-666).  Integer abs-diff work: no MFMA;
// the levers are coalesced 256-B descriptor rows, wave64 DPP reductions and
// keeping every problem's working set inside one XCD's L2.
//
// Kernels
//   pack_desc_kernel    f32 N x dlen (boundary layout)  ->  u16 N x 128 rows (+bias)
//   match_u16_kernel    neighbour gate + epipolar gate + SAD + best/2nd-best  (hot)
//   match_f32_kernel    same walk, double-accumulated SAD for non-integer data
//   sort_matches_kernel (dist,i1)-ordered match list, inverse permutation, count
//
// Equivalence with the reference's list walk (proved in DESIGN.md §3):
// cvflann returns in-radius targets ordered by key=(L1 distance, index), keeps
// the first K, and match_desc walks them while index > 0 (Q1).  Hence the
// scored set is { t : key(t) < min(key_K, key(target 0 if in radius)) } and,
// because the best/second-best update is order independent except for ties
// (`<=`: the LAST equal candidate wins, Q2), the winner is the candidate of
// minimal SAD with the LARGEST key.  No neighbour list is materialised.
#include "common.h"

#include <math.h>

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ int lane_id() { return __lane_id(); }

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// sum over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ uint32_t row16_sum(uint32_t v) {
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    return v;
}

// cvflann::L1<float> over 2 elements: result = 0; result += |a0-b0|; result += |a1-b1|
__device__ __forceinline__ float l1_kp(float qx, float qy, float2 t) {
    float r = fabsf(qx - t.x);
    r += fabsf(qy - t.y);
    return r;
}

// sampsonDistance + algebricDistance, src/viso.cpp:655-666, 390-407 — same
// operation order and the same float roundings (Q4).
__device__ __forceinline__ double sampson_dev(const double* F, float p1x, float p1y, float p2x,
                                              float p2y) {
    double Fx0 = F[0] * p1x + F[1] * p1y + F[2];
    double Fx1 = F[3] * p1x + F[4] * p1y + F[5];
    double Ftx0 = F[0] * p2x + F[3] * p2y + F[6];
    double Ftx1 = F[1] * p2x + F[4] * p2y + F[7];
    float a0 = p1x, a1 = p1y, a2 = 1.f, b0 = p2x, b1 = p2y, b2 = 1.f;
    double adv = b0 * F[0] * a0 + b0 * F[1] * a1 + b0 * F[2] * a2 + b1 * F[3] * a0 +
                 b1 * F[4] * a1 + b1 * F[5] * a2 + b2 * F[6] * a0 + b2 * F[7] * a1 +
                 b2 * F[8] * a2;
    float ad = (float)adv;
    float ad2 = ad * ad;
    return ad2 / (Fx0 * Fx0 + Fx1 * Fx1 + Ftx0 * Ftx0 + Ftx1 * Ftx1);
}

__device__ __forceinline__ bool key_less(uint32_t ad, uint32_t ai, uint32_t bd, uint32_t bi) {
    return ad < bd || (ad == bd && ai < bi);
}

// ------------------------------------------------------------------ pack
// One thread = 8 consecutive u16 of one row (a 16-B store).
__global__ __launch_bounds__(256) void pack_desc_kernel(const float* __restrict__ src,
                                                        uint16_t* __restrict__ dst,
                                                        const int* __restrict__ n_rows,
                                                        int n_img, int cap, int dlen,
                                                        int* __restrict__ bad) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)n_img * cap * (VISO_ROW / 8);
    if (gid >= total) return;
    const int chunk = (int)(gid % (VISO_ROW / 8));
    const long long row = gid / (VISO_ROW / 8);
    const int img = (int)(row / cap);
    const int r = (int)(row % cap);
    if (r >= n_rows[img]) return;
    const float* s = src + row * dlen;
    uint32_t w[4];
    bool isbad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t pair = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = chunk * 8 + 2 * k + h;
            uint32_t u = VISO_BIAS;
            if (c < dlen) {
                const float v = s[c];
                const float vr = rintf(v);
                if (!(v == vr) || v < -32768.f || v > 32767.f) isbad = true;
                else u = (uint32_t)((int)vr + VISO_BIAS);
            }
            pair |= (u & 0xffffu) << (16 * h);
        }
        w[k] = pair;
    }
    *reinterpret_cast<uint4*>(dst + row * VISO_ROW + chunk * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    if (isbad) atomicOr(bad, 1);
}

int launch_pack(hipStream_t s, const float* src, uint16_t* dst, const int* n_rows_per_img,
                int n_img, int cap, int dlen, int* bad) {
    if (dlen > VISO_ROW) {  // rows do not fit the packed format: force the general path
        int one = 1;
        HIP_TRY(hipMemcpyAsync(bad, &one, sizeof(int), hipMemcpyHostToDevice, s));
        return VISO_OK;
    }
    const long long total = (long long)n_img * cap * (VISO_ROW / 8);
    if (total == 0) return VISO_OK;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_desc_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n_rows_per_img,
                       n_img, cap, dlen, bad);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// ------------------------------------------------------------------ scorers
// Best / second-best bookkeeping (src/viso.cpp:703-709), order independent:
// d1 = min SAD, d2 = second order statistic WITH multiplicity, key = largest
// (distance,index) among the candidates whose SAD equals d1.
struct TrackU {
    uint32_t d1, d2, kd, ki;
    __device__ __forceinline__ void init() { d1 = d2 = 0xffffffffu; kd = ki = 0; }
    __device__ __forceinline__ void add(uint32_t s, uint32_t sd, uint32_t si) {
        const bool lt = s < d1, eq = s == d1;
        const bool kgt = key_less(kd, ki, sd, si);
        const uint32_t nd2 = (lt || eq) ? d1 : min(d2, s);
        const bool take = lt || (eq && kgt);
        d2 = nd2;
        d1 = lt ? s : d1;
        kd = take ? sd : kd;
        ki = take ? si : ki;
    }
    __device__ __forceinline__ void merge(uint32_t od1, uint32_t od2, uint32_t okd, uint32_t oki) {
        if (od1 < d1) {
            d2 = min(d1, od2); d1 = od1; kd = okd; ki = oki;
        } else if (od1 == d1) {
            if (d1 != 0xffffffffu) {
                d2 = d1;
                if (key_less(kd, ki, okd, oki)) { kd = okd; ki = oki; }
            }
        } else {
            d2 = min(d2, od1);
        }
    }
};

// Fast path: 16 lanes per candidate, one 16-B load each (a 256-B row per
// 16-lane DPP row), 4 x v_sad_u16, 4 DPP adds.  4 candidates per wave pass.
struct ScorerU16 {
    const uint16_t* d2;
    uint4 q;        // this lane's 8 query elements
    TrackU t;
    int g, sub;
    __device__ __forceinline__ void begin(const MatchProblem& P, int i, int lane) {
        g = lane >> 4; sub = lane & 15;
        d2 = P.d2;
        q = *reinterpret_cast<const uint4*>(P.d1 + (size_t)i * VISO_ROW + sub * 8);
        t.init();
    }
    __device__ __forceinline__ void score(const uint2* queue, int n) {
        for (int b = 0; b < n; b += 16) {
            uint2 e[4];
            uint4 r[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int j = min(b + p * 4 + g, n - 1);
                e[p] = queue[j];
                r[p] = *reinterpret_cast<const uint4*>(d2 + (size_t)e[p].x * VISO_ROW + sub * 8);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t s = __builtin_amdgcn_sad_u16(r[p].x, q.x, 0u);
                s = __builtin_amdgcn_sad_u16(r[p].y, q.y, s);
                s = __builtin_amdgcn_sad_u16(r[p].z, q.z, s);
                s = __builtin_amdgcn_sad_u16(r[p].w, q.w, s);
                s = row16_sum(s);
                const bool valid = (b + p * 4 + g) < n;
                t.add(valid ? s : 0xffffffffu, e[p].y, e[p].x);
            }
        }
    }
    // combine the four 16-lane groups; every lane ends with the wave result
    __device__ __forceinline__ void finish(int& idx, int& dist, double& bd1, double& bd2) {
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            const uint32_t od1 = __shfl_xor(t.d1, m), od2 = __shfl_xor(t.d2, m);
            const uint32_t okd = __shfl_xor(t.kd, m), oki = __shfl_xor(t.ki, m);
            t.merge(od1, od2, okd, oki);
        }
        const bool any = t.d1 != 0xffffffffu;
        idx = any ? (int)t.ki : -1;
        dist = (int)t.d1;
        bd1 = (double)t.d1;
        bd2 = (t.d2 == 0xffffffffu) ? 1.7976931348623157e308 : (double)t.d2;
    }
};

// General path (descriptors that are not int16-valued, or dlen > 128): one
// lane per candidate, |a-b| in float summed in double in index order — the
// exact arithmetic of cv::norm(d2.row - d1.row, NORM_L1) at src/viso.cpp:702.
struct TrackD {
    double d1, d2;
    uint32_t kd, ki;
    bool any;
    __device__ __forceinline__ void init() { d1 = d2 = 1.7976931348623157e308; kd = ki = 0; any = false; }
    __device__ __forceinline__ void add(double s, uint32_t sd, uint32_t si) {
        if (!any) { d1 = s; kd = sd; ki = si; any = true; return; }
        if (s < d1) { d2 = d1; d1 = s; kd = sd; ki = si; }
        else if (s == d1) { d2 = d1; if (key_less(kd, ki, sd, si)) { kd = sd; ki = si; } }
        else if (s < d2) d2 = s;
    }
    __device__ __forceinline__ void merge(double od1, double od2, uint32_t okd, uint32_t oki, bool oany) {
        if (!oany) return;
        if (!any) { d1 = od1; d2 = od2; kd = okd; ki = oki; any = true; return; }
        if (od1 < d1) { d2 = fmin(d1, od2); d1 = od1; kd = okd; ki = oki; }
        else if (od1 == d1) { d2 = d1; if (key_less(kd, ki, okd, oki)) { kd = okd; ki = oki; } }
        else d2 = fmin(d2, od1);
    }
};

struct ScorerF32 {
    const float* f1row;
    const float* f2;
    int dlen, lane;
    TrackD t;
    __device__ __forceinline__ void begin(const MatchProblem& P, int i, int lane_, int dlen_) {
        dlen = dlen_; lane = lane_;
        f1row = P.f1 + (size_t)i * dlen;
        f2 = P.f2;
        t.init();
    }
    __device__ __forceinline__ void score(const uint2* queue, int n) {
        for (int j = lane; j < n; j += VISO_WAVE) {
            const uint2 e = queue[j];
            const float* a = f2 + (size_t)e.x * dlen;
            double s = 0;
            for (int c = 0; c < dlen; ++c) {
                const float df = a[c] - f1row[c];
                s += (double)fabsf(df);
            }
            t.add(s, e.y, e.x);
        }
    }
    __device__ __forceinline__ void finish(int& idx, int& dist, double& bd1, double& bd2) {
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) {
            const double od1 = __shfl_xor(t.d1, m), od2 = __shfl_xor(t.d2, m);
            const uint32_t okd = __shfl_xor(t.kd, m), oki = __shfl_xor(t.ki, m);
            const int oany = __shfl_xor((int)t.any, m);
            t.merge(od1, od2, okd, oki, oany != 0);
        }
        idx = t.any ? (int)t.ki : -1;
        // Vec3i(i, idx, double): conversion truncates; saturate instead of UB
        double c = t.d1 > 2147483647.0 ? 2147483647.0 : t.d1;
        dist = t.any ? (int)c : 0;
        bd1 = t.d1; bd2 = t.d2;
    }
};

// ------------------------------------------------------------------ the walk
// Target keypoints come either from LDS (staged once per workgroup) or global.
template <bool STAGED>
struct KpSrc {
    const float2* g;
    const float2* s;
    __device__ __forceinline__ float2 at(int t) const { return STAGED ? s[t] : g[t]; }
};

// Count the targets whose key=(dist,idx) is < (kd,ki), among those in radius
// and below the Q1 cut.  Wave-uniform result.
template <bool STAGED>
__device__ int count_below(const KpSrc<STAGED>& kp, int n2, float qx, float qy, float radius,
                           float d0cut, uint32_t kd, uint32_t ki, int lane) {
    int c = 0;
    for (int base = 0; base < n2; base += VISO_WAVE) {
        const int t = base + lane;
        bool in = false;
        if (t < n2) {
            const float d = l1_kp(qx, qy, kp.at(t));
            in = (d <= radius) && (d < d0cut) && key_less(__float_as_uint(d), (uint32_t)t, kd, ki);
        }
        c += __popcll(__ballot(in));
    }
    return c;
}

template <bool STAGED, class Scorer>
__device__ void match_query(const MatchProblem& P, const MatchParamsDev& mp, int i, int n2,
                            const KpSrc<STAGED>& kp, uint2* queue, Scorer& sc, int lane,
                            unsigned long long& scored) {
    const float2 q = P.kp1[i];
    const float qx = q.x, qy = q.y;
    const float radius = mp.radius;
    // Q1 (src/viso.cpp:693): the walk stops at target index 0, i.e. everything
    // at or behind key(0) = (d0, 0) is cut: dist < d0 strictly.
    float d0cut = __builtin_huge_valf();
    if (n2 > 0) {
        const float d0 = l1_kp(qx, qy, P.kp2[0]);
        if (d0 <= radius) d0cut = d0;
    }
    // ---- phase A: scan, queue the in-radius candidates
    int cnt = 0;
    for (int base = 0; base < n2; base += VISO_WAVE) {
        const int t = base + lane;
        bool in = false;
        float d = 0.f;
        if (t < n2) {
            d = l1_kp(qx, qy, kp.at(t));
            in = (d <= radius) && (d < d0cut);
        }
        const unsigned long long m = __ballot(in);
        if (m) {
            const int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                  __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (in && pos < VISO_QCAP) queue[pos] = make_uint2((uint32_t)t, __float_as_uint(d));
            cnt += __popcll(m);
        }
    }
    const int K = mp.K;
    const double* F = mp.F;
    if (cnt <= K && cnt <= VISO_QCAP) {
        // ---- fast path: whole candidate set is in the queue
        int n = cnt;
        if (mp.epi) {
            int w = 0;
            for (int b = 0; b < n; b += VISO_WAVE) {
                const int j = b + lane;
                bool pass = false;
                uint2 e = make_uint2(0, 0);
                if (j < n) {
                    e = queue[j];
                    const float2 t2 = kp.at((int)e.x);
                    const double s = sampson_dev(F, qx, qy, t2.x, t2.y);
                    pass = isfinite(s) && !(s > mp.sampson_thresh);
                }
                const unsigned long long m = __ballot(pass);
                const int pos = w + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                __builtin_amdgcn_wave_barrier();
                if (pass) queue[pos] = e;   // pos <= j: in-place compaction is safe
                w += __popcll(m);
            }
            n = w;
        }
        __builtin_amdgcn_wave_barrier();
        if (n > 0) sc.score(queue, n);
        scored += (unsigned long long)n;
        return;
    }
    // ---- slow path (dense clusters): apply the K cap, then stream in batches
    uint32_t tkd = 0xffffffffu, tki = 0xffffffffu;   // threshold key (exclusive)
    if (cnt > K) {
        // key_K = largest v with |{key < v}| <= K, built bit by bit (64-bit key)
        unsigned long long v = 0;
        for (int bit = 63; bit >= 0; --bit) {
            const unsigned long long trial = v | (1ull << bit);
            const int c = count_below(kp, n2, qx, qy, radius, d0cut, (uint32_t)(trial >> 32),
                                      (uint32_t)trial, lane);
            if (c <= K) v = trial;
        }
        tkd = (uint32_t)(v >> 32); tki = (uint32_t)v;
    }
    int qn = 0;
    for (int base = 0; base < n2; base += VISO_WAVE) {
        const int t = base + lane;
        bool in = false;
        float d = 0.f;
        if (t < n2) {
            const float2 t2 = kp.at(t);
            d = l1_kp(qx, qy, t2);
            in = (d <= radius) && (d < d0cut) && key_less(__float_as_uint(d), (uint32_t)t, tkd, tki);
            if (in && mp.epi) {
                const double s = sampson_dev(F, qx, qy, t2.x, t2.y);
                in = isfinite(s) && !(s > mp.sampson_thresh);
            }
        }
        const unsigned long long m = __ballot(in);
        const int pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                              __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (in) queue[pos] = make_uint2((uint32_t)t, __float_as_uint(d));   // qn < 64 => pos < 128
        qn += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (qn >= VISO_WAVE) {
            sc.score(queue, VISO_WAVE);
            scored += VISO_WAVE;
            const int rest = qn - VISO_WAVE;
            uint2 mv = make_uint2(0, 0);
            if (lane < rest) mv = queue[VISO_WAVE + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) queue[lane] = mv;
            __builtin_amdgcn_wave_barrier();
            qn = rest;
        }
    }
    if (qn > 0) {
        sc.score(queue, qn);
        scored += (unsigned long long)qn;
    }
}

// blockIdx -> (problem, query block).  Blocks b and b+8 share an XCD (round
// robin dispatch; speed only), so problem = f(b % 8, b / 8): all query blocks of
// one problem run on one XCD and re-read its target rows from that XCD's L2.
__device__ __forceinline__ void block_to_problem(int n_probs, int bpp, int& prob, int& qblk) {
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    prob = (slot / bpp) * 8 + xcd;
    qblk = slot % bpp;
    if (prob >= n_probs) prob = -1;
}

struct MatchArgs {
    const MatchProblem* probs;
    int n_probs, bpp, dlen, kp_lds;   // kp_lds: target keypoints the LDS staging area holds
    const int* bad;
    MatchParamsDev mp[2];
};

template <bool GENERAL>
__global__ __launch_bounds__(VISO_MATCH_THREADS) void match_kernel(MatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the pack kernel decides which variant does the work (no host round trip)
    const bool is_bad = *a.bad != 0;
    if (is_bad != GENERAL) return;
    int prob, qblk;
    block_to_problem(a.n_probs, a.bpp, prob, qblk);
    if (prob < 0) return;
    const MatchProblem P = a.probs[prob];
    const int n1 = *P.n1p, n2 = *P.n2p;
    const int q0 = qblk * VISO_QPB;
    if (q0 >= n1) return;
    const MatchParamsDev& mp = a.mp[P.pidx];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint2* queue = reinterpret_cast<uint2*>(smem) + wave * VISO_QCAP;
    float2* skp = reinterpret_cast<float2*>(smem + (VISO_MATCH_THREADS / VISO_WAVE) * VISO_QCAP * sizeof(uint2));
    const bool staged = n2 <= a.kp_lds;
    if (staged) {
        for (int t = threadIdx.x; t < n2; t += VISO_MATCH_THREADS) skp[t] = P.kp2[t];
        __syncthreads();
    }
    unsigned long long scored = 0;
    const int q1 = min(q0 + VISO_QPB, n1);
    for (int i = q0 + wave; i < q1; i += VISO_MATCH_THREADS / VISO_WAVE) {
        int idx, dist;
        double bd1, bd2;
        if constexpr (GENERAL) {
            ScorerF32 sc;
            sc.begin(P, i, lane, a.dlen);
            if (staged) { KpSrc<true> kp{P.kp2, skp}; match_query(P, mp, i, n2, kp, queue, sc, lane, scored); }
            else { KpSrc<false> kp{P.kp2, skp}; match_query(P, mp, i, n2, kp, queue, sc, lane, scored); }
            sc.finish(idx, dist, bd1, bd2);
        } else {
            ScorerU16 sc;
            sc.begin(P, i, lane);
            if (staged) { KpSrc<true> kp{P.kp2, skp}; match_query(P, mp, i, n2, kp, queue, sc, lane, scored); }
            else { KpSrc<false> kp{P.kp2, skp}; match_query(P, mp, i, n2, kp, queue, sc, lane, scored); }
            sc.finish(idx, dist, bd1, bd2);
        }
        if (lane == 0) {
            bool accept = idx >= 0;
            // src/viso.cpp:713-716 — ratio test in double (Q3)
            if (accept && mp.second) accept = bd1 < bd2 * mp.ratio;
            P.res[i] = make_int2(accept ? idx : -1, dist);
        }
    }
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
}

int launch_match_timed(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max,
                       int n2_max, int dlen, const MatchParamsDev mp[2], const int* bad,
                       hipEvent_t e0, hipEvent_t e1) {
    if (n_probs <= 0 || cap_max <= 0) return VISO_OK;
    MatchArgs a;
    a.probs = probs_dev;
    a.n_probs = n_probs;
    a.bpp = (cap_max + VISO_QPB - 1) / VISO_QPB;
    a.dlen = dlen;
    a.bad = bad;
    a.mp[0] = mp[0];
    a.mp[1] = mp[1];
    const int groups = (n_probs + 7) / 8;
    const long long blocks = (long long)groups * 8 * a.bpp;
    if (blocks > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    const int staged = n2_max <= VISO_KP_LDS_MAX ? n2_max : 0;
    a.kp_lds = staged;
    const size_t lds = (VISO_MATCH_THREADS / VISO_WAVE) * VISO_QCAP * sizeof(uint2) + (size_t)staged * sizeof(float2);
    if (lds > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void*)match_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)match_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (e0) HIP_TRY(hipEventRecord(e0, s));
    hipLaunchKernelGGL(match_kernel<false>, dim3((unsigned)blocks), dim3(VISO_MATCH_THREADS), lds, s, a);
    HIP_TRY(hipGetLastError());
    if (e1) HIP_TRY(hipEventRecord(e1, s));
    hipLaunchKernelGGL(match_kernel<true>, dim3((unsigned)blocks), dim3(VISO_MATCH_THREADS), lds, s, a);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

int launch_match(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max,
                 int n2_max, int dlen, const MatchParamsDev mp[2], const int* bad) {
    return launch_match_timed(s, probs_dev, n_probs, cap_max, n2_max, dlen, mp, bad, nullptr, nullptr);
}

extern "C" const char* viso_matcher_kernel_name(void) { return "match_kernel<false>"; }

// ------------------------------------------------------------------ sort
// std::sort(match, by [2]) of src/viso.cpp:724 with the documented total order
// (dist asc, i1 asc).  One workgroup per problem; keys (dist<<32 | i1) in LDS,
// bitonic network; rejected queries carry ~0 and sink to the end, so the sort
// is also the compaction.  Also emits pos[i1] (row of query i1, or -1) for the
// circle join, and the match count.
#define VISO_SORT_THREADS 512
#define VISO_SORT_MAX 16384

__global__ __launch_bounds__(VISO_SORT_THREADS) void sort_matches_kernel(const MatchProblem* probs,
                                                                         int n_probs, int npad_alloc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    const int prob = blockIdx.x;
    if (prob >= n_probs) return;
    const MatchProblem P = probs[prob];
    const int n1 = *P.n1p;
    int npad = 64;
    while (npad < n1) npad <<= 1;
    for (int i = threadIdx.x; i < npad; i += VISO_SORT_THREADS) {
        unsigned long long k = ~0ull;
        if (i < n1) {
            const int2 r = P.res[i];
            if (r.x >= 0) k = ((unsigned long long)(uint32_t)r.y << 32) | (uint32_t)i;
            P.pos[i] = -1;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npad >> 1); t += VISO_SORT_THREADS) {
                const int i = ((t / j) * 2 * j) + (t % j);
                const int l = i + j;
                const bool up = (i & k) == 0;
                const unsigned long long a = keys[i], b = keys[l];
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    int local = 0;
    for (int r = threadIdx.x; r < n1; r += VISO_SORT_THREADS) {
        const unsigned long long k = keys[r];
        if (k != ~0ull) {
            const int i1 = (int)(uint32_t)k;
            const int2 rr = P.res[i1];
            P.sorted[3 * r + 0] = i1;
            P.sorted[3 * r + 1] = rr.x;
            P.sorted[3 * r + 2] = (int)(uint32_t)(k >> 32);
            P.pos[i1] = r;
            ++local;
        }
    }
    // count = number of valid keys (tail word of the dynamic LDS area)
    int* s_cnt = reinterpret_cast<int*>(keys + npad_alloc);
    if (threadIdx.x == 0) *s_cnt = 0;
    __syncthreads();
    if (local) atomicAdd(s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) *P.m_cnt = *s_cnt;
}

int launch_sort(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max) {
    if (n_probs <= 0) return VISO_OK;
    if (cap_max > VISO_SORT_MAX) {
        viso_set_error("match_desc: more than %d queries per call is not supported by this build", VISO_SORT_MAX);
        return VISO_ERR_UNSUPPORTED;
    }
    int npad = 64;
    while (npad < cap_max) npad <<= 1;
    if ((size_t)npad * 8 + 16 > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)sort_matches_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, npad * 8 + 16));
    hipLaunchKernelGGL(sort_matches_kernel, dim3(n_probs), dim3(VISO_SORT_THREADS),
                       (size_t)npad * sizeof(unsigned long long) + 16, s, probs_dev, n_probs, npad);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

void fill_match_params(MatchParamsDev* d, const viso_match_params* h) {
    d->epi = h->enforce_epipolar != 0;
    d->second = h->enforce_2nd_best != 0;
    d->K = h->max_neighbors;
    d->_pad = 0;
    d->radius = (float)h->radius;   // src/viso.cpp:685: double -> float parameter
    d->_padf = 0;
    for (int i = 0; i < 9; ++i) d->F[i] = h->F[i];
    d->sampson_thresh = h->sampson_thresh;
    d->ratio = h->ratio_2nd_best;
}
