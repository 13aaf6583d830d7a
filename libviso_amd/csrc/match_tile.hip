// match_tile.hip — LDS-resident window variant of the u16 matcher (the hot
// kernel).  Same results as match_kernel<false> (match.hip); different data
// movement:
//
//   * a workgroup (8 waves) owns a tile of 128 x-sorted queries; the target
//     rows of the tile's +-radius column window are staged ONCE into LDS
//     (coalesced 16-B pieces) and re-used by all 128 queries (~12x reuse), so
//     the scoring loop never waits on L2;
//   * scoring is one LANE per candidate: no cross-lane reduction.  Rows are
//     stored with their 16-B chunks rotated by the row index
//     (slot = (chunk + row) & 15) and lane l walks its row starting at slot
//     (l & 15), so the 16 lanes of every ds_read_b128 group touch 16 distinct
//     slots: conflict-free for ANY candidate set.  The matching query chunk is
//     read from a per-wave LDS copy of the query row (same address ->
//     broadcast, different chunk -> different banks: also conflict-free);
//   * windows larger than the LDS budget (dense keypoints, e.g. 8000/image) are
//     processed in chunks with the per-query best/second-best state kept in LDS;
//   * anything irregular — more than K in-radius candidates, more than 128
//     candidates in a chunk, an exact SAD tie — is left to
//     match_overflow_kernel (exact K-cap selection and largest-key tie rule).
#include "common.h"
#include "match_dev.h"

#define T3_THREADS 1024
#define T3_WAVES 16
#define T3_QB 128          // queries per tile
#define T3_WCAP 448        // window rows resident in LDS per chunk
#define T3_QCAP 128        // candidates per query per chunk
#define T3_ROWB 256        // bytes per row

struct TileArgs {
    const MatchProblem* probs;
    int n_probs, bpp, _p0, _p1;
    const int* bad;
    MatchParamsDev mp[2];
};

// LDS carve (bytes)
#define T3_OFF_ROWS 0
#define T3_OFF_QROW (T3_OFF_ROWS + T3_WCAP * T3_ROWB)            // [8][512] query row twice
#define T3_OFF_KP (T3_OFF_QROW + T3_WAVES * 512)                // float2[WCAP]
#define T3_OFF_IDX (T3_OFF_KP + T3_WCAP * 8)                    // int[WCAP]
#define T3_OFF_QUEUE (T3_OFF_IDX + T3_WCAP * 4)                 // uint2[8][QCAP]
#define T3_OFF_STATE (T3_OFF_QUEUE + T3_WAVES * T3_QCAP * 8)    // uint32[6][QB]
#define T3_OFF_MISC (T3_OFF_STATE + 6 * T3_QB * 4)
#define T3_LDS_BYTES (T3_OFF_MISC + 64)

// wave-wide unsigned min, result uniform (DPP; gfx9 row_bcast forms)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    const int ident = -1;   // 0xffffffff
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x140, 0xf, 0xf, false));  // row_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(ident, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(T3_THREADS) void match_tile_kernel(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (*a.bad != 0) return;   // non-integer descriptors: the general kernel does the work
    int prob, tile;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        prob = (slot / a.bpp) * 8 + xcd;
        tile = slot % a.bpp;
        if (prob >= a.n_probs) return;
    }
    const MatchProblem P = a.probs[prob];
    const int n1 = *P.q.n, n2 = *P.t.n;
    const int q0 = tile * T3_QB;
    if (q0 >= n1) return;
    const int q1 = min(q0 + T3_QB, n1);
    const MatchParamsDev& mp = a.mp[P.pidx];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    unsigned char* s_rows = smem + T3_OFF_ROWS;
    unsigned char* s_qrow = smem + T3_OFF_QROW + wave * 512;
    float2* s_kp = reinterpret_cast<float2*>(smem + T3_OFF_KP);
    int* s_idx = reinterpret_cast<int*>(smem + T3_OFF_IDX);
    uint2* queue = reinterpret_cast<uint2*>(smem + T3_OFF_QUEUE) + wave * T3_QCAP;
    uint32_t* st_d1 = reinterpret_cast<uint32_t*>(smem + T3_OFF_STATE);
    uint32_t* st_d2 = st_d1 + T3_QB;
    uint32_t* st_bi = st_d2 + T3_QB;     // ORIGINAL index of a target reaching d1
    uint32_t* st_fl = st_bi + T3_QB;     // bit0 tie, bit1 overflow
    uint32_t* st_cnt = st_fl + T3_QB;    // in-radius candidates so far (K cap)
    uint32_t* st_sc = st_cnt + T3_QB;    // SAD evaluations so far (counted only if the result stands)
    float* s_xr = reinterpret_cast<float*>(smem + T3_OFF_MISC);

    // ---- tile x range (queries are x-sorted; NaNs sort last and are ignored)
    if (wave == 0) {
        const float nanv = __builtin_nanf("");
        float x0 = (q0 + lane < q1) ? P.q.skp[q0 + lane].x : nanv;
        float x1 = (q0 + 64 + lane < q1) ? P.q.skp[q0 + 64 + lane].x : nanv;
        float mn = fminf(x0, x1), mx = fmaxf(x0, x1);
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, m));
            mx = fmaxf(mx, __shfl_xor(mx, m));
        }
        if (lane == 0) { s_xr[0] = mn; s_xr[1] = mx; }
    }
    if (tid < T3_QB) {
        st_d1[tid] = 0xffffffffu; st_d2[tid] = 0xffffffffu; st_bi[tid] = 0; st_fl[tid] = 0; st_cnt[tid] = 0; st_sc[tid] = 0;
    }
    __syncthreads();
    int lo = 0, W = 0;
    float xslack = 1e-6f;
    {
        const float xa = s_xr[0], xb = s_xr[1];
        xslack = (fabsf(xa) + fabsf(xb) + fabsf(mp.radius)) * 1e-6f + 1e-6f;
        const float r = mp.radius;
        if (n2 > 0 && xa == xa && r >= 0.f) {
            const float slack = (fabsf(xa) + fabsf(xb) + fabsf(r)) * 1e-6f + 1e-6f;
            const float x0 = P.t.xinfo[0], scale = P.t.xinfo[1];
            const int blo = bucket_of(xa - r - slack, x0, scale);
            const int bhi = bucket_of(xb + r + slack, x0, scale);
            lo = P.t.bstart[blo];
            W = P.t.bstart[bhi + 1] - lo;
        }
    }
    float2 kp0 = make_float2(0.f, 0.f);
    const bool has0 = n2 > 0;
    if (has0) kp0 = P.t.skp[P.t.rank[0]];
    const float radius = mp.radius;
    const int K = mp.K;

    for (int c0 = 0; c0 < W; c0 += T3_WCAP) {
        const int Wc = min(T3_WCAP, W - c0);
        __syncthreads();   // everyone is done with the previous chunk
        // ---- stage the chunk: 16-B pieces, chunk c of local row rl goes to slot (c + rl) & 15
        {
            const unsigned char* g = reinterpret_cast<const unsigned char*>(P.t.rows) + (size_t)(lo + c0) * T3_ROWB;
            const int pieces = Wc * 16;
            for (int p = tid; p < pieces; p += T3_THREADS) {
                const int rl = p >> 4, slot = p & 15;
                const int c = (slot - rl) & 15;
                const uint4 v = *reinterpret_cast<const uint4*>(g + (size_t)rl * T3_ROWB + c * 16);
                *reinterpret_cast<uint4*>(s_rows + rl * T3_ROWB + slot * 16) = v;
            }
            for (int w = tid; w < Wc; w += T3_THREADS) {
                s_kp[w] = P.t.skp[lo + c0 + w];
                s_idx[w] = P.t.sidx[lo + c0 + w];
            }
        }
        __syncthreads();
        // ---- every wave walks its queries against the resident chunk; the next
        // query's keypoint and row dword are fetched while the current one is scored
        float2 q_next = make_float2(0.f, 0.f);
        uint32_t v_next = 0;
        if (q0 + wave < q1) {
            q_next = P.q.skp[q0 + wave];
            v_next = reinterpret_cast<const uint32_t*>(P.q.rows + (size_t)(q0 + wave) * VISO_ROW)[lane];
        }
        for (int j = q0 + wave; j < q1; j += T3_WAVES) {
            const int qi = j - q0;
            const float2 q = q_next;
            const float qx = q.x, qy = q.y;
            // query row -> LDS, twice back to back (chunk walk never wraps)
            {
                const uint32_t v = v_next;
                reinterpret_cast<uint32_t*>(s_qrow)[lane] = v;
                reinterpret_cast<uint32_t*>(s_qrow)[64 + lane] = v;
            }
            if (j + T3_WAVES < q1) {
                q_next = P.q.skp[j + T3_WAVES];
                v_next = reinterpret_cast<const uint32_t*>(P.q.rows + (size_t)(j + T3_WAVES) * VISO_ROW)[lane];
            }
            float d0cut = __builtin_huge_valf();   // Q1, src/viso.cpp:693
            if (has0) {
                const float d0 = l1_kp(qx, qy, kp0);
                if (d0 <= radius) d0cut = d0;
            }
            // ---- scan the chunk's keypoints, queue the in-radius ones
            int cnt = 0;
            const float xlo = qx - radius - xslack, xhi = qx + radius + xslack;
            for (int base = 0; base < Wc; base += VISO_WAVE) {
                // the chunk is x-sorted: skip 64-blocks entirely left of the query's column range, stop right of it
                if (s_kp[min(base + VISO_WAVE - 1, Wc - 1)].x < xlo) continue;
                if (s_kp[base].x > xhi) break;
                const int w = base + lane;
                bool in = false;
                float d = 0.f;
                if (w < Wc) {
                    d = l1_kp(qx, qy, s_kp[w]);
                    in = (d <= radius) && (d < d0cut);
                }
                const unsigned long long m = __ballot(in);
                if (m) {
                    const int pos = cnt + mbcnt(m);
                    if (in && pos < T3_QCAP) queue[pos] = make_uint2((uint32_t)w, __float_as_uint(d));
                    cnt += __popcll(m);
                }
            }
            if (cnt == 0) continue;
            if (cnt > T3_QCAP) {
                if (lane == 0) st_fl[qi] |= 2u;
                continue;
            }
            if (lane == 0) st_cnt[qi] += (uint32_t)cnt;
            int n = cnt;
            if (mp.epi) {   // Sampson gate, one candidate per lane, in-place compaction
                int wr = 0;
                for (int b = 0; b < n; b += VISO_WAVE) {
                    const int k = b + lane;
                    bool pass = false;
                    uint2 e = make_uint2(0, 0);
                    if (k < n) {
                        e = queue[k];
                        const float2 t2 = s_kp[e.x];
                        const double s = sampson_dev(mp.F, qx, qy, t2.x, t2.y);
                        pass = isfinite(s) && !(s > mp.sampson_thresh);
                    }
                    const unsigned long long m = __ballot(pass);
                    const int pos = wr + mbcnt(m);
                    __builtin_amdgcn_wave_barrier();
                    if (pass) queue[pos] = e;
                    wr += __popcll(m);
                }
                n = wr;
            }
            __builtin_amdgcn_wave_barrier();
            if (n == 0) continue;
            if (lane == 0) st_sc[qi] += (uint32_t)n;
            // ---- score: one lane per candidate, rounds of 64
            uint32_t r_d1 = 0xffffffffu, r_d2 = 0xffffffffu, r_bi = 0, r_tie = 0;
            for (int b = 0; b < n; b += VISO_WAVE) {
                const bool valid = (b + lane) < n;
                const uint2 e = queue[min(b + lane, n - 1)];
                const int rl = (int)e.x;
                const int l15 = lane & 15;
                const unsigned char* a1 = s_rows + rl * T3_ROWB + l15 * 16;      // slot (l&15) of my row
                const unsigned char* qa = s_qrow + ((l15 - rl) & 15) * 16;       // the query chunk stored there
                uint32_t s = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    // slot (l15 + k) & 15: wraps back by 256 B for lanes with l15 + k >= 16
                    const unsigned char* pa = (l15 + k >= 16) ? (a1 + k * 16 - T3_ROWB) : (a1 + k * 16);
                    const uint4 rv = *reinterpret_cast<const uint4*>(pa);
                    const uint4 qv = *reinterpret_cast<const uint4*>(qa + k * 16);
                    s = __builtin_amdgcn_sad_u16(rv.x, qv.x, s);
                    s = __builtin_amdgcn_sad_u16(rv.y, qv.y, s);
                    s = __builtin_amdgcn_sad_u16(rv.z, qv.z, s);
                    s = __builtin_amdgcn_sad_u16(rv.w, qv.w, s);
                }
                s = valid ? s : 0xffffffffu;
                const uint32_t m1 = wave_min_u32(s);
                const bool eq = valid && s == m1;
                const unsigned long long em = __ballot(eq);
                const int c = __popcll(em);
                const uint32_t m2 = wave_min_u32(eq ? 0xffffffffu : s);
                const int first = __ffsll((long long)em) - 1;
                const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane(s_idx[rl], first);
                // merge the round (m1, c>1 ? m1 : m2, bi, tie) into the running result
                const uint32_t o2 = c > 1 ? m1 : m2;
                const uint32_t ot = c > 1 ? 1u : 0u;
                if (m1 < r_d1) { r_d2 = min(r_d1, o2); r_d1 = m1; r_bi = bi; r_tie = ot; }
                else if (m1 == r_d1) { r_d2 = r_d1; r_tie = 1; }
                else r_d2 = min(r_d2, m1);
            }
            // ---- merge into the per-query state (other chunks)
            if (lane == 0) {
                const uint32_t d1 = st_d1[qi], d2 = st_d2[qi];
                if (r_d1 < d1) { st_d2[qi] = min(d1, r_d2); st_d1[qi] = r_d1; st_bi[qi] = r_bi; st_fl[qi] = (st_fl[qi] & ~1u) | r_tie; }
                else if (r_d1 == d1) { st_d2[qi] = d1; st_fl[qi] |= 1u; }
                else st_d2[qi] = min(d2, r_d1);
            }
        }
    }
    __syncthreads();
    // ---- results
    unsigned long long scored = 0;
    if (tid < q1 - q0) {
        const int qi = tid, j = q0 + qi;
        const uint32_t d1 = st_d1[qi], d2 = st_d2[qi], fl = st_fl[qi];
        if (fl != 0 || st_cnt[qi] > (uint32_t)K) {
            P.ovf[atomicAdd(P.ovf_cnt, 1)] = j;   // K cap / queue overflow / exact tie: overflow kernel
        } else {
            bool accept = d1 != 0xffffffffu;
            if (accept && mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                const double bd2 = d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)d2;
                accept = (double)d1 < bd2 * mp.ratio;
            }
            P.res[P.q.sidx[j]] = make_int2(accept ? (int)st_bi[qi] : -1, (int)d1);
            scored = st_sc[qi];
        }
    }
    if (wave < 2) {   // threads 0..127 hold the per-query counts
#pragma unroll
        for (int m = 1; m < VISO_WAVE; m <<= 1) scored += __shfl_xor(scored, m);
        if (lane == 0 && scored) atomicAdd(P.scored, scored);
    }
}

int launch_match_tile(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max,
                      const MatchParamsDev mp[2], const int* bad) {
    TileArgs a;
    a.probs = probs_dev;
    a.n_probs = n_probs;
    a.bpp = (cap_max + T3_QB - 1) / T3_QB;
    a._p0 = a._p1 = 0;
    a.bad = bad;
    a.mp[0] = mp[0];
    a.mp[1] = mp[1];
    const int groups = (n_probs + 7) / 8;
    const long long blocks = (long long)groups * 8 * a.bpp;
    if (blocks > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)match_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(match_tile_kernel, dim3((unsigned)blocks), dim3(T3_THREADS), T3_LDS_BYTES, s, a);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}
