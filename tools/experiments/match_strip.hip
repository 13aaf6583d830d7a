// match_strip.hip — temporal (no epipolar gate) matcher with the target window's descriptor rows RESIDENT IN LDS,
// one persistent workgroup per run of tiles.
//
// match_union_kernel (match_union.hip) gathers every candidate row from the XCD's L2 (8 lanes x 2 x 16 B per row):
// 4.9 GB of L2 reads per 512-problem launch, and its scoring phase runs AT the chip's L2 row-gather rate
// (~18 TB/s, MI355X_MICROARCH.md "Indexed rows").  The rows a tile of x-adjacent queries can ever need are the
// CONTIGUOUS range [lo, hi) of the x-sorted target image (its +-radius column window, ~390 rows = 100 KB for 128
// queries at 2000 keypoints/image): they fit a CU's 160 KB of LDS, and the windows of consecutive tiles overlap by
// two thirds.  So
//
//   block   one 1024-thread workgroup (one per CU) walks a SEGMENT of consecutive 128-query tiles of one problem;
//   ring    the window rows live in a ring of 512 row slots (128 KB; slot = sorted position & 511), filled by
//           asynchronous linear copies (global_load_lds_dwordx4: 1 KiB = 4 rows per wave-instruction, no VGPRs);
//           while a tile is being matched, the rows its successor adds to the window (a third of it) are already
//           in flight into the slots the current tile no longer needs: no load latency on the critical path,
//           ~1/8 of the gather kernel's L2 traffic, none of it indexed;
//   index   per tile the window's keypoints are bucket-sorted by y (as in match_union_kernel): a round only scans
//           the buckets its four diamonds touch;
//   round   four y-adjacent queries per wave (y order inside 64-blocks: ImageView::qord, from sort_kp_kernel): one
//           scan -> membership masks + union list; rolling pipeline over the list, 8 lanes per row, rows read from
//           LDS by ds_read_b128 (lanes with bit 4 set read the two 128-B halves in the opposite order: every
//           16-lane service group of the instruction then hits 64 distinct banks), the four query rows in
//           registers, 8 x v_sad_u16 per query, transposing DPP reduction, packed-key (min, second min, tie)
//           tracker; merge, ratio test, store.
//
// Windows wider than the ring (dense keypoints) are processed in chunks with the per-query state carried in LDS.
// Same results as the other matcher kernels (bit-exact, parity tests).  Irregular queries (more than K in-radius
// candidates, a union list that does not fit, an exact tie of the minimum) go to match_overflow_kernel.
#include "common.h"      // libviso_amd/csrc (built with -I of that directory: make DEBUG_VARIANTS=1)
#include "match_dev.h"

#include <stdlib.h>

#define MS_THREADS 1024
#define MS_WAVES 16
#define MS_QPB 128         // queries per tile
#define MS_SUB 64          // queries per y-ranked sub-block
#define MS_G 4             // queries per round
#define MS_ROUNDS (MS_QPB / (MS_WAVES * MS_G))   // 2 rounds per wave and tile
#define MS_RING 512        // row slots (power of two)
#define MS_CHUNK 504       // rows of a window processed at once (+ alignment to 4-row pieces <= MS_RING)
#define MS_NBY 64          // y buckets of a chunk
#define MS_UCAP 224        // union list entries per round
#define MS_PAD 32          // list padding: the pipeline runs passes of 8 rows past the end
#define MS_NP 3            // passes in flight
#define MS_TMAX 64         // tiles per segment (LDS metadata)

// LDS carve (bytes); the ring sits at offset 0 so that (slot << 8) IS the LDS address of a row
#define MS_OFF_ROWS 0
#define MS_OFF_KPR (MS_OFF_ROWS + MS_RING * 256)                       // float2[MS_RING]: keypoints, same slots
#define MS_OFF_YKP (MS_OFF_KPR + MS_RING * 8)                          // float2[MS_RING]: chunk keypoints in y-bucket order
#define MS_OFF_YPOS (MS_OFF_YKP + MS_RING * 8)                         // uint16[MS_RING]: their ring slots
#define MS_OFF_YS (MS_OFF_YPOS + MS_RING * 2)                          // int[MS_NBY + 1] (+ pad)
#define MS_OFF_UL (MS_OFF_YS + 272)
#define MS_OFF_STATE (MS_OFF_UL + MS_WAVES * (MS_UCAP + MS_PAD) * 4)   // uint4 per query (chunked windows only):
                                                                       //   d1, d2, winner position, count | tie << 29 | force << 30
#define MS_OFF_TLO (MS_OFF_STATE + MS_QPB * 16)                        // int[MS_TMAX] window start per tile of the segment
#define MS_OFF_THI (MS_OFF_TLO + MS_TMAX * 4)
#define MS_OFF_QORD (MS_OFF_THI + MS_TMAX * 4)                         // uint8[2][MS_QPB]: y order of this and the next tile
#define MS_LDS_BYTES (MS_OFF_QORD + 2 * MS_QPB)
static_assert(MS_LDS_BYTES <= 163840, "LDS budget of one CU");

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

// packed-key order statistics (see match_union.hip): key = SAD << 9 | position in the round's union list
struct MsTrack { uint32_t m1, m2; };

__device__ __forceinline__ void ms_update(MsTrack& t, uint32_t key) {
    uint32_t med;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(t.m1), "v"(t.m2), "v"(key));
    t.m2 = med;
    t.m1 = min(t.m1, key);
}

__device__ __forceinline__ void ms_merge(MsTrack& a, const MsTrack& b) {
    const uint32_t hi = max(a.m1, b.m1);
    a.m2 = min(hi, min(a.m2, b.m2));
    a.m1 = min(a.m1, b.m1);
}

__device__ __forceinline__ int ms_ybucket(float y, float y0, float scale) {   // monotone in y
    if (y != y) return MS_NBY - 1;
    const float f = floorf((y - y0) * scale);
    return f <= 0.f ? 0 : (f >= (float)(MS_NBY - 1) ? MS_NBY - 1 : (int)f);
}

struct StripArgs {
    BatchMatchArgs b;      // b.bpp = segments per problem
    int tiles_per_seg;
    int debug;
};

__global__ __launch_bounds__(MS_THREADS) void match_strip_kernel(StripArgs sa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const BatchMatchArgs& a = sa.b;
    unsigned char* const s_rows = smem + MS_OFF_ROWS;
    float2* const s_kpr = reinterpret_cast<float2*>(smem + MS_OFF_KPR);
    float2* const s_ykp = reinterpret_cast<float2*>(smem + MS_OFF_YKP);
    uint16_t* const s_ypos = reinterpret_cast<uint16_t*>(smem + MS_OFF_YPOS);
    int* const s_ys = reinterpret_cast<int*>(smem + MS_OFF_YS);
    uint4* const s_state = reinterpret_cast<uint4*>(smem + MS_OFF_STATE);
    int* const s_tlo = reinterpret_cast<int*>(smem + MS_OFF_TLO);
    int* const s_thi = reinterpret_cast<int*>(smem + MS_OFF_THI);
    uint8_t* const s_qord = smem + MS_OFF_QORD;
    int prob, seg;
    {
        const int b = blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        const int g = slot / a.bpp;
        prob = ((g / a.gc) * a.gs + a.gf + g % a.gc) * 8 + xcd;
        seg = slot % a.bpp;
        if (prob >= a.n_probs) return;
    }
    const MatchProblem P = a.probs[prob];
    if ((*P.q.bad | *P.t.bad) != 0) return;   // non-integer descriptors: the general kernel does this problem
    const int n1 = *P.q.n, n2 = *P.t.n;
    const MatchParamsDev& mp = a.mp[P.pidx];
    if (mp.epi != 0) return;   // stereo problems: match_stereo_kernel / match_batch_kernel<1>
    const int t_first = seg * sa.tiles_per_seg;
    const int t_end = min((n1 + MS_QPB - 1) / MS_QPB, t_first + sa.tiles_per_seg);
    if (t_first >= t_end) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float radius = mp.radius;
    const int K = mp.K;
    // ---- segment prologue: window [lo, hi) of every tile (queries are x-sorted: a tile's x range is its first and
    // its last keypoint with a real x); the keypoint ring starts out as NaNs (never in radius)
    if (tid < t_end - t_first) {
        const int q0 = (t_first + tid) * MS_QPB;
        const int q1v = min(min(q0 + MS_QPB, n1), (int)P.q.xinfo[4]);
        int lo = 0, hi = 0;
        if (q1v > q0 && n2 > 0 && radius >= 0.f) {
            const float xa = P.q.skp[q0].x, xb = P.q.skp[q1v - 1].x;
            const float slack = (fabsf(xa) + fabsf(xb) + fabsf(radius)) * 1e-6f + 1e-6f;
            const float x0 = P.t.xinfo[0], scale = P.t.xinfo[1];
            lo = P.t.bstart[bucket_of(xa - radius - slack, x0, scale)];
            hi = P.t.bstart[bucket_of(xb + radius + slack, x0, scale) + 1];
        }
        s_tlo[tid] = lo; s_thi[tid] = max(hi, lo);
    }
    if (tid < MS_RING) s_kpr[tid] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
    float2 kp0 = make_float2(0.f, 0.f);
    const bool has0 = n2 > 0;
    if (has0) kp0 = P.t.skp[P.t.rank[0]];
    const float ty0 = P.t.xinfo[2];
    float yscale = 0.f;
    {
        const float ty1 = P.t.xinfo[3];
        if (ty1 > ty0) yscale = (float)MS_NBY / (ty1 - ty0);
        if (!(yscale > 0.f) || !(yscale < 3.0e38f)) yscale = 0.f;
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* grow_t;
    typedef const __attribute__((address_space(1))) char* gbytes_t;
    typedef __attribute__((address_space(3))) void* lds_t;
    const gbytes_t trows = (gbytes_t)reinterpret_cast<const char*>(P.t.rows);
    uint32_t* const ul = reinterpret_cast<uint32_t*>(smem + MS_OFF_UL) + wave * (MS_UCAP + MS_PAD);
    const int g8 = lane >> 3, sub = lane & 7;
    // the lane's two 16-B chunks of a 256-B row, in the order it reads them: lanes with bit 4 set take the upper
    // 128-B half first (conflict-free ds_read_b128, see the header)
    const uint32_t sw = (lane >> 4) & 1u;
    const uint32_t off_a = (uint32_t)(sub + 8 * sw) << 4, off_b = (uint32_t)(sub + 8 * (1 - sw)) << 4;
    // which of the round's four queries this lane tracks after the transposing reduction: lanes 4..7 mirror 3..0
    const bool sel0 = ((lane ^ (lane >> 2)) & 1) != 0, sel1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0;
    const int msh = 31 - ((sel0 ? 1 : 0) + (sel1 ? 2 : 0));   // its membership bit in a list entry
    unsigned long long scored = 0;
    // the wave's rounds of a tile: ranks [rbase + r * 4, +4) of the tile's y order (two 64-blocks per tile)
    const int rbase = (wave / (MS_WAVES / 2)) * MS_SUB + (wave % (MS_WAVES / 2)) * (MS_ROUNDS * MS_G);
    // rows / keypoints of sorted positions [pa, pb) (multiples of 4, at most MS_RING of them) into their ring slots;
    // every wave takes pieces.  The keypoint of position pa + tid is only LOADED here (kp_p / kp_v); kp_commit() stores
    // it to the ring later, so that no wave waits for a load right behind the asynchronous row copies it has just issued.
    int kp_p = -1;
    float2 kp_v = make_float2(0.f, 0.f);
    auto fill = [&](int pa, int pb) {
        if (sa.debug & 2) return;
        kp_p = pa + tid < pb ? pa + tid : -1;
        kp_v = make_float2(__builtin_nanf(""), __builtin_nanf(""));
        if (kp_p >= 0 && kp_p < n2) kp_v = P.t.skp[kp_p];
        for (int p = (pa >> 2) + wave; p < (pb >> 2); p += MS_WAVES) {
            const int row = min(p * 4 + (lane >> 4), max(n2 - 1, 0));   // positions past the image: any valid row (never referenced)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(trows + (size_t)row * 256 + (lane & 15) * 16),
                                             (lds_t)(s_rows + (p & (MS_RING / 4 - 1)) * 1024), 16, 0, 0);
        }
    };
    auto kp_commit = [&]() {
        if (kp_p >= 0) s_kpr[kp_p & (MS_RING - 1)] = kp_v;
        kp_p = -1;
    };
    // query data of a round, loaded one round ahead: lane l carries offset / keypoint / original index of query (l & 3)
    int pf_t = -1, pf_r = -1, pli = 0, po = -1;
    float2 pq = make_float2(0.f, 0.f);
    auto prefetch = [&](int t, int r) {
        const int q0 = t * MS_QPB, q1 = min(q0 + MS_QPB, n1);
        const int rk = rbase + r * MS_G + (lane & (MS_G - 1));
        pli = (rk & ~(MS_SUB - 1)) + (int)s_qord[(t & 1) * MS_QPB + rk];
        const int j_ = q0 + pli;
        const int jc_ = min(j_, q1 - 1);
        pq = P.q.skp[jc_];
        po = j_ < q1 ? P.q.sidx[jc_] : -1;
        pf_t = t; pf_r = r;
    };
    int have_a = 0, have_b = 0;   // sorted positions whose rows are (or are being) loaded: [have_a, have_b), uniform
    // y order of a tile's two 64-blocks (ImageView::qord; identity past the image) into the LDS buffer of its parity
    const int n64 = (n1 + MS_SUB - 1) & ~(MS_SUB - 1);
    auto load_qord = [&](int t) {
        if (tid < MS_QPB) {
            const int j = t * MS_QPB + tid;
            s_qord[(t & 1) * MS_QPB + tid] = j < n64 ? P.q.qord[j] : (uint8_t)(tid & (MS_SUB - 1));
        }
    };
    load_qord(t_first);
    __syncthreads();
    prefetch(t_first, 0);

    for (int t = t_first; t < t_end; ++t) {
        const int q0 = t * MS_QPB, q1 = min(q0 + MS_QPB, n1);
        const int lo = __builtin_amdgcn_readfirstlane(s_tlo[t - t_first]);
        const int hi = __builtin_amdgcn_readfirstlane(s_thi[t - t_first]);
        for (int cl = lo; cl == lo || cl < hi; cl += MS_CHUNK) {
            const int ch = min(hi, cl + MS_CHUNK);
            const int cw = ch - cl;
            const bool first = cl == lo, last = ch >= hi;
            __syncthreads();   // S0: every wave is done with the previous chunk (rows, y index)
            kp_commit();       // keypoints of the rows prefetched during the previous tile
            // ---- make sure [cl, ch) is in the ring: usually the prefetch issued during the previous tile covered it
            {
                const int ca = cl & ~3, cbb = (ch + 3) & ~3;
                if (ca >= have_a && ca <= have_b) {
                    if (cbb > have_b) { fill(have_b, cbb); have_b = cbb; have_a = max(have_a, have_b - MS_RING); }
                } else {
                    fill(ca, cbb); have_a = ca; have_b = cbb;
                }
            }
            kp_commit();
            if (first && t + 1 < t_end) load_qord(t + 1);
            if (first && !last && tid < MS_QPB)   // chunked window: per-query state carried in LDS
                s_state[tid] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
            if (tid <= MS_NBY) s_ys[tid] = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // S1: rows and keypoints of the chunk have landed
            // ---- y index of the chunk (bucket sort: histogram with returning LDS atomics, scan, scatter)
            const int cwpad = (cw + 127) & ~127;
            float2 e_kp = make_float2(0.f, 0.f);
            int e_b = 0, e_r = 0;
            if (tid < cw && !(sa.debug & 16)) {
                e_kp = s_kpr[(cl + tid) & (MS_RING - 1)];
                e_b = ms_ybucket(e_kp.y, ty0, yscale);
                e_r = atomicAdd(&s_ys[e_b], 1);
            }
            __syncthreads();   // S2
            if (wave == 0) {
                const int h = s_ys[lane];
                int incl = h;
#pragma unroll
                for (int d = 1; d < VISO_WAVE; d <<= 1) {
                    const int o = __shfl_up(incl, d);
                    if (lane >= d) incl += o;
                }
                s_ys[lane] = incl - h;
                if (lane == VISO_WAVE - 1) s_ys[MS_NBY] = incl;
            }
            __syncthreads();   // S3
            if (tid < cw) {
                const int p = s_ys[e_b] + e_r;
                s_ykp[p] = e_kp;
                s_ypos[p] = (uint16_t)((cl + tid) & (MS_RING - 1));
            } else if (tid < cwpad) {
                s_ykp[tid] = make_float2(__builtin_nanf(""), __builtin_nanf(""));   // [cw, cwpad): never in radius
                s_ypos[tid] = 0;
            }
            __syncthreads();   // S4
            // ---- prefetch: what the next tile adds to the window, into slots this chunk does not use
            if (last && t + 1 < t_end) {
                const int nlo = __builtin_amdgcn_readfirstlane(s_tlo[t + 1 - t_first]);
                const int nhi = __builtin_amdgcn_readfirstlane(s_thi[t + 1 - t_first]);
                const int na = nlo & ~3;
                const int nb = min((min(nhi, nlo + MS_CHUNK) + 3) & ~3, (cl & ~3) + MS_RING);
                if (na >= have_a && na <= have_b && nb > have_b) {
                    fill(have_b, nb); have_b = nb; have_a = max(have_a, have_b - MS_RING);
                }
            }

            for (int r = 0; r < MS_ROUNDS && !(sa.debug & 8); ++r) {
                // ---------------- round setup: scalars of the four queries (loaded one round ahead)
                if (pf_t != t || pf_r != r) prefetch(t, r);
                float2 qk[MS_G];
                int orig[MS_G], jq[MS_G], cnt[MS_G], pl[MS_G];
                uint32_t thr[MS_G];
                bool any_live = false;
#pragma unroll
                for (int k = 0; k < MS_G; ++k) {
                    qk[k].x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.x), k));
                    qk[k].y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pq.y), k));
                    orig[k] = __builtin_amdgcn_readlane(po, k);
                    pl[k] = __builtin_amdgcn_readlane(pli, k);
                    jq[k] = q0 + pl[k];
                    cnt[k] = 0;
                    // (d <= radius && d < d0cut) as ONE unsigned compare of the bits of d = |dx| + |dy| (see match_union.hip)
                    uint32_t tt = __float_as_uint(radius) + 1u;
                    if (has0) {
                        const float d0 = l1_kp(qk[k].x, qk[k].y, kp0);
                        if (d0 <= radius) tt = __float_as_uint(d0);
                    }
                    thr[k] = orig[k] >= 0 ? tt : 0u;
                    any_live = any_live || orig[k] >= 0;
                }
                // next round of this wave: this tile, the tile's next chunk, or the next tile
                if (r + 1 < MS_ROUNDS) prefetch(t, r + 1);
                else if (!last) prefetch(t, 0);
                else if (t + 1 < t_end) prefetch(t + 1, 0);
                if (!any_live) continue;   // wave uniform
                // query rows into registers: this lane's two chunks of each (the loads land during the scan)
                u32x4 qa[MS_G], qb[MS_G];
#pragma unroll
                for (int k = 0; k < MS_G; ++k) {
                    const gbytes_t qrow = (gbytes_t)reinterpret_cast<const char*>(P.q.rows) + (size_t)min(jq[k], q1 - 1) * (VISO_ROW * 2);
                    qa[k] = *(grow_t)(qrow + off_a);
                    qb[k] = *(grow_t)(qrow + off_b);
                }
                // ---------------- phase 1: scan the buckets the round's diamonds touch -> membership masks, union list.
                // entry = mask << 28 | ring slot << 8  (= LDS address of the row)
                int ucnt = 0;
                if (!(sa.debug & 4)) {
                    const float ymn = fminf(fminf(qk[0].y, qk[1].y), fminf(qk[2].y, qk[3].y));
                    const float ymx = fmaxf(fmaxf(qk[0].y, qk[1].y), fmaxf(qk[2].y, qk[3].y));
                    const float ys = (fabsf(ymn) + fabsf(ymx) + fabsf(radius)) * 1e-6f + 1e-6f;   // covers the rounding of dy in the test
                    const int sc0 = s_ys[ms_ybucket(ymn - radius - ys, ty0, yscale)] & ~(2 * VISO_WAVE - 1);
                    const int sc1 = s_ys[ms_ybucket(ymx + radius + ys, ty0, yscale) + 1];
                    for (int base = sc0; base < sc1; base += 2 * VISO_WAVE) {
                        const float2 ta = s_ykp[base + lane], tb = s_ykp[base + VISO_WAVE + lane];
                        const uint32_t pa = s_ypos[base + lane], pb = s_ypos[base + VISO_WAVE + lane];
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
                        if (ma) ul[min(ucnt + mbcnt(ua), MS_UCAP - 1)] = (ma << 28) | (pa << 8);
                        if (mb) ul[min(ucnt + ca + mbcnt(ub), MS_UCAP - 1)] = (mb << 28) | (pb << 8);
                        ucnt += ca + __popcll(ub);
                    }
                }
                const bool list_ovf = ucnt > MS_UCAP;
                const int nu = (list_ovf || (sa.debug & 1)) ? 0 : ucnt;
                __builtin_amdgcn_wave_barrier();
                if (nu > 0 && lane < MS_PAD) ul[nu + lane] = ul[nu - 1] & 0x0fffffffu;   // padding: scored, never counted
                __builtin_amdgcn_wave_barrier();
                // ---------------- phase 2: rolling pipeline over the union list, rows from the LDS ring
                MsTrack tr;
                tr.m1 = 0xffffffffu; tr.m2 = 0xffffffffu;
                {
                    const int npass = (nu + 7) >> 3;
                    u32x4 r0[MS_NP], r1[MS_NP];
                    uint32_t ent[MS_NP];
#define MS_ISSUE(SLOT, T)                                                                                  \
                    do {                                                                                   \
                        ent[SLOT] = ul[(T) * 8 + g8];                                                      \
                        const uint32_t ro_ = ent[SLOT] & 0x0fffff00u;                                      \
                        r0[SLOT] = *reinterpret_cast<const u32x4*>(s_rows + (ro_ | off_a));                \
                        r1[SLOT] = *reinterpret_cast<const u32x4*>(s_rows + (ro_ | off_b));                \
                    } while (0)
#define MS_SAD(K, SLOT)                                                                                    \
                    ({                                                                                     \
                        uint32_t s_ = __builtin_amdgcn_sad_u16(r0[SLOT].x, qa[K].x, 0u);                   \
                        s_ = __builtin_amdgcn_sad_u16(r0[SLOT].y, qa[K].y, s_);                            \
                        s_ = __builtin_amdgcn_sad_u16(r0[SLOT].z, qa[K].z, s_);                            \
                        s_ = __builtin_amdgcn_sad_u16(r0[SLOT].w, qa[K].w, s_);                            \
                        s_ = __builtin_amdgcn_sad_u16(r1[SLOT].x, qb[K].x, s_);                            \
                        s_ = __builtin_amdgcn_sad_u16(r1[SLOT].y, qb[K].y, s_);                            \
                        s_ = __builtin_amdgcn_sad_u16(r1[SLOT].z, qb[K].z, s_);                            \
                        s_ = __builtin_amdgcn_sad_u16(r1[SLOT].w, qb[K].w, s_);                            \
                        s_;                                                                                \
                    })
                    // four partial SADs per lane -> every lane of the 8-lane group holds the total of "its" query
                    // (transposing reduction: 6 selects + 4 DPP adds), then two instructions update the tracker
#define MS_REDUCE(SLOT, U)                                                                                 \
                    do {                                                                                   \
                        const uint32_t s0_ = MS_SAD(0, SLOT), s1_ = MS_SAD(1, SLOT), s2_ = MS_SAD(2, SLOT), s3_ = MS_SAD(3, SLOT); \
                        uint32_t a01_ = sel0 ? s1_ : s0_, a23_ = sel0 ? s3_ : s2_;                         \
                        const uint32_t b01_ = sel0 ? s0_ : s1_, b23_ = sel0 ? s2_ : s3_;                   \
                        a01_ += ms_dpp<0xB1>(b01_);   /* quad_perm 1,0,3,2 */                              \
                        a23_ += ms_dpp<0xB1>(b23_);                                                        \
                        uint32_t m_ = sel1 ? a23_ : a01_;                                                  \
                        const uint32_t o_ = sel1 ? a01_ : a23_;                                            \
                        m_ += ms_dpp<0x4E>(o_);       /* quad_perm 2,3,0,1 */                              \
                        m_ += ms_dpp<0x141>(m_);      /* row_half_mirror: lanes i and 7 - i track the same query */ \
                        const bool member_ = ((ent[SLOT] >> msh) & 1u) != 0;                               \
                        ms_update(tr, member_ ? ((m_ << 9) | (uint32_t)(U)) : 0xffffffffu);                \
                    } while (0)
                    if (npass > 0) {
#pragma unroll
                        for (int p = 0; p < MS_NP; ++p) MS_ISSUE(p, p);
                    }
                    int tp = 0;
                    for (; tp + MS_NP < npass; tp += MS_NP) {
#pragma unroll
                        for (int p = 0; p < MS_NP; ++p) {
                            MS_REDUCE(p, (tp + p) * 8 + g8);
                            __builtin_amdgcn_sched_barrier(0);   // keep the refill of the slot right behind its reduce
                            MS_ISSUE(p, tp + p + MS_NP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (npass > 0) {
#pragma unroll
                        for (int p = 0; p < MS_NP; ++p) MS_REDUCE(p, (tp + p) * 8 + g8);
                    }
#undef MS_REDUCE
#undef MS_SAD
#undef MS_ISSUE
                }
                // ---------------- phase 3: merge the 8 lane groups; lanes 0..3 end up with queries 0..3 of the round
#pragma unroll
                for (int m = 8; m < VISO_WAVE; m <<= 1) {
                    MsTrack o;
                    o.m1 = (uint32_t)__shfl_xor((int)tr.m1, m);
                    o.m2 = (uint32_t)__shfl_xor((int)tr.m2, m);
                    ms_merge(tr, o);
                }
                {
                    int my_orig = -1, my_j = 0, my_cnt = 0, my_pl = 0;
#pragma unroll
                    for (int k = 0; k < MS_G; ++k)
                        if (lane == k) { my_orig = orig[k]; my_j = jq[k]; my_cnt = cnt[k]; my_pl = pl[k]; }
                    if (lane < MS_G && my_orig >= 0) {
                        const bool cnone = tr.m1 == 0xffffffffu;
                        uint32_t d1 = cnone ? 0xffffffffu : tr.m1 >> 9;
                        uint32_t d2 = tr.m2 == 0xffffffffu ? 0xffffffffu : tr.m2 >> 9;
                        bool tie = !cnone && d2 == d1;
                        bool force = list_ovf;
                        // list position -> ring slot -> sorted position of the winner's row (the chunk starts at cl)
                        uint32_t wpos = 0;
                        if (!cnone) wpos = (uint32_t)cl + ((((ul[tr.m1 & 511u] >> 8) & (MS_RING - 1)) - (uint32_t)cl) & (MS_RING - 1));
                        if (!first || !last) {   // several chunks: fold into the state carried in LDS
                            const uint4 sv = s_state[my_pl];
                            const uint32_t pd1 = sv.x, pd2 = sv.y, pw = sv.z;
                            const bool ptie = (sv.w >> 29) & 1u, pforce = (sv.w >> 30) & 1u;
                            uint32_t nd1, nd2, nw; bool ntie;
                            if (d1 < pd1) { nd1 = d1; nd2 = min(d2, pd1); nw = wpos; ntie = tie; }
                            else if (d1 == pd1) { nd1 = pd1; nd2 = pd1; nw = pw; ntie = pd1 != 0xffffffffu; }
                            else { nd1 = pd1; nd2 = min(pd2, d1); nw = pw; ntie = ptie; }
                            d1 = nd1; d2 = nd2; wpos = nw; tie = ntie;
                            force = force || pforce;
                            my_cnt += (int)(sv.w & 0x1fffffffu);
                            if (!last) s_state[my_pl] = make_uint4(d1, d2, wpos, (uint32_t)my_cnt | (tie ? 1u << 29 : 0u) | (force ? 1u << 30 : 0u));
                        }
                        if (last) {
                            const bool none = d1 == 0xffffffffu;
                            if (my_cnt > K || force || tie) {
                                // more than K candidates / union too long / exact tie of the minimum (largest-key rule): overflow kernel
                                P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, my_j);
                            } else {
                                bool accept = !none;
                                int idx = -1;
                                if (accept) {
                                    idx = P.t.sidx[wpos];
                                    if (mp.second) {   // src/viso.cpp:713-716 — ratio test in double (Q3)
                                        const double bd2 = d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)d2;
                                        accept = (double)d1 < bd2 * mp.ratio;
                                    }
                                }
                                P.res[my_orig] = make_int2(accept ? idx : -1, none ? -1 : (int)d1);
                                scored += (unsigned long long)my_cnt;
                            }
                        }
                    }
                }
            }
        }
    }
    // scored pairs of the segment's queries whose result stands (lanes 0..3 of every wave hold partial sums)
#pragma unroll
    for (int m = 1; m < MS_G; m <<= 1) scored += (unsigned long long)__shfl_xor((long long)scored, m);
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
}

int launch_match_strip_temporal(hipStream_t s, const BatchMatchArgs& a64, int cap_max) {
    // same problem enumeration as the 64-query kernels; one workgroup per (problem, segment of consecutive tiles).
    // Segments are as long as possible (the ring is re-used along a segment) while the launch still has about two
    // workgroups per CU.
    StripArgs sa;
    sa.b = a64;
    const int tiles = (cap_max + MS_QPB - 1) / MS_QPB;
    const int groups = (a64.n_probs + 7) / 8;
    const long long nprob = a64.gs == 3 ? (long long)((groups + 2) / 3) * a64.gc * 8 : (long long)groups * 8;
    int nseg = (int)((512 + nprob - 1) / (nprob > 0 ? nprob : 1));
    if (nseg < 1) nseg = 1;
    if (nseg > tiles) nseg = tiles;
    int tps = (tiles + nseg - 1) / nseg;
    if (tps > MS_TMAX) tps = MS_TMAX;
    nseg = (tiles + tps - 1) / tps;
    sa.b.bpp = nseg;
    sa.tiles_per_seg = tps;
    {   // timing experiments only (results are wrong with any bit set): 1 = no scoring, 2 = no row staging, 4 = no scan
        const char* e = getenv("VISO_STRIP_DEBUG");
        sa.debug = e ? atoi(e) : 0;
    }
    const long long blocks = nprob * nseg;
    if (blocks > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    static unsigned long long attr_set = 0;   // bit d: done for device d (the attribute is per function AND device)
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_set >> dev) & 1ull)) {
        HIP_TRY(hipFuncSetAttribute((const void*)match_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MS_LDS_BYTES));
        if (dev < 64) attr_set |= 1ull << dev;
    }
    hipLaunchKernelGGL(match_strip_kernel, dim3((unsigned)blocks), dim3(MS_THREADS), MS_LDS_BYTES, s, sa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_strip_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
