// circle.hip — circular-match join, match gathering and rectified
// triangulation (reference src/viso.cpp:207-243 match_circle, :501-514
// collect_matches, :1137-1162 triangulate_rectified, :1292-1305 gather).
#include "common.h"

#include <string.h>

#define CIRC_THREADS 256

// ------------------------------------------------------------------ general
// Literal semantics of the reference's four nested loops for ARBITRARY lists
// (duplicate keys included): thread i owns row i of match_lr, enumerates its
// (j,k,l) hits in loop order; a workgroup scan keeps the output in i order.

template <bool WRITE>
__device__ __forceinline__ int circle_row(const CircleArgs& a, int i, int off) {
    int n = 0;
    const int ileft = a.lr[3 * i], iright = a.lr[3 * i + 1];
    for (int j = 0; j < a.n11; ++j) {
        if (a.m11[3 * j] != ileft) continue;
        const int ileft_prev = a.m11[3 * j + 1];
        for (int k = 0; k < a.n_lrp; ++k) {
            if (a.lrp[3 * k] != ileft_prev) continue;
            const int iright_prev = a.lrp[3 * k + 1];
            for (int l = 0; l < a.n22; ++l) {
                if (a.m22[3 * l + 1] == iright_prev && a.m22[3 * l] == iright) {
                    if (WRITE) {
                        const int o = off + n;
                        if (o < a.cap) {
                            a.rows[6 * o + 0] = ileft; a.rows[6 * o + 1] = iright;
                            a.rows[6 * o + 2] = ileft_prev; a.rows[6 * o + 3] = iright_prev;
                            a.rows[6 * o + 4] = i; a.rows[6 * o + 5] = k;
                        }
                    }
                    ++n;
                }
            }
        }
    }
    return n;
}

// ------------------------------------------------------------------ plain family: tables when the keys allow it
// viso_match_circle for the lists match_desc produces: at most one row per query index in match11 (key [0]), match_lr_prev
// (key [0]) and match22 (key [0]), so every hop of the reference's nested loops (:215-240) hits at most one row and the
// loops collapse to three table lookups per row of match_lr, in row order.  ONE workgroup: build the three tables
// (key -> row, -1 = absent) in global scratch with compare-and-swap -- a second row with the same key, or a key outside
// [0, tabn), sets `dup` and the SAME kernel falls back to the literal nested loops above (arbitrary lists, duplicate
// keys included: the reference's semantics for any caller).  The general path on 1 500-row lists costs 200 ms (three nested
// linear scans per row on one workgroup); the tables 10 us.  Result block: the row count, then 6 ints per row (circ_match | match_pcl).
#define CIRCT_THREADS 1024
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* total, int* scratch) {   // scratch: 16 ints of LDS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int incl = (int)viso_wave_scan((uint32_t)v);
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CIRCT_THREADS / 64; ++w) { const int t = scratch[w]; if (w < wave) base += t; tot += t; }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

// LDS = true (the tables fit the CU's LDS -- every list of fewer than CIRCT_LDS_TABN keys): six tables there, key -> row AND
// key -> the row's value, so that a row of match_lr is joined by three LDS lookups behind one global load, and the thread
// that finds a joined row gathers its columns on the spot (no second pass over the rows).  The kernel is a chain of
// dependent memory operations on one workgroup: with the tables in global scratch it was seven L2 round trips long
// (18 us per frame of the per-call loop; ~10 with the tables in LDS).  LDS = false: three key -> row tables in `tab`.
#define CIRCT_LDS_TABN 6144   // 6 tables x 4 B x 6144 = 144 KB of the CU's 160
template <bool LDS>
__global__ __launch_bounds__(CIRCT_THREADS) void circle_table_kernel(CircleArgs a, int* tab, int tabn) {
    extern __shared__ int s_tab[];
    __shared__ int scratch[16];
    __shared__ int s_dup;
    if (blockIdx.x > 0) { plain_out_blocks(*a.ride, blockIdx.x - 1); return; }   // riders: the copy-out of the kernel before (common.h, OutArgs)
    if (a.n_lr_p) a.n_lr = *a.n_lr_p;      // lists that an earlier kernel of the chain produced: their lengths are on the device
    if (a.n_lrp_p) a.n_lrp = *a.n_lrp_p;
    if (a.n11_p) a.n11 = *a.n11_p;
    if (a.n22_p) a.n22 = *a.n22_p;
    int* const T = LDS ? s_tab : tab;
    int* t11 = T; int* tlrp = T + tabn; int* t22 = T + 2 * tabn;
    int* v11 = T + 3 * tabn; int* vlrp = T + 4 * tabn; int* v22 = T + 5 * tabn;   // LDS only: the rows' values ([1] of the row)
    if (threadIdx.x == 0) s_dup = 0;
    for (int i = threadIdx.x; i < 3 * tabn; i += CIRCT_THREADS) T[i] = -1;
    __syncthreads();
    int dup = 0;
    for (int j = threadIdx.x; j < a.n11; j += CIRCT_THREADS) {
        const int key = a.m11[3 * j], val = LDS ? a.m11[3 * j + 1] : 0;
        if (key < 0 || key >= tabn || atomicCAS(&t11[key], -1, j) != -1) dup = 1;
        else if (LDS) v11[key] = val;
    }
    for (int k = threadIdx.x; k < a.n_lrp; k += CIRCT_THREADS) {
        const int key = a.lrp[3 * k], val = LDS ? a.lrp[3 * k + 1] : 0;
        if (key < 0 || key >= tabn || atomicCAS(&tlrp[key], -1, k) != -1) dup = 1;
        else if (LDS) vlrp[key] = val;
    }
    for (int l = threadIdx.x; l < a.n22; l += CIRCT_THREADS) {
        const int key = a.m22[3 * l], val = LDS ? a.m22[3 * l + 1] : 0;
        if (key < 0 || key >= tabn || atomicCAS(&t22[key], -1, l) != -1) dup = 1;
        else if (LDS) v22[key] = val;
    }
    if (dup) s_dup = 1;
    __threadfence_block();
    __syncthreads();
    int running = 0;
    const bool gather_here = LDS && a.g_x != nullptr && !s_dup;   // uniform
    if (s_dup) {   // arbitrary lists: the literal loops
        for (int base = 0; base < a.n_lr; base += CIRCT_THREADS) {
            const int i = base + threadIdx.x;
            const int cnt = (i < a.n_lr) ? circle_row<false>(a, i, 0) : 0;
            int total;
            const int off = running + block_exclusive_scan_1024(cnt, &total, scratch);
            if (cnt) circle_row<true>(a, i, off);
            running += total;
        }
    } else {
        for (int base = 0; base < a.n_lr; base += CIRCT_THREADS) {
            const int i = base + threadIdx.x;
            int ok = 0, ileft = 0, iright = 0, ileft_prev = 0, iright_prev = 0, k = 0;
            if (i < a.n_lr) {
                ileft = a.lr[3 * i]; iright = a.lr[3 * i + 1];
                const int j = (ileft >= 0 && ileft < tabn) ? t11[ileft] : -1;
                if (j >= 0) {
                    ileft_prev = LDS ? v11[ileft] : a.m11[3 * j + 1];
                    k = (ileft_prev >= 0 && ileft_prev < tabn) ? tlrp[ileft_prev] : -1;
                    if (k >= 0) {
                        iright_prev = LDS ? vlrp[ileft_prev] : a.lrp[3 * k + 1];
                        const int l = (iright >= 0 && iright < tabn) ? t22[iright] : -1;
                        ok = l >= 0 && (LDS ? v22[iright] : a.m22[3 * l + 1]) == iright_prev;
                    }
                }
            }
            double gv[7];
            if (gather_here && ok) {   // the joined row's columns, asked for before the scan's barriers
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[r] = a.g_x[(size_t)r * a.g_ldx + i];
#pragma unroll
                for (int r = 0; r < 3; ++r) gv[4 + r] = a.g_Xp[(size_t)r * a.g_ldXp + k];
            }
            int total;
            const int o = running + block_exclusive_scan_1024(ok, &total, scratch);
            if (ok && o < a.cap) {
                a.rows[6 * o + 0] = ileft; a.rows[6 * o + 1] = iright; a.rows[6 * o + 2] = ileft_prev; a.rows[6 * o + 3] = iright_prev;
                a.rows[6 * o + 4] = i; a.rows[6 * o + 5] = k;
                if (gather_here) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a.g_xc[(size_t)r * a.g_ldc + o] = gv[r];
#pragma unroll
                    for (int r = 0; r < 3; ++r) a.g_Xpc[(size_t)r * a.g_ldc + o] = gv[4 + r];
                }
            }
            running += total;
        }
    }
    if (threadIdx.x == 0) *a.out_n = running;
    if (a.g_x && !gather_here) {   // uniform: the gather of the joined rows' columns (the rows were written by this workgroup)
        __threadfence_block();
        __syncthreads();
        const int n = running < a.cap ? running : a.cap;
        for (int i = threadIdx.x; i < n; i += CIRCT_THREADS) {
            const int ci = a.rows[6 * i + 4], k = a.rows[6 * i + 5];
#pragma unroll
            for (int r = 0; r < 4; ++r) a.g_xc[(size_t)r * a.g_ldc + i] = a.g_x[(size_t)r * a.g_ldx + ci];
#pragma unroll
            for (int r = 0; r < 3; ++r) a.g_Xpc[(size_t)r * a.g_ldc + i] = a.g_Xp[(size_t)r * a.g_ldXp + k];
        }
    }
}

int launch_circle_table(hipStream_t s, const CircleArgs& a, int* tab, int tabn) {
    const dim3 grid(1 + (a.ride ? a.ride_blocks : 0));
    if (tabn > 0 && tabn <= CIRCT_LDS_TABN) {
        const size_t lds = sizeof(int) * 6 * (size_t)tabn;
        if (lds > 40 * 1024)   // (per device, and cheap: asked for whenever the launch needs it, like launch_sort does)
            HIP_TRY(hipFuncSetAttribute((const void*)circle_table_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * 6 * CIRCT_LDS_TABN)));
        hipLaunchKernelGGL(circle_table_kernel<true>, grid, dim3(CIRCT_THREADS), lds, s, a, tab, tabn);
    } else {
        hipLaunchKernelGGL(circle_table_kernel<false>, grid, dim3(CIRCT_THREADS), 0, s, a, tab, tabn);
    }
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// ------------------------------------------------------------------ table join
// Batch pipeline form.  match_desc emits at most one match per query index, so
// each of the three hops hits at most one row and becomes a table lookup:
//   res11[ileft] -> ileft_prev ; pos_lr_prev[ileft_prev] -> k, iright_prev ;
//   res22[iright] == iright_prev.
// One workgroup per frame.  x_c / Xp_c (src/viso.cpp:1292-1305) are computed here from the keypoints: x_c column =
// collect_matches of the joined stereo match of frame t (:501-514), Xp_c column = triangulate_rectified<double> of the
// joined stereo match of frame t-1 (:1137-1162, no clamp) — the same expressions, evaluated only for the rows the
// solver will read, so the batch path needs no collect / triangulate launch of its own.
#define CIRC_NCH 8   // chunks of CIRC_THREADS stereo matches whose lookups are in flight together
__global__ __launch_bounds__(CIRC_THREADS) void circle_join_kernel(const JoinItem* items, int n_items, SolverParamsDev sp) {
    __shared__ int s_tot[CIRC_NCH][CIRC_THREADS / 64];
    if ((int)blockIdx.x >= n_items) return;
    const JoinItem J = items[blockIdx.x];
    const int M = *J.lr_cnt;
    int running = 0;
    // The join of a match is a chain of four dependent table lookups, and the output order needs a scan over the chunk:
    // chunk by chunk (a chain, a scan, the stores, the next chain ...) the kernel was seven chains long for the bench's
    // ~1 700 matches per frame (59 us, 4 us of it arithmetic).  The lookups of CIRC_NCH chunks are asked for together —
    // one chain's latency for all of them — then the chunks are scanned and stored in order.
    for (int base = 0; base < M; base += CIRC_NCH * CIRC_THREADS) {
        int ok[CIRC_NCH], ileft[CIRC_NCH], iright[CIRC_NCH], ileft_prev[CIRC_NCH], iright_prev[CIRC_NCH], k[CIRC_NCH];
#pragma unroll
        for (int c = 0; c < CIRC_NCH; ++c) {
            const int r = base + c * CIRC_THREADS + (int)threadIdx.x;
            ok[c] = 0; ileft[c] = 0; iright[c] = 0; ileft_prev[c] = 0; iright_prev[c] = 0; k[c] = 0;
            if (r < M) {
                ileft[c] = J.lr[3 * r]; iright[c] = J.lr[3 * r + 1];
                const int2 a = J.res11[ileft[c]];
                if (a.x >= 0) {
                    ileft_prev[c] = a.x;
                    k[c] = J.pos_lrp[ileft_prev[c]];
                    if (k[c] >= 0) {
                        iright_prev[c] = J.res_lrp[ileft_prev[c]].x;
                        ok[c] = J.res22[iright[c]].x == iright_prev[c];
                    }
                }
            }
        }
        // output rows: the matches in order (chunk after chunk, thread after thread): ONE pass over the workgroup for all
        // CIRC_NCH chunks — every wave scans its 64 flags per chunk and leaves its chunk totals in LDS
        int o[CIRC_NCH];
        {
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            int excl[CIRC_NCH];
#pragma unroll
            for (int c = 0; c < CIRC_NCH; ++c) {
                const unsigned long long m = __ballot(ok[c] != 0);
                excl[c] = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (lane == 0) s_tot[c][wave] = __popcll(m);
            }
            __syncthreads();
            int before = running;
#pragma unroll
            for (int c = 0; c < CIRC_NCH; ++c) {
                int inwg = 0;
#pragma unroll
                for (int w = 0; w < CIRC_THREADS / 64; ++w) {
                    const int t = s_tot[c][w];
                    if (w < wave) inwg += t;
                    running += t;
                }
                o[c] = before + inwg + excl[c];
                before = running;
            }
            __syncthreads();   // s_tot is rewritten by the next super-chunk
        }
        // the joined rows' keypoints of all chunks asked for together, then triangulated and stored
#pragma unroll
        for (int c = 0; c < CIRC_NCH; ++c) {
            if (ok[c]) {
                const int r = base + c * CIRC_THREADS + (int)threadIdx.x;
                const int oc = o[c];
                J.circ[4 * oc + 0] = ileft[c]; J.circ[4 * oc + 1] = iright[c];
                J.circ[4 * oc + 2] = ileft_prev[c]; J.circ[4 * oc + 3] = iright_prev[c];
                J.pcl[2 * oc + 0] = r; J.pcl[2 * oc + 1] = k[c];
                const float2 a1 = J.kp1[ileft[c]], a2 = J.kp2[iright[c]];
                J.x_c[0 * J.ldc + oc] = (double)a1.x; J.x_c[1 * J.ldc + oc] = (double)a1.y;
                J.x_c[2 * J.ldc + oc] = (double)a2.x; J.x_c[3 * J.ldc + oc] = (double)a2.y;
                const float2 p1 = J.kp1p[ileft_prev[c]], p2 = J.kp2p[iright_prev[c]];
                const double uL = p1.x, vL = p1.y, uR = p2.x;
                const double d = uL - uR;                       // src/viso.cpp:1148-1151, no clamp
                J.Xp_c[0 * J.ldc + oc] = sp.base * (uL - sp.cu) / d;
                J.Xp_c[1 * J.ldc + oc] = sp.base * (vL - sp.cv) / d;
                J.Xp_c[2 * J.ldc + oc] = sp.f * sp.base / d;
            }
        }
    }
    if (threadIdx.x == 0) *J.mc = running;
}

int launch_circle_join(hipStream_t s, const JoinItem* items_dev, int n_items, const SolverParamsDev& sp) {
    if (n_items <= 0) return VISO_OK;
    hipLaunchKernelGGL(circle_join_kernel, dim3(n_items), dim3(CIRC_THREADS), 0, s, items_dev, n_items, sp);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// ------------------------------------------------- collect + triangulate
__global__ __launch_bounds__(256) void collect_triangulate_kernel(const TriItem* items, int n_items,
                                                                 SolverParamsDev sp, int cap) {
    const int item = blockIdx.y;
    if (item >= n_items) return;
    const TriItem T = items[item];
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= cap || r >= *T.m_cnt) return;
    const int i1 = T.match[3 * r], i2 = T.match[3 * r + 1];
    const float2 a = T.kp1[i1], b = T.kp2[i2];
    const double uL = a.x, vL = a.y, uR = b.x, vR = b.y;
    T.x[0 * T.ld + r] = uL; T.x[1 * T.ld + r] = vL; T.x[2 * T.ld + r] = uR; T.x[3 * T.ld + r] = vR;
    if (T.X) {
        const double d = uL - uR;                       // src/viso.cpp:1148-1151, no clamp
        T.X[0 * T.ld + r] = sp.base * (uL - sp.cu) / d;
        T.X[1 * T.ld + r] = sp.base * (vL - sp.cv) / d;
        T.X[2 * T.ld + r] = sp.f * sp.base / d;
    }
}

int launch_collect_triangulate(hipStream_t s, const TriItem* items_dev, int n_items,
                               const SolverParamsDev& sp, int cap) {
    if (n_items <= 0 || cap <= 0) return VISO_OK;
    hipLaunchKernelGGL(collect_triangulate_kernel, dim3((cap + 255) / 256, n_items), dim3(256), 0, s,
                       items_dev, n_items, sp, cap);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// The plain family's own collect_matches / triangulate_rectified calls (a caller outside the loop's pattern): ONE kernel each,
// which reads its inputs from the call's pinned block over PCIe, writes the result into pinned memory and signals -- the copy
// kernels that used to run in front of and behind it are gone (three launches and two waits were 29 / 22 us per call).
__global__ __launch_bounds__(256) void collect_direct_kernel(const float2* kp1, const float2* kp2, const int* match, int n, double* x, PlainSignal sig) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) {
        const int i1 = match[3 * r], i2 = match[3 * r + 1];
        const float2 a = kp1[i1], b = kp2[i2];
        x[0 * (size_t)n + r] = (double)a.x; x[1 * (size_t)n + r] = (double)a.y; x[2 * (size_t)n + r] = (double)b.x; x[3 * (size_t)n + r] = (double)b.y;
    }
    plain_signal_done(sig, gridDim.x);
}
__global__ __launch_bounds__(256) void triangulate_direct_kernel(const double* x, int m, SolverParamsDev sp, double* X, PlainSignal sig) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        const double uL = x[0 * (size_t)m + i], vL = x[1 * (size_t)m + i], uR = x[2 * (size_t)m + i];
        const double d = uL - uR;                       // src/viso.cpp:1148-1151, no clamp
        X[0 * (size_t)m + i] = sp.base * (uL - sp.cu) / d;
        X[1 * (size_t)m + i] = sp.base * (vL - sp.cv) / d;
        X[2 * (size_t)m + i] = sp.f * sp.base / d;
    }
    plain_signal_done(sig, gridDim.x);
}

// ------------------------------------------------------------ plain family
// Every call: inputs packed into the context's pinned block and sent with ONE copy (PlainStage), results fetched with
// ONE copy behind ONE synchronize.
#define CIRC_TAB_MAX (1 << 20)   // keys below this take the tables (3 ints each); larger or negative ones the literal loops

extern "C" int viso_match_circle(const int32_t* lr, int n_lr, const int32_t* lr_prev, int n_lrp,
                                 const int32_t* m11, int n11, const int32_t* m22, int n22,
                                 int32_t* circ, int32_t* pcl, int cap, int* out_n) {
    if (n_lr < 0 || n_lrp < 0 || n11 < 0 || n22 < 0 || cap < 0 || !out_n ||
        (n_lr && !lr) || (n_lrp && !lr_prev) || (n11 && !m11) || (n22 && !m22) || (cap && (!circ || !pcl))) {
        viso_set_error("viso_match_circle: bad argument");
        return VISO_ERR_ARG;
    }
    *out_n = 0;
    if (n_lr == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    HIP_TRY(hipSetDevice(c->device));
    PlainProf pp(VISO_PLAIN_MATCH_CIRCLE, c->stream);
    {   // the frame's stereo call may have joined exactly these lists already (plain.hip)
        int ret = VISO_OK;
        if (plain_try_circle(c, lr, n_lr, lr_prev, n_lrp, m11, n11, m22, n22, circ, pcl, cap, out_n, &ret)) return ret;
    }
    // size of the tables: the largest key of the three keyed lists (a look at ~4 500 ints; argument inspection, no arithmetic of the path)
    long long kmax = -1;
    bool tables = true;
    for (int j = 0; j < n11 && tables; ++j) { const int k = m11[3 * j]; if (k < 0 || k >= CIRC_TAB_MAX) tables = false; else if (k > kmax) kmax = k; }
    for (int j = 0; j < n_lrp && tables; ++j) { const int k = lr_prev[3 * j]; if (k < 0 || k >= CIRC_TAB_MAX) tables = false; else if (k > kmax) kmax = k; }
    for (int j = 0; j < n22 && tables; ++j) { const int k = m22[3 * j]; if (k < 0 || k >= CIRC_TAB_MAX) tables = false; else if (k > kmax) kmax = k; }
    const int tabn = tables ? (int)(kmax + 1) : 0;   // 0: every key is "out of range" -> the kernel takes the literal loops
    int r;
    PlainStage in;
    const size_t in_bytes = PlainStage::need(sizeof(int) * 3 * (size_t)n_lr) + PlainStage::need(sizeof(int) * 3 * (size_t)n_lrp) +
                            PlainStage::need(sizeof(int) * 3 * (size_t)n11) + PlainStage::need(sizeof(int) * 3 * (size_t)n22);
    if ((r = in.begin(c, in_bytes)) < 0) return r;
    CircleArgs a{};
    a.lr = in.put(lr, 3 * (size_t)n_lr); a.lrp = in.put(lr_prev, 3 * (size_t)n_lrp);
    a.m11 = in.put(m11, 3 * (size_t)n11); a.m22 = in.put(m22, 3 * (size_t)n22);
    a.n_lr = n_lr; a.n_lrp = n_lrp; a.n11 = n11; a.n22 = n22; a.cap = cap;
    char *dout, *hout; int* dtab;
    const size_t out_bytes = 256 + sizeof(int) * 6 * (size_t)cap;
    if ((r = ctx_scratch(c, PLAIN_SLOT_OUT, out_bytes, (void**)&dout)) < 0) return r;
    if ((r = ctx_pinned(c, 1, out_bytes, &hout)) < 0) return r;
    if ((r = ctx_scratch(c, 16, sizeof(int) * 3 * (size_t)(tabn + 1), (void**)&dtab)) < 0) return r;
    a.out_n = reinterpret_cast<int*>(dout); a.rows = reinterpret_cast<int*>(dout + 256);
    if ((r = in.flush(c->stream)) < 0) return r;
    pp.mark(1);
    if ((r = launch_circle_table(c->stream, a, dtab, tabn)) < 0) return r;
    pp.mark(2);
    // the rows that exist (the count is on the device), by a copy kernel into pinned memory
    PlainSignal sig_;
    if ((r = plain_signal_next(c, &sig_)) < 0) return r;
    if ((r = plain_blit(c->stream, dout, hout, 64, a.out_n, 6, cap, &sig_)) < 0) return r;
    pp.wait_begin();
    if ((r = plain_signal_wait(c, c->stream, sig_.seq)) < 0) return r;
    const int cnt = *reinterpret_cast<const int*>(hout);
    const int w = cnt < cap ? cnt : cap;
    pp.wait_end();
    *out_n = cnt;
    plain_note_circle(c, cnt);
    const int* rows = reinterpret_cast<const int*>(hout + 256);
    for (int i = 0; i < w; ++i) {
        circ[4 * i + 0] = rows[6 * i + 0]; circ[4 * i + 1] = rows[6 * i + 1]; circ[4 * i + 2] = rows[6 * i + 2]; circ[4 * i + 3] = rows[6 * i + 3];
        pcl[2 * i + 0] = rows[6 * i + 4]; pcl[2 * i + 1] = rows[6 * i + 5];
    }
    pp.mark(3);
    if (cnt > cap) { viso_set_error("viso_match_circle: %d rows needed, cap %d", cnt, cap); return VISO_ERR_ARG; }
    return VISO_OK;
}

extern "C" int viso_collect_matches(const float* kp1, int n1, const float* kp2, int n2,
                                    const int32_t* match, int n, double* x) {
    if (n1 < 0 || n2 < 0 || n < 0 || (n && (!match || !x || !kp1 || !kp2))) {
        viso_set_error("viso_collect_matches: bad argument");
        return VISO_ERR_ARG;
    }
    for (int i = 0; i < n; ++i)
        if (match[3 * i] < 0 || match[3 * i] >= n1 || match[3 * i + 1] < 0 || match[3 * i + 1] >= n2) {
            viso_set_error("viso_collect_matches: match index out of range (std::vector::at would throw)");
            return VISO_ERR_ARG;
        }
    if (n == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    HIP_TRY(hipSetDevice(c->device));
    PlainProf pp(VISO_PLAIN_COLLECT_MATCHES, c->stream);
    if (plain_try_collect(c, kp1, n1, kp2, n2, match, n, x)) return VISO_OK;   // the frame's stereo call computed it (plain.hip)
    int r;
    PlainStage in;
    const size_t in_bytes = PlainStage::need(sizeof(float2) * (size_t)n1) + PlainStage::need(sizeof(float2) * (size_t)n2) +
                            PlainStage::need(sizeof(int) * 3 * (size_t)n);
    if ((r = in.begin(c, in_bytes)) < 0) return r;
    char* hout;
    const size_t out_bytes = sizeof(double) * 4 * (size_t)n;
    if ((r = ctx_pinned(c, 1, out_bytes, &hout)) < 0) return r;
    const float2* hkp1 = in.host_of(in.put(reinterpret_cast<const float2*>(kp1), (size_t)n1));   // the kernel reads the pinned block itself
    const float2* hkp2 = in.host_of(in.put(reinterpret_cast<const float2*>(kp2), (size_t)n2));
    const int* hmatch = in.host_of(in.put(match, 3 * (size_t)n));
    pp.mark(1);
    PlainSignal sig_;
    if ((r = plain_signal_next(c, &sig_)) < 0) return r;
    hipLaunchKernelGGL(collect_direct_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, hkp1, hkp2, hmatch, n, reinterpret_cast<double*>(hout), sig_);
    HIP_TRY(hipGetLastError());
    pp.mark(2);
    pp.wait_begin();
    if ((r = plain_signal_wait(c, c->stream, sig_.seq)) < 0) return r;
    pp.wait_end();
    memcpy(x, hout, out_bytes);
    pp.mark(3);
    return VISO_OK;
}

extern "C" int viso_triangulate_rectified(const double* x, int m, const viso_param* p, double* X) {
    if (m < 0 || !p || (m && (!x || !X))) { viso_set_error("viso_triangulate_rectified: bad argument"); return VISO_ERR_ARG; }
    if (m == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    HIP_TRY(hipSetDevice(c->device));
    PlainProf pp(VISO_PLAIN_TRIANGULATE, c->stream);
    if (plain_try_triangulate(c, x, m, p, X)) return VISO_OK;   // the frame's stereo call computed it (plain.hip)
    int r;
    PlainStage in;
    if ((r = in.begin(c, PlainStage::need(sizeof(double) * 4 * (size_t)m))) < 0) return r;
    char* hout;
    const size_t out_bytes = sizeof(double) * 3 * (size_t)m;
    if ((r = ctx_pinned(c, 1, out_bytes, &hout)) < 0) return r;
    const double* hx = in.host_of(in.put(x, 4 * (size_t)m));   // the kernel reads the pinned block itself
    pp.mark(1);
    SolverParamsDev sp;
    fill_solver_params(&sp, p);
    PlainSignal sig_;
    if ((r = plain_signal_next(c, &sig_)) < 0) return r;
    hipLaunchKernelGGL(triangulate_direct_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, hx, m, sp, reinterpret_cast<double*>(hout), sig_);
    HIP_TRY(hipGetLastError());
    pp.mark(2);
    pp.wait_begin();
    if ((r = plain_signal_wait(c, c->stream, sig_.seq)) < 0) return r;
    pp.wait_end();
    memcpy(X, hout, out_bytes);
    pp.mark(3);
    return VISO_OK;
}
