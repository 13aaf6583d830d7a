// circle.hip — circular-match join, match gathering and rectified
// triangulation (reference src/viso.cpp:207-243 match_circle, :501-514
// collect_matches, :1137-1162 triangulate_rectified, :1292-1305 gather).
#include "common.h"

#define CIRC_THREADS 256

// ------------------------------------------------------------------ general
// Literal semantics of the reference's four nested loops for ARBITRARY lists
// (duplicate keys included): thread i owns row i of match_lr, enumerates its
// (j,k,l) hits in loop order; a workgroup scan keeps the output in i order.
struct CircleArgs {
    const int* lr; const int* lrp; const int* m11; const int* m22;
    int n_lr, n_lrp, n11, n22;
    int* circ; int* pcl; int cap; int* out_n;
};

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
                            a.circ[4 * o + 0] = ileft; a.circ[4 * o + 1] = iright;
                            a.circ[4 * o + 2] = ileft_prev; a.circ[4 * o + 3] = iright_prev;
                            a.pcl[2 * o + 0] = i; a.pcl[2 * o + 1] = k;
                        }
                    }
                    ++n;
                }
            }
        }
    }
    return n;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* scratch) {
    // 256 threads = 4 waves; scratch: 8 ints of LDS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += scratch[w];
    *total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(CIRC_THREADS) void circle_general_kernel(CircleArgs a) {
    __shared__ int scratch[8];
    int running = 0;
    for (int base = 0; base < a.n_lr; base += CIRC_THREADS) {
        const int i = base + threadIdx.x;
        const int cnt = (i < a.n_lr) ? circle_row<false>(a, i, 0) : 0;
        int total;
        const int off = running + block_exclusive_scan(cnt, &total, scratch);
        if (cnt) circle_row<true>(a, i, off);
        running += total;
    }
    if (threadIdx.x == 0) *a.out_n = running;
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

__global__ __launch_bounds__(256) void triangulate_kernel(const double* x, int m, SolverParamsDev sp, double* X) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const double d = x[0 * m + i] - x[2 * m + i];
    X[0 * m + i] = sp.base * (x[0 * m + i] - sp.cu) / d;
    X[1 * m + i] = sp.base * (x[1 * m + i] - sp.cv) / d;
    X[2 * m + i] = sp.f * sp.base / d;
}

// ------------------------------------------------------------ plain family
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
    int *d[4], *dcirc, *dpcl, *dn;
    const int32_t* h[4] = {lr, lr_prev, m11, m22};
    const int n[4] = {n_lr, n_lrp, n11, n22};
    int r;
    for (int k = 0; k < 4; ++k) {
        if ((r = ctx_scratch(c, k, sizeof(int) * 3 * (size_t)(n[k] + 1), (void**)&d[k])) < 0) return r;
        if (n[k]) HIP_TRY(hipMemcpyAsync(d[k], h[k], sizeof(int) * 3 * (size_t)n[k], hipMemcpyHostToDevice, c->stream));
    }
    if ((r = ctx_scratch(c, 4, sizeof(int) * 4 * (size_t)(cap + 1), (void**)&dcirc)) < 0) return r;
    if ((r = ctx_scratch(c, 5, sizeof(int) * 2 * (size_t)(cap + 1), (void**)&dpcl)) < 0) return r;
    if ((r = ctx_scratch(c, 6, sizeof(int) * 4, (void**)&dn)) < 0) return r;
    CircleArgs a{d[0], d[1], d[2], d[3], n_lr, n_lrp, n11, n22, dcirc, dpcl, cap, dn};
    hipLaunchKernelGGL(circle_general_kernel, dim3(1), dim3(CIRC_THREADS), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    int cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, dn, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out_n = cnt;
    const int w = cnt < cap ? cnt : cap;
    if (w > 0) {
        HIP_TRY(hipMemcpy(circ, dcirc, sizeof(int) * 4 * (size_t)w, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(pcl, dpcl, sizeof(int) * 2 * (size_t)w, hipMemcpyDeviceToHost));
    }
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
    float2 *dk1, *dk2; int *dm, *dcnt; double* dx; TriItem* dit;
    int r;
    if ((r = ctx_scratch(c, 0, sizeof(float2) * (size_t)n1, (void**)&dk1)) < 0) return r;
    if ((r = ctx_scratch(c, 1, sizeof(float2) * (size_t)n2, (void**)&dk2)) < 0) return r;
    if ((r = ctx_scratch(c, 2, sizeof(int) * 3 * (size_t)n, (void**)&dm)) < 0) return r;
    if ((r = ctx_scratch(c, 3, sizeof(double) * 4 * (size_t)n, (void**)&dx)) < 0) return r;
    if ((r = ctx_scratch(c, 4, sizeof(int) * 4, (void**)&dcnt)) < 0) return r;
    if ((r = ctx_scratch(c, 5, sizeof(TriItem), (void**)&dit)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(dk1, kp1, sizeof(float2) * (size_t)n1, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dk2, kp2, sizeof(float2) * (size_t)n2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dm, match, sizeof(int) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dcnt, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    TriItem it{dk1, dk2, dm, dcnt, dx, nullptr, n};
    HIP_TRY(hipMemcpyAsync(dit, &it, sizeof(it), hipMemcpyHostToDevice, c->stream));
    SolverParamsDev sp{};
    if ((r = launch_collect_triangulate(c->stream, dit, 1, sp, n)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(x, dx, sizeof(double) * 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VISO_OK;
}

extern "C" int viso_triangulate_rectified(const double* x, int m, const viso_param* p, double* X) {
    if (m < 0 || !p || (m && (!x || !X))) { viso_set_error("viso_triangulate_rectified: bad argument"); return VISO_ERR_ARG; }
    if (m == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    double *dx, *dX;
    int r;
    if ((r = ctx_scratch(c, 0, sizeof(double) * 4 * (size_t)m, (void**)&dx)) < 0) return r;
    if ((r = ctx_scratch(c, 1, sizeof(double) * 3 * (size_t)m, (void**)&dX)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(dx, x, sizeof(double) * 4 * (size_t)m, hipMemcpyHostToDevice, c->stream));
    SolverParamsDev sp;
    fill_solver_params(&sp, p);
    hipLaunchKernelGGL(triangulate_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, dx, m, sp, dX);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(X, dX, sizeof(double) * 3 * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VISO_OK;
}
