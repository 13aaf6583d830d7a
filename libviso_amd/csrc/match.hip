// match.hip — descriptor-window SAD matcher for gfx950 (MI355X).
//
// Replaces match_desc + radiusSearch + sampsonDistance of the reference
// (src/viso.cpp:669-726, 170-203, 655-666).  Integer abs-diff work: no MFMA;
// the levers are column-bucketed images (a query tile only scans the +-radius column
// window of the target image), coalesced 256-B descriptor rows, wave64
// ballot/DPP reductions and keeping a problem's working set inside one XCD's L2.
//
// Kernels
//   sort_kp_kernel       per image: keypoints grouped into 256 column buckets (counting sort; no order inside a
//                        bucket, NaN x in the last one) + inverse permutation, bucket index, y order per 64-block
//   pack_desc_kernel     f32 N x dlen (boundary layout) -> u16 N x 128 rows (+bias), bucket order
//   match_kernel<false>  neighbour gate + epipolar gate + SAD + best/2nd-best   (hot, u16)
//   match_kernel<true>   same walk, double-accumulated SAD for non-integer data
//   sort_matches_kernel  (dist,i1)-ordered match list, inverse permutation, count
//
// Equivalence with the reference's list walk:
// cvflann returns in-radius targets ordered by key=(L1 distance, index), keeps
// the first K, and match_desc walks them while index > 0 (Q1).  Hence the
// scored set is { t : key(t) < min(key_K, key(target 0 if in radius)) } and,
// because the best/second-best update is order independent except for ties
// (`<=`: the LAST equal candidate wins, Q2), the winner is the candidate of
// minimal SAD with the LARGEST key.  No neighbour list is materialised, and the
// order in which candidates are visited (here: column buckets) is irrelevant.
#include "common.h"
#include <stdlib.h>
#include "match_dev.h"

#include <math.h>

// ------------------------------------------------------------------ helpers
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// ------------------------------------------------------------------ column index
// One workgroup per image: the keypoints grouped by a 256-bucket column map of their x (counting sort: histogram,
// scan, scatter with LDS atomics), NOT fully sorted — nothing downstream needs more: a tile is any 64 consecutive
// entries (its x extent is taken by a reduction), its target window is the range of whole buckets that covers that
// extent +- radius (bstart), and every matcher result is independent of the order in which candidates are visited.
// Which entry lands where inside a bucket depends on the atomics' order; results do not.  NaN x goes to the last
// bucket.  Also: the image's y range (the matchers bucket their windows by y), the number of non-NaN x, and the y
// order inside every block of 64 consecutive entries (ImageView::qord).
#ifndef VISO_IMG_THREADS
#define VISO_IMG_THREADS 512
#endif
#ifndef VISO_KP_REGS
#define VISO_KP_REGS 4
#endif

#ifdef VISO_DEBUG_VARIANTS   // timing aid (tools/experiments/sortkp_phases.py): 100 MHz time stamps of sort_kp_kernel's phases, workgroup 0
__device__ unsigned long long viso_dbg_sortkp_clk[8];
extern "C" int viso_debug_sortkp_clocks(unsigned long long* out8) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(viso_dbg_sortkp_clk), sizeof(unsigned long long) * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? VISO_OK : VISO_ERR_HIP;
}
#define SK_CLK(I) do { if (blockIdx.x == 0 && threadIdx.x == 0) viso_dbg_sortkp_clk[I] = wall_clock64(); } while (0)
#else
#define SK_CLK(I) do {} while (0)
#endif
struct KpImport2 { KpImport k[2]; };
__global__ __launch_bounds__(VISO_IMG_THREADS) void sort_kp_kernel(const ImageView* imgs, int n_img, int n64_alloc,
                                                                   uint32_t* zero_words, int n_zero, int* r8zero, KpImport2 imps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* ykey = reinterpret_cast<uint32_t*>(smem);   // [n rounded up to 64] sortable y of the entry at each position
    __shared__ int s_cnt[VISO_NB + 1];                    // bucket counts -> starts -> running offsets
    __shared__ float s_red[5][VISO_IMG_THREADS / 64];
    __shared__ float s_x[2];
    if ((int)blockIdx.x >= n_img) return;
    SK_CLK(0);
    if (r8zero && blockIdx.x == 0 && threadIdx.x < 4) r8zero[threadIdx.x] = 0;   // a run whose pack kernels count magnitudes (VISO_R8_*)
    if (zero_words) {   // this workgroup's slice of the run's counters (first kernel of a run: everything that counts comes later)
        const int per = (n_zero + n_img - 1) / n_img;
        for (int i = threadIdx.x; i < per; i += VISO_IMG_THREADS) {
            const int j = (int)blockIdx.x * per + i;
            if (j < n_zero) zero_words[j] = 0u;
        }
    }
    // One image straight from the caller's side (the plain family, KpImport): the view comes with the launch, the keypoints
    // are read from pinned host memory and left in the image's device array on the way, the header words are written here --
    // no copy kernel in front of this one
    const bool importing = imps.k[0].src_kp != nullptr;
    const KpImport& imp = imps.k[blockIdx.x & 1];   // (two at most: the two images of a plain-family call)
    const ImageView I = importing ? imp.view : imgs[blockIdx.x];
    const int n = importing ? imp.n : *I.n;
    if (importing) {
        if (threadIdx.x == 0) { *const_cast<int*>(I.n) = n; *I.bad = imp.bad0; }
        if (threadIdx.x < sizeof(ImageView) / 4) reinterpret_cast<uint32_t*>(imp.view_dst)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&imp.view)[threadIdx.x];
        // keypoints past the registers' share are walked from memory three times: bring them over first
        for (int i = threadIdx.x + VISO_KP_REGS * VISO_IMG_THREADS; i < n; i += VISO_IMG_THREADS) const_cast<float2*>(I.kp)[i] = imp.src_kp[i];
    }
    const float2* kp_in = importing ? imp.src_kp : I.kp;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // ---- x extent (x0 = smallest x, x1 = largest non-NaN x: the bucket map), finite y extent, number of non-NaN x
    float xmn = __builtin_huge_valf(), xmx = -__builtin_huge_valf(), ymn = __builtin_huge_valf(), ymx = -__builtin_huge_valf();
    float nvx = 0.f;
    // the kernel walks the keypoints three times (extent, histogram, scatter): a thread's first VISO_KP_REGS of them stay
    // in registers (all of them up to 2048 keypoints per image), so that only the first walk waits for memory
    float2 kreg[VISO_KP_REGS];
#pragma unroll
    for (int u = 0; u < VISO_KP_REGS; ++u) {
        const int i = threadIdx.x + u * VISO_IMG_THREADS;
        kreg[u] = i < n ? kp_in[i] : make_float2(0.f, 0.f);
        if (importing && i < n) const_cast<float2*>(I.kp)[i] = kreg[u];
    }
    // visit(i, k) for every keypoint of this thread: the register ones (compile-time slots), then the rest from memory
    auto walk = [&](auto visit) {
#pragma unroll
        for (int u = 0; u < VISO_KP_REGS; ++u) {
            const int i = threadIdx.x + u * VISO_IMG_THREADS;
            if (i < n) visit(i, kreg[u]);
        }
        for (int i = threadIdx.x + VISO_KP_REGS * VISO_IMG_THREADS; i < n; i += VISO_IMG_THREADS) visit(i, I.kp[i]);
    };
    walk([&](int, float2 k) {
        if (k.x == k.x) { xmn = fminf(xmn, k.x); xmx = fmaxf(xmx, k.x); nvx += 1.f; }
        if (fabsf(k.y) < 3.0e38f) { ymn = fminf(ymn, k.y); ymx = fmaxf(ymx, k.y); }
    });
    SK_CLK(1);
    xmn = viso_wave_fext<false>(xmn); xmx = viso_wave_fext<true>(xmx);
    ymn = viso_wave_fext<false>(ymn); ymx = viso_wave_fext<true>(ymx);
    nvx = viso_wave_fsum63(nvx);   // a count of at most a few thousand, as a float: exact in any order
    if (lane == 63) { s_red[0][wv] = xmn; s_red[1][wv] = xmx; s_red[2][wv] = ymn; s_red[3][wv] = ymx; s_red[4][wv] = nvx; }
    for (int b = threadIdx.x; b <= VISO_NB; b += VISO_IMG_THREADS) s_cnt[b] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = __builtin_huge_valf(), b = -__builtin_huge_valf(), c = __builtin_huge_valf(), d = -__builtin_huge_valf(), e = 0.f;
        for (int w = 0; w < VISO_IMG_THREADS / 64; ++w) {
            a = fminf(a, s_red[0][w]); b = fmaxf(b, s_red[1][w]); c = fminf(c, s_red[2][w]); d = fmaxf(d, s_red[3][w]); e += s_red[4][w];
        }
        float x0 = 0.f, scale = 0.f;
        if (e > 0.f) {
            x0 = a;
            if (b > a) scale = (float)VISO_NB / (b - a);
            if (!(scale > 0.f) || !(scale < 3.0e38f)) scale = 0.f;
        }
        if (!(c <= d)) { c = 0.f; d = 0.f; }   // no finite y at all
        s_x[0] = x0; s_x[1] = scale;
        I.xinfo[0] = x0; I.xinfo[1] = scale; I.xinfo[2] = c; I.xinfo[3] = d; I.xinfo[4] = e;
    }
    __syncthreads();
    SK_CLK(2);
    const float x0 = s_x[0], scale = s_x[1];
    // ---- counting sort by column bucket
    walk([&](int, float2 k) { atomicAdd(&s_cnt[bucket_of(k.x, x0, scale)], 1); });
    __syncthreads();
    if (wv == 0) {   // exclusive scan of the 256 counts: 4 per lane + wave scan
        int c[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = s_cnt[lane * 4 + k]; tot += c[k]; }
        const int incl = (int)viso_wave_scan((uint32_t)tot);
        int run = incl - tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_cnt[lane * 4 + k] = run; I.bstart[lane * 4 + k] = run; run += c[k]; }
        if (lane == 63) { s_cnt[VISO_NB] = run; I.bstart[VISO_NB] = n; }
    }
    __syncthreads();
    SK_CLK(3);
    const int n64 = (n + 63) & ~63;
    walk([&](int i, float2 k) {
        const int p = atomicAdd(&s_cnt[bucket_of(k.x, x0, scale)], 1);   // running offset of the bucket
        I.skp[p] = k;
        I.sidx[p] = i;
        I.rank[i] = p;
        const uint32_t yb = __float_as_uint(k.y);
        uint32_t yk = yb ^ ((yb >> 31) ? 0xffffffffu : 0x80000000u);
        if (yk == 0xffffffffu) yk = 0xfffffffeu;   // keep "past n" strictly last
        ykey[p] = yk;
    });
    for (int j = n + threadIdx.x; j < n64; j += VISO_IMG_THREADS) ykey[j] = 0xffffffffu;
    __syncthreads();
    SK_CLK(4);
    // ---- y order inside every block of 64 positions (the matcher kernels score rounds of y-adjacent queries of such
    // a block: their candidate sets overlap by ~2/3).  Any total order is valid; (y without its six lowest bits, position) is used:
    // ONE 32-bit key per entry, unique inside the block, so that a rank is a count of plain unsigned compares.
    // A wave takes a block: every lane holds the key of its own position and meets the block's 64 keys as SCALARS (v_readlane with a
    // constant lane: no LDS read per comparison -- walking the block through LDS was 6.8 of this kernel's 17.7 us for an image
    // of 2000 keypoints, its largest phase).  j walks in steps of the workgroup size, a multiple of 64: a wave's lanes are one block.
    static_assert(VISO_IMG_THREADS % 64 == 0, "a wave must cover one 64-block");
    for (int j = threadIdx.x; j < n64; j += VISO_IMG_THREADS) {
        const int base = j & ~63, me = j & 63;
        const uint32_t key = (ykey[j] & ~63u) | (uint32_t)me;   // (entries past n keep the largest keys: they are the block's last positions)
        int rank = 0;
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const uint32_t km = (uint32_t)__builtin_amdgcn_readlane((int)key, m);
            rank += km < key ? 1 : 0;
        }
        I.qord[base + rank] = (uint8_t)me;
    }
    SK_CLK(5);
}

int launch_sort_kp(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap_max, uint32_t* zero_words, int n_zero, int* r8zero, const KpImport* imp, int n_imp) {
    if (n_img <= 0) return VISO_OK;
    if (imp && (n_imp != n_img || n_img > 2)) { viso_set_error("launch_sort_kp: imports are one or two images, the whole launch"); return VISO_ERR_ARG; }
    KpImport2 ki{};
    if (imp) { ki.k[0] = imp[0]; if (n_imp > 1) ki.k[1] = imp[1]; }
    if (cap_max > VISO_SORT_MAX) {
        viso_set_error("more than %d keypoints per image is not supported by this build", VISO_SORT_MAX);
        return VISO_ERR_UNSUPPORTED;
    }
    const int n64 = (cap_max + 63) & ~63;
    const size_t lds = (size_t)n64 * sizeof(uint32_t) + 16;
    if (lds > 40 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)sort_kp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sort_kp_kernel, dim3(n_img), dim3(VISO_IMG_THREADS), lds, s, imgs_dev, n_img, n64, zero_words, n_zero, r8zero, ki);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// ------------------------------------------------------------------ pack
// One wave = VISO_PACK_RPW consecutive ORIGINAL rows = one contiguous, 16-B aligned run of RPW * dlen floats of
// the boundary-layout matrix: streamed in with 16-B loads per lane, staged in LDS (the rows are 121 floats long,
// so row boundaries fall anywhere in a lane's 16 B), then lane l converts floats 2l, 2l+1 of every row into one
// packed dword and the wave writes each 256-B row to its bucket-order position rank[i] (two full cache lines).
#define VISO_PACK_RPW 8   // rows per wave

__global__ __launch_bounds__(256) void pack_desc_kernel(const ImageView* __restrict__ imgs, int n_img,
                                                        int cap, int dlen, int* __restrict__ bad_any, int extras, int r8s, int* __restrict__ r8cnt, unsigned r8m) {
    __shared__ __attribute__((aligned(16))) float s_buf[4][VISO_PACK_RPW * VISO_ROW];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f32x4* gvec_t;
    typedef const __attribute__((address_space(1))) float* gflt_t;
    typedef const __attribute__((address_space(1))) int* gint_t;
    typedef __attribute__((address_space(1))) uint32_t* gout_t;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long row0 = wave * VISO_PACK_RPW;
    if (row0 >= (long long)n_img * cap) return;
    const int img = (int)(row0 / cap);      // cap is a multiple of VISO_PACK_RPW: a wave never straddles images
    const ImageView I = imgs[img];
    const int n = *I.n;
    const int r0 = (int)(row0 % cap);
    if (r0 >= n) return;
    const int nrows = min(VISO_PACK_RPW, n - r0);
    const int nf = nrows * dlen;            // floats of this wave's rows, contiguous in the matrix
    const float* src = I.frows + (size_t)r0 * dlen;
    float* buf = s_buf[wv];
    if ((reinterpret_cast<size_t>(src) & 15) == 0) {
        const gvec_t v = (gvec_t)reinterpret_cast<const f32x4*>(src);
        const int nq = nf >> 2;
#pragma unroll
        for (int t = 0; t < (VISO_PACK_RPW * VISO_ROW / 4 + 63) / 64; ++t) {
            const int q = lane + 64 * t;
            if (q < nq) *reinterpret_cast<f32x4*>(buf + 4 * q) = v[q];
        }
        const int e = (nq << 2) + lane;     // the up to 3 floats behind the last full 16 B
        if (e < nf) buf[e] = ((gflt_t)src)[e];
    } else {
        for (int e = lane; e < nf; e += 64) buf[e] = ((gflt_t)src)[e];
    }
    int dst = 0;
    if (lane < nrows) dst = ((gint_t)I.rank)[r0 + lane];
    __builtin_amdgcn_wave_barrier();
    const int c = 2 * lane;
    bool isbad = false;
#pragma unroll
    for (int k = 0; k < VISO_PACK_RPW; ++k) {
        if (k >= nrows) break;              // wave uniform
        const float a = c < dlen ? buf[k * dlen + c] : 0.f;
        const float b = c + 1 < dlen ? buf[k * dlen + c + 1] : 0.f;
        const float ar = rintf(a), br = rintf(b);
        if (!(a == ar) || a < -32768.f || a > 32767.f || !(b == br) || b < -32768.f || b > 32767.f) isbad = true;
        const uint32_t ua = (uint32_t)((int)ar + VISO_BIAS) & 0xffffu, ub = (uint32_t)((int)br + VISO_BIAS) & 0xffffu;
        const int d = __builtin_amdgcn_readlane(dst, k);
#ifdef VISO_DEBUG_VARIANTS   // timing experiment only ($VISO_EXP_PACK_NO_U16, HISTORY.md round 6): what the kernel would cost without the u16 rows
        if (!(extras & 0x100))
#endif
        ((gout_t)reinterpret_cast<uint32_t*>(I.rows + (size_t)d * VISO_ROW))[lane] = ua | (ub << 16);
        // block sums (ImageView::sums; only match_prune_kernel reads them): lanes 16b..16b+15 hold block b; meaningless
        // for flagged images (never read then)
        if (extras & VISO_PACK_SUMS) {   // uniform
            const uint2 bs = pack_block_sums((int)ar + (int)br);
            if (lane == 0) I.sums[d] = bs;
        }
        if (extras & VISO_PACK_ROWS8) store_row8(I.rows8, (size_t)d, lane, (int)ar, (int)br, r8s);   // uniform
    }
    if (__any(isbad) && lane == 0) { atomicOr(I.bad, 1); atomicOr(bad_any, 1); }
    if (r8cnt && ((unsigned)wave & r8m) == 0) {   // uniform, ~256 waves of the launch: the rows once more (they are still in LDS), counted
        R8Count r8c = {0, 0, 0, 0};
        for (int k = 0; k < nrows; ++k)
            r8_count(r8c, c < dlen ? (int)rintf(buf[k * dlen + c]) : 0, c + 1 < dlen ? (int)rintf(buf[k * dlen + c + 1]) : 0);
        r8_flush(r8c, r8cnt, lane);
    }
}

// The same rows from int16 descriptors (viso_batch_upload_i16*: the lossless encoding of the reference's N x 121
// CV_32F Sobel windows, half the PCIe bytes).  desc16: [n_img][cap][dlen] int16, tightly packed; one wave = 8
// consecutive rows = one contiguous 16-B aligned run of 8 * dlen * 2 bytes.  An int16 always fits the rows: no flag.
__global__ __launch_bounds__(256) void pack_desc_i16_kernel(const ImageView* __restrict__ imgs, int n_img, int cap, int cap_stride,
                                                            int dlen, const int16_t* __restrict__ desc16, int extras, int r8s, int* __restrict__ r8cnt, unsigned r8m) {
    __shared__ __attribute__((aligned(16))) uint16_t s_buf[4][VISO_PACK_RPW * VISO_ROW];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* gvec_t;
    typedef const __attribute__((address_space(1))) uint16_t* gu16_t;
    typedef const __attribute__((address_space(1))) int* gint_t;
    typedef __attribute__((address_space(1))) uint32_t* gout_t;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long row0 = wave * VISO_PACK_RPW;
    if (row0 >= (long long)n_img * cap) return;
    const int img = (int)(row0 / cap);      // cap is a multiple of VISO_PACK_RPW: a wave never straddles images
    const ImageView I = imgs[img];
    const int n = *I.n;
    const int r0 = (int)(row0 % cap);
    if (r0 >= n) return;
    const int nrows = min(VISO_PACK_RPW, n - r0);
    const int ne = nrows * dlen;            // int16 elements of this wave's rows, contiguous
    const int16_t* src = desc16 + ((size_t)img * cap_stride + r0) * dlen;
    uint16_t* buf = s_buf[wv];
    if ((reinterpret_cast<size_t>(src) & 15) == 0) {
        const gvec_t v = (gvec_t)reinterpret_cast<const u32x4*>(src);
        const int nq = ne >> 3;             // 16-B pieces
        for (int q = lane; q < nq; q += 64) *reinterpret_cast<u32x4*>(buf + 8 * q) = v[q];
        const int e = (nq << 3) + lane;     // the up to 7 elements behind the last full 16 B
        if (e < ne) buf[e] = ((gu16_t)reinterpret_cast<const uint16_t*>(src))[e];
    } else {
        for (int e = lane; e < ne; e += 64) buf[e] = ((gu16_t)reinterpret_cast<const uint16_t*>(src))[e];
    }
    int dst = 0;
    if (lane < nrows) dst = ((gint_t)I.rank)[r0 + lane];
    __builtin_amdgcn_wave_barrier();
    const int c = 2 * lane;
#pragma unroll
    for (int k = 0; k < VISO_PACK_RPW; ++k) {
        if (k >= nrows) break;              // wave uniform
        const int a = c < dlen ? (int)(int16_t)buf[k * dlen + c] : 0;
        const int b = c + 1 < dlen ? (int)(int16_t)buf[k * dlen + c + 1] : 0;
        const uint32_t ua = (uint32_t)(a + VISO_BIAS) & 0xffffu, ub = (uint32_t)(b + VISO_BIAS) & 0xffffu;
        const int d = __builtin_amdgcn_readlane(dst, k);
        ((gout_t)reinterpret_cast<uint32_t*>(I.rows + (size_t)d * VISO_ROW))[lane] = ua | (ub << 16);
        if (extras & VISO_PACK_SUMS) {   // uniform
            const uint2 bs = pack_block_sums(a + b);
            if (lane == 0) I.sums[d] = bs;
        }
        if (extras & VISO_PACK_ROWS8) store_row8(I.rows8, (size_t)d, lane, a, b, r8s);   // uniform
    }
    if (r8cnt && ((unsigned)wave & r8m) == 0) {   // uniform, ~256 waves of the launch: the rows once more (still in LDS), counted
        R8Count r8c = {0, 0, 0, 0};
        for (int k = 0; k < nrows; ++k)
            r8_count(r8c, c < dlen ? (int)(int16_t)buf[k * dlen + c] : 0, c + 1 < dlen ? (int)(int16_t)buf[k * dlen + c + 1] : 0);
        r8_flush(r8c, r8cnt, lane);
    }
}

int launch_pack_i16(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap, int dlen, const int16_t* desc16, int extras, int r8s, int* r8cnt) {
    if (n_img <= 0) return VISO_OK;
    if (dlen > VISO_ROW) { viso_set_error("int16 descriptors longer than %d are not supported", VISO_ROW); return VISO_ERR_UNSUPPORTED; }
    const int capp = (cap + VISO_PACK_RPW - 1) / VISO_PACK_RPW * VISO_PACK_RPW;
    const long long waves = (long long)n_img * capp / VISO_PACK_RPW;
    if (waves == 0) return VISO_OK;
    hipLaunchKernelGGL(pack_desc_i16_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, imgs_dev, n_img, capp, cap, dlen, desc16, extras, r8s, (extras & VISO_PACK_ROWS8) ? r8cnt : nullptr, r8_mask(waves));
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

__global__ void flag_all_kernel(int* flags, int n, int* any) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = 1;
    if (i == 0) *any = 1;
}

int launch_pack(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap, int dlen, int* bad_img, int* bad_any, int extras, int r8s, int* r8cnt) {
    if (n_img <= 0) return VISO_OK;
    if (dlen > VISO_ROW) {  // rows do not fit the packed format: every image takes the general path
        hipLaunchKernelGGL(flag_all_kernel, dim3((n_img + 255) / 256), dim3(256), 0, s, bad_img, n_img, bad_any);
        HIP_TRY(hipGetLastError());
        return VISO_OK;
    }
    // the kernel addresses rows as img * capp + r with capp a multiple of the rows per wave,
    // so that a wave never straddles two images
    const int capp = (cap + VISO_PACK_RPW - 1) / VISO_PACK_RPW * VISO_PACK_RPW;
    const long long waves = (long long)n_img * capp / VISO_PACK_RPW;
    if (waves == 0) return VISO_OK;
    const int blocks = (int)((waves + 3) / 4);
#ifdef VISO_DEBUG_VARIANTS
    { static const int no_u16 = [] { const char* e = getenv("VISO_EXP_PACK_NO_U16"); return e && *e == '1'; }(); if (no_u16) extras |= 0x100; }
#endif
    hipLaunchKernelGGL(pack_desc_kernel, dim3(blocks), dim3(256), 0, s, imgs_dev, n_img, capp, dlen, bad_any, extras, r8s, (extras & VISO_PACK_ROWS8) ? r8cnt : nullptr, r8_mask(waves));
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// ------------------------------------------------------------------ trackers
// Best / second-best bookkeeping (src/viso.cpp:703-709), order independent:
// d1 = min SAD, d2 = second order statistic WITH multiplicity, winner = the
// candidate with SAD == d1 and the largest key (distance bits, original index).

// Exact tracker (keys compared on every update).
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

// Cheap tracker: remembers ANY position reaching d1 and whether d1 was reached
// more than once; ties (rare) are resolved by re-scoring with TrackU.
struct TrackT {
    uint32_t d1, d2, bp, tie;
    __device__ __forceinline__ void init() { d1 = d2 = 0xffffffffu; bp = 0; tie = 0; }
    __device__ __forceinline__ void add(uint32_t s, uint32_t p) {   // s == 0xffffffff: invalid slot
        const bool lt = s < d1;
        const bool eq = (s == d1) && (s != 0xffffffffu);
        d2 = (lt || eq) ? d1 : min(d2, s);
        tie = lt ? 0u : (tie | (uint32_t)eq);
        d1 = lt ? s : d1;
        bp = lt ? p : bp;
    }
    __device__ __forceinline__ void merge(uint32_t od1, uint32_t od2, uint32_t obp, uint32_t otie) {
        if (od1 < d1) { d2 = min(d1, od2); d1 = od1; bp = obp; tie = otie; }
        else if (od1 == d1) { if (d1 != 0xffffffffu) { d2 = d1; tie = 1; } }
        else d2 = min(d2, od1);
    }
};

// ------------------------------------------------------------------ scorers
// Queue entries: (sorted target position p, float bits of the keypoint distance).
//
// Fast path: VISO_LPC lanes per candidate, 16 / VISO_LPC 16-B loads each (load k of
// lane `sub` fetches chunk k * VISO_LPC + sub of the 256-B row, so the lanes of a
// candidate read VISO_LPC * 16 contiguous bytes per instruction), v_sad_u16 chain,
// log2(VISO_LPC) DPP adds; 64 / VISO_LPC candidates per wave pass.
// Measured (MI355X, bench.py defaults): 8 lanes -> 0.90 ms per launch, 4 lanes -> 1.24 ms although it
// issues fewer instructions: every load then touches 16 rows and only half of each 128-B line.
#ifndef VISO_LPC
#define VISO_LPC 8
#endif
#define VISO_CPP (64 / VISO_LPC)                 // candidates per wave pass
#define VISO_NQ (16 / VISO_LPC)                  // 16-B chunks per lane
#define VISO_NPASS (VISO_LPC == 8 ? 2 : 1)       // passes whose loads are in flight together (16 candidates)

__device__ __forceinline__ uint32_t lpc_sum(uint32_t v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
#if VISO_LPC == 8
    v += dpp_mov<0x141>(v);  // row_half_mirror (quads are uniform by now)
#endif
    return v;
}

struct ScorerU16 {
    const uint16_t* d2;
    const int* sidx2;
    uint4 q[VISO_NQ];   // this lane's query elements
    int g, sub;
    __device__ __forceinline__ void begin(const MatchProblem& P, int j, int lane) {
        g = lane / VISO_LPC; sub = lane % VISO_LPC;
        d2 = P.t.rows; sidx2 = P.t.sidx;
        const uint16_t* qr = P.q.rows + (size_t)j * VISO_ROW + sub * 8;
#pragma unroll
        for (int k = 0; k < VISO_NQ; ++k) q[k] = *reinterpret_cast<const uint4*>(qr + k * VISO_LPC * 8);
    }
    __device__ __forceinline__ void load_row(uint32_t p, uint4 (&r)[VISO_NQ]) const {
        const uint16_t* row = d2 + (size_t)p * VISO_ROW + sub * 8;
#pragma unroll
        for (int k = 0; k < VISO_NQ; ++k) r[k] = *reinterpret_cast<const uint4*>(row + k * VISO_LPC * 8);
    }
    __device__ __forceinline__ uint32_t sad_row(const uint4 (&r)[VISO_NQ]) const {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < VISO_NQ; ++k) {
            s = __builtin_amdgcn_sad_u16(r[k].x, q[k].x, s);
            s = __builtin_amdgcn_sad_u16(r[k].y, q[k].y, s);
            s = __builtin_amdgcn_sad_u16(r[k].z, q[k].z, s);
            s = __builtin_amdgcn_sad_u16(r[k].w, q[k].w, s);
        }
        return lpc_sum(s);
    }
    __device__ __forceinline__ void score_fast(const uint2* queue, int n, TrackT& t) const {
        for (int b = 0; b < n; b += VISO_CPP * VISO_NPASS) {
            uint2 e[VISO_NPASS];
            uint4 r[VISO_NPASS][VISO_NQ];
#pragma unroll
            for (int p = 0; p < VISO_NPASS; ++p) {
                e[p] = queue[min(b + p * VISO_CPP + g, n - 1)];
                load_row(e[p].x, r[p]);
            }
#pragma unroll
            for (int p = 0; p < VISO_NPASS; ++p) {
                const uint32_t s = sad_row(r[p]);
                t.add((b + p * VISO_CPP + g) < n ? s : 0xffffffffu, e[p].x);
            }
        }
    }
    __device__ __forceinline__ void score_exact(const uint2* queue, int n, TrackU& t) const {
        for (int b = 0; b < n; b += VISO_CPP) {
            const uint2 e = queue[min(b + g, n - 1)];
            uint4 r[VISO_NQ];
            load_row(e.x, r);
            const uint32_t oi = (uint32_t)sidx2[e.x];
            const uint32_t s = sad_row(r);
            t.add((b + g) < n ? s : 0xffffffffu, e.y, oi);
        }
    }
};

__device__ __forceinline__ void merge_groups(TrackT& t) {
#pragma unroll
    for (int m = VISO_LPC; m <= 32; m <<= 1) {
        const uint32_t od1 = __shfl_xor(t.d1, m), od2 = __shfl_xor(t.d2, m);
        const uint32_t obp = __shfl_xor(t.bp, m), ot = __shfl_xor(t.tie, m);
        t.merge(od1, od2, obp, ot);
    }
}
__device__ __forceinline__ void merge_groups(TrackU& t) {
#pragma unroll
    for (int m = VISO_LPC; m <= 32; m <<= 1) {
        const uint32_t od1 = __shfl_xor(t.d1, m), od2 = __shfl_xor(t.d2, m);
        const uint32_t okd = __shfl_xor(t.kd, m), oki = __shfl_xor(t.ki, m);
        t.merge(od1, od2, okd, oki);
    }
}

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
    const int* sidx2;
    int dlen, lane;
    __device__ __forceinline__ void begin(const MatchProblem& P, int orig_q, int lane_, int dlen_) {
        dlen = dlen_; lane = lane_;
        f1row = P.q.frows + (size_t)orig_q * dlen;
        f2 = P.t.frows; sidx2 = P.t.sidx;
    }
    __device__ __forceinline__ void score(const uint2* queue, int n, TrackD& t) const {
        for (int j = lane; j < n; j += VISO_WAVE) {
            const uint2 e = queue[j];
            const uint32_t oi = (uint32_t)sidx2[e.x];
            const float* a = f2 + (size_t)oi * dlen;
            double s = 0;
            for (int c = 0; c < dlen; ++c) {
                const float df = a[c] - f1row[c];
                s += (double)fabsf(df);
            }
            t.add(s, e.y, oi);
        }
    }
};

__device__ __forceinline__ void merge_lanes(TrackD& t) {
#pragma unroll
    for (int m = 1; m < VISO_WAVE; m <<= 1) {
        const double od1 = __shfl_xor(t.d1, m), od2 = __shfl_xor(t.d2, m);
        const uint32_t okd = __shfl_xor(t.kd, m), oki = __shfl_xor(t.ki, m);
        const int oany = __shfl_xor((int)t.any, m);
        t.merge(od1, od2, okd, oki, oany != 0);
    }
}

// ------------------------------------------------------------------ the walk
// Window of the target image a query tile has to look at: sorted positions
// [lo, lo+W).  The first `cap` entries are staged in LDS, the rest (dense
// clusters only) is read from global memory.
struct Window {
    const float2* gkp; const int* gidx;   // global, sorted order
    const float2* skp; const int* sidx;   // LDS copies of [lo, lo+cap)
    int lo, W, cap;
    __device__ __forceinline__ float2 kp(int w) const { return w < cap ? skp[w] : gkp[lo + w]; }
    __device__ __forceinline__ int idx(int w) const { return w < cap ? sidx[w] : gidx[lo + w]; }
};

struct QueryResult { int idx; int dist; double bd1, bd2; };

// SLOW = false: the tile kernel's path; returns false (nothing scored) when the
// candidate set needs the K cap or does not fit the queue — the query is then
// handed to the overflow kernel, which runs the SLOW = true instantiation.
template <bool GENERAL, bool SLOW, int EPI>
__device__ bool match_query(const MatchProblem& P, const MatchParamsDev& mp, int j, const Window& win,
                            float2 kp0, bool has0, uint2* queue, int lane, int dlen,
                            unsigned long long& scored, QueryResult& out, uint4* stage = nullptr, int stage_cap = 0) {
    const float2 q = P.q.skp[j];
    const float qx = q.x, qy = q.y;
    const float radius = mp.radius;
    // Q1 (src/viso.cpp:693): the walk stops at target index 0: everything at or
    // behind key(0) = (d0, 0) is cut, i.e. dist < d0 strictly.
    float d0cut = __builtin_huge_valf();
    if (has0) {
        const float d0 = l1_kp(qx, qy, kp0);
        if (d0 <= radius) d0cut = d0;
    }
    // ---- phase A: scan the window, queue the in-radius candidates
    int cnt = 0;
    for (int base = 0; base < win.W; base += VISO_WAVE) {
        const int w = base + lane;
        bool in = false;
        float d = 0.f;
        if (w < win.W) {
            d = l1_kp(qx, qy, win.kp(w));
            in = (d <= radius) && (d < d0cut);
        }
        const unsigned long long m = __ballot(in);
        if (m) {
            const int pos = cnt + mbcnt(m);
            if (!SLOW && in && pos < VISO_QCAP) queue[pos] = make_uint2((uint32_t)(win.lo + w), __float_as_uint(d));
            // slow path: the in-radius set (position, distance bits, original index) is staged in LDS once, so that
            // the 64 selection steps and the final pass below do not walk the window in global memory 65 times
            if (SLOW && in && pos < stage_cap) stage[pos] = make_uint4((uint32_t)(win.lo + w), __float_as_uint(d), (uint32_t)win.idx(w), 0u);
            cnt += __popcll(m);
        }
    }
    const int K = mp.K;
    const double* F = mp.F;
    ScorerU16 su;
    ScorerF32 sf;
    TrackU tu;
    TrackD td;
    if constexpr (GENERAL) { sf.begin(P, P.q.sidx[j], lane, dlen); td.init(); }
    else { su.begin(P, j, lane); tu.init(); }
    if constexpr (!SLOW) {
        if (cnt > K || cnt > VISO_QCAP) return false;
        // ---- fast path: the whole candidate set is in the queue
        int n = cnt;
        if (EPI < 0 ? (mp.epi != 0) : (EPI != 0)) {
            int wr = 0;
            for (int b = 0; b < n; b += VISO_WAVE) {
                const int k = b + lane;
                bool pass = false;
                uint2 e = make_uint2(0, 0);
                if (k < n) {
                    e = queue[k];
                    const float2 t2 = win.kp((int)e.x - win.lo);
                    const double s = sampson_dev(F, qx, qy, t2.x, t2.y);
                    pass = isfinite(s) && !(s > mp.sampson_thresh);
                }
                const unsigned long long m = __ballot(pass);
                const int pos = wr + mbcnt(m);
                __builtin_amdgcn_wave_barrier();
                if (pass) queue[pos] = e;   // pos <= k: in-place compaction is safe
                wr += __popcll(m);
            }
            n = wr;
        }
        __builtin_amdgcn_wave_barrier();
        scored += (unsigned long long)n;
        if constexpr (GENERAL) {
            if (n > 0) sf.score(queue, n, td);
        } else {
            TrackT tt;
            tt.init();
            if (n > 0) su.score_fast(queue, n, tt);
            merge_groups(tt);
            if (tt.tie) {
                su.score_exact(queue, n, tu);   // rare: equal minimal SADs -> largest key wins
            } else {
                out.idx = tt.d1 != 0xffffffffu ? win.idx((int)tt.bp - win.lo) : -1;
                out.dist = (int)tt.d1;
                out.bd1 = (double)tt.d1;
                out.bd2 = tt.d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)tt.d2;
                return true;
            }
        }
    } else {
        // ---- slow path (dense clusters): apply the K cap, then stream in batches
        uint32_t tkd = 0xffffffffu, tki = 0xffffffffu;   // threshold key (exclusive)
        const bool staged = cnt <= stage_cap;
        if (cnt > K && staged) {
            // key_K = largest v with |{key < v}| <= K, built bit by bit (64-bit key) over the staged set
            unsigned long long v = 0;
            for (int bit = 63; bit >= 0; --bit) {
                const unsigned long long trial = v | (1ull << bit);
                const uint32_t kd = (uint32_t)(trial >> 32), ki = (uint32_t)trial;
                int c = 0;
                for (int base = 0; base < cnt; base += VISO_WAVE) {
                    const int i = base + lane;
                    bool in = false;
                    if (i < cnt) { const uint4 e = stage[i]; in = key_less(e.y, e.z, kd, ki); }
                    c += __popcll(__ballot(in));
                }
                if (c <= K) v = trial;
            }
            tkd = (uint32_t)(v >> 32); tki = (uint32_t)v;
        } else if (cnt > K) {
            // key_K = largest v with |{key < v}| <= K, built bit by bit (64-bit key)
            unsigned long long v = 0;
            for (int bit = 63; bit >= 0; --bit) {
                const unsigned long long trial = v | (1ull << bit);
                const uint32_t kd = (uint32_t)(trial >> 32), ki = (uint32_t)trial;
                int c = 0;
                for (int base = 0; base < win.W; base += VISO_WAVE) {
                    const int w = base + lane;
                    bool in = false;
                    if (w < win.W) {
                        const float d = l1_kp(qx, qy, win.kp(w));
                        in = (d <= radius) && (d < d0cut) &&
                             key_less(__float_as_uint(d), (uint32_t)win.idx(w), kd, ki);
                    }
                    c += __popcll(__ballot(in));
                }
                if (c <= K) v = trial;
            }
            tkd = (uint32_t)(v >> 32); tki = (uint32_t)v;
        }
        int qn = 0;
        const int n_walk = staged ? cnt : win.W;   // staged: the in-radius set in LDS; otherwise the window in global memory
        for (int base = 0; base < n_walk; base += VISO_WAVE) {
            const int w = base + lane;
            bool in = false;
            uint32_t e_pos = 0, e_d = 0;
            if (w < n_walk) {
                float2 t2;
                if (staged) {
                    const uint4 e = stage[w];
                    e_pos = e.x; e_d = e.y;
                    in = key_less(e.y, e.z, tkd, tki);
                    if (in && mp.epi) t2 = win.gkp[e.x];
                } else {
                    t2 = win.kp(w);
                    const float d = l1_kp(qx, qy, t2);
                    e_pos = (uint32_t)(win.lo + w); e_d = __float_as_uint(d);
                    in = (d <= radius) && (d < d0cut) &&
                         key_less(__float_as_uint(d), (uint32_t)win.idx(w), tkd, tki);
                }
                if (in && mp.epi) {
                    const double s = sampson_dev(F, qx, qy, t2.x, t2.y);
                    in = isfinite(s) && !(s > mp.sampson_thresh);
                }
            }
            const unsigned long long m = __ballot(in);
            const int pos = qn + mbcnt(m);
            if (in) queue[pos] = make_uint2(e_pos, e_d);   // qn < 64 => pos < 128
            qn += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (qn >= VISO_WAVE) {
                if constexpr (GENERAL) sf.score(queue, VISO_WAVE, td); else su.score_exact(queue, VISO_WAVE, tu);
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
            if constexpr (GENERAL) sf.score(queue, qn, td); else su.score_exact(queue, qn, tu);
            scored += (unsigned long long)qn;
        }
    }
    if constexpr (GENERAL) {
        merge_lanes(td);
        out.idx = td.any ? (int)td.ki : -1;
        // Vec3i(i, idx, double): conversion truncates; saturate instead of UB
        const double c = td.d1 > 2147483647.0 ? 2147483647.0 : td.d1;
        out.dist = td.any ? (int)c : 0;
        out.bd1 = td.d1; out.bd2 = td.d2;
    } else {
        merge_groups(tu);
        out.idx = tu.d1 != 0xffffffffu ? (int)tu.ki : -1;
        out.dist = (int)tu.d1;
        out.bd1 = (double)tu.d1;
        out.bd2 = tu.d2 == 0xffffffffu ? 1.7976931348623157e308 : (double)tu.d2;
    }
    return true;
}

// blockIdx -> (problem, query tile).  Blocks b and b+8 share an XCD (round
// robin dispatch; speed only), so problem = f(b % 8, b / 8): all tiles of one
// problem run on one XCD and re-read its target rows from that XCD's L2.
// Groups of 8 problems can be strided so that a launch only enumerates the
// problems of one kind: group = (g / gc) * gs + gf + g % gc  (batches lay the
// problems out as [8 stereo][8 temporal-left][8 temporal-right] per 8 frames).
__device__ __forceinline__ void block_to_problem(int b, int n_probs, int bpp, int gs, int gf, int gc, int& prob, int& qblk) {
    const int xcd = b & 7, slot = b >> 3;
    const int g = slot / bpp;
    prob = ((g / gc) * gs + gf + g % gc) * 8 + xcd;
    qblk = slot % bpp;
    if (prob >= n_probs) prob = -1;
}

struct MatchArgs {
    const MatchProblem* probs;
    int n_probs, bpp, dlen, gs, gf, gc;
    int vblocks;   // number of (problem, tile) slots; the grid may be smaller (blocks stride over the slots)
    const int* bad;
    const int2* ovf_q;   // the launch's overflow queue and its count (MatchProblem::ovf / ovf_cnt)
    const int* ovf_cnt;
    MatchParamsDev mp[2];
};

#define VISO_MATCH_WAVES (VISO_MATCH_THREADS / VISO_WAVE)

// EPI is a compile-time copy of MatchParams::enforce_epipolar: the temporal
// instantiation carries no fp64 Sampson code and needs fewer registers (more
// waves per SIMD to hide the L2 gather latency); each instantiation skips the
// problems of the other kind.
// One (problem, query tile) slot: the body of match_kernel.  `vb` is the slot number (block index of a full grid).
template <bool GENERAL, int EPI>
__device__ __forceinline__ void match_tile_slot(const MatchArgs& a, int vb, uint2 (*s_queue)[VISO_QCAP], float2* s_kp,
                                                int* s_idx, float* s_xr) {
    int prob, qblk;
    block_to_problem(vb, a.n_probs, a.bpp, a.gs, a.gf, a.gc, prob, qblk);
    if (prob < 0) return;
    const MatchProblem P = a.probs[prob];
    if (((*P.q.bad | *P.t.bad) != 0) != GENERAL) return;   // decided per problem by the pack kernel (no host round trip)
    const int n1 = *P.q.n, n2 = *P.t.n;
    const int q0 = qblk * VISO_QPB;
    if (q0 >= n1) return;
    const int q1 = min(q0 + VISO_QPB, n1);
    const MatchParamsDev& mp = a.mp[P.pidx];
    if (EPI >= 0 && (mp.epi != 0) != (EPI != 0)) return;   // EPI < 0: the gate is a run-time matter (match_query), every problem is this kernel's
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // x range of the tile (any 64 consecutive bucket-order entries; the reductions ignore NaN x)
    if (wave == 0) {
        float x = (q0 + lane < q1) ? P.q.skp[q0 + lane].x : __builtin_nanf("");
        const float mn = viso_wave_fext<false>(x), mx = viso_wave_fext<true>(x);
        if (lane == 0) { s_xr[0] = mn; s_xr[1] = mx; }
    }
    __syncthreads();
    Window win;
    win.gkp = P.t.skp; win.gidx = P.t.sidx; win.skp = s_kp; win.sidx = s_idx;
    win.lo = 0; win.W = 0; win.cap = 0;
    const float xa = s_xr[0], xb = s_xr[1];
    if (n2 > 0 && xa == xa) {
        const float r = mp.radius;
        const float slack = (fabsf(xa) + fabsf(xb) + fabsf(r)) * 1e-6f + 1e-6f;
        const float x0 = P.t.xinfo[0], scale = P.t.xinfo[1];
        if (r >= 0.f) {
            const int blo = bucket_of(xa - r - slack, x0, scale);
            const int bhi = bucket_of(xb + r + slack, x0, scale);
            win.lo = P.t.bstart[blo];
            win.W = P.t.bstart[bhi + 1] - win.lo;
        }
    }
    win.cap = min(win.W, VISO_KPCAP);
    for (int w = threadIdx.x; w < win.cap; w += VISO_MATCH_THREADS) {
        s_kp[w] = P.t.skp[win.lo + w];
        s_idx[w] = P.t.sidx[win.lo + w];
    }
    float2 kp0 = make_float2(0.f, 0.f);
    const bool has0 = n2 > 0;
    if (has0) kp0 = P.t.skp[P.t.rank[0]];
    __syncthreads();
    unsigned long long scored = 0;
    for (int j = q0 + wave; j < q1; j += VISO_MATCH_WAVES) {
        QueryResult r;
        const bool done = match_query<GENERAL, false, EPI>(P, mp, j, win, kp0, has0, s_queue[wave], lane, a.dlen, scored, r);
        if (lane == 0) {
            if (done) {
                bool accept = r.idx >= 0;
                // src/viso.cpp:713-716 — ratio test in double (Q3)
                if (accept && mp.second) accept = r.bd1 < r.bd2 * mp.ratio;
                P.res[P.q.sidx[j]] = make_int2(accept ? r.idx : -1, r.dist);
            } else {
                P.ovf[atomicAdd(P.ovf_cnt, 1)] = make_int2(prob, j);   // dense cluster: overflow kernel
            }
        }
    }
    if (lane == 0 && scored) atomicAdd(P.scored, scored);
}

template <bool GENERAL, int EPI>
__global__ __attribute__((amdgpu_waves_per_eu(GENERAL ? 4 : 8, 8))) __launch_bounds__(VISO_MATCH_THREADS) void match_kernel(MatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint2 s_queue[VISO_MATCH_WAVES][VISO_QCAP];
    __shared__ float2 s_kp[VISO_KPCAP];
    __shared__ int s_idx[VISO_KPCAP];
    __shared__ float s_xr[2];
    if constexpr (GENERAL) {
        if (*a.bad == 0) return;   // no image of this run is flagged: nothing to do
        // normally idle: launched on a small grid that strides over the slots when it does have work
        for (int vb = blockIdx.x; vb < a.vblocks; vb += gridDim.x) {
            __syncthreads();   // LDS of the previous slot is free
            match_tile_slot<GENERAL, EPI>(a, vb, s_queue, s_kp, s_idx, s_xr);
        }
    } else {
        match_tile_slot<GENERAL, EPI>(a, blockIdx.x, s_queue, s_kp, s_idx, s_xr);
    }
}

// Queries whose in-radius set exceeds K or an LDS list (dense keypoint clusters), or whose minimum is an exact tie:
// exact K-cap selection + streaming, reading the window from global memory.  ONE queue for the whole launch
// (MatchProblem::ovf): a fixed grid of waves strides over it, so the work is shared evenly whichever problems it comes
// from; a handful of entries for ordinary data.
#define VISO_OVF_GRID 4096   // one-wave workgroups
#define VISO_OVF_STAGE 768   // in-radius targets of one query staged in LDS (12 KB per wave); more: the window is re-walked

// One wave per workgroup: the kernel has a few microseconds of work but sits on the batch's critical path behind the
// tile kernels, usually while ANOTHER batch's tile kernel owns the GPU (7 workgroups of 4 x 72 registers and 21 KB of
// LDS per CU, refilled the moment one leaves).  A workgroup only starts when its whole footprint is free on one CU: a
// 4-wave workgroup with 57 KB of LDS waited for milliseconds (kernel trace), a single wave with 14 KB
// takes the first slot any finishing tile workgroup leaves.
template <bool GENERAL>
__global__ __launch_bounds__(VISO_WAVE) void match_overflow_kernel(MatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint2 s_queue[VISO_QCAP];
    __shared__ __attribute__((aligned(16))) uint4 s_stage[VISO_OVF_STAGE];
    if (GENERAL && *a.bad == 0) return;
    const int n_ovf = *a.ovf_cnt;
    const int lane = threadIdx.x;
    for (int k = (int)blockIdx.x; k < n_ovf; k += (int)gridDim.x) {
        const int2 e = a.ovf_q[k];
        const MatchProblem P = a.probs[e.x];
        if (((*P.q.bad | *P.t.bad) != 0) != GENERAL) continue;   // the other instantiation's
        const int j = e.y;
        const int n2 = *P.t.n;
        const MatchParamsDev& mp = a.mp[P.pidx];
        float2 kp0 = make_float2(0.f, 0.f);
        const bool has0 = n2 > 0;
        if (has0) kp0 = P.t.skp[P.t.rank[0]];
        unsigned long long scored = 0;
        Window win;
        win.gkp = P.t.skp; win.gidx = P.t.sidx; win.skp = nullptr; win.sidx = nullptr;
        win.lo = 0; win.W = 0; win.cap = 0;   // the query's own +-radius column window, global reads
        {
            const float qx = P.q.skp[j].x, r = mp.radius;
            if (n2 > 0 && r >= 0.f) {
                if (qx == qx) {
                    const float slack = (2.f * fabsf(qx) + fabsf(r)) * 1e-6f + 1e-6f;
                    const float x0 = P.t.xinfo[0], scale = P.t.xinfo[1];
                    win.lo = P.t.bstart[bucket_of(qx - r - slack, x0, scale)];
                    win.W = P.t.bstart[bucket_of(qx + r + slack, x0, scale) + 1] - win.lo;
                }   // NaN x: no target is in radius
            }
        }
        QueryResult r;
        match_query<GENERAL, true, -1>(P, mp, j, win, kp0, has0, s_queue, lane, a.dlen, scored, r, s_stage, VISO_OVF_STAGE);
        if (lane == 0) {
            bool accept = r.idx >= 0;
            if (accept && mp.second) accept = r.bd1 < r.bd2 * mp.ratio;
            P.res[P.q.sidx[j]] = make_int2(accept ? r.idx : -1, r.dist);
            if (scored) atomicAdd(P.scored, scored);
        }
        __builtin_amdgcn_wave_barrier();   // the wave's LDS queue is reused by its next entry
    }
}

// Which kernel takes the temporal problems of the u16 path (viso_ctx_set_matcher, per context):
//   6 = match_union8_kernel (match_union8.hip: the union pass on the rows' 8-bit planes, the two best scored exactly)  <- default
//   3 = match_union_kernel  (match_union.hip: rows gathered from L2, every row scored against eight queries on the u16 rows)
//   5 = match_prune_kernel  (match_prune.hip: exact successive elimination on block sums, cell-granular scorer)
//   4 = match_strip_kernel  (window rows resident in LDS, tools/experiments/match_strip.hip: 0.60 ms against 0.44 ms)
//   2 = match_batch_kernel<0> (rows gathered from L2, one pair per 8-lane group, match_batch.hip: 0.63 ms)
// 2 and 4 exist in -DVISO_DEBUG_VARIANTS builds only (make DEBUG_VARIANTS=1).
// The stereo problems always take match_stereo_kernel / match_batch_kernel<1>.  Same results from all of them (the parity
// tests run over viso_matcher_variants()).
const char* matcher_kernel_name(int variant) {
    return variant == 2 ? "match_batch_kernel<0>" : variant == 3 ? "match_union_kernel" : variant == 5 ? "match_prune_kernel" : variant == 6 ? "match_union8_kernel" : "match_strip_kernel";
}

// layout 0: problems in any order (both instantiations enumerate all of them);
// layout 1: the batch order [8 stereo][8 temporal-left][8 temporal-right] per 8 frames.
// e0/e1 (may be null) bracket the kernel that takes the temporal problems: the dominant kernel.
int launch_match_timed(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max,
                       int dlen, const MatchParamsDev mp[2], int* bad,
                       hipEvent_t e0, hipEvent_t e1, int layout, int variant,
                       const int2* ovf_q, const int* ovf_cnt, int r8s, int general_possible, int kinds) {
    if (n_probs <= 0 || cap_max <= 0) return VISO_OK;
    variant = matcher_effective(variant, dlen);
    MatchArgs a;
    a.probs = probs_dev;
    a.n_probs = n_probs;
    a.bpp = (cap_max + VISO_QPB - 1) / VISO_QPB;
    a.dlen = dlen;
    a.gs = 1; a.gf = 0; a.gc = 1;
    a.bad = bad;
    a.ovf_q = ovf_q; a.ovf_cnt = ovf_cnt;
    a.mp[0] = mp[0];
    a.mp[1] = mp[1];
    const int groups = (n_probs + 7) / 8;
    const long long blocks = (long long)groups * 8 * a.bpp;
    if (blocks > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    a.vblocks = (int)blocks;
    if (e0) HIP_TRY(hipEventRecord(e0, s));
    {
        const int r = launch_match_batch(s, probs_dev, n_probs, cap_max, mp, bad, layout, e1, variant, r8s, kinds);
        if (r < 0) return r;
    }
    // general (non-u16) path: normally idle (no image is flagged and every block leaves at once), so
    // it gets a small grid that strides over the (problem, tile) slots when it does have work
    const unsigned gblocks = (unsigned)(blocks < 512 ? blocks : 512);
    // A launch with both kinds of problems (every batch) takes ONE general kernel whose gate is decided per problem at run
    // time: the kernel is idle unless an image is flagged, and each idle launch is ~4.7 us of the step's chain
    if (general_possible && (kinds & VISO_KIND_ALL) == VISO_KIND_ALL) {
        hipLaunchKernelGGL((match_kernel<true, -1>), dim3(gblocks), dim3(VISO_MATCH_THREADS), 0, s, a);
        HIP_TRY(hipGetLastError());
    } else if (general_possible && (kinds & VISO_KIND_TEMPORAL)) {
        hipLaunchKernelGGL((match_kernel<true, 0>), dim3(gblocks), dim3(VISO_MATCH_THREADS), 0, s, a);
        HIP_TRY(hipGetLastError());
    } else if (general_possible && (kinds & VISO_KIND_STEREO)) {
        hipLaunchKernelGGL((match_kernel<true, 1>), dim3(gblocks), dim3(VISO_MATCH_THREADS), 0, s, a);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(match_overflow_kernel<false>, dim3(VISO_OVF_GRID), dim3(VISO_WAVE), 0, s, a);
    HIP_TRY(hipGetLastError());
    if (general_possible) {
        hipLaunchKernelGGL(match_overflow_kernel<true>, dim3(VISO_OVF_GRID), dim3(VISO_WAVE), 0, s, a);
        HIP_TRY(hipGetLastError());
    }
    return VISO_OK;
}

int launch_match(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max, int dlen,
                 const MatchParamsDev mp[2], int* bad, int variant, const int2* ovf_q, const int* ovf_cnt, int r8s) {
    return launch_match_timed(s, probs_dev, n_probs, cap_max, dlen, mp, bad, nullptr, nullptr, 0, variant, ovf_q, ovf_cnt, r8s);
}


// ------------------------------------------------------------------ sort
// std::sort(match, by [2]) of src/viso.cpp:724 with the documented total order
// (dist asc, i1 asc).  One workgroup per problem; keys (dist<<32 | i1) of the accepted queries in LDS (compacted:
// rejected queries carry no key).  Also emits pos[i1] (row of query i1, or -1) for the circle join, and the match
// count.
#define VISO_SORT_THREADS 512
#define VISO_SORT_NB 2048          // distance buckets of the fast path:
#define VISO_SORT_FINE 1024        //   equal-width ones over the bulk, then 32 x 32 log-linear ones for the tail
#define VISO_SORT_BMAX 128         // a fuller bucket sends the problem to the bitonic network
#define VISO_SORT_FAST_MAX 8192    // keys the fast path's LDS arrays are sized for

// Fast path: the keys are (dist << 32 | i1) with distinct i1.  Bucket the distances (power-of-two bucket width over
// the bulk of the distribution, log-linear buckets for the tail; LDS histogram with returning atomics, scan, scatter of 16-bit indices), then every
// key counts the smaller keys of ITS bucket (a handful): final row = bucket start + that count.  Five LDS round
// trips instead of the 66 stages of a 2048-key bitonic network.  Distances piled into one bucket (all equal, say)
// would make the counting quadratic: such problems (and those beyond VISO_SORT_FAST_MAX keys) take the network.
// tri (plain family's frames, or null): collect_matches / triangulate_rectified (src/viso.cpp:501-514, 1137-1162) of problem 0's
// list, row by row as the rows are written -- the thread that stores row r has i1 and i2 in hand: no kernel of its own
__device__ __forceinline__ void sort_tri_row(const TriItem& T, const SolverParamsDev& sp, int r, int i1, int i2) {
    const float2 a = T.kp1[i1], b = T.kp2[i2];
    const double uL = a.x, vL = a.y, uR = b.x, vR = b.y;
    T.x[0 * T.ld + r] = uL; T.x[1 * T.ld + r] = vL; T.x[2 * T.ld + r] = uR; T.x[3 * T.ld + r] = vR;
    if (T.X) {
        const double d = uL - uR;                       // :1148-1151, no clamp
        T.X[0 * T.ld + r] = sp.base * (uL - sp.cu) / d;
        T.X[1 * T.ld + r] = sp.base * (vL - sp.cv) / d;
        T.X[2 * T.ld + r] = sp.f * sp.base / d;
    }
}

#ifdef VISO_DEBUG_VARIANTS   // timing aid (tools/experiments/sortkp_phases.py): time stamps of sort_matches_kernel's phases, workgroup 1
__device__ unsigned long long viso_dbg_sortm_clk[12];
extern "C" int viso_debug_sortm_clocks(unsigned long long* out12) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out12, HIP_SYMBOL(viso_dbg_sortm_clk), sizeof(unsigned long long) * 12, 0, hipMemcpyDeviceToHost) == hipSuccess ? VISO_OK : VISO_ERR_HIP;
}
#define SM_CLK(I) do { if (blockIdx.x == 1 && threadIdx.x == 0) viso_dbg_sortm_clk[I] = wall_clock64(); } while (0)
#else
#define SM_CLK(I) do {} while (0)
#endif
__global__ __launch_bounds__(VISO_SORT_THREADS) void sort_matches_kernel(const MatchProblem* probs,
                                                                         int n_probs, int npad_alloc, int fast, int flagged_empty,
                                                                         const TriItem* tri, SolverParamsDev tri_sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ int s_cnt;
    __shared__ int s_start[VISO_SORT_NB + 1];
    __shared__ uint32_t s_red[2][VISO_SORT_THREADS / 64];
    __shared__ unsigned long long s_sum[VISO_SORT_THREADS / 64];
    __shared__ int s_maxb;
    const int prob = blockIdx.x;
    if (prob >= n_probs) return;
    SM_CLK(0);
    const MatchProblem P = probs[prob];
    const bool do_tri = tri != nullptr && prob == 0;   // uniform
    TriItem T{};
    if (do_tri) T = *tri;
    // a launch that left the general kernels out (plain.hip: every image so far fitted the u16 rows, the new ones are
    // expected to) has NO results for a problem with a flagged image: an empty list, the host sees the flag and repeats
    if (flagged_empty && (*P.q.bad | *P.t.bad) != 0) { if (threadIdx.x == 0) *P.m_cnt = 0; return; }
    const int n1 = *P.q.n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // compaction first: only the accepted queries carry a key (the ratio test rejects about half of the temporal ones)
    if (threadIdx.x == 0) s_cnt = 0;
    for (int b = threadIdx.x; b <= VISO_SORT_NB; b += VISO_SORT_THREADS) s_start[b] = 0;
    __syncthreads();
    uint32_t dmn = 0xffffffffu, dmx = 0u;
    unsigned long long dsum = 0;
    for (int base0 = 0; base0 < n1; base0 += 4 * VISO_SORT_THREADS) {
        int2 rv[4];   // four result loads in flight before the (serialising) LDS counter is touched
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base0 + u * VISO_SORT_THREADS + threadIdx.x;
            rv[u] = i < n1 ? P.res[i] : make_int2(-1, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base0 + u * VISO_SORT_THREADS + threadIdx.x;
            if (base0 + u * VISO_SORT_THREADS >= n1) break;   // uniform
            unsigned long long k = ~0ull;
            if (i < n1) {
                if (rv[u].x >= 0) k = ((unsigned long long)(uint32_t)rv[u].y << 32) | (uint32_t)i;
                else P.pos[i] = -1;   // accepted queries get their row below
            }
            const bool valid = k != ~0ull;
            if (valid) { dmn = min(dmn, (uint32_t)(k >> 32)); dmx = max(dmx, (uint32_t)(k >> 32)); dsum += k >> 32; }
            const unsigned long long m = __ballot(valid);
            int wbase = 0;
            if (lane == 0 && m) wbase = atomicAdd(&s_cnt, __popcll(m));
            wbase = __shfl(wbase, 0);
            if (valid) keys[wbase + mbcnt(m)] = k;
        }
    }
    SM_CLK(1);
    dmn = viso_wave_min63(dmn); dmx = viso_wave_max63(dmx); dsum = viso_wave_sum63(dsum);
    if (lane == 63) { s_red[0][wv] = dmn; s_red[1][wv] = dmx; s_sum[wv] = dsum; }
    __syncthreads();
    const int mv = s_cnt;
    if (threadIdx.x == 0) *P.m_cnt = mv;
    if (mv == 0) return;
    bool use_fast = fast != 0;
    if (use_fast) {
        unsigned long long* bkeys = keys + npad_alloc;                     // the keys in bucket order
        dsum = 0;
#pragma unroll
        for (int w = 0; w < VISO_SORT_THREADS / 64; ++w) { dmn = min(dmn, s_red[0][w]); dmx = max(dmx, s_red[1][w]); dsum += s_sum[w]; }
        // Bucket map, monotone in the distance: the accepted matches sit in a narrow peak with a few far outliers, so
        // VISO_SORT_FINE buckets of one power-of-two width cover [dmn, dmn + 2 (mean - dmn)] (at most [dmn, dmx]) and
        // the tail behind them goes to log-linear buckets (32 per octave).
        SM_CLK(2);
        unsigned long long mean_off = dsum / (unsigned long long)mv - dmn;
        for (int trim = 0; trim < 2; ++trim) {   // the mean of what lies within twice the mean, twice: outliers drop out
            unsigned long long ts = 0;
            unsigned int tn = 0;
            for (int e = threadIdx.x; e < mv; e += VISO_SORT_THREADS) {
                const unsigned long long off = (keys[e] >> 32) - dmn;
                if (off <= 2 * mean_off) { ts += off; ++tn; }
            }
            ts = viso_wave_sum63(ts); tn = viso_wave_scan(tn);
            __syncthreads();
            if (lane == 63) { s_sum[wv] = ts; s_red[0][wv] = tn; }
            __syncthreads();
            ts = 0; tn = 0;
#pragma unroll
            for (int w = 0; w < VISO_SORT_THREADS / 64; ++w) { ts += s_sum[w]; tn += s_red[0][w]; }
            mean_off = tn ? ts / tn : 0;
        }
        SM_CLK(3);
        const unsigned long long span = min((unsigned long long)(dmx - dmn), 2 * mean_off + 1);
        int sh = 0;
        while (sh < 32 && (span >> sh) >= (unsigned long long)VISO_SORT_FINE) ++sh;
        auto bucket_of_dist = [&](uint32_t d) -> int {
            const uint32_t off = d - dmn, f = off >> sh;
            if (f < (uint32_t)VISO_SORT_FINE) return (int)f;
            const uint32_t t = off - ((uint32_t)VISO_SORT_FINE << sh) + 1u;     // >= 1
            const int lg = 31 - __clz((int)t);
            const uint32_t frac = lg >= 5 ? (t >> (lg - 5)) & 31u : (t << (5 - lg)) & 31u;
            return VISO_SORT_FINE + lg * 32 + (int)frac;                         // < VISO_SORT_FINE + 1024
        };
#define SORT_BUCKET(D) bucket_of_dist((uint32_t)(D))
        for (int e = threadIdx.x; e < mv; e += VISO_SORT_THREADS) atomicAdd(&s_start[SORT_BUCKET(keys[e] >> 32)], 1);
        __syncthreads();
        SM_CLK(4);
        if (wv == 0) {   // exclusive scan of the bucket counts (32 per lane + wave scan), largest bucket
            int c[VISO_SORT_NB / 64], tot = 0, big = 0;
#pragma unroll
            for (int q = 0; q < VISO_SORT_NB / 64; ++q) { c[q] = s_start[lane * (VISO_SORT_NB / 64) + q]; tot += c[q]; big = max(big, c[q]); }
            const int incl = (int)viso_wave_scan((uint32_t)tot);
            big = (int)viso_wave_max63((uint32_t)big);
            int run = incl - tot;
#pragma unroll
            for (int q = 0; q < VISO_SORT_NB / 64; ++q) { s_start[lane * (VISO_SORT_NB / 64) + q] = run; run += c[q]; }
            if (lane == 63) s_maxb = big;
        }
        __syncthreads();
        use_fast = s_maxb <= VISO_SORT_BMAX;
        SM_CLK(5);
        if (use_fast) {
            // scatter into bucket order: the running offset of bucket b ends up at the bucket's END = start of b + 1
            for (int e = threadIdx.x; e < mv; e += VISO_SORT_THREADS) {
                const unsigned long long k = keys[e];
                bkeys[atomicAdd(&s_start[SORT_BUCKET(k >> 32)], 1)] = k;
            }
            __syncthreads();
            SM_CLK(6);
            for (int p = threadIdx.x; p < mv; p += VISO_SORT_THREADS) {
                const unsigned long long k = bkeys[p];
                const int b = SORT_BUCKET(k >> 32);
                const int s0 = b ? s_start[b - 1] : 0, s1 = s_start[b];
                int r = s0;
                int q = s0;
                for (; q + 4 <= s1; q += 4) {   // independent LDS reads: four in flight
                    const unsigned long long k0 = bkeys[q], k1 = bkeys[q + 1], k2 = bkeys[q + 2], k3 = bkeys[q + 3];
                    r += (k0 < k ? 1 : 0) + (k1 < k ? 1 : 0) + (k2 < k ? 1 : 0) + (k3 < k ? 1 : 0);
                }
                for (; q < s1; ++q) r += bkeys[q] < k ? 1 : 0;
                const int i1 = (int)(uint32_t)k;
                const int2 rr = P.res[i1];
                P.sorted[3 * r + 0] = i1;
                P.sorted[3 * r + 1] = rr.x;
                P.sorted[3 * r + 2] = (int)(uint32_t)(k >> 32);
                P.pos[i1] = r;
                if (do_tri) sort_tri_row(T, tri_sp, r, i1, rr.x);
            }
            SM_CLK(7);
            return;
        }
#undef SORT_BUCKET
    }
    // the network: keys padded with ~0 to a power of two
    int npad = 64;
    while (npad < mv) npad <<= 1;
    for (int i = mv + threadIdx.x; i < npad; i += VISO_SORT_THREADS) keys[i] = ~0ull;
    __syncthreads();
    bitonic_sort_lds<VISO_SORT_THREADS>(keys, npad);
    for (int r = threadIdx.x; r < mv; r += VISO_SORT_THREADS) {
        const unsigned long long k = keys[r];
        const int i1 = (int)(uint32_t)k;
        const int2 rr = P.res[i1];
        P.sorted[3 * r + 0] = i1;
        P.sorted[3 * r + 1] = rr.x;
        P.sorted[3 * r + 2] = (int)(uint32_t)(k >> 32);
        P.pos[i1] = r;
        if (do_tri) sort_tri_row(T, tri_sp, r, i1, rr.x);
    }
}

int launch_sort(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max, int flagged_empty, const TriItem* tri, const SolverParamsDev* tri_sp) {
    if (n_probs <= 0) return VISO_OK;
    if (cap_max > VISO_SORT_MAX) {
        viso_set_error("match_desc: more than %d queries per call is not supported by this build", VISO_SORT_MAX);
        return VISO_ERR_UNSUPPORTED;
    }
    int npad = 64;
    while (npad < cap_max) npad <<= 1;
    const int fast = npad <= VISO_SORT_FAST_MAX ? 1 : 0;
    const size_t lds = (size_t)npad * sizeof(unsigned long long) * (fast ? 2 : 1) + 16;   // keys (+ the keys in bucket order)
    if (lds > 40 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)sort_matches_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SolverParamsDev sp{};
    if (tri && tri_sp) sp = *tri_sp;
    hipLaunchKernelGGL(sort_matches_kernel, dim3(n_probs), dim3(VISO_SORT_THREADS), lds, s, probs_dev, n_probs, npad, fast, flagged_empty,
                       tri_sp ? tri : nullptr, sp);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

void fill_match_params(MatchParamsDev* d, const viso_match_params* h) {
    d->epi = h->enforce_epipolar != 0;
    d->second = h->enforce_2nd_best != 0;
    d->K = h->max_neighbors;
    d->_pad = 0;
    d->radius = (float)h->radius;   // src/viso.cpp:685: double -> float parameter
    if (d->radius == 0.f) d->radius = 0.f;   // -0.0 -> +0.0: the kernels compare bit patterns of distances with the radius'
    d->_padf = 0;
    for (int i = 0; i < 9; ++i) d->F[i] = h->F[i];
    d->sampson_thresh = h->sampson_thresh;
    d->ratio = h->ratio_2nd_best;
}
