// common.h — shared declarations of libviso_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/viso_hip.h"

#define VISO_ROW 128          // packed descriptor row: 128 x u16 = 256 B (121 used, rest = bias)
#define VISO_ROW8 128         // bytes of a row's 8-bit plane (ImageView::rows8)
// The planes' element is h_s(v) = clamp((v + (128 << s)) >> s, 0, 255) with a shift s in 0..3 chosen per RUN from the data:
// every 16th run of a batch (and its first) the pack kernels count, over a sample of rows, how many element pairs reach
// 128 / 256 / 512 (VISO_R8_*), the counts come back through a pinned buffer behind an event, and the host gives the runs
// after that the smallest s that clamps at most 1 pair in 256 — as a kernel argument.  (SAD8 << s) - VISO_ROW8_SLACK(s) <=
// SAD for ANY descriptors and any s: a wrong s costs time (more candidates scored exactly), never a result.
#define VISO_ROW8_SLACK(S) (((1 << (S)) - 1) * VISO_ROW8)   // at most 128 elements count, each off by at most 2^s - 1
#define VISO_R8_C128 0        // int cnt[4] per batch (8-byte aligned): [0..2] sampled element pairs (lanes of the pack kernels) whose
#define VISO_R8_ROWS 3        //   larger magnitude reaches 128 / 256 / 512, [3] sampled rows
#define VISO_R8_DEFAULT 3     // shift before any statistics exist (covers the whole range of a 3x3 Sobel of uint8)
#define VISO_R8_EVERY 16      // runs between two looks at the data
#define VISO_PACK_SUMS 1      // pack kernels, `extras`: also write ImageView::sums (matcher variant 5)
#define VISO_PACK_ROWS8 2     //                         also write ImageView::rows8 (matcher variant 6)
#define VISO_BIAS 32768       // u16 = int16 value + 32768 (SAD is translation invariant)
#define VISO_WAVE 64
#define VISO_QCAP 256         // per-wave candidate queue entries
#define VISO_QPB 32           // queries per workgroup in the matcher
#define VISO_MATCH_THREADS 256
#define VISO_KPCAP 768        // target-window keypoints staged in LDS per query tile
#define VISO_NB 256           // column buckets of the per-image x index
#define VISO_SORT_MAX 16384   // keypoints per image / queries per call (LDS bitonic sorts)

void viso_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            viso_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,           \
                           hipGetErrorString(_e));                                 \
            return VISO_ERR_HIP;                                                   \
        }                                                                          \
    } while (0)

// ---- device-side views -----------------------------------------------------
struct MatchParamsDev {          // viso_match_params, device copy (kernarg)
    int epi, second, K, _pad;
    float radius, _padf;
    double F[9];
    double sampson_thresh, ratio;
};

struct ImageView {               // one image's keypoints + descriptors on the device
    const float2* kp;            // [n] boundary order (x,y)
    const float* frows;          // [n][dlen] boundary-layout descriptors (original order)
    const int* n;                // keypoint count (device, ragged batches)
    float2* skp;                 // [n] keypoints grouped by column bucket (counting sort by bucket_of(x): 256 buckets over
                                 //     the image's x range).  NO order inside a bucket (scattered with LDS atomics), and
                                 //     keypoints with a NaN x share the last bucket with the largest columns.
    int* sidx;                   // [n] bucket-order position -> original index
    int* rank;                   // [n] original index -> bucket-order position
    int* bstart;                 // [VISO_NB+1] first sorted position of each column bucket
    float* xinfo;                // [8] x0, scale of the column bucket map; smallest and largest finite keypoint y;
                                 //     [4] number of keypoints whose x is not NaN, as a float: a COUNT only, not a prefix
                                 //     length (NaN-x keypoints sit anywhere inside the last bucket)
    uint8_t* qord;               // [n rounded up to 64] y order inside every block of 64 bucket-order positions: entry B*64 + r =
                                 //     offset (0..63) in block B of the keypoint with y rank r (positions past n rank last)
    uint16_t* rows;              // [n][128] packed descriptor rows, bucket order
    uint2* sums;                 // [n] bucket order: sums of the row's four 32-element blocks, each clamped to int16 and
                                 //     biased like the elements (4 x u16).  sum_k |S_q,k - S_t,k| <= SAD(q, t) (triangle
                                 //     inequality per block; clamping is 1-Lipschitz): the lower bound match_prune_kernel
                                 //     prunes candidates with.  Written by the pack kernels together with the rows.
    uint8_t* rows8;              // [n][128] bucket order: the rows' 8-bit plane, element h_s(v) = clamp((v + (128 << s)) >> s, 0, 255)
                                 //     (pad = h_s(0)), s = the run's shift (a kernel argument, see VISO_R8_*).  2^s |h(a) - h(b)| - (2^s - 1) <= |a - b|
                                 //     for ANY a, b (floor and clamp are monotone and 1-Lipschitz in units of 2^s), so
                                 //     (SAD8 << s) - VISO_ROW8_SLACK(s) <= SAD: the lower bound match_union8_kernel ranks candidates
                                 //     with.  Written by the pack kernels when matcher variant 6 is selected.
    int* bad;                    // [1] != 0: this image's descriptors do not fit the u16 rows (pack_desc_kernel);
                                 //     every problem that reads the image then takes the general (double) kernel
};

struct MatchProblem {            // one match_desc call (reference src/viso.cpp:669)
    ImageView q, t;              // queries (kp1,d1) and targets (kp2,d2)
    int2* res;                   // per ORIGINAL query index: (accepted target or -1, (int)best SAD)
    int* sorted;                 // out: M x 3 (i1,i2,dist) sorted by (dist,i1)
    int* pos;                    // out: per query, row in `sorted` or -1
    int* m_cnt;                  // out: M
    unsigned long long* scored;  // out: number of SAD evaluations (C of SURVEY 8(d))
    int2* ovf;                   // scratch shared by ALL problems of a launch: (problem, sorted position) of every query
    int* ovf_cnt;                //     left to the overflow kernel, and their count (zeroed before every run) — one
                                 //     queue, so that the overflow kernel's waves share the work evenly whatever
                                 //     problem it comes from (dense keypoint clusters concentrate it in a few problems)
    int* tile_flag;              // scratch, stereo call: per 64-query tile, 1 = match_batch_kernel<1> does the tile
                                 //     (written by match_stereo_kernel for every tile, read by the kernel behind it)
    int pidx;                    // 0 = stereo params, 1 = temporal params
    int cap;                     // row capacity of res/sorted/pos
};

struct SolverParamsDev {         // viso_param, device copy
    double base, f, cu, cv, inlier_threshold, thresh;
    int ransac_iter, _pad;
};

struct viso_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int matcher_variant;         // viso_ctx_set_matcher
    int row8_force;              // viso_ctx_set_row8_shift: -1 = chosen from the data (default), 0..3 = fixed
    int gn_split;                // viso_ctx_set_gn_split
    // second stream of the context: the RANSAC stage of its batches runs here, beside the next run's
    // matcher on `stream` (viso_ctx_synchronize waits for both).  One per CONTEXT, not per batch: the runtime maps
    // streams onto a handful of hardware queues, and two busy RANSAC streams that land on one queue serialise
    // (measured: 9 streams for 3 busy batches -> two chains on one queue, 0.43 -> 0.68 ms per step).
    hipStream_t solver_stream;
    // grow-only scratch for the plain (host-pointer) family
    void* scratch[24];
    size_t scratch_bytes[24];
    // pinned staging of the plain family (grow-only): [0] a call's inputs, packed back to back and sent with ONE
    // host-to-device copy; [1] a call's results, fetched with ONE device-to-host copy behind ONE synchronize
    char* pin[2];
    size_t pin_bytes[2];
    struct PlainCache* plain;    // the plain family's image cache (plain.hip), created on first use
    // completion signal of the plain family's calls (PlainSignal below): a word of pinned memory the call's LAST kernel writes,
    // a device counter of that kernel's finished workgroups, the sequence number of the last signal asked for
    int* sig_flag; int* sig_ctr; int sig_seq;
};

// handle registry (ctx.hip): the live contexts, the live batches of each, tombstones of batches their context took along
struct viso_batch;
bool viso_ctx_live(const viso_ctx* c);
bool viso_batch_live(const viso_batch* b);
bool viso_batch_register(viso_ctx* c, viso_batch* b);   // false: the context is not live
int viso_batch_unregister(viso_batch* b);               // 1 live (now the caller's to free), 0 tombstone (consumed), -1 unknown
int viso_batch_free(viso_batch* b, bool keep_shell);    // batch.hip: frees a batch that has left the registry (keep_shell: the host object stays for the caller's late destroy)

struct PlainLock { PlainLock(); ~PlainLock(); };   // serialises the plain family on the default context
// viso_plain_profile: phases of one plain-family call, bracketed by four events on the call's stream (ctx.hip).  Used
// inside a PlainLock.  mark(1) = inputs enqueued, mark(2) = kernels enqueued, mark(3) = results on the host.
struct PlainProf {
    PlainProf(int fn, hipStream_t s);
    ~PlainProf();
    void mark(int k);
    void wait_begin();
    void wait_end();
    int fn; hipStream_t s; bool on; int marks; double t0, tw, wait;
};
int ctx_scratch(viso_ctx* c, int slot, size_t bytes, void** out, bool zero_new = false);   // zero_new: a block that is (re)allocated starts zeroed
int ctx_pinned(viso_ctx* c, int which, size_t bytes, char** out);
void plain_cache_free(viso_ctx* c);   // plain.hip; from viso_ctx_destroy
// plain.hip: a call of the reference's loop that the frame's stereo call has already answered (1 = served, 0 = go to the device)
int plain_try_collect(viso_ctx* c, const float* kp1, int n1, const float* kp2, int n2, const int32_t* match, int n, double* x);
int plain_try_triangulate(viso_ctx* c, const double* x, int m, const viso_param* p, double* X);
int plain_try_circle(viso_ctx* c, const int32_t* lr, int n_lr, const int32_t* lr_prev, int n_lrp, const int32_t* m11, int n11,
                     const int32_t* m22, int n22, int32_t* circ, int32_t* pcl, int cap, int* out_n, int* ret);
void plain_note_circle(viso_ctx* c, int cnt);
int plain_try_ransac(viso_ctx* c, const double* X, const double* obs, int m, double best_tr[6], int32_t* best_inl, int* n_inl,
                     const viso_param* p, const int32_t* samples, uint64_t seed, uint64_t frame, int* ret);

struct PlainSignal { int* ctr; int* flag; int seq; };   // a call's completion signal (described below); flag == nullptr: no signal
struct OutArgs;                                          // a copy-out into the call's pinned mirror (below)
// match_circle on lists in device memory (circle.hip): counts by value, or read from the device when the pointers are set
struct CircleArgs {
    const int* lr; const int* lrp; const int* m11; const int* m22;
    int n_lr, n_lrp, n11, n22;
    const int* n_lr_p; const int* n_lrp_p; const int* n11_p; const int* n22_p;   // device counts (override the values), or null
    int* rows;      // out: 6 ints per joined row: circ_match (ileft, iright, ileft_prev, iright_prev) | match_pcl (i, k)
    int cap; int* out_n;
    // optional (the plain family's frames; g_x == null: no gather): x_c / Xp_c of src/viso.cpp:1292-1305 -- columns of this frame's x
    // and of the previous frame's X picked by match_pcl -- written by the kernel's tail instead of a kernel of their own
    const double* g_x; const double* g_Xp; double* g_xc; double* g_Xpc; int g_ldx, g_ldXp, g_ldc;
    // optional (plain family): a copy-out of the kernel BEFORE this one rides in the launch as ride_blocks extra workgroups
    // (OutArgs in device memory; plain_out_blocks)
    const OutArgs* ride; int ride_blocks;
};
// tab: 3 * tabn ints of device scratch; keys outside [0, tabn) or duplicate keys take the literal nested loops
int launch_circle_table(hipStream_t s, const CircleArgs& a, int* tab, int tabn);
viso_ctx* viso_default_ctx();

// One plain-family call's inputs: appended to the context's pinned block (256-B aligned pieces), mirrored at the same
// offsets in one device block, sent with ONE hipMemcpyAsync.  A pageable hipMemcpyAsync costs 5-10 us of host time
// per call whatever its size; the plain family used to issue 4-7 of them per function (tools/h2d_probe.hip).
#define PLAIN_SLOT_IN 14      // ctx_scratch slots of the staging blocks
#define PLAIN_SLOT_OUT 15
struct PlainStage {
    char* h = nullptr; char* d = nullptr; size_t off = 0, cap = 0;
    static size_t need(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    int begin(viso_ctx* c, size_t bytes) {
        cap = bytes; off = 0;
        int r = ctx_pinned(c, 0, bytes, &h);
        if (r < 0) return r;
        return ctx_scratch(c, PLAIN_SLOT_IN, bytes, (void**)&d);
    }
    template <class T> T* put(const T* src, size_t count) {   // returns the DEVICE address
        const size_t b = sizeof(T) * count;
        if (b) __builtin_memcpy(h + off, src, b);
        T* r = reinterpret_cast<T*>(d + off);
        off += need(b);
        return r;
    }
    template <class T> T* host_of(const T* dev) const { return reinterpret_cast<T*>(h + (reinterpret_cast<const char*>(dev) - d)); }
    // Small blocks go through a copy KERNEL that reads the pinned block over PCIe itself: the copy engine costs ~10 us
    // of latency per transfer whatever its size, a launch 2-3 (tools/h2d_probe.hip: H2D + kernel + D2H + synchronize
    // 29 us by copy engine, 18 us by copy kernels).
    int flush(hipStream_t s);
};
// words = 32-bit units.  dst[0..head) = src[0..head), then min(*n_rows, max_rows) rows of row_words behind them (n_rows
// may be null: head only).  Either side may be pinned host memory.
// A plain-family call returns when its results are in pinned host memory.  hipStreamSynchronize sees that ~5 us after the
// fact (an empty kernel + synchronize: 10.9 us; the same kernel writing a word of pinned memory that the host spins on:
// 6.2 us -- tools/h2d_probe.hip).  So the call's LAST kernel -- a copy kernel into pinned memory -- signals itself: every thread
// fences its stores at system scope, the last workgroup to finish (a device counter) release-stores the call's sequence
// number into a pinned word, and the host spins on that word with an acquire load (plain_signal_wait: bounded, then
// hipStreamSynchronize as the fallback, so a lost signal costs time, never a hang or a result).
int plain_signal_next(viso_ctx* c, PlainSignal* out);   // the next sequence number of the context (allocates on first use)
int plain_signal_wait(viso_ctx* c, hipStream_t s, int seq);
int plain_blit(hipStream_t s, const void* src, void* dst, size_t head_words, const int* n_rows = nullptr, int row_words = 0, int max_rows = 0,
               const PlainSignal* sig = nullptr);
#ifdef __HIPCC__
// Wave-wide steps as DPP operands (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast:15 / :31): the total -- or the
// inclusive prefix -- is in lane 63 after six dependent VALU instructions, where six ds_bpermute round trips (__shfl_xor /
// __shfl_up) take about ten times as long.  The per-image, per-tile and per-problem kernels are chains of such steps between
// their loads; in the latency-bound ones (match_stereo_kernel, the sorts) and for ONE frame (the per-call path) those chains
// are the kernel's duration.  Lanes without a source take the identity.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t viso_dpp(uint32_t v, uint32_t ident) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROWMASK, 0xf, false);
}
#define VISO_WAVE_STEPS(OP) OP(0x111, 0xf); OP(0x112, 0xf); OP(0x114, 0xf); OP(0x118, 0xf); OP(0x142, 0xa); OP(0x143, 0xc)
__device__ __forceinline__ uint32_t viso_wave_min63(uint32_t v) {   // valid in lane 63
#define OP_(C, M) v = min(v, viso_dpp<C, M>(v, 0xffffffffu))
    VISO_WAVE_STEPS(OP_);
#undef OP_
    return v;
}
__device__ __forceinline__ uint32_t viso_wave_max63(uint32_t v) {   // valid in lane 63
#define OP_(C, M) v = max(v, viso_dpp<C, M>(v, 0u))
    VISO_WAVE_STEPS(OP_);
#undef OP_
    return v;
}
__device__ __forceinline__ uint32_t viso_wave_scan(uint32_t v) {    // inclusive prefix sum (lane 63: the total)
#define OP_(C, M) v += viso_dpp<C, M>(v, 0u)
    VISO_WAVE_STEPS(OP_);
#undef OP_
    return v;
}
__device__ __forceinline__ unsigned long long viso_wave_sum63(unsigned long long v) {   // valid in lane 63
#define OP_(C, M) v += ((unsigned long long)viso_dpp<C, M>((uint32_t)(v >> 32), 0u) << 32) | viso_dpp<C, M>((uint32_t)v, 0u)
    VISO_WAVE_STEPS(OP_);
#undef OP_
    return v;
}
__device__ __forceinline__ unsigned long long viso_wave_max63(unsigned long long v) {   // valid in lane 63
#define OP_(C, M) do { const unsigned long long o_ = ((unsigned long long)viso_dpp<C, M>((uint32_t)(v >> 32), 0u) << 32) | viso_dpp<C, M>((uint32_t)v, 0u); v = o_ > v ? o_ : v; } while (0)
    VISO_WAVE_STEPS(OP_);
#undef OP_
    return v;
}
__device__ __forceinline__ float viso_wave_fsum63(float v) {        // valid in lane 63 (the order of the additions is this function's)
#define OP_(C, M) v += __uint_as_float(viso_dpp<C, M>(__float_as_uint(v), 0u))
    VISO_WAVE_STEPS(OP_);
#undef OP_
    return v;
}
// minimum / maximum of a float over the wave, to EVERY lane: quad, half-row and row exchanges as DPP operands, then the four rows'
// values as scalars (fminf / fmaxf ignore a NaN operand: the same result as any other order of the same operations)
template <bool MAX>
__device__ __forceinline__ float viso_wave_fext(float v) {
#define OP_(C) do { const float o_ = __uint_as_float(viso_dpp<C, 0xf>(__float_as_uint(v), 0u)); v = MAX ? fmaxf(v, o_) : fminf(v, o_); } while (0)
    OP_(0xB1); OP_(0x4E); OP_(0x141); OP_(0x140);   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
#undef OP_
    const float a = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0));
    const float b = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
    const float c = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32));
    const float d = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 48));
    return MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : fminf(fminf(a, b), fminf(c, d));
}
// Called by EVERY thread of the kernel's every workgroup, behind its last store.
__device__ __forceinline__ void plain_signal_done(const PlainSignal& g, unsigned nblocks) {
    if (!g.flag) return;                       // uniform
    __threadfence_system();                    // this thread's stores (to pinned host memory) are out, system scope
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(g.ctr, 1) == (int)nblocks - 1) {   // the last workgroup of the launch: every other one has fenced and counted
            __threadfence_system();                      // (acquire side of the counter: what the others fenced is ordered before the flag)
            *g.ctr = 0;                                  // for the next signalling kernel (streams run them one after the other)
            __hip_atomic_store(g.flag, g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif
// ---- results out: every region's rows that exist (counts on the device) into the call's pinned mirror ----------------------
// A workgroup's path to host memory is narrow (a single one needs 15 us for what 39 spread over the chip write in 4), so the
// copy is many small workgroups -- and where the chain has a next kernel they RIDE in its launch as extra workgroups (the
// copy-out of the sort kernel's lists in the join kernel's launch, the join's in ransac_coop_kernel's): they depend on the
// kernel before, not on the one they ride in, and the 8.5 us + launch gap of a copy kernel of their own leave the chain.
// The last kernel of the chain, ransac_refit_kernel, writes its few hundred bytes itself (RefitMirror).
struct OutRegion { const uint32_t* src; uint32_t* dst; const int* cnt; int row_words, max_rows; };
#define OUT_REGIONS 16
struct OutArgs { OutRegion r[OUT_REGIONS]; int n_regions, gx; PlainSignal sig; };   // gx workgroups of 256 threads' worth per region
#ifdef __HIPCC__
// workgroup `b` of n_regions * gx (any workgroup size; every thread calls)
__device__ __forceinline__ void plain_out_blocks(const OutArgs& a, unsigned b) {
    const OutRegion R = a.r[b / (unsigned)a.gx];
    const unsigned bx = b % (unsigned)a.gx;
    int n = R.max_rows;
    if (R.cnt) { const int c = *R.cnt; n = c < 0 ? 0 : c < n ? c : n; }
    const unsigned total = (unsigned)n * (unsigned)R.row_words;
    for (unsigned i = bx * blockDim.x + threadIdx.x; i < total; i += (unsigned)a.gx * blockDim.x) R.dst[i] = R.src[i];
    plain_signal_done(a.sig, (unsigned)(a.n_regions * a.gx));
}
#endif
struct RefitMirror {                 // ransac_refit_kernel, n_items == 1: the result block and the inliers that exist, copied by the workgroup, which signals
    const uint32_t* res_src; uint32_t* res_dst; int res_words;
    const int* n_inl; const uint32_t* inl_src; uint32_t* inl_dst; int max_inl;
    PlainSignal sig;
};

// ---- launchers (host) -------------------------------------------------------
// group every image's keypoints by column bucket (+ inverse permutation, column index, y order inside 64-blocks)
// zero_words / n_zero: optional 32-bit words the kernel zeroes on the way (a run's counters: one memset less in front of it)
// r8zero: the batch's four VISO_R8_* counters, zeroed too when the run is one that counts (or null)
// imp (plain family, n_img == n_imp <= 2, imgs_dev unused): the images come from pinned HOST memory -- an image's view rides in the kernel arguments, the
// kernel reads the keypoints over PCIe, leaves them in view.kp, writes *view.n = n, *view.bad = bad0 and a copy of the view
// at view_dst (where the pack kernel's launch finds it): the copy kernel that used to run in front is gone
struct KpImport { const float2* src_kp; int n, bad0; ImageView* view_dst; ImageView view; };
int launch_sort_kp(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap_max, uint32_t* zero_words = nullptr, int n_zero = 0,
                   int* r8zero = nullptr, const KpImport* imp = nullptr, int n_imp = 0);
// pack boundary-layout float descriptors into biased u16 rows in bucket order;
// sets the image's own flag (ImageView::bad) and *bad_any when a value is not an integer in [-32768, 32767];
// dlen > 128 (rows do not fit) flags every image.  bad_img: the n_img flags, contiguous (for that case).
// extras: VISO_PACK_SUMS / VISO_PACK_ROWS8 — what the selected matcher variant reads beside the u16 rows (pack_extras)
int launch_pack(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap, int dlen, int* bad_img, int* bad_any, int extras, int r8s, int* r8cnt);
// match_union8_kernel keeps a cell's SAD8 >> 7 in a byte whose value 255 marks "scored exactly": 127 elements x 255 = 32385
// stays below that, 128 x 255 does not.  Descriptors of 128 elements (the reference's have 121) take match_union_kernel
static inline int matcher_effective(int variant, int dlen) { return (variant == 6 && dlen > 127) ? 3 : variant; }
static inline int pack_extras(int variant, int dlen) { variant = matcher_effective(variant, dlen); return variant == 5 ? VISO_PACK_SUMS : variant == 6 ? VISO_PACK_ROWS8 : 0; }
// the same from int16 descriptors [n_img][cap][dlen] (viso_batch_upload_i16*): never flags anything
int launch_pack_i16(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap, int dlen, const int16_t* desc16, int extras, int r8s, int* r8cnt);
// bad: int[2] zeroed before the run ([0] any image flagged by the pack kernel, [1] scratch counter of the stereo kernels)
int launch_match(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max, int dlen,
                 const MatchParamsDev mp[2], int* bad, int variant, const int2* ovf_q, const int* ovf_cnt, int r8s);
// kinds: which problems the launch can contain (the plain family knows: one call = one problem) -- kernels that would
// find nothing to do are not launched.  VISO_KIND_TEMPORAL: problems without the epipolar gate, VISO_KIND_STEREO: with it.
#define VISO_KIND_TEMPORAL 1
#define VISO_KIND_STEREO 2
#define VISO_KIND_ALL 3
#define VISO_KIND_NO_WIDE 4      // (plain family) leave match_batch_kernel<1> out: the launcher expects no tile with a wide epipolar band --
                                 // the count of declined tiles comes back with the results, and the call is repeated if there was one
// general_possible = 0: the rows cannot be flagged (descriptors extracted on the device from uint8 images), the
// kernels of the general (double) path are not even launched
int launch_match_timed(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max, int dlen,
                       const MatchParamsDev mp[2], int* bad, hipEvent_t e0, hipEvent_t e1, int layout, int variant,
                       const int2* ovf_q, const int* ovf_cnt, int r8s, int general_possible = 1, int kinds = VISO_KIND_ALL);
const char* matcher_kernel_name(int variant);
#define VISO_MATCHER_DEFAULT 6
struct TriItem;
// tri / tri_sp (plain family, or null): collect_matches + triangulate_rectified of problem 0's list inside the sort kernel
int launch_sort(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max, int flagged_empty = 0,
                const TriItem* tri = nullptr, const SolverParamsDev* tri_sp = nullptr);
void fill_match_params(MatchParamsDev* d, const viso_match_params* h);
void fill_solver_params(SolverParamsDev* d, const viso_param* h);


// ---- solver / circle / triangulation items (device-resident descriptors) ---
struct SolverItem {            // one ransac_minimize_reproj problem (one frame)
    const double* X;           // 3 x ld  previous-frame 3-D points
    const double* obs;         // 4 x ld  observations (uL,vL,uR,vR)
    const int* m_ptr;          // number of points (device)
    int ld;
    int _pad;
    const int* samples;        // iters x 3, or NULL -> splitmix64 stream
    int* samp_h;               // iters x 3   the triples in use (ransac_hyp_kernel: drawn, or copied from `samples`)
    unsigned long long frame;  // stream key
    double* tr_h;              // iters x 6   hypothesis transforms
    int* ok_h;                 // iters
    int* cnt_h;                // iters
    char* rot;                 // viso_rot_bytes(iters): the hypotheses' rotations for the counting kernel (ransac_rot_kernel)
    double* tr;                // 6  in/out (best_tr)
    int* ok;                   // 1
    int* n_inl;                // 1
    int* inl;                  // up to m
    int* kept;                 // null, or 1 word: set to 1 when the stage leaves best_tr as the caller passed it -- no hypothesis found any
                               //     support (best_tr is assigned only on improvement, src/viso.cpp:1564-1568) or m < 3 -- and `tr` is then
                               //     NOT written; 0 otherwise.  Null (the batch family: sequence_odometry passes zeros, :1312): `tr` gets zeros.
};

struct JoinItem {
    const int* lr; const int* lr_cnt;        // sorted stereo matches of frame t
    const int2* res11;                       // temporal-left results, per query
    const int2* res22;                       // temporal-right results, per query
    const int* pos_lrp; const int2* res_lrp; // previous frame's stereo: row of query / result
    const float2* kp1; const float2* kp2;    // keypoints of frame t (left, right): x_c is collect_matches of the joined rows
    const float2* kp1p; const float2* kp2p;  // keypoints of frame t-1: Xp_c is triangulate_rectified of the joined rows,
                                             // computed in the join itself (src/viso.cpp:501-514, 1137-1162, 1292-1305)
    int* circ; int* pcl; int* mc;            // outputs
    double* x_c; double* Xp_c; int ldc;      // 4 x ldc, 3 x ldc
};

struct TriItem {
    const float2* kp1; const float2* kp2;
    const int* match; const int* m_cnt;   // n x 3 (i1,i2,dist)
    double* x; double* X; int ld;         // 4 x ld, 3 x ld (X may be NULL)
};

// queue: device scratch of 2 + n_items * iters ints (list of the hypotheses stage 1 leaves undecided); queue[0] must be ZERO
// when the chain starts: zero it at allocation -- every chain leaves it zero behind its reader (ransac_rot_kernel)
// split: iterations the lane-per-hypothesis kernel runs before it hands undecided hypotheses to the wave-per-hypothesis
// kernel (viso_ctx::gn_split; 100 = the lane kernel does everything)
int launch_ransac(hipStream_t s, const SolverItem* items_dev, int n_items, int iters,
                  unsigned long long seed, const SolverParamsDev& sp, int* queue, int split, int max_points,
                  const RefitMirror* mir = nullptr, const OutArgs* ride = nullptr, int ride_blocks = 0);
size_t viso_rot_bytes(int iters);   // bytes of SolverItem::rot
int launch_circle_join(hipStream_t s, const JoinItem* items_dev, int n_items, const SolverParamsDev& sp);
int launch_collect_triangulate(hipStream_t s, const TriItem* items_dev, int n_items,
                               const SolverParamsDev& sp, int cap);
struct BatchMatchArgs {          // kernarg of match_batch_kernel / match_union_kernel (tiles of 64 queries)
    const MatchProblem* probs;
    int n_probs, bpp, gs, gf, gc;
    int vblocks;                 // match_batch_kernel: (problem, tile) slots the grid walks (the stereo instantiation gets a small grid)
    int* bad;                    // [0] "some image of this run is flagged": lets the (normally idle) general kernels leave at once
                                 // [1] tiles match_stereo_kernel left to match_batch_kernel<1> (0: that kernel leaves at once)
    int r8s, _pad8;              // the shift of the 8-bit planes this run (match_union8_kernel)
    MatchParamsDev mp[2];
};
int launch_match_batch(hipStream_t s, const MatchProblem* probs_dev, int n_probs, int cap_max,
                       const MatchParamsDev mp[2], int* bad, int layout, hipEvent_t e_mid, int variant, int r8s,
                       int kinds = VISO_KIND_ALL);
int launch_match_union_temporal(hipStream_t s, const BatchMatchArgs& a, long long blocks);
int launch_match_prune_temporal(hipStream_t s, const BatchMatchArgs& a, long long blocks);
int launch_match_union8_temporal(hipStream_t s, const BatchMatchArgs& a, long long blocks);
int launch_match_strip_temporal(hipStream_t s, const BatchMatchArgs& a64, int cap_max);
int launch_match_stereo(hipStream_t s, const BatchMatchArgs& a64, int cap_max);
// match_frame.hip: the stereo and the temporal problems of a handful of problems (one frame) in ONE launch
int launch_match_frame(hipStream_t s, const BatchMatchArgs& at, const BatchMatchArgs& as64, long long blocks_t, int cap_max);
#define VISO_FRAME_MAX_BLOCKS 256   // temporal grids up to this size (8 problem slots of 2000 keypoints) take the one-launch kernel
int launch_extract_pack(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap, const uint8_t* images,
                        int rows, int cols, int extras, int r8s, int* r8cnt);
int launch_harris_response(hipStream_t s, const uint8_t* images, int n_img, int rows, int cols, double k, float* resp);
size_t harris_fused_lds(int rows, int cols, int nbinx, int nbiny, int per);
// part: harris_strip_bytes(...) bytes of device scratch for the strip kernel (waves over 58-column strips instead of bins), or
// null: the wave-per-bin kernel.  harris_strip_bytes returns 0 where the strip kernel does not apply or would not pay.
size_t harris_strip_bytes(int n_img, int rows, int cols, int nbinx, int nbiny, int per);
int launch_harris_detect(hipStream_t s, const uint8_t* images, int n_img, int rows, int cols, int n_features, int nbinx,
                         int nbiny, double k, float2* tmp_kp, float* tmp_resp, int* cnt, float2* kp_out, float* resp_out,
                         int* n_out, int cap, size_t kp_stride, void* part = nullptr);
int launch_harris_bins(hipStream_t s, const float* resp, int n_img, int rows, int cols, int n_features, int nbinx,
                       int nbiny, float2* tmp_kp, float* tmp_resp, int* cnt, float2* kp_out, float* resp_out,
                       int* n_out, int cap, size_t kp_stride);
