/*
 * viso_hip.h — C-ABI of libviso_hip.so, the MI355X (gfx950) implementation of
 * libviso's per-frame hot path: descriptor-window SAD matcher, circular-match
 * join, rectified triangulation and the RANSAC + Gauss-Newton stereo
 * reprojection pose solver.
 *
 * Every entry point replaces one free function of the reference
 * (alexkreimer/libviso, paths relative to its root); the reference has no FFI
 * of its own (plain C++ free functions over cv::Mat / std::vector), so this
 * header is what a cgo/ctypes/C++ adapter binds instead.  INTEGRATION.md shows
 * the reference-side adapter.
 *
 * Conventions
 *   - plain pointers and sizes only; all matrices are row-major and contiguous;
 *   - "4xM" / "3xM" double matrices are SoA exactly like the reference's
 *     cv::Mat(4,M,CV_64F): row r starts at p + r*M;
 *   - return value: 1 = true, 0 = false (the reference's bool), negative =
 *     VISO_ERR_* (the reference would assert/abort; we never abort across the ABI);
 *   - the *_dev / batch family takes DEVICE pointers and a context (persistent
 *     device buffers + one HIP stream); the plain family takes HOST pointers and
 *     runs on a lazily created default context.
 *   - nothing here computes on the CPU: if the HIP device is missing the calls
 *     fail with VISO_ERR_HIP.
 */
#ifndef VISO_HIP_H_
#define VISO_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VISO_OK 1
#define VISO_FALSE 0
#define VISO_ERR_ARG (-1)         /* bad argument (the reference asserts: src/viso.cpp:676,174-175) */
#define VISO_ERR_HIP (-2)         /* HIP runtime error / no device */
#define VISO_ERR_UNSUPPORTED (-3) /* size beyond what this build handles */
#define VISO_ERR_NOMEM (-4)

#define VISO_DESC_LEN 121 /* (2*5+1)^2, src/viso.cpp:1001,1174 */

/* Mirrors `struct MatchParams` (src/viso.cpp:48-75).  `alg_thresh` and
 * `allow_ann` are never read by match_desc and are omitted. */
typedef struct viso_match_params {
    int32_t enforce_epipolar;   /* src/viso.cpp:50 */
    int32_t enforce_2nd_best;   /* :55 */
    int32_t max_neighbors;      /* :59  K, columns of the neighbour matrix */
    int32_t _pad;
    double F[9];                /* :51  row-major fundamental matrix, x2' F x1 = 0 */
    double sampson_thresh;      /* :53 */
    double ratio_2nd_best;      /* :56 */
    double radius;              /* :60  L1 pixel radius (cast to float at :685) */
} viso_match_params;

/* Mirrors `struct param` (src/viso.h:58-72). */
typedef struct viso_param {
    double base;                /* :61 */
    int32_t ransac_iter;        /* :62 default 50 */
    int32_t save_debug;         /* :65 unused by the hot path, kept for layout parity */
    double inlier_threshold;    /* :63 default 2 */
    double thresh;              /* :64 default 1e-4 */
    double f, cu, cv;           /* :66-71 calib.{f,cu,cv} */
} viso_param;

/* MatchParams(F) ctor, src/viso.cpp:62-71: stereo L->R (epipolar on, K=200, r=80). */
void viso_match_params_stereo(viso_match_params* mp, const double F[9]);
/* MatchParams() ctor, src/viso.cpp:72-74: temporal (2nd-best 0.9, K=250, r=80). */
void viso_match_params_temporal(viso_match_params* mp);
/* param() ctor, src/viso.h:60. base/f/cu/cv are left 0. */
void viso_param_default(viso_param* p);

/* ---------------------------------------------------------------- context */
typedef struct viso_ctx viso_ctx;

/* device: HIP device ordinal.  stream: a hipStream_t to run on, or NULL to let
 * the context create its own.  Returns NULL on failure (viso_last_error()). */
viso_ctx* viso_ctx_create(int device, void* stream);
/* Waits for the streams, frees the context.  VISO_OK, or VISO_ERR_HIP with the first HIP error met in
 * viso_last_error() (everything that can be freed still is).  Batches of the context that are still alive are destroyed
 * with it, FIRST (the library keeps a registry of its live handles): the intended order is batches, then context, but the
 * other order costs nothing worse than a return code -- a later viso_batch_destroy of such a batch is a no-op (VISO_OK, once),
 * every other call on it returns VISO_ERR_ARG, and so does anything on a context or batch handle that was destroyed before
 * or never existed.  Destroy handles before the process starts exiting: not from static destructors that may run after the
 * HIP runtime's own. */
int viso_ctx_destroy(viso_ctx* ctx);
/* hipStream_t the context launches on.  Every context also owns a second stream for the RANSAC
 * stage of its batches (it runs beside the next run's matcher; ordered by events, waited for by
 * viso_ctx_synchronize and by every getter): a host that orders its own work against viso_ctx_stream() must
 * use viso_ctx_synchronize (or a result getter) to see poses, not a bare hipStreamSynchronize of that stream. */
void* viso_ctx_stream(viso_ctx* ctx);
int viso_ctx_synchronize(viso_ctx* ctx);
/* Which kernel takes the temporal match_desc calls of this context (ctx == NULL: the default context of the
 * plain family).  The product build offers three: 6 = match_union8_kernel (the default: a wave ranks every row of a
 * round's union list against eight y-adjacent queries on the rows' 8-bit planes, scores the two best candidates of a
 * query exactly, and lets a rigorous lower bound of the rest decide whether that settles match_desc),
 * 3 = match_union_kernel (the same round structure on the u16 rows: every pair scored exactly; the default until
 * round 4) and 5 = match_prune_kernel (exact successive elimination on block sums in front of a cell-granular scorer).
 * Further variants (2 = match_batch_kernel<0>, 4 = match_strip_kernel) exist in -DVISO_DEBUG_VARIANTS builds only.
 * Every variant gives identical results, and the parity tests run over whatever viso_matcher_variants() reports for
 * the build under test.  Returns VISO_ERR_ARG for a variant this build does not have. */
int viso_ctx_set_matcher(viso_ctx* ctx, int variant);
/* The variants of this build: fills out[0..cap), returns their number.  Needs no device. */
int viso_matcher_variants(int* out, int cap);
/* The variant a new context starts with: the build's default, or $VISO_MATCHER when that names a variant of this build
 * (an A/B and test aid).  Needs no device. */
int viso_matcher_default(void);
/* How the RANSAC stage splits the <= 100 Gauss-Newton iterations of a 3-point hypothesis (src/viso.cpp:1593) between
 * its two kernels: the lane-per-hypothesis kernel runs the first `split`, the wave-per-hypothesis kernel the rest of
 * the few that are still undecided.  0 = the build's default (10); 100 = the lane kernel alone.  Every split gives
 * bit-identical hypotheses (tests/test_gpu_solver_edges.py); this is a tuning / test knob.  ctx == NULL: the default
 * context of the plain family. */
int viso_ctx_set_gn_split(viso_ctx* ctx, int split);
/* match_union8_kernel (matcher variant 6) ranks candidates on 8-bit planes h(v) = clamp((v + (128 << s)) >> s, 0, 255) of the
 * descriptor rows.  shift = -1 (default): s is chosen per run from the descriptor magnitudes the previous run of the batch
 * packed (the smallest s that clamps at most one element pair in 256; 3 before any statistics exist and in the one-call
 * plain family); 0..3: fixed.  Every s gives the same results — the bound behind the ranking holds for any s and any
 * descriptors — it only decides how often the kernel has to score more than two candidates of a query exactly.
 * $VISO_ROW8_SHIFT sets the same for every new context (test / A-B aid). */
int viso_ctx_set_row8_shift(viso_ctx* ctx, int shift);
/* Name of that kernel as it appears in rocprofv3 summaries. */
const char* viso_ctx_matcher_kernel_name(viso_ctx* ctx);
const char* viso_last_error(void);
/* "libviso_hip <version> gfx950 ..." */
const char* viso_version(void);

/* ------------------------------------------------------ plain (host) family */

/* match_desc, src/viso.cpp:669-726 (+ radiusSearch :170-203, sampsonDistance
 * :655-666, kp2mat :246-256).  kp*: n x 2 float (x,y).  d*: n x dlen float.
 * out_match: up to n1 rows (i1,i2,(int)dist), sorted by (dist asc, i1 asc) —
 * the reference's std::sort is unstable (:724); this is the documented
 * tightening.  *out_n = number of matches. */
int viso_match_desc(const float* kp1, int n1, const float* kp2, int n2,
                    const float* d1, const float* d2, int dlen,
                    const viso_match_params* mp,
                    int32_t* out_match, int* out_n);

/* match_circle, src/viso.cpp:207-243.  Lists are n x 3 int32 (i1,i2,dist).
 * circ: up to cap x 4, pcl: up to cap x 2 (positions into lr / lr_prev).
 * Exact for arbitrary lists (duplicate keys included): output order is the
 * reference's nested-loop order.  Returns VISO_ERR_ARG if more than cap rows
 * would be produced (*out_n then holds the required count). */
int viso_match_circle(const int32_t* lr, int n_lr, const int32_t* lr_prev, int n_lrp,
                      const int32_t* m11, int n11, const int32_t* m22, int n22,
                      int32_t* circ, int32_t* pcl, int cap, int* out_n);

/* collect_matches(...,Mat& x) src/viso.cpp:501-514: x = 4 x n double
 * (uL,vL,uR,vR). */
int viso_collect_matches(const float* kp1, int n1, const float* kp2, int n2,
                         const int32_t* match, int n, double* x4xn);

/* triangulate_rectified<double>, src/viso.cpp:1137-1162. */
int viso_triangulate_rectified(const double* x4xM, int m, const viso_param* p,
                               double* X3xM);

/* minimize_reproj, src/viso.cpp:1583-1623.  tr is in/out.  1 = converged,
 * 0 = singular system or 100 iterations exhausted. */
int viso_minimize_reproj(const double* X3xM, const double* obs4xM, int m,
                         double tr[6], const viso_param* p,
                         const int32_t* active, int n_active);

/* get_inliers, src/viso.cpp:1509-1537.  inliers: up to m ascending indices.
 * rms (may be NULL) reproduces the reference's last-point value (:1535). */
int viso_get_inliers(const double* X3xM, const double* obs4xM, int m,
                     const double tr[6], const viso_param* p,
                     int32_t* inliers, int* n_inliers, double* rms);

/* ransac_minimize_reproj, src/viso.cpp:1543-1580.  samples: ransac_iter x 3
 * ascending distinct indices (what randomsample(3,m,.) :87-107 yields), or
 * NULL to draw them from viso_ransac_samples(seed, frame, ...).  best_tr is
 * in/out like the reference's: it is assigned only when a hypothesis improves the
 * support (:1564-1568), so the caller's values survive when no hypothesis finds any
 * (return 0, *n_inl = 0) and for m < 3; with a support of 1..5 it is the best
 * hypothesis' motion (return 0, :1571), otherwise the refit's (the partly iterated
 * value when the refit fails, :1572).  best_inl: up to m indices. */
int viso_ransac_minimize_reproj(const double* X3xM, const double* obs4xM, int m,
                                double best_tr[6], int32_t* best_inl, int* n_inl,
                                const viso_param* p, const int32_t* samples,
                                uint64_t seed, uint64_t frame);

/* Support sizes of n_h given motions tr_h[n_h][6] over one point set, through the RANSAC stage's counting kernel
 * (diagnostics / tests): cnt[h] = number of inliers get_inliers (src/viso.cpp:1509-1537) finds for tr_h[h]. */
int viso_support_sizes(const double* X3xM, const double* obs4xM, int m, const double* tr_h, int n_h,
                       const viso_param* p, int32_t* cnt);

/* Deterministic replacement for randomsample's per-call random_device
 * (src/viso.cpp:87-107: a uniformly distributed 3-subset of 0..m-1, ascending).
 * Triple h = the first three outputs of a splitmix64 stream keyed on
 * (seed, frame, h) through Floyd's subset sampling -- t_i = floor(draw_i * (j+1) / 2^64)
 * for j = m-3, m-2, m-1, taken, or j itself if t_i was taken already -- sorted
 * ascending: the same distribution in three draws (the reference's algorithm S
 * needs ~m/2; rounds 1-5 ran it over the same stream: other triples, same law).
 * m < 3: zeros.  out: iters x 3. */
void viso_ransac_samples(uint64_t seed, uint64_t frame, int iters, int m, int32_t* out);

/* tr2mat, src/viso.cpp:109-133. T: 4x4 row-major. Host arithmetic (six
 * sin/cos; nothing to offload). */
void viso_tr2mat(const double tr[6], double T[16]);
/* pose <- pose * inv(tr2mat(tr)), src/viso.cpp:1315-1321. */
void viso_pose_update(const double pose[16], const double tr[6], double out[16]);
/* F_from_P<double>, src/mvg.h:41-66, followed by the normalisation of
 * src/viso.cpp:1177-1180.  P1,P2: 3x4 row-major. */
void viso_F_from_P(const double P1[12], const double P2[12], double F[9]);

/* MyFeatureExtractor::computeImpl, src/viso.cpp:1004-1024: Sobel-x 3x3
 * (BORDER_REFLECT_101) then (2r+1)^2 window per keypoint; zero where
 * y<=0|y>=rows|x<=0|x>=cols.  img: rows x cols uint8.  desc: n x (2r+1)^2. */
int viso_extract_descriptors(const uint8_t* img, int rows, int cols,
                             const float* kp, int n, int radius, float* desc);

/* cv::cornerHarris(img, R, 3, 5, k, BORDER_DEFAULT) restated in OpenCV's evaluation order (scale folded into the float
 * smoothing taps, row pass then column pass with the symmetric grouping, box filter as row sums then column sums; what
 * an algorithm-level restatement cannot pin is listed in oracle/viso_oracle.c); resp: rows x cols float.
 * ($VISO_HARRIS_BAND = rows per wave of the response kernel, 1..4096: a tuning / test aid, the image does not depend on it.) */
int viso_harris_response(const uint8_t* img, int rows, int cols, double k, float* resp);
/* HarrisBinnedFeatureDetector::detectImpl, src/viso.cpp:926-975 (reference defaults:
 * n_features 1200, nbinx 24, nbiny 5).  kp: up to n_features x 2 (x,y);
 * resp_out (may be NULL): |response| per keypoint. */
int viso_detect_harris_binned(const uint8_t* img, int rows, int cols, int n_features, int nbinx, int nbiny,
                              double k, float* kp, float* resp_out, int* n_out);

/* Where a plain-family call's time goes (diagnostics; bench.py `drop_in_per_call`, viso_host_gputest).  The plain family is
 * what the patched reference loop calls once per function per frame (src/viso.cpp:1240-1313: match_desc x3,
 * collect_matches, triangulate_rectified, match_circle, ransac_minimize_reproj), so a frame's cost there is seven
 * synchronous host->device->host round trips.  With profiling on, every call brackets its phases with hipEvents on the
 * default context's stream: h2d_us = input copies, kernel_us = the kernels, d2h_us = result copies (up to the last byte
 * on the host), wait_us = host time blocked in hipStreamSynchronize / blocking copies, host_us = wall time of the call.
 * Sums over the calls since profiling was switched on.  Profiling costs a few microseconds per call: rates are quoted
 * with it off. */
#define VISO_PLAIN_MATCH_DESC 0
#define VISO_PLAIN_COLLECT_MATCHES 1
#define VISO_PLAIN_TRIANGULATE 2
#define VISO_PLAIN_MATCH_CIRCLE 3
#define VISO_PLAIN_RANSAC 4
#define VISO_PLAIN_MINIMIZE 5
#define VISO_PLAIN_GET_INLIERS 6
#define VISO_PLAIN_N 7
typedef struct viso_plain_times {
    int64_t calls;
    double host_us, h2d_us, kernel_us, d2h_us, wait_us;
} viso_plain_times;
/* What the plain family does behind one function per call (libviso_amd/csrc/plain.hip; DESIGN.md "the drop-in path"):
 *   image cache    match_desc recognises an image it has been given before -- the loop passes every (keypoints,
 *                  descriptors) set three times, under changing addresses (`d1.copyTo(d1_prev)`, src/viso.cpp:1213) -- by
 *                  comparing its bytes with a pinned host shadow (memcmp: exact), and then neither uploads nor sorts nor
 *                  packs it again.  viso_plain_cache(0) / $VISO_PLAIN_CACHE=0: every image uploaded again.
 *   frames         the stereo call of a frame (enforce_epipolar != 0) also runs what the loop asks for next: the two
 *                  temporal match_desc problems against the previous stereo call's images, collect_matches /
 *                  triangulate_rectified of its own matches, match_circle of the four lists, the gather of :1292-1305
 *                  and ransac_minimize_reproj with the parameters and stream key of the previous frame's call (key
 *                  advanced by its last step).  A later call is answered from those results ONLY if its arguments are
 *                  byte for byte what was assumed (the functions are pure: same results as the direct path, which any
 *                  other argument takes).  Guessing starts after the call sequence has been seen once and stops when a
 *                  guess goes unused.  viso_plain_speculate(0) / $VISO_PLAIN_SPECULATE=0: every call direct.
 *   waiting        a call returns when its results are in pinned host memory: the workgroups that write them there (they ride in
 *                  the launch of the chain's next kernel; the last kernel writes its own) say so in a pinned word the host
 *                  spins on (a few microseconds sooner than hipStreamSynchronize sees it; after 20 ms the stream is
 *                  synchronised instead).  $VISO_PLAIN_SIGNAL=0: always hipStreamSynchronize.
 * Both only change when the work is done, never a result (tests/test_gpu_drop_in.py runs every combination).  The
 * speculation statistics: served[0..3] = calls answered from a frame (temporal match_desc, collect_matches,
 * triangulate_rectified + match_circle, ransac_minimize_reproj), wasted[0..3] = results computed ahead and never asked for. */
int viso_plain_cache(int enable);
int viso_plain_cache_stats(int64_t* hits, int64_t* misses);
int64_t viso_plain_general_reruns(void);   /* calls repeated because their launch had left out a kernel the data then needed: an image
                                              that unexpectedly did not fit the u16 rows, or a stereo pair with a wide epipolar band
                                              after several rectified ones */
int viso_plain_speculate(int enable);
int viso_plain_speculate_stats(int64_t served_wasted[8]);
/* $VISO_PLAIN_TRACE=1: host microseconds of viso_match_desc by phase, of the temporal calls answered from a frame, and the waits of
   match_circle / ransac_minimize_reproj behind the stereo call; this prints and zeroes them (stderr). */
void viso_plain_trace_dump(void);
int viso_plain_profile(int enable);                        /* 1: zero the sums and start; 0: stop */
int viso_plain_profile_get(int fn, viso_plain_times* out); /* fn: VISO_PLAIN_* */
const char* viso_plain_profile_name(int fn);

/* ------------------------------------------- batched, device-resident family
 *
 * A frame set holds `n_frames` stereo frames resident in HBM:
 *   kp   [n_frames][2][cap][2]   float   (x,y)   image 0 = left, 1 = right
 *   desc [n_frames][2][cap][121] float            (the reference's boundary layout)
 *   n    [n_frames][2]           int32   keypoints actually present (<= cap)
 * viso_batch_* processes the pairs (t-1,t) for t = 1..n_frames-1 exactly like
 * one iteration each of sequence_odometry's loop body (src/viso.cpp:1205-1327):
 * stereo match of every frame, two temporal matches, circle join, gather,
 * RANSAC/GN.  All stages run on the context's stream without host round trips.
 */
typedef struct viso_batch viso_batch;

viso_batch* viso_batch_create(viso_ctx* ctx, int n_frames, int cap, int dlen);
/* Waits for the context's streams, frees the batch; return value as viso_ctx_destroy (VISO_OK also for a batch its context
 * has already taken along; VISO_ERR_ARG for a handle that is not, or no longer, a batch). */
int viso_batch_destroy(viso_batch* b);

/* Stream rule for everything below: a batch's kernels run asynchronously on its context's stream.  Every call
 * that writes batch inputs (upload*, set_params, detect) is ordered against that stream — the synchronous ones
 * wait for it, the *_async ones are enqueued on it — so it is safe to call them while a run is in flight.
 * Every entry point selects the context's device (hipSetDevice) first: one process may drive several GPUs. */

/* Upload host data for frames [f0, f0+nf) (layout as above, tightly packed
 * over nf frames).  Synchronous: returns when the data is on the device. */
int viso_batch_upload(viso_batch* b, int f0, int nf, const float* kp,
                      const float* desc, const int32_t* n);
/* Same, enqueued on the context's stream: returns at once.  The host buffers must stay untouched until the
 * stream has passed the copies (viso_ctx_synchronize / any result getter of a later run).  Buffers from
 * viso_host_alloc (pinned) are copied by DMA and overlap kernels of other contexts: the streaming mode of a
 * host that feeds new frames every step (sequence_odometry consumes fresh data per frame, src/viso.cpp:1205-1231). */
int viso_batch_upload_async(viso_batch* b, int f0, int nf, const float* kp,
                            const float* desc, const int32_t* n);
/* The same uploads with the descriptors as int16 (N x dlen, tightly packed over [nf][2][cap][dlen]): the lossless
 * encoding of what MyFeatureExtractor produces (3x3 Sobel of uint8: integers in [-1020, 1020], src/viso.cpp:1004-1024)
 * at half the bytes of the reference's CV_32F rows (:995,1008) — the PCIe-bound streaming mode moves half the data.
 * Same results as the f32 uploads of the same values.  All frames of a batch must come through ONE of the two
 * families (the int16 rows live in the f32 rows' device buffer): the batch remembers per frame which family filled it,
 * and viso_batch_run* returns VISO_ERR_ARG when the frames of a run disagree.  Needs dlen <= 128. */
int viso_batch_upload_i16(viso_batch* b, int f0, int nf, const float* kp,
                          const int16_t* desc16, const int32_t* n);
int viso_batch_upload_i16_async(viso_batch* b, int f0, int nf, const float* kp,
                                const int16_t* desc16, const int32_t* n);
void* viso_host_alloc(size_t bytes);
int viso_host_free(void* p);
/* Device pointers of the boundary-layout buffers, for producers that already
 * live on the GPU (a device-side extractor, torch): kp, desc, n as above. */
int viso_batch_device_ptrs(viso_batch* b, void** kp, void** desc, void** n);

int viso_batch_set_params(viso_batch* b, const viso_match_params* stereo,
                          const viso_match_params* temporal, const viso_param* p,
                          uint64_t seed, uint64_t first_frame_index);

/* Matcher stage only (BASELINE config 2): pack + 3 match_desc per frame
 * (stereo for all frames, temporal L and R for t>=1) incl. the final sort. */
int viso_batch_run_matcher(viso_batch* b);
/* Full per-frame path (BASELINE config 3): matcher + triangulation + circle
 * join + RANSAC/GN.  Asynchronous on the context's stream. */
int viso_batch_run(viso_batch* b);

/* Image-in mode: upload uint8 images [nf][2][rows][cols]
 * and keypoints; viso_batch_run_images extracts the 11x11 Sobel-x descriptor
 * windows on the device (MyFeatureExtractor, src/viso.cpp:1004-1024) straight
 * into the matcher's row format and then runs like viso_batch_run_matcher
 * (matcher_only != 0) or viso_batch_run.  Needs dlen == 121. */
int viso_batch_upload_images(viso_batch* b, int f0, int nf, const uint8_t* images, int rows, int cols,
                             const float* kp, const int32_t* n);
int viso_batch_run_images(viso_batch* b, int matcher_only);
/* viso_batch_upload_images enqueued on the stream (see viso_batch_upload_async); the image buffers must already
 * exist for this geometry (one synchronous viso_batch_upload_images call allocates them). */
int viso_batch_upload_images_async(viso_batch* b, int f0, int nf, const uint8_t* images, int rows, int cols,
                                   const float* kp, const int32_t* n);
/* HarrisBinnedFeatureDetector::detectImpl (src/viso.cpp:926-975) on every
 * uploaded image (pass kp = n = NULL to viso_batch_upload_images): fills the
 * batch's keypoints on the device, one workgroup per (image, bin), without a response image in memory (bins up to
 * 62 pixels wide with up to 32 corners each; larger ones take a response image + a selection kernel).
 * cv::cornerHarris(blockSize 3, ksize 5, k)
 * restated; the reference leaves k uninitialised (:915-919,978) — its intended
 * default is 0.04f — and the order inside a bin unspecified (:963); here:
 * (|response| desc, push order asc).  n_features/(nbinx*nbiny) corners per bin. */
int viso_batch_detect(viso_batch* b, int n_features, int nbinx, int nbiny, double k);
int viso_batch_get_keypoints(viso_batch* b, int t, int side, float* kp, int* n_out);

/* Results (host copies; they synchronise the stream).
 * which: 0 = stereo L->R of frame t, 1 = temporal left (t vs t-1), 2 = temporal right. */
int viso_batch_get_matches(viso_batch* b, int which, int t, int32_t* out_match, int* out_n);
int viso_batch_get_circle(viso_batch* b, int t, int32_t* circ, int32_t* pcl, int* out_n);
/* tr[6], ok (1/0, 0 also when <3 circle matches, src/viso.cpp:1283), inliers. */
int viso_batch_get_pose(viso_batch* b, int t, double tr[6], int* ok,
                        int32_t* inliers, int* n_inl);
/* All frames at once: tr [n_frames][6], ok [n_frames], n_inl [n_frames]
 * (entry 0 is zero/0: the first frame has no predecessor, :1256-1260).  Waits for the batch's streams, then reads a
 * pinned host mirror that the last kernel of every viso_batch_run / viso_batch_run_images fills: no device copy. */
int viso_batch_get_poses(viso_batch* b, double* tr, int32_t* ok, int32_t* n_inl);
/* Per-hypothesis state of the last run's RANSAC stage (diagnostics / tests): tr_h [n_frames][ransac_iter][6],
 * ok_h, cnt_h [n_frames][ransac_iter] (support sizes; frame 0 unused), *n_undecided = hypotheses that needed the
 * wave-per-hypothesis kernel.  Any pointer may be NULL.  ransac_iter is the one given to viso_batch_set_params;
 * viso_batch_get_hypotheses2 takes the capacity of the caller's arrays (in hypotheses per frame): the arrays are
 * [n_frames][iters_capacity] (x 6 for tr_h) and every frame's row is written at THAT stride (entries beyond ransac_iter
 * are left alone); VISO_ERR_ARG when iters_capacity < ransac_iter. */
int viso_batch_get_hypotheses(viso_batch* b, double* tr_h, int32_t* ok_h, int32_t* cnt_h, int32_t* n_undecided);
int viso_batch_get_hypotheses2(viso_batch* b, int iters_capacity, double* tr_h, int32_t* ok_h, int32_t* cnt_h,
                               int32_t* n_undecided);
/* Work counters of the last run, for the algorithmic-bytes model of
 * per (which,t) the number of scored (query,candidate) pairs C
 * and matches emitted M_out.  scored/m_out: [3][n_frames] int64. */
int viso_batch_get_counters(viso_batch* b, int64_t* scored, int64_t* m_out);
/* flags [n_frames][2]: 1 where the last run found descriptor values of that image that the packed u16 rows cannot
 * hold (not integers in [-32768, 32767], or dlen > 128).  Only the match_desc calls that read such an image take
 * the general kernel (float differences summed in double, the arithmetic of cv::norm at src/viso.cpp:702); all
 * other calls of the batch stay on the u16 kernels.  Same results either way. */
int viso_batch_get_general_path_flags(viso_batch* b, int32_t* flags);
/* Number of queries of the last run that took match_overflow_kernel (exact K-cap selection / largest-key tie rule
 * / candidate lists beyond the tile kernels' LDS slots): a few for sparse features, a sizeable share where keypoints
 * cluster densely.  Same results either way; this is the data-dependent cost to watch. */
int viso_batch_get_overflow_count(viso_batch* b, int32_t* n);
/* Diagnostics: the shift of the 8-bit planes the batch's last run used (viso_ctx_set_row8_shift). */
int viso_batch_get_row8_shift(viso_batch* b, int* shift);
/* Duration of the kernel that takes the temporal calls (viso_ctx_matcher_kernel_name), measured with hipEvents
 * on the context's stream: average in ms over the runs since the last viso_batch_kernel_ms call. */
int viso_batch_kernel_timing(viso_batch* b, int enable);
int viso_batch_kernel_ms(viso_batch* b, double* matcher_ms_avg, int* n_launches);
/* Time stamps of a run (hipEvents on the batch's streams), for hosts that want to know where a chunk's time went
 * (the KITTI runner's per-rank report): viso_batch_stamp(b, 0) before the run's uploads, viso_batch_stamp(b, 1) after
 * them; viso_batch_run* stamps the end of the run itself.  viso_batch_stamp_ms waits for the run and returns
 * ms[0] = the uploads, ms[1] = everything the run launched behind them. */
int viso_batch_stamp(viso_batch* b, int which);
int viso_batch_stamp_ms(viso_batch* b, double ms[2]);

#ifdef __cplusplus
}
#endif
#endif /* VISO_HIP_H_ */
