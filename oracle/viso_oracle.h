/*
 * viso_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, single-threaded) of the libviso hot path, used
 * as the parity checker by tests/, __graft_entry__.smoke() and as bench.py's
 * `cpu_baseline` leg.  Nothing under libviso_amd/ may include, link or call
 * this.  Same POD signatures as include/viso_hip.h so a test can call both
 * sides with identical arguments.
 *
 * PARITY PINNING: the reference (alexkreimer/libviso) cannot be compiled here
 * (OpenCV, Boost, Eigen absent; no network) and its own tests pin no numeric
 * result (test/test.cpp:152-168 only asserts `== true` on a file that is not
 * in the repo).  Matcher: "parity unpinned" by reference fixtures; pinned by
 * this restatement + hand-built quirk cases + an independent numpy
 * restatement (tests/test_oracle_*.py).  Solver: pinned by the analytic
 * known answers the reference's disabled tests describe
 * (test/test.cpp:51-114, :171-205; src/mvg.cpp:73-89).
 */
#ifndef VISO_ORACLE_H_
#define VISO_ORACLE_H_

#include "../include/viso_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/viso.cpp:170-203 + cvflann LinearIndex/L1/RadiusUniqueResultSet
 * (OpenCV flann, un-vendored): neighbours n1 x K int32, -1 padded.
 * Returns 0. found (may be NULL): per query the total number in radius. */
int oracle_radius_search(const float* kp1, int n1, const float* kp2, int n2,
                         float radius, int K, int32_t* neighbors, int32_t* found);

/* src/viso.cpp:655-666 (+ :390-407). p1 = (x,y) of the query (left), p2 target. */
double oracle_sampson_distance(const double F[9], float p1x, float p1y, float p2x, float p2y);

/* src/viso.cpp:669-726. scored (may be NULL) receives the number of SAD
 * evaluations (line :702 executions) — SURVEY.md 8(d)'s C. */
int oracle_match_desc(const float* kp1, int n1, const float* kp2, int n2,
                      const float* d1, const float* d2, int dlen,
                      const viso_match_params* mp,
                      int32_t* out_match, int* out_n, int64_t* scored);

/* src/viso.cpp:207-243 (literal nested loops). */
int oracle_match_circle(const int32_t* lr, int n_lr, const int32_t* lr_prev, int n_lrp,
                        const int32_t* m11, int n11, const int32_t* m22, int n22,
                        int32_t* circ, int32_t* pcl, int cap, int* out_n);

/* src/viso.cpp:501-514 */
int oracle_collect_matches(const float* kp1, int n1, const float* kp2, int n2,
                           const int32_t* match, int n, double* x4xn);
/* src/viso.cpp:1137-1162 */
int oracle_triangulate_rectified(const double* x4xM, int m, const viso_param* p, double* X3xM);

/* src/viso.cpp:1401-1497. J: 4n x 6, predict: 4 x n, residual: 4n. */
void oracle_compute_J(const double* X, const double* obs, int m, const double tr[6],
                      const viso_param* p, const int32_t* active, int n,
                      double* J, double* predict, double* residual);
/* src/viso.cpp:1583-1623. iters (may be NULL): compute_J evaluations done. */
int oracle_minimize_reproj(const double* X, const double* obs, int m, double tr[6],
                           const viso_param* p, const int32_t* active, int n, int* iters);
/* src/viso.cpp:1509-1537 */
int oracle_get_inliers(const double* X, const double* obs, int m, const double tr[6],
                       const viso_param* p, int32_t* inliers, int* n_inl, double* rms);
/* src/viso.cpp:1543-1580 */
int oracle_ransac_minimize_reproj(const double* X, const double* obs, int m,
                                  double best_tr[6], int32_t* best_inl, int* n_inl,
                                  const viso_param* p, const int32_t* samples,
                                  uint64_t seed, uint64_t frame);
/* randomsample(3, m, .) src/viso.cpp:87-107: uniform 3-subsets, ascending; three splitmix64 draws keyed on
 * (seed, frame, hypothesis) through Floyd's subset sampling (viso_oracle.c).  _algorithm_s: the reference's algorithm
 * over the same stream (the definition of rounds 1-5), for the distribution test. */
void oracle_ransac_samples(uint64_t seed, uint64_t frame, int iters, int m, int32_t* out);
void oracle_ransac_samples_algorithm_s(uint64_t seed, uint64_t frame, int iters, int m, int32_t* out);

/* 6x6 LU solve as cv::solve(DECOMP_LU) does it (OpenCV 3.0 LUImpl, un-vendored;
 * singular iff |pivot| < DBL_EPSILON). A (36) and b (6) are overwritten; x in b.
 * Returns 1, or 0 when singular. */
int oracle_lu_solve6(double* A, double* b);

/* src/viso.cpp:109-133 */
void oracle_tr2mat(const double tr[6], double T[16]);
/* src/viso.cpp:1315-1321 */
void oracle_pose_update(const double pose[16], const double tr[6], double out[16]);
/* src/mvg.h:41-66 + src/viso.cpp:1177-1180 */
void oracle_F_from_P(const double P1[12], const double P2[12], double F[9]);
/* src/viso.cpp:1004-1024 */
int oracle_extract_descriptors(const uint8_t* img, int rows, int cols,
                               const float* kp, int n, int radius, float* desc);

/* cv::cornerHarris(blockSize 3, ksize 5, k, BORDER_DEFAULT) restated; resp: rows x cols float. */
int oracle_harris_response(const uint8_t* img, int rows, int cols, double k, float* resp);
/* the rounds 1-3 restatement (exact integer Sobel sums times the scale, nine box taps row-major), for comparison */
int oracle_harris_response_v1(const uint8_t* img, int rows, int cols, double k, float* resp);
/* HarrisBinnedFeatureDetector::detectImpl, src/viso.cpp:926-975 (k explicit). */
int oracle_detect_harris_binned(const uint8_t* img, int rows, int cols, int n_features, int nbinx, int nbiny,
                                double k, float* kp, float* resp_out, int* n_out);

/* One sequence_odometry loop body over in-memory frames (src/viso.cpp:1205-1327),
 * without the front-end: frames laid out as viso_batch (kp [nf][2][cap][2],
 * desc [nf][2][cap][dlen], n [nf][2]).  Outputs per frame t: tr [nf][6],
 * ok [nf], n_inl [nf]; optional per-stage seconds in stage_s[5] (match_desc, circle, gather+triangulate, RANSAC/GN, neighbour-search share of match_desc)
 * (neighbour search+SAD+sort, circle, triangulate/gather, RANSAC/GN).
 * matcher_only != 0 stops after the three match_desc calls. */
int oracle_sequence(const float* kp, const float* desc, const int32_t* n,
                    int nf, int cap, int dlen,
                    const viso_match_params* stereo, const viso_match_params* temporal,
                    const viso_param* p, uint64_t seed, uint64_t first_frame,
                    int matcher_only, double* tr, int32_t* ok, int32_t* n_inl,
                    int64_t* scored /* [3][nf] or NULL */, int64_t* m_out /* [3][nf] or NULL */,
                    double* stage_s);

#ifdef __cplusplus
}
#endif
#endif
