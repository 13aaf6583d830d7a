/*
 * viso_oracle.c — TEST INFRASTRUCTURE ONLY (see viso_oracle.h).
 *
 * Single-threaded C99 restatement of libviso's per-frame hot path, quirks
 * included (Q1..Q9 of SURVEY.md 8(a)).  Compile with -ffp-contract=off: the
 * reference is built for baseline x86-64 (no FMA), so no product-sum is fused.
 *
 * "parity unpinned" for the matcher by reference fixtures (there are none);
 * see the header for what pins it instead.
 */
#include "viso_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ params */
/* These three also exist in the HIP library (host side); restated here so the
 * oracle is self-contained. */
static void o_params_stereo(viso_match_params* mp, const double F[9]) {
    memset(mp, 0, sizeof(*mp));
    mp->enforce_epipolar = 1;      /* src/viso.cpp:62 */
    mp->sampson_thresh = 1;        /* :63 */
    mp->enforce_2nd_best = 0;      /* :64 */
    mp->ratio_2nd_best = .8;       /* :65 */
    mp->max_neighbors = 200;       /* :67 */
    mp->radius = 80;               /* :68 */
    memcpy(mp->F, F, 9 * sizeof(double));
}
static void o_params_temporal(viso_match_params* mp) {
    memset(mp, 0, sizeof(*mp));
    mp->enforce_epipolar = 0;      /* src/viso.cpp:72 */
    mp->enforce_2nd_best = 1;
    mp->ratio_2nd_best = .9;       /* :73 */
    mp->max_neighbors = 250;
    mp->radius = 80;               /* :74 */
}
void oracle_match_params_stereo(viso_match_params* mp, const double F[9]) { o_params_stereo(mp, F); }
void oracle_match_params_temporal(viso_match_params* mp) { o_params_temporal(mp); }

/* ----------------------------------------------------------- radiusSearch */
typedef struct { float d; int32_t idx; } dist_index;

/* cvflann DistIndex::operator< : (dist, index) lexicographic */
static int cmp_dist_index(const void* a, const void* b) {
    const dist_index* x = (const dist_index*)a;
    const dist_index* y = (const dist_index*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int oracle_radius_search(const float* kp1, int n1, const float* kp2, int n2,
                         float radius, int K, int32_t* neighbors, int32_t* found) {
    dist_index* set = (dist_index*)malloc(sizeof(dist_index) * (size_t)(n2 > 0 ? n2 : 1));
    if (!set) return -1;
    for (int i = 0; i < n1; ++i) {
        const float qx = kp1[2 * i], qy = kp1[2 * i + 1];
        int cnt = 0;
        /* LinearIndex::findNeighbors: every dataset point, in index order */
        for (int t = 0; t < n2; ++t) {
            /* cvflann::L1<float>::operator() with size 2: tail loop only,
             * result = 0; result += |a0-b0|; result += |a1-b1|  (float) */
            float result = 0.f;
            result += fabsf(qx - kp2[2 * t]);
            result += fabsf(qy - kp2[2 * t + 1]);
            /* RadiusUniqueResultSet::addPoint: dist <= radius_ */
            if (result <= radius) { set[cnt].d = result; set[cnt].idx = t; ++cnt; }
        }
        /* std::set<DistIndex> iteration order */
        qsort(set, (size_t)cnt, sizeof(dist_index), cmp_dist_index);
        int32_t* row = neighbors + (size_t)i * K;
        int j = 0;
        for (; j < K && j < cnt; ++j) row[j] = set[j].idx;
        /* src/viso.cpp:182-186: pad the tail with -1 (the matrix also starts
         * as Scalar(-1), :681) */
        for (; j < K; ++j) row[j] = -1;
        if (found) found[i] = cnt;
    }
    free(set);
    return 0;
}

/* ------------------------------------------------------------- sampson */
/* src/viso.cpp:390-407 */
static double algebric_distance(const double F[9], float p1x, float p1y, float p2x, float p2y) {
    float a0 = p1x, a1 = p1y, a2 = 1, b0 = p2x, b1 = p2y, b2 = 1;
    return b0 * F[0] * a0 +
           b0 * F[1] * a1 +
           b0 * F[2] * a2 +
           b1 * F[3] * a0 +
           b1 * F[4] * a1 +
           b1 * F[5] * a2 +
           b2 * F[6] * a0 +
           b2 * F[7] * a1 +
           b2 * F[8] * a2;
}

/* src/viso.cpp:655-666 */
double oracle_sampson_distance(const double F[9], float p1x, float p1y, float p2x, float p2y) {
    double Fx0 = F[0] * p1x + F[1] * p1y + F[2],
           Fx1 = F[3] * p1x + F[4] * p1y + F[5],
           Ftx0 = F[0] * p2x + F[3] * p2y + F[6],
           Ftx1 = F[1] * p2x + F[4] * p2y + F[7];
    float ad = (float)algebric_distance(F, p1x, p1y, p2x, p2y); /* Q4: rounded to float */
    float ad2 = ad * ad;                                          /* float * float */
    return ad2 / (Fx0 * Fx0 + Fx1 * Fx1 + Ftx0 * Ftx0 + Ftx1 * Ftx1);
}

/* ------------------------------------------------------------ match_desc */
static int cmp_match(const void* a, const void* b) {
    const int32_t* x = (const int32_t*)a;
    const int32_t* y = (const int32_t*)b;
    /* src/viso.cpp:724 sorts by [2] only and is unstable (Q5); the documented
     * total order used on both sides is (dist asc, i1 asc). */
    if (x[2] != y[2]) return (x[2] > y[2]) - (x[2] < y[2]);
    return (x[0] > y[0]) - (x[0] < y[0]);
}

/* wall time spent in the neighbour search (:685) of the oracle_match_desc calls of this thread, for the per-stage
 * CPU baseline (the reference logs only whole-call times, :674,725) */
static __thread double g_search_s = 0;
static double mono_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int oracle_match_desc(const float* kp1, int n1, const float* kp2, int n2,
                      const float* d1, const float* d2, int dlen,
                      const viso_match_params* mp,
                      int32_t* out_match, int* out_n, int64_t* scored) {
    if (n1 < 0 || n2 < 0 || dlen <= 0 || !mp || mp->max_neighbors <= 0) return VISO_ERR_ARG;
    const int K = mp->max_neighbors;
    int32_t* neighbors = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1) * K);
    if (!neighbors) return VISO_ERR_NOMEM;
    /* :685 — sp.radius (double) is passed as `float radius` */
    const double ts0 = mono_s();
    oracle_radius_search(kp1, n1, kp2, n2, (float)mp->radius, K, neighbors, NULL);
    g_search_s += mono_s() - ts0;
    int m = 0;
    int64_t nscored = 0;
    for (int i = 0; i < n1; ++i) {
        const float p1x = kp1[2 * i], p1y = kp1[2 * i + 1];
        double best_d1 = DBL_MAX, best_d2 = DBL_MAX;
        int best_idx = -1;
        const int32_t* row = neighbors + (size_t)i * K;
        /* :692-693 — Q1: the walk stops at -1 AND at target index 0 */
        for (int j = 0; j < K && row[j] > 0; ++j) {
            const int nind = row[j];
            if (mp->enforce_epipolar) {
                double s = oracle_sampson_distance(mp->F, p1x, p1y, kp2[2 * nind], kp2[2 * nind + 1]);
                if (!isfinite(s) || s > mp->sampson_thresh) continue;
            }
            /* :702 — cv::norm(d2.row - d1.row, NORM_L1): difference in float,
             * |.| summed in double */
            const float* a = d2 + (size_t)nind * dlen;
            const float* b = d1 + (size_t)i * dlen;
            double d = 0;
            for (int c = 0; c < dlen; ++c) {
                float df = a[c] - b[c];
                d += (double)fabsf(df);
            }
            ++nscored;
            if (d <= best_d1) {            /* Q2: <= , the later candidate wins ties */
                best_d2 = best_d1;
                best_d1 = d;
                best_idx = nind;
            } else if (d <= best_d2) {
                best_d2 = d;
            }
        }
        if (best_idx >= 0) {
            int accept = 1;
            if (mp->enforce_2nd_best)
                accept = best_d1 < best_d2 * mp->ratio_2nd_best; /* Q3 */
            if (accept) {
                out_match[3 * m + 0] = i;
                out_match[3 * m + 1] = best_idx;
                out_match[3 * m + 2] = (int32_t)best_d1; /* Vec3i(i, idx, double) */
                ++m;
            }
        }
    }
    qsort(out_match, (size_t)m, 3 * sizeof(int32_t), cmp_match);
    *out_n = m;
    if (scored) *scored = nscored;
    free(neighbors);
    return VISO_OK;
}

/* ---------------------------------------------------------- match_circle */
int oracle_match_circle(const int32_t* lr, int n_lr, const int32_t* lr_prev, int n_lrp,
                        const int32_t* m11, int n11, const int32_t* m22, int n22,
                        int32_t* circ, int32_t* pcl, int cap, int* out_n) {
    int n = 0;
    for (int i = 0; i < n_lr; ++i) {
        int ileft = lr[3 * i], iright = lr[3 * i + 1];
        for (int j = 0; j < n11; ++j) {
            if (m11[3 * j] != ileft) continue;
            int ileft_prev = m11[3 * j + 1];
            for (int k = 0; k < n_lrp; ++k) {
                if (lr_prev[3 * k] != ileft_prev) continue;
                int iright_prev = lr_prev[3 * k + 1];
                for (int l = 0; l < n22; ++l) {
                    if (m22[3 * l + 1] == iright_prev && m22[3 * l] == iright) {
                        if (n < cap) {
                            circ[4 * n + 0] = ileft;
                            circ[4 * n + 1] = iright;
                            circ[4 * n + 2] = ileft_prev;
                            circ[4 * n + 3] = iright_prev;
                            pcl[2 * n + 0] = i;
                            pcl[2 * n + 1] = k;
                        }
                        ++n;
                    }
                }
            }
        }
    }
    *out_n = n;
    return n <= cap ? VISO_OK : VISO_ERR_ARG;
}

/* -------------------------------------------- collect + triangulate */
int oracle_collect_matches(const float* kp1, int n1, const float* kp2, int n2,
                           const int32_t* match, int n, double* x) {
    for (int i = 0; i < n; ++i) {
        int i1 = match[3 * i], i2 = match[3 * i + 1];
        if (i1 < 0 || i1 >= n1 || i2 < 0 || i2 >= n2) return VISO_ERR_ARG;
        x[0 * n + i] = kp1[2 * i1];
        x[1 * n + i] = kp1[2 * i1 + 1];
        x[2 * n + i] = kp2[2 * i2];
        x[3 * n + i] = kp2[2 * i2 + 1];
    }
    return VISO_OK;
}

int oracle_triangulate_rectified(const double* x, int m, const viso_param* p, double* X) {
    for (int i = 0; i < m; ++i) {
        double d = x[0 * m + i] - x[2 * m + i];
        X[0 * m + i] = p->base * (x[0 * m + i] - p->cu) / d;
        X[1 * m + i] = p->base * (x[1 * m + i] - p->cv) / d;
        X[2 * m + i] = p->f * p->base / d;
    }
    return VISO_OK;
}

/* ------------------------------------------------------------- compute_J */
void oracle_compute_J(const double* X, const double* obs, int m, const double tr[6],
                      const viso_param* param, const int32_t* active, int n,
                      double* J, double* predict, double* residual) {
    double rx = tr[0], ry = tr[1], rz = tr[2];
    double tx = tr[3], ty = tr[4], tz = tr[5];
    double sx = sin(rx), cx = cos(rx), sy = sin(ry);
    double cy = cos(ry), sz = sin(rz), cz = cos(rz);

    double r00 = +cy * cz, r01 = -cy * sz, r02 = +sy;
    double r10 = +sx * sy * cz + cx * sz, r11 = -sx * sy * sz + cx * cz, r12 = -sx * cy;
    double r20 = -cx * sy * cz + sx * sz, r21 = +cx * sy * sz + sx * cz, r22 = +cx * cy;
    double rdrx10 = +cx * sy * cz - sx * sz, rdrx11 = -cx * sy * sz - sx * cz, rdrx12 = -cx * cy;
    double rdrx20 = +sx * sy * cz + cx * sz, rdrx21 = -sx * sy * sz + cx * cz, rdrx22 = -sx * cy;
    double rdry00 = -sy * cz, rdry01 = +sy * sz, rdry02 = +cy;
    double rdry10 = +sx * cy * cz, rdry11 = -sx * cy * sz, rdry12 = +sx * sy;
    double rdry20 = -cx * cy * cz, rdry21 = +cx * cy * sz, rdry22 = -cx * sy;
    double rdrz00 = -cy * sz, rdrz01 = -cy * cz;
    double rdrz10 = -sx * sy * sz + cx * cz, rdrz11 = -sx * sy * cz - cx * sz;
    double rdrz20 = +cx * sy * sz + sx * cz, rdrz21 = +cx * sy * cz - sx * sz;

    double X1p, Y1p, Z1p, X1c, Y1c, Z1c, X2c, X1cd = 0, Y1cd = 0, Z1cd = 0;
    for (int i = 0; i < n; i++) {
        X1p = X[0 * m + active[i]];
        Y1p = X[1 * m + active[i]];
        Z1p = X[2 * m + active[i]];

        X1c = r00 * X1p + r01 * Y1p + r02 * Z1p + tx;
        Y1c = r10 * X1p + r11 * Y1p + r12 * Z1p + ty;
        Z1c = r20 * X1p + r21 * Y1p + r22 * Z1p + tz;

        /* :1449 — Q6: column i of observe, NOT active[i] */
        double weight = 1.0 / (fabs(obs[0 * m + i] - param->cu) / fabs(param->cu) + 0.05);

        X2c = X1c - param->base;
        for (int j = 0; j < 6; j++) {
            switch (j) {
            case 0: X1cd = 0;
                Y1cd = rdrx10 * X1p + rdrx11 * Y1p + rdrx12 * Z1p;
                Z1cd = rdrx20 * X1p + rdrx21 * Y1p + rdrx22 * Z1p;
                break;
            case 1: X1cd = rdry00 * X1p + rdry01 * Y1p + rdry02 * Z1p;
                Y1cd = rdry10 * X1p + rdry11 * Y1p + rdry12 * Z1p;
                Z1cd = rdry20 * X1p + rdry21 * Y1p + rdry22 * Z1p;
                break;
            case 2: X1cd = rdrz00 * X1p + rdrz01 * Y1p;
                Y1cd = rdrz10 * X1p + rdrz11 * Y1p;
                Z1cd = rdrz20 * X1p + rdrz21 * Y1p;
                break;
            case 3: X1cd = 1; Y1cd = 0; Z1cd = 0; break;
            case 4: X1cd = 0; Y1cd = 1; Z1cd = 0; break;
            case 5: X1cd = 0; Y1cd = 0; Z1cd = 1; break;
            }
            J[(4 * i + 0) * 6 + j] = weight * param->f * (X1cd * Z1c - X1c * Z1cd) / (Z1c * Z1c);
            J[(4 * i + 1) * 6 + j] = weight * param->f * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
            J[(4 * i + 2) * 6 + j] = weight * param->f * (X1cd * Z1c - X2c * Z1cd) / (Z1c * Z1c);
            J[(4 * i + 3) * 6 + j] = weight * param->f * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
        }
        predict[0 * n + i] = param->f * X1c / Z1c + param->cu;
        predict[1 * n + i] = param->f * Y1c / Z1c + param->cv;
        predict[2 * n + i] = param->f * X2c / Z1c + param->cu;
        predict[3 * n + i] = param->f * Y1c / Z1c + param->cv;

        residual[4 * i + 0] = weight * (obs[0 * m + active[i]] - predict[0 * n + i]);
        residual[4 * i + 1] = weight * (obs[1 * m + active[i]] - predict[1 * n + i]);
        residual[4 * i + 2] = weight * (obs[2 * m + active[i]] - predict[2 * n + i]);
        residual[4 * i + 3] = weight * (obs[3 * m + active[i]] - predict[3 * n + i]);
    }
}

/* ------------------------------------------------------------ LU (cv::solve) */
/* OpenCV 3.0 modules/core/src/lapack.cpp LUImpl<double> restated (the
 * reference's build points at /home/kreimer/opencv3.0, src/CMakeLists.txt:1).
 * Partial pivoting by first strict maximum; singular iff |pivot| < DBL_EPSILON;
 * multiplies by d = -1/pivot; back substitution multiplies by the stored
 * reciprocal. */
int oracle_lu_solve6(double* A, double* b) {
    const int m = 6;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
        if (fabs(A[k * m + i]) < DBL_EPSILON) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) { double t = A[i * m + j]; A[i * m + j] = A[k * m + j]; A[k * m + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / A[i * m + i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j * m + i] * d;
            for (int c = i + 1; c < m; c++) A[j * m + c] += alpha * A[i * m + c];
            b[j] += alpha * b[i];
        }
        A[i * m + i] = -d;
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int c = i + 1; c < m; c++) s -= A[i * m + c] * b[c];
        b[i] = s * A[i * m + i];
    }
    return 1;
}

/* -------------------------------------------------------- minimize_reproj */
int oracle_minimize_reproj(const double* X, const double* obs, int m, double tr[6],
                           const viso_param* param, const int32_t* active, int n, int* iters) {
    if (n <= 0) { if (iters) *iters = 0; return 0; }
    double* J = (double*)malloc(sizeof(double) * (size_t)n * 4 * 6);
    double* residual = (double*)malloc(sizeof(double) * (size_t)n * 4);
    double* predict = (double*)malloc(sizeof(double) * (size_t)n * 4);
    int ret = 0, it = 0;
    const double step_size = 1.0f;
    for (int i = 0; i < 100; ++i) {
        oracle_compute_J(X, obs, m, tr, param, active, n, J, predict, residual);
        ++it;
        double JtJ[36], p_gn[6];
        /* mulTransposed(J,JtJ,true) and J.t()*residual: plain f64 sums in row order */
        for (int a = 0; a < 6; ++a) {
            for (int b = a; b < 6; ++b) {
                double s = 0;
                for (int k = 0; k < 4 * n; ++k) s += J[k * 6 + a] * J[k * 6 + b];
                JtJ[a * 6 + b] = s;
                JtJ[b * 6 + a] = s;
            }
            double s = 0;
            for (int k = 0; k < 4 * n; ++k) s += J[k * 6 + a] * residual[k];
            p_gn[a] = s;
        }
        if (!oracle_lu_solve6(JtJ, p_gn)) { ret = 0; goto done; } /* :1602-1606 */
        int converged = 1;
        for (int j = 0; j < 6; ++j) {
            /* :1610 — Q7: fabs(p > thresh): a bool goes through fabs */
            if (fabs((double)(p_gn[j] > param->thresh))) { converged = 0; break; }
        }
        if (converged) { ret = 1; goto done; } /* :1616-1617, step NOT applied */
        for (int j = 0; j < 6; ++j) tr[j] = tr[j] + step_size * p_gn[j];
    }
    ret = 0; /* :1622 */
done:
    if (iters) *iters = it;
    free(J); free(residual); free(predict);
    return ret;
}

/* ------------------------------------------------------------ get_inliers */
int oracle_get_inliers(const double* X, const double* obs, int m, const double tr[6],
                       const viso_param* param, int32_t* inliers, int* n_inl, double* rms) {
    int n = 0;
    double err2 = 0;
    if (m > 0) {
        int32_t* active = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
        double* J = (double*)malloc(sizeof(double) * (size_t)m * 4 * 6); /* "will not be used", :1517 */
        double* residual = (double*)malloc(sizeof(double) * (size_t)m * 4);
        double* predict = (double*)malloc(sizeof(double) * (size_t)m * 4);
        for (int i = 0; i < m; ++i) active[i] = i;
        oracle_compute_J(X, obs, m, tr, param, active, m, J, predict, residual);
        for (int i = 0; i < m; ++i) {
            /* pow(x,2) == x*x (gcc expands integer exponent 2 exactly) */
            double e0 = obs[0 * m + i] - predict[0 * m + i];
            double e1 = obs[1 * m + i] - predict[1 * m + i];
            double e2 = obs[2 * m + i] - predict[2 * m + i];
            double e3 = obs[3 * m + i] - predict[3 * m + i];
            err2 = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
            if (err2 < param->inlier_threshold * param->inlier_threshold) inliers[n++] = i;
        }
        free(active); free(J); free(residual); free(predict);
    }
    *n_inl = n;
    if (rms) *rms = sqrt(err2 / m); /* Q8: error of the LAST point only */
    return VISO_OK;
}

/* ---------------------------------------------------------------- RANSAC */
static inline uint64_t splitmix64_next(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* randomsample(3,N,.) src/viso.cpp:87-107 returns a uniformly distributed 3-subset of 0..N-1 in ascending order
 * (Knuth's selection sampling, algorithm S) from a per-call random_device/mt19937 (Q9: irreproducible).  WHICH
 * generator drives it is therefore this build's definition, not the reference's; since round 6 the definition is: the
 * first three outputs of a splitmix64 stream keyed on (seed, frame, hypothesis) -- both sides draw the same triples and
 * the result does not depend on how frames are partitioned over ranks -- through Floyd's subset sampling (three draws,
 * whatever N: algorithm S walks ~N/2 candidates per triple), then sorted ascending:
 *     for i = 0, 1, 2:  j = N - 3 + i;  t = floor(draw_i * (j + 1) / 2^64)  (uniform on 0..j);  pick t, or j if t was picked
 * Every 3-subset has probability 1 / C(N,3) up to the 2^-64 granularity of the draws: the distribution of
 * randomsample(3, N) (tests/test_oracle.py checks subsets and marginals empirically).  Rounds 1-5 ran algorithm S over
 * the same stream; the triples differ, their distribution does not. */
static void sample3_floyd(uint64_t seed, uint64_t frame, int h, int N, int32_t out[3]) {
    uint64_t s = seed ^ (0xD1B54A32D192ED03ULL * (frame + 1)) ^ (0x8CB92BA72F3D8DD7ULL * ((uint64_t)h + 1));
    out[0] = out[1] = out[2] = 0;
    if (N < 3) return;
    int32_t pick[3];
    for (int i = 0; i < 3; ++i) {
        const uint64_t j = (uint64_t)(N - 3 + i);
        const int32_t t = (int32_t)(((unsigned __int128)splitmix64_next(&s) * (j + 1)) >> 64);
        int taken = 0;
        for (int k = 0; k < i; ++k) taken |= pick[k] == t;
        pick[i] = taken ? (int32_t)j : t;
    }
    /* ascending, like randomsample's output */
    if (pick[0] > pick[1]) { int32_t x = pick[0]; pick[0] = pick[1]; pick[1] = x; }
    if (pick[1] > pick[2]) { int32_t x = pick[1]; pick[1] = pick[2]; pick[2] = x; }
    if (pick[0] > pick[1]) { int32_t x = pick[0]; pick[0] = pick[1]; pick[1] = x; }
    out[0] = pick[0]; out[1] = pick[1]; out[2] = pick[2];
}

void oracle_ransac_samples(uint64_t seed, uint64_t frame, int iters, int N, int32_t* out) {
    for (int h = 0; h < iters; ++h) sample3_floyd(seed, frame, h, N, out + 3 * h);
}

/* The literal algorithm S of src/viso.cpp:87-107 over the same stream (one uniform double per candidate index): the
 * rounds 1-5 definition, kept for the distribution test that compares the two samplers. */
void oracle_ransac_samples_algorithm_s(uint64_t seed, uint64_t frame, int iters, int N, int32_t* out) {
    for (int h = 0; h < iters; ++h) {
        uint64_t s = seed ^ (0xD1B54A32D192ED03ULL * (frame + 1)) ^ (0x8CB92BA72F3D8DD7ULL * ((uint64_t)h + 1));
        int n = 3, t = 0, m = 0;
        if (N < n) { out[3 * h] = out[3 * h + 1] = out[3 * h + 2] = 0; continue; }
        while (m < n) {
            double u = (double)(splitmix64_next(&s) >> 11) * (1.0 / 9007199254740992.0);
            if ((N - t) * u >= n - m) {
                t++;
            } else {
                out[3 * h + m] = t;
                t++; m++;
            }
        }
    }
}

int oracle_ransac_minimize_reproj(const double* X, const double* obs, int m,
                                  double best_tr[6], int32_t* best_inl, int* n_inl,
                                  const viso_param* param, const int32_t* samples,
                                  uint64_t seed, uint64_t frame) {
    *n_inl = 0;
    if (m < 3) return 0; /* the reference's randomsample would never return; sequence_odometry guards (:1283) */
    const int iters = param->ransac_iter;
    int32_t* own = NULL;
    if (!samples) {
        own = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(iters > 0 ? iters : 1));
        oracle_ransac_samples(seed, frame, iters, m, own);
        samples = own;
    }
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
    int best_n = 0;
    for (int i = 0; i < iters; ++i) {
        double tr[6] = {0, 0, 0, 0, 0, 0};
        if (!oracle_minimize_reproj(X, obs, m, tr, param, samples + 3 * i, 3, NULL)) continue;
        int n = 0;
        oracle_get_inliers(X, obs, m, tr, param, cur, &n, NULL);
        if (n > best_n) { /* strict: the first of equals is kept */
            best_n = n;
            memcpy(best_inl, cur, sizeof(int32_t) * (size_t)n);
            memcpy(best_tr, tr, sizeof(tr));
        }
    }
    free(cur);
    free(own);
    *n_inl = best_n;
    if (best_n < 6 || !oracle_minimize_reproj(X, obs, m, best_tr, param, best_inl, best_n, NULL))
        return 0;
    oracle_get_inliers(X, obs, m, best_tr, param, best_inl, n_inl, NULL);
    return 1;
}

/* ------------------------------------------------------------ pose algebra */
void oracle_tr2mat(const double tr[6], double T[16]) {
    double rx = tr[0], ry = tr[1], rz = tr[2], tx = tr[3], ty = tr[4], tz = tr[5];
    double sx = sin(rx), cx = cos(rx), sy = sin(ry), cy = cos(ry), sz = sin(rz), cz = cos(rz);
    T[0] = +cy * cz;                 T[1] = -cy * sz;                 T[2] = +sy;       T[3] = tx;
    T[4] = +sx * sy * cz + cx * sz;  T[5] = -sx * sy * sz + cx * cz;  T[6] = -sx * cy;  T[7] = ty;
    T[8] = -cx * sy * cz + sx * sz;  T[9] = +cx * sy * sz + sx * cz;  T[10] = +cx * cy; T[11] = tz;
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

/* generic n x n inverse by Gauss-Jordan with partial pivoting (cv::Mat::inv()
 * default DECOMP_LU; any correct inverse is within the 1e-5 pose tolerance) */
static int invert4(const double A[16], double Ainv[16]) {
    double M[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { M[i][j] = A[4 * i + j]; M[i][4 + j] = (i == j); }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (fabs(M[p][c]) < DBL_EPSILON) return 0;
        if (p != c) for (int j = 0; j < 8; ++j) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
        double inv = 1.0 / M[c][c];
        for (int j = 0; j < 8; ++j) M[c][j] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) {
            double f = M[r][c];
            for (int j = 0; j < 8; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Ainv[4 * i + j] = M[i][4 + j];
    return 1;
}

void oracle_pose_update(const double pose[16], const double tr[6], double out[16]) {
    double T[16], Ti[16], r[16];
    oracle_tr2mat(tr, T);
    if (!invert4(T, Ti)) { memcpy(out, pose, sizeof(r)); return; }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += pose[4 * i + k] * Ti[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(out, r, sizeof(r));
}

static double det4(const double* r0, const double* r1, const double* r2, const double* r3) {
    double M[4][4];
    memcpy(M[0], r0, 32); memcpy(M[1], r1, 32); memcpy(M[2], r2, 32); memcpy(M[3], r3, 32);
    double det = 1;
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (M[p][c] == 0) return 0;
        if (p != c) { for (int j = 0; j < 4; ++j) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; } det = -det; }
        det *= M[c][c];
        for (int r = c + 1; r < 4; ++r) {
            double f = M[r][c] / M[c][c];
            for (int j = c; j < 4; ++j) M[r][j] -= f * M[c][j];
        }
    }
    return det;
}

/* src/mvg.h:41-66: F(i,j) = det [ P1 without row j ; P2 without row i ] with the
 * cyclic row orders {1,2},{2,0},{0,1}; then src/viso.cpp:1177-1180. */
void oracle_F_from_P(const double P1[12], const double P2[12], double F[9]) {
    static const int pick[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    for (int i = 0; i < 3; ++i)      /* Y index (rows of P2) */
        for (int j = 0; j < 3; ++j)  /* X index (rows of P1) */
            F[3 * i + j] = det4(P1 + 4 * pick[j][0], P1 + 4 * pick[j][1],
                                P2 + 4 * pick[i][0], P2 + 4 * pick[i][1]);
    if (F[8] > DBL_MIN) {
        const double s = F[8];
        for (int k = 0; k < 9; ++k) F[k] /= s;
    }
}

/* ---------------------------------------------------------- descriptors */
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

int oracle_extract_descriptors(const uint8_t* img, int rows, int cols,
                               const float* kp, int n, int radius, float* desc) {
    if (rows <= 0 || cols <= 0 || radius < 0) return VISO_ERR_ARG;
    const int side = 2 * radius + 1, dlen = side * side;
    float* sob = (float*)malloc(sizeof(float) * (size_t)rows * cols);
    if (!sob) return VISO_ERR_NOMEM;
    /* cv::Sobel(image, sob, CV_32F, 1, 0, 3, 1, 0, BORDER_REFLECT_101) :1010 */
    for (int y = 0; y < rows; ++y) {
        const uint8_t* r0 = img + (size_t)reflect101(y - 1, rows) * cols;
        const uint8_t* r1 = img + (size_t)y * cols;
        const uint8_t* r2 = img + (size_t)reflect101(y + 1, rows) * cols;
        for (int x = 0; x < cols; ++x) {
            int xm = reflect101(x - 1, cols), xp = reflect101(x + 1, cols);
            int v = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
            sob[(size_t)y * cols + x] = (float)v;
        }
    }
    for (int k = 0; k < n; ++k) {
        /* Point2i p = kp.pt (:1013): saturate_cast<int>(float) == cvRound */
        int px = (int)lrintf(kp[2 * k]), py = (int)lrintf(kp[2 * k + 1]);
        int col = 0;
        for (int i = -radius; i <= radius; ++i)
            for (int j = -radius; j <= radius; ++j, ++col) {
                int y = py + i, x = px + j;
                /* :1018 — strict > 0: row 0 and column 0 count as outside */
                float val = (y > 0 && y < rows && x > 0 && x < cols) ? sob[(size_t)y * cols + x] : 0.f;
                desc[(size_t)k * dlen + col] = val;
            }
    }
    free(sob);
    return VISO_OK;
}

/* ------------------------------------------------------- Harris (binned) */
/* HarrisBinnedFeatureDetector::detectImpl, reference src/viso.cpp:926-975, with
 * cv::cornerHarris(image, R, blockSize=3, ksize=5, k, BORDER_DEFAULT) restated from OpenCV's published algorithm
 * (imgproc corner.cpp / deriv.cpp / filter.cpp / smooth.cpp of the 3.0 line; un-vendored, the reference's CMake points at
 * an OpenCV 3.0 tree), IN OPENCV'S EVALUATION ORDER as far as that order is a function of the pixel's neighbourhood:
 *   scale = 1 / (2^(ksize-1) * blockSize * 255) (double), handed to Sobel(..., CV_32F, ..., scale), which multiplies
 *     the SMOOTHING kernel by it (deriv.cpp: "if( dx == 0 ) kx *= scale; else ky *= scale;", Mat *= double on a CV_32F
 *     kernel = convertTo with the factor cast to float: tap_i = fl32(s_i * fl32(scale)), s = [1,4,6,4,1]);
 *   sepFilter2D, 8U -> 32F, float row buffer, BORDER_REFLECT_101 on the source:
 *     Dx: row pass with the derivative kernel [-1,-2,0,2,1] (RowFilter<uchar,float>: k0*S0, += k1*S1, ... left to right;
 *         exact, the values are small integers), column pass with the scaled symmetric smoothing kernel
 *         (SymmColumnFilter: f0*S0 + delta, += f1*(S1 + S-1), += f2*(S2 + S-2));
 *     Dy: row pass with the scaled smoothing kernel (five float products added left to right), column pass with the
 *         anti-symmetric derivative kernel (SymmColumnFilter: delta, += 2*(S1 - S-1), += 1*(S2 - S-2));
 *   cov = (dx*dx, dx*dy, dy*dy) in float;
 *   boxFilter(cov, 3x3, normalize = false, BORDER_REFLECT_101 on cov): row sums (S[x-1] + S[x]) + S[x+1], then column
 *     sums (rs[y-1] + rs[y]) + rs[y+1];
 *   R = (float)(a*c - b*b - k*(a+c)*(a+c)): a*c, b*b and their difference in float, k double.
 * What is NOT reproducible from the algorithm alone: OpenCV's ColumnSum keeps RUNNING column sums down the image
 * (SUM += new row; out = SUM; SUM -= old row), so its float rounding depends on every row above in the stripe its
 * thread was given; the restatement takes the local sum.  Harris parity with a real OpenCV stays unpinned (no OpenCV
 * here); the previous restatement (exact integer Sobel sums times the scale, nine taps row-major) is kept as
 * oracle_harris_response_v1 for comparison: the two agree to ~1e-6 relative.
 * The reference never initialises its k (src/viso.cpp:915-919,978): k is an explicit input here. */
static void harris_taps(float tap[5]) {
    static const float sm[5] = {1.f, 4.f, 6.f, 4.f, 1.f};
    const float scale = (float)(1.0 / (16.0 * 3.0 * 255.0));
    for (int i = 0; i < 5; ++i) tap[i] = sm[i] * scale;
}

static void harris_cov(const uint8_t* img, int rows, int cols, float* cov /* rows*cols*3 */) {
    float tap[5];
    harris_taps(tap);
    float* H = (float*)malloc(sizeof(float) * (size_t)rows * cols);   /* row pass, derivative kernel */
    float* G = (float*)malloc(sizeof(float) * (size_t)rows * cols);   /* row pass, scaled smoothing kernel */
    for (int y = 0; y < rows; ++y) {
        const uint8_t* r = img + (size_t)y * cols;
        for (int x = 0; x < cols; ++x) {
            float p[5];
            for (int j = 0; j < 5; ++j) p[j] = (float)r[reflect101(x + j - 2, cols)];
            float h = -1.f * p[0];
            h += -2.f * p[1]; h += 0.f * p[2]; h += 2.f * p[3]; h += 1.f * p[4];
            float g = tap[0] * p[0];
            g += tap[1] * p[1]; g += tap[2] * p[2]; g += tap[3] * p[3]; g += tap[4] * p[4];
            H[(size_t)y * cols + x] = h;
            G[(size_t)y * cols + x] = g;
        }
    }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const size_t y0 = (size_t)y * cols + x;
            const size_t ym1 = (size_t)reflect101(y - 1, rows) * cols + x, yp1 = (size_t)reflect101(y + 1, rows) * cols + x;
            const size_t ym2 = (size_t)reflect101(y - 2, rows) * cols + x, yp2 = (size_t)reflect101(y + 2, rows) * cols + x;
            float dx = tap[2] * H[y0] + 0.f;
            dx += tap[3] * (H[yp1] + H[ym1]);
            dx += tap[4] * (H[yp2] + H[ym2]);
            float dy = 0.f;
            dy += 2.f * (G[yp1] - G[ym1]);
            dy += 1.f * (G[yp2] - G[ym2]);
            float* c = cov + y0 * 3;
            c[0] = dx * dx; c[1] = dx * dy; c[2] = dy * dy;
        }
    free(H); free(G);
}

int oracle_harris_response(const uint8_t* img, int rows, int cols, double k, float* resp) {
    if (rows <= 0 || cols <= 0) return VISO_ERR_ARG;
    float* cov = (float*)malloc(sizeof(float) * (size_t)rows * cols * 3);
    float* rs = (float*)malloc(sizeof(float) * (size_t)rows * cols * 3);
    if (!cov || !rs) { free(cov); free(rs); return VISO_ERR_NOMEM; }
    harris_cov(img, rows, cols, cov);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float* l = cov + ((size_t)y * cols + reflect101(x - 1, cols)) * 3;
            const float* m = cov + ((size_t)y * cols + x) * 3;
            const float* r = cov + ((size_t)y * cols + reflect101(x + 1, cols)) * 3;
            float* o = rs + ((size_t)y * cols + x) * 3;
            for (int ch = 0; ch < 3; ++ch) o[ch] = (l[ch] + m[ch]) + r[ch];
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float* u = rs + ((size_t)reflect101(y - 1, rows) * cols + x) * 3;
            const float* m = rs + ((size_t)y * cols + x) * 3;
            const float* d = rs + ((size_t)reflect101(y + 1, rows) * cols + x) * 3;
            const float a = (u[0] + m[0]) + d[0], b = (u[1] + m[1]) + d[1], c = (u[2] + m[2]) + d[2];
            const float t1 = a * c, t2 = b * b;
            const float t3 = t1 - t2;
            const float tr = a + c;
            resp[(size_t)y * cols + x] = (float)((double)t3 - k * (double)tr * (double)tr);
        }
    free(cov); free(rs);
    return VISO_OK;
}

/* The previous restatement (rounds 1-3): exact integer 5x5 Sobel sums, dx = (float)Dx * (float)scale, the nine box
 * taps added row-major.  Kept for the cross-check in tests/test_oracle.py. */
int oracle_harris_response_v1(const uint8_t* img, int rows, int cols, double k, float* resp) {
    static const int d[5] = {-1, -2, 0, 2, 1}, sm[5] = {1, 4, 6, 4, 1};
    if (rows <= 0 || cols <= 0) return VISO_ERR_ARG;
    float* cov = (float*)malloc(sizeof(float) * (size_t)rows * cols * 3);
    if (!cov) return VISO_ERR_NOMEM;
    const float scale = (float)(1.0 / (16.0 * 3.0 * 255.0));
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int Dx = 0, Dy = 0;
            for (int i = -2; i <= 2; ++i) {
                const uint8_t* r = img + (size_t)reflect101(y + i, rows) * cols;
                int hd = 0, hs = 0;
                for (int j = -2; j <= 2; ++j) {
                    const int p = r[reflect101(x + j, cols)];
                    hd += d[j + 2] * p;
                    hs += sm[j + 2] * p;
                }
                Dx += sm[i + 2] * hd;
                Dy += d[i + 2] * hs;
            }
            const float dx = (float)Dx * scale, dy = (float)Dy * scale;
            float* c = cov + ((size_t)y * cols + x) * 3;
            c[0] = dx * dx; c[1] = dx * dy; c[2] = dy * dy;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float a = 0.f, b = 0.f, c = 0.f;
            for (int i = -1; i <= 1; ++i)
                for (int j = -1; j <= 1; ++j) {
                    const float* q = cov + ((size_t)reflect101(y + i, rows) * cols + reflect101(x + j, cols)) * 3;
                    a += q[0]; b += q[1]; c += q[2];
                }
            const float t1 = a * c, t2 = b * b;
            const float t3 = t1 - t2;
            const float tr = a + c;
            resp[(size_t)y * cols + x] = (float)((double)t3 - k * (double)tr * (double)tr);
        }
    free(cov);
    return VISO_OK;
}

typedef struct { float v; int pos; int x, y; } harris_elem;
static int cmp_harris(const void* pa, const void* pb) {
    const harris_elem* a = (const harris_elem*)pa;
    const harris_elem* b = (const harris_elem*)pb;
    if (a->v != b->v) return a->v > b->v ? -1 : 1;   /* |response| descending */
    return (a->pos > b->pos) - (a->pos < b->pos);    /* then the reference's push order (x outer, y inner) */
}

/* src/viso.cpp:931-975.  kp: up to n_features x 2 (x,y); resp_out (may be NULL): |response| per keypoint.
 * Bins in (binx outer, biny inner) order; within a bin the corners_per_block largest
 * |response| != 0, ordered by (|response| desc, push order asc) — the reference's
 * nth_element leaves that order unspecified (:963). */
int oracle_detect_harris_binned(const uint8_t* img, int rows, int cols, int n_features, int nbinx, int nbiny,
                                double k, float* kp, float* resp_out, int* n_out) {
    if (rows <= 0 || cols <= 0 || nbinx <= 0 || nbiny <= 0 || n_features < 0) return VISO_ERR_ARG;
    const int stridex = cols / nbinx, stridey = rows / nbiny;
    if (stridex <= 0 || stridey <= 0) return VISO_ERR_ARG;       /* assert(stridex>0 && stridey>0), :934 */
    const int per = n_features / (nbinx * nbiny);                /* corners_per_block, :943 */
    float* resp = (float*)malloc(sizeof(float) * (size_t)rows * cols);
    harris_elem* v = (harris_elem*)malloc(sizeof(harris_elem) * (size_t)stridex * stridey);
    if (!resp || !v) { free(resp); free(v); return VISO_ERR_NOMEM; }
    oracle_harris_response(img, rows, cols, k, resp);
    int n = 0;
    for (int bx = 0; bx < nbinx; ++bx)
        for (int by = 0; by < nbiny; ++by) {
            int cnt = 0;
            for (int x = bx * stridex; x < (bx + 1) * stridex && x < cols; ++x)
                for (int y = by * stridey; y < (by + 1) * stridey && y < rows; ++y) {
                    const float r = fabsf(resp[(size_t)y * cols + x]);
                    if (fabsf(r - 0.f) <= 1e-6f * fabsf(r)) continue;          /* isEqual(response, .0f), src/misc.cpp:10-14 */
                    v[cnt].v = r; v[cnt].pos = cnt; v[cnt].x = x; v[cnt].y = y;
                    ++cnt;
                }
            /* pos must be the push index among ALL pixels of the bin (not only the kept ones) to
             * be reproducible on the device: recompute it from coordinates */
            for (int i = 0; i < cnt; ++i) v[i].pos = (v[i].x - bx * stridex) * stridey + (v[i].y - by * stridey);
            qsort(v, (size_t)cnt, sizeof(harris_elem), cmp_harris);
            for (int i = 0; i < cnt && i < per; ++i) {
                kp[2 * n] = (float)v[i].x; kp[2 * n + 1] = (float)v[i].y;
                if (resp_out) resp_out[n] = v[i].v;
                ++n;
            }
        }
    *n_out = n;
    free(resp); free(v);
    return VISO_OK;
}

/* ---------------------------------------------- sequence_odometry loop body */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int oracle_sequence(const float* kp, const float* desc, const int32_t* n,
                    int nf, int cap, int dlen,
                    const viso_match_params* stereo, const viso_match_params* temporal,
                    const viso_param* p, uint64_t seed, uint64_t first_frame,
                    int matcher_only, double* tr_out, int32_t* ok, int32_t* n_inl_out,
                    int64_t* scored, int64_t* m_out, double* stage_s) {
    const size_t kps = (size_t)cap * 2, dss = (size_t)cap * dlen;
    int32_t* mlr = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)cap);
    int32_t* mlr_prev = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)cap);
    int32_t* m11 = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)cap);
    int32_t* m22 = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)cap);
    int32_t* circ = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)cap);
    int32_t* pcl = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)cap);
    int32_t* inl = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    double* x = (double*)malloc(sizeof(double) * 4 * (size_t)cap);
    double* X = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
    double* X_prev = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
    double* x_c = (double*)malloc(sizeof(double) * 4 * (size_t)cap);
    double* Xp_c = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
    int n_lr = 0, n_lr_prev = 0;
    double st[5] = {0, 0, 0, 0, 0};   /* match_desc total, circle, gather+triangulate, RANSAC/GN, of [0]: neighbour search */
    g_search_s = 0;
    for (int t = 0; t < nf; ++t) {
        const float* kp1 = kp + ((size_t)t * 2 + 0) * kps;
        const float* kp2 = kp + ((size_t)t * 2 + 1) * kps;
        const float* d1 = desc + ((size_t)t * 2 + 0) * dss;
        const float* d2 = desc + ((size_t)t * 2 + 1) * dss;
        const int n1 = n[2 * t], n2 = n[2 * t + 1];
        for (int j = 0; j < 6; ++j) tr_out[6 * t + j] = 0;
        ok[t] = 0; n_inl_out[t] = 0;
        if (scored) { scored[0 * nf + t] = scored[1 * nf + t] = scored[2 * nf + t] = 0; }
        if (m_out) { m_out[0 * nf + t] = m_out[1 * nf + t] = m_out[2 * nf + t] = 0; }
        /* carry state, src/viso.cpp:1208-1222 */
        { int32_t* tmp = mlr_prev; mlr_prev = mlr; mlr = tmp; n_lr_prev = n_lr; }
        { double* tmp = X_prev; X_prev = X; X = tmp; }
        double t0 = now_s();
        int64_t sc = 0;
        oracle_match_desc(kp1, n1, kp2, n2, d1, d2, dlen, stereo, mlr, &n_lr, &sc); /* :1240 */
        if (scored) scored[0 * nf + t] = sc;
        if (m_out) m_out[0 * nf + t] = n_lr;
        st[0] += now_s() - t0; t0 = now_s();
        if (!matcher_only) {
            oracle_collect_matches(kp1, n1, kp2, n2, mlr, n_lr, x);                /* :1245 */
            oracle_triangulate_rectified(x, n_lr, p, X);                           /* :1247 */
        }
        st[2] += now_s() - t0;
        if (t == 0) continue;                                                      /* :1256-1260 */
        const float* kp1p = kp + ((size_t)(t - 1) * 2 + 0) * kps;
        const float* kp2p = kp + ((size_t)(t - 1) * 2 + 1) * kps;
        const float* d1p = desc + ((size_t)(t - 1) * 2 + 0) * dss;
        const float* d2p = desc + ((size_t)(t - 1) * 2 + 1) * dss;
        const int n1p = n[2 * (t - 1)], n2p = n[2 * (t - 1) + 1];
        int n11 = 0, n22 = 0;
        t0 = now_s();
        oracle_match_desc(kp1, n1, kp1p, n1p, d1, d1p, dlen, temporal, m11, &n11, &sc); /* :1264 */
        if (scored) scored[1 * nf + t] = sc;
        if (m_out) m_out[1 * nf + t] = n11;
        oracle_match_desc(kp2, n2, kp2p, n2p, d2, d2p, dlen, temporal, m22, &n22, &sc); /* :1275 */
        if (scored) scored[2 * nf + t] = sc;
        if (m_out) m_out[2 * nf + t] = n22;
        st[0] += now_s() - t0;
        if (matcher_only) continue;
        t0 = now_s();
        int nc = 0;
        oracle_match_circle(mlr, n_lr, mlr_prev, n_lr_prev, m11, n11, m22, n22, circ, pcl, cap, &nc); /* :1282 */
        st[1] += now_s() - t0;
        if (nc < 3) continue;                                                       /* :1283-1288 */
        t0 = now_s();
        for (int i = 0; i < nc; ++i) {                                              /* :1292-1305 */
            for (int r = 0; r < 4; ++r) x_c[r * nc + i] = x[r * n_lr + pcl[2 * i]];
            for (int r = 0; r < 3; ++r) Xp_c[r * nc + i] = X_prev[r * n_lr_prev + pcl[2 * i + 1]];
        }
        st[2] += now_s() - t0; t0 = now_s();
        double tr[6] = {0, 0, 0, 0, 0, 0};
        int ni = 0;
        int r = oracle_ransac_minimize_reproj(Xp_c, x_c, nc, tr, inl, &ni, p, NULL, seed,
                                              first_frame + (uint64_t)t);           /* :1313 */
        st[3] += now_s() - t0;
        ok[t] = r; n_inl_out[t] = ni;
        for (int j = 0; j < 6; ++j) tr_out[6 * t + j] = tr[j];
    }
    st[4] = g_search_s;
    if (stage_s) memcpy(stage_s, st, sizeof(st));
    free(mlr); free(mlr_prev); free(m11); free(m22); free(circ); free(pcl); free(inl);
    free(x); free(X); free(X_prev); free(x_c); free(Xp_c);
    return VISO_OK;
}
