// solver_dev.h — device functions of the stereo reprojection Gauss-Newton
// solver: compute_J / minimize_reproj / get_inliers (reference
// src/viso.cpp:1401-1497, 1583-1623, 1509-1537).  fp64 throughout; every
// expression keeps the reference's operand order (the library is built with
// -ffp-contract=off), so the only differences from the CPU path are the
// last-ulp behaviour of sin/cos and the summation tree of the block reduction.
#pragma once
#include "common.h"

#include <float.h>
#include <math.h>

struct RotDev {
    double r00, r01, r02, r10, r11, r12, r20, r21, r22;
    double rdrx10, rdrx11, rdrx12, rdrx20, rdrx21, rdrx22;
    double rdry00, rdry01, rdry02, rdry10, rdry11, rdry12, rdry20, rdry21, rdry22;
    double rdrz00, rdrz01, rdrz10, rdrz11, rdrz20, rdrz21;
    double tx, ty, tz;
};

// The rotation entries and their derivatives from the six sines / cosines (src/viso.cpp:1410-1424) — the ONE place
// this table exists in the library (make_rot and ransac_coop_kernel both call it).
__device__ __forceinline__ void rot_from_sincos(double sx, double cx, double sy, double cy, double sz, double cz,
                                                const double* tr, RotDev& R) {
    R.tx = tr[3]; R.ty = tr[4]; R.tz = tr[5];
    R.r00 = +cy * cz;                R.r01 = -cy * sz;                R.r02 = +sy;
    R.r10 = +sx * sy * cz + cx * sz; R.r11 = -sx * sy * sz + cx * cz; R.r12 = -sx * cy;
    R.r20 = -cx * sy * cz + sx * sz; R.r21 = +cx * sy * sz + sx * cz; R.r22 = +cx * cy;
    R.rdrx10 = +cx * sy * cz - sx * sz; R.rdrx11 = -cx * sy * sz - sx * cz; R.rdrx12 = -cx * cy;
    R.rdrx20 = +sx * sy * cz + cx * sz; R.rdrx21 = -sx * sy * sz + cx * cz; R.rdrx22 = -sx * cy;
    R.rdry00 = -sy * cz;      R.rdry01 = +sy * sz;      R.rdry02 = +cy;
    R.rdry10 = +sx * cy * cz; R.rdry11 = -sx * cy * sz; R.rdry12 = +sx * sy;
    R.rdry20 = -cx * cy * cz; R.rdry21 = +cx * cy * sz; R.rdry22 = -cx * sy;
    R.rdrz00 = -cy * sz;                R.rdrz01 = -cy * cz;
    R.rdrz10 = -sx * sy * sz + cx * cz; R.rdrz11 = -sx * sy * cz - cx * sz;
    R.rdrz20 = +cx * sy * sz + sx * cz; R.rdrz21 = +cx * sy * cz - sx * sz;
}

// src/viso.cpp:1405-1424
__device__ __forceinline__ void make_rot(const double* tr, RotDev& R) {
    double sx, cx, sy, cy, sz, cz;   // sincos shares the argument reduction of sin and cos
    sincos(tr[0], &sx, &cx);
    sincos(tr[1], &sy, &cy);
    sincos(tr[2], &sz, &cz);
    rot_from_sincos(sx, cx, sy, cy, sz, cz, tr, R);
}

// prediction of one point (src/viso.cpp:1441-1443, 1452, 1486-1489)
__device__ __forceinline__ void predict_point(const RotDev& R, const SolverParamsDev& sp,
                                              double X1p, double Y1p, double Z1p, double pred[4],
                                              double& X1c, double& Y1c, double& Z1c, double& X2c) {
    X1c = R.r00 * X1p + R.r01 * Y1p + R.r02 * Z1p + R.tx;
    Y1c = R.r10 * X1p + R.r11 * Y1p + R.r12 * Z1p + R.ty;
    Z1c = R.r20 * X1p + R.r21 * Y1p + R.r22 * Z1p + R.tz;
    X2c = X1c - sp.base;
    pred[0] = sp.f * X1c / Z1c + sp.cu;
    pred[1] = sp.f * Y1c / Z1c + sp.cv;
    pred[2] = sp.f * X2c / Z1c + sp.cu;
    pred[3] = sp.f * Y1c / Z1c + sp.cv;
}

// squared reprojection error test of get_inliers (src/viso.cpp:1524-1533)
// get_inliers' test of one point (src/viso.cpp:1527-1533): sum_k (observe_k - predict_k)^2 < inlier_threshold^2, strict.
// The reference's three divisions by Z (:1486-1489) cost ~30 fp64 instructions each here.  The verdict only needs them
// when the sum is within rounding distance of the threshold: first the sum with ONE reciprocal of Z (refined to < 1 ulp)
// and its error bound — f X rz is off by < 4 ulps of ITS magnitude (<= |predict| + |c|, c the principal point: the sum
// cancels near the image origin), the sum of squares by far less than 1e-12 (S + 1), S = sum_k (|observe_k| + |predict_k|
// + |c_k|)^2 —, and only a sum inside that band (or not a number) is decided
// by the reference's own expression.  Same verdicts, bit for bit; err2_out (the Q8 rms, :1535) always takes the exact path.
__device__ __forceinline__ bool is_inlier_pt(const RotDev& R, const SolverParamsDev& sp, double X0, double X1, double X2,
                                             double o0, double o1, double o2, double o3, double* err2_out) {
    const double thr2 = sp.inlier_threshold * sp.inlier_threshold;
    if (!err2_out) {
        const double X1c = R.r00 * X0 + R.r01 * X1 + R.r02 * X2 + R.tx;
        const double Y1c = R.r10 * X0 + R.r11 * X1 + R.r12 * X2 + R.ty;
        const double Z1c = R.r20 * X0 + R.r21 * X1 + R.r22 * X2 + R.tz;
        double rz = __builtin_amdgcn_rcp(Z1c);
        rz = fma(fma(-Z1c, rz, 1.0), rz, rz);          // two Newton steps: < 1 ulp from 1 / Z1c for any normal Z1c
        rz = fma(fma(-Z1c, rz, 1.0), rz, rz);
        const double p0 = sp.f * X1c * rz + sp.cu, p1 = sp.f * Y1c * rz + sp.cv, p2 = sp.f * (X1c - sp.base) * rz + sp.cu;
        const double e0 = o0 - p0, e1 = o1 - p1, e2 = o2 - p2, e3 = o3 - p1;
        const double approx = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
        // |cu|, |cv|: p = f X rz + c cancels when the projection lies near the image origin, and the rounding of
        // f X rz is relative to ITS magnitude (<= |p| + |c|), not to |p|
        const double acu = fabs(sp.cu), acv = fabs(sp.cv);
        const double a0 = fabs(o0) + fabs(p0) + acu, a1 = fabs(o1) + fabs(p1) + acv, a2 = fabs(o2) + fabs(p2) + acu, a3 = fabs(o3) + fabs(p1) + acv;
        const double band = 1e-12 * (a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 + 1.0);
        if (approx < thr2 - band) return true;         // any NaN / inf makes both tests false: the exact path decides
        if (approx > thr2 + band) return false;
    }
    double pred[4], X1c, Y1c, Z1c, X2c;
    predict_point(R, sp, X0, X1, X2, pred, X1c, Y1c, Z1c, X2c);
    const double e0 = o0 - pred[0];
    const double e1 = o1 - pred[1];
    const double e2 = o2 - pred[2];
    const double e3 = o3 - pred[3];
    const double err2 = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    if (err2_out) *err2_out = err2;
    return err2 < thr2;
}
__device__ __forceinline__ bool is_inlier(const RotDev& R, const SolverParamsDev& sp, const double* X,
                                          const double* obs, int ld, int i, double* err2_out) {
    return is_inlier_pt(R, sp, X[0 * ld + i], X[1 * ld + i], X[2 * ld + i],
                        obs[0 * ld + i], obs[1 * ld + i], obs[2 * ld + i], obs[3 * ld + i], err2_out);
}

// Adds the 4 Jacobian rows and residuals of one active point to the normal
// equations A (upper triangle, 21 sums) and B (6 sums), rows in the
// reference's order 4i..4i+3 (row 4i+3 equals row 4i+1, src/viso.cpp:1479,1481).
// `pos` is the position in the active list: the weight reads observe(0,pos),
// not observe(0,active[pos]) (Q6, src/viso.cpp:1449).
__device__ __forceinline__ void accumulate_point(const RotDev& R, const SolverParamsDev& sp,
                                                 const double* X, const double* obs, int ld,
                                                 int a, int pos, double A[6][6], double B[6]) {
    const double X1p = X[0 * ld + a], Y1p = X[1 * ld + a], Z1p = X[2 * ld + a];
    double pred[4], X1c, Y1c, Z1c, X2c;
    predict_point(R, sp, X1p, Y1p, Z1p, pred, X1c, Y1c, Z1c, X2c);
    const double weight = 1.0 / (fabs(obs[0 * ld + pos] - sp.cu) / fabs(sp.cu) + 0.05);
    const double wf = weight * sp.f, z2 = Z1c * Z1c;
    double Jr[3][6];   // rows u_left, v_left, u_right
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double X1cd, Y1cd, Z1cd;
        switch (j) {
        case 0: X1cd = 0;
            Y1cd = R.rdrx10 * X1p + R.rdrx11 * Y1p + R.rdrx12 * Z1p;
            Z1cd = R.rdrx20 * X1p + R.rdrx21 * Y1p + R.rdrx22 * Z1p;
            break;
        case 1: X1cd = R.rdry00 * X1p + R.rdry01 * Y1p + R.rdry02 * Z1p;
            Y1cd = R.rdry10 * X1p + R.rdry11 * Y1p + R.rdry12 * Z1p;
            Z1cd = R.rdry20 * X1p + R.rdry21 * Y1p + R.rdry22 * Z1p;
            break;
        case 2: X1cd = R.rdrz00 * X1p + R.rdrz01 * Y1p;
            Y1cd = R.rdrz10 * X1p + R.rdrz11 * Y1p;
            Z1cd = R.rdrz20 * X1p + R.rdrz21 * Y1p;
            break;
        case 3: X1cd = 1; Y1cd = 0; Z1cd = 0; break;
        case 4: X1cd = 0; Y1cd = 1; Z1cd = 0; break;
        default: X1cd = 0; Y1cd = 0; Z1cd = 1; break;
        }
        // weight*f*(..)/(Z1c*Z1c), src/viso.cpp:1478-1481: the reference's operand order, divisions included
        // (round 1 multiplied by one reciprocal of Z1c^2 instead: <= 1 ulp per entry, but the convergence test of
        // :1610 is a threshold on values derived from these)
        Jr[0][j] = wf * (X1cd * Z1c - X1c * Z1cd) / z2;
        Jr[1][j] = wf * (Y1cd * Z1c - Y1c * Z1cd) / z2;
        Jr[2][j] = wf * (X1cd * Z1c - X2c * Z1cd) / z2;
    }
    double res[4];
    res[0] = weight * (obs[0 * ld + a] - pred[0]);
    res[1] = weight * (obs[1 * ld + a] - pred[1]);
    res[2] = weight * (obs[2 * ld + a] - pred[2]);
    res[3] = weight * (obs[3 * ld + a] - pred[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int jr = (r == 3) ? 1 : r;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int q = p; q < 6; ++q) A[p][q] += Jr[jr][p] * Jr[jr][q];
            B[p] += Jr[jr][p] * res[r];
        }
    }
}

// The same sums for the all-inlier refit (gn_block), arranged for fewer live registers: the 21 + 6 sums as one flat
// array S (upper triangle row by row, then J^T r), ONE Jacobian row alive at a time, rows in the order u_left, u_right,
// v_left, v_right so that the row the two v observations share (src/viso.cpp:1479,1481) is built once and used twice.
// The refit's sums go through a workgroup reduction tree anyway: this path is compared with the CPU within the 1e-5 pose
// tolerance, not bit for bit (the 3-point hypotheses, whose iteration counts decide inlier sets, keep accumulate_point
// and the reference's every operation).  So the arithmetic here is the cheap one: a point's 18 Jacobian entries share ONE
// division (wf / Z1c^2; the reference divides each entry, :1478-1481 -- a division is a dozen dependent fp64
// instructions, and 17 of them were two fifths of this function), and the 27 sums take fused multiply-adds.  A single
// frame's refit of ~1200 inliers is bound by the fp64 rate of the ONE compute unit its workgroup runs on: 10.3 -> 4 us
// per Gauss-Newton iteration for this function.
__device__ __forceinline__ void accumulate_point_rows(const RotDev& R, const SolverParamsDev& sp,
                                                      const double* X, const double* obs, int ld,
                                                      int a, int pos, double (&S)[27]) {
    const double X1p = X[0 * ld + a], Y1p = X[1 * ld + a], Z1p = X[2 * ld + a];
    double pred[4], X1c, Y1c, Z1c, X2c;
    predict_point(R, sp, X1p, Y1p, Z1p, pred, X1c, Y1c, Z1c, X2c);
    const double weight = 1.0 / (fabs(obs[0 * ld + pos] - sp.cu) / fabs(sp.cu) + 0.05);   // Q6: position, not index
    const double wz = weight * sp.f / (Z1c * Z1c);
    const double res0 = weight * (obs[0 * ld + a] - pred[0]), res1 = weight * (obs[1 * ld + a] - pred[1]);
    const double res2 = weight * (obs[2 * ld + a] - pred[2]), res3 = weight * (obs[3 * ld + a] - pred[3]);
    // d(X1c, Y1c, Z1c) / d(rx, ry, rz, tx, ty, tz), the switch of accumulate_point
    const double Xd[6] = {0.0, R.rdry00 * X1p + R.rdry01 * Y1p + R.rdry02 * Z1p, R.rdrz00 * X1p + R.rdrz01 * Y1p, 1.0, 0.0, 0.0};
    const double Yd[6] = {R.rdrx10 * X1p + R.rdrx11 * Y1p + R.rdrx12 * Z1p, R.rdry10 * X1p + R.rdry11 * Y1p + R.rdry12 * Z1p,
                          R.rdrz10 * X1p + R.rdrz11 * Y1p, 0.0, 1.0, 0.0};
    const double Zd[6] = {R.rdrx20 * X1p + R.rdrx21 * Y1p + R.rdrx22 * Z1p, R.rdry20 * X1p + R.rdry21 * Y1p + R.rdry22 * Z1p,
                          R.rdrz20 * X1p + R.rdrz21 * Y1p, 0.0, 0.0, 1.0};
    double J[6];
    auto add_row = [&](double res) {
        int c = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int q = p; q < 6; ++q) { S[c] = fma(J[p], J[q], S[c]); ++c; }
            S[21 + p] = fma(J[p], res, S[21 + p]);
        }
    };
#pragma unroll
    for (int j = 0; j < 6; ++j) J[j] = (Xd[j] * Z1c - X1c * Zd[j]) * wz;   // u_left, :1478
    add_row(res0);
#pragma unroll
    for (int j = 0; j < 6; ++j) J[j] = (Xd[j] * Z1c - X2c * Zd[j]) * wz;   // u_right, :1480
    add_row(res2);
#pragma unroll
    for (int j = 0; j < 6; ++j) J[j] = (Yd[j] * Z1c - Y1c * Zd[j]) * wz;   // v_left = v_right, :1479,1481
    add_row(res1);
    add_row(res3);
}

// cv::solve(A, b, x, DECOMP_LU) for 6x6 (OpenCV 3.0 LUImpl: first strict
// maximum pivot, singular iff |pivot| < DBL_EPSILON).  All indices static so
// the matrix stays in registers; row swaps are predicated.
__device__ __forceinline__ int lu_solve6(double A[6][6], double b[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int k = i;
        double best = fabs(A[i][i]);
#pragma unroll
        for (int j = i + 1; j < 6; ++j) {
            const double v = fabs(A[j][i]);
            if (v > best) { best = v; k = j; }
        }
        if (best < DBL_EPSILON) return 0;
#pragma unroll
        for (int j = i + 1; j < 6; ++j) {   // selects, not a branch: the compiler turns a predicated swap into a
            const bool sw = k == j;         // dynamically indexed row access, i.e. the matrix into scratch memory
#pragma unroll
            for (int c = i; c < 6; ++c) {
                const double ti = A[i][c], tj = A[j][c];
                A[i][c] = sw ? tj : ti;
                A[j][c] = sw ? ti : tj;
            }
            const double ti = b[i], tj = b[j];
            b[i] = sw ? tj : ti;
            b[j] = sw ? ti : tj;
        }
        const double d = -1 / A[i][i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) {
            const double alpha = A[j][i] * d;
#pragma unroll
            for (int c = i + 1; c < 6; ++c) A[j][c] += alpha * A[i][c];
            b[j] += alpha * b[i];
        }
        A[i][i] = -d;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int c = i + 1; c < 6; ++c) s -= A[i][c] * b[c];
        b[i] = s * A[i][i];
    }
    return 1;
}

__device__ __forceinline__ void symmetrize(double A[6][6]) {
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = 0; q < p; ++q) A[p][q] = A[q][p];
}

// One thread runs the whole Gauss-Newton loop over `n` active points in the
// reference's summation order (used for the 3-point RANSAC hypotheses: one
// lane per hypothesis).  Returns 1 (converged) / 0; tr is in/out.
// Iterations [it_begin, it_end) of the reference's 100; returns 2 when it_end < 100 is reached without a
// verdict (tr then holds the state after it_end iterations: ransac_coop_kernel continues from there).
template <int N>
__device__ inline int gn_serial(const double* X, const double* obs, int ld, const int (&active)[N],
                                double tr[6], const SolverParamsDev& sp, int it_begin = 0, int it_end = 100) {
    for (int it = it_begin; it < it_end; ++it) {
        RotDev R;
        make_rot(tr, R);
        double A[6][6], B[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            B[p] = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) A[p][q] = 0;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) accumulate_point(R, sp, X, obs, ld, active[i], i, A, B);
        symmetrize(A);
        if (!lu_solve6(A, B)) return 0;           // src/viso.cpp:1602-1606
        bool converged = true;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (B[j] > sp.thresh) converged = false;   // Q7: fabs(p > thresh), :1610
        if (converged) return 1;                  // step not applied, :1616-1617
#pragma unroll
        for (int j = 0; j < 6; ++j) tr[j] = tr[j] + B[j];
    }
    return it_end >= 100 ? 0 : 2;                 // :1622
}

// ---- the sample triples: randomsample(3, N, .), src/viso.cpp:87-107 ----------------------------------------------------
// The reference returns a uniformly distributed 3-subset of 0..N-1, ascending, from a per-call random_device (Q9): which
// generator drives it is this build's definition.  Since round 6: the first three outputs of a splitmix64 stream keyed
// on (seed, frame, hypothesis) through Floyd's subset sampling -- pick t uniform on 0..j, or j itself if t is taken, for
// j = N-3, N-2, N-1 -- then sorted: the same distribution in O(1) (rounds 1-5 walked the reference's algorithm S over
// the stream: ~N/2 draws per triple, 16.9 M vector instructions per 25 600 triples in a kernel of its own).  Host twin
// viso_ransac_samples (hostmath.cpp), oracle twin oracle_ransac_samples: the same integers on every side.
__host__ __device__ inline unsigned long long viso_splitmix64(unsigned long long* s) {
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ inline unsigned long long viso_mulhi64(unsigned long long a, unsigned long long b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (unsigned long long)(((unsigned __int128)a * b) >> 64);
#endif
}

__host__ __device__ inline void viso_sample3(unsigned long long seed, unsigned long long frame, int h,
                                             int N, int out[3]) {
    unsigned long long s = seed ^ (0xD1B54A32D192ED03ULL * (frame + 1)) ^
                           (0x8CB92BA72F3D8DD7ULL * ((unsigned long long)h + 1));
    out[0] = out[1] = out[2] = 0;
    if (N < 3) return;
    const int a = (int)viso_mulhi64(viso_splitmix64(&s), (unsigned long long)(N - 2));            // uniform on 0..N-3
    int b = (int)viso_mulhi64(viso_splitmix64(&s), (unsigned long long)(N - 1));                  // 0..N-2
    int c = (int)viso_mulhi64(viso_splitmix64(&s), (unsigned long long)N);                        // 0..N-1
    b = b == a ? N - 2 : b;
    c = (c == a || c == b) ? N - 1 : c;
    // ascending, like randomsample's output: min / median / max
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    out[0] = lo < c ? lo : c;
    out[2] = hi > c ? hi : c;
    out[1] = a + b + c - out[0] - out[2];
}
