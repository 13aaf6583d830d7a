// hostmath.cpp — once-per-call host arithmetic of the boundary: parameter
// constructors, tr2mat, the pose chain step, F_from_P and the RANSAC sample
// stream.  A handful of flops each; nothing here is worth a kernel launch.
#include "solver_dev.h"

#include <float.h>
#include <math.h>
#include <string.h>

extern "C" void viso_match_params_stereo(viso_match_params* mp, const double F[9]) {
    memset(mp, 0, sizeof(*mp));       // MatchParams(Mat F), reference src/viso.cpp:62-71
    mp->enforce_epipolar = 1;
    mp->sampson_thresh = 1;
    mp->enforce_2nd_best = 0;
    mp->ratio_2nd_best = .8;
    mp->max_neighbors = 200;
    mp->radius = 80;
    if (F) memcpy(mp->F, F, 9 * sizeof(double));
}

extern "C" void viso_match_params_temporal(viso_match_params* mp) {
    memset(mp, 0, sizeof(*mp));       // MatchParams(), reference src/viso.cpp:72-74
    mp->enforce_epipolar = 0;
    mp->enforce_2nd_best = 1;
    mp->ratio_2nd_best = .9;
    mp->max_neighbors = 250;
    mp->radius = 80;
}

extern "C" void viso_param_default(viso_param* p) {
    memset(p, 0, sizeof(*p));         // param(), reference src/viso.h:60
    p->ransac_iter = 50;
    p->inlier_threshold = 2;
    p->save_debug = 1;
    p->thresh = 1e-4;
}

// tr2mat, reference src/viso.cpp:109-133
extern "C" void viso_tr2mat(const double tr[6], double T[16]) {
    const double rx = tr[0], ry = tr[1], rz = tr[2], tx = tr[3], ty = tr[4], tz = tr[5];
    const double sx = sin(rx), cx = cos(rx), sy = sin(ry), cy = cos(ry), sz = sin(rz), cz = cos(rz);
    T[0] = +cy * cz;                 T[1] = -cy * sz;                 T[2] = +sy;       T[3] = tx;
    T[4] = +sx * sy * cz + cx * sz;  T[5] = -sx * sy * sz + cx * cz;  T[6] = -sx * cy;  T[7] = ty;
    T[8] = -cx * sy * cz + sx * sz;  T[9] = +cx * sy * sz + sx * cz;  T[10] = +cx * cy; T[11] = tz;
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// pose * inv(tr2mat(tr)), reference src/viso.cpp:1316-1319.  tr2mat is rigid, so
// the inverse is [R' | -R' t] (cv::Mat::inv's LU result agrees to round-off).
extern "C" void viso_pose_update(const double pose[16], const double tr[6], double out[16]) {
    double T[16], Ti[16], r[16];
    viso_tr2mat(tr, T);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Ti[4 * i + j] = T[4 * j + i];
        Ti[4 * i + 3] = -(T[0 * 4 + i] * T[3] + T[1 * 4 + i] * T[7] + T[2 * 4 + i] * T[11]);
    }
    Ti[12] = Ti[13] = Ti[14] = 0; Ti[15] = 1;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += pose[4 * i + k] * Ti[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(out, r, sizeof(r));
}

static double det3(const double a[3], const double b[3], const double c[3]) {
    return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) +
           a[2] * (b[0] * c[1] - b[1] * c[0]);
}

// 4x4 determinant by cofactor expansion along the first row
static double det4rows(const double* r0, const double* r1, const double* r2, const double* r3) {
    double d = 0;
    for (int c = 0; c < 4; ++c) {
        double a[3], b[3], e[3];
        int k = 0;
        for (int j = 0; j < 4; ++j) {
            if (j == c) continue;
            a[k] = r1[j]; b[k] = r2[j]; e[k] = r3[j];
            ++k;
        }
        const double minor = det3(a, b, e);
        d += ((c & 1) ? -1.0 : 1.0) * r0[c] * minor;
    }
    return d;
}

// F_from_P<double>, reference src/mvg.h:41-66, then src/viso.cpp:1177-1180.
extern "C" void viso_F_from_P(const double P1[12], const double P2[12], double F[9]) {
    static const int pick[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            F[3 * i + j] = det4rows(P1 + 4 * pick[j][0], P1 + 4 * pick[j][1],
                                    P2 + 4 * pick[i][0], P2 + 4 * pick[i][1]);
    if (F[8] > DBL_MIN) {
        const double s = F[8];
        for (int k = 0; k < 9; ++k) F[k] /= s;
    }
}

// randomsample(3, m, .) with a reproducible stream, reference src/viso.cpp:87-107
extern "C" void viso_ransac_samples(uint64_t seed, uint64_t frame, int iters, int m, int32_t* out) {
    for (int h = 0; h < iters; ++h) {
        int s3[3];
        viso_sample3(seed, frame, h, m, s3);
        out[3 * h] = s3[0]; out[3 * h + 1] = s3[1]; out[3 * h + 2] = s3[2];
    }
}
