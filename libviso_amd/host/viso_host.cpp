// viso_host.cpp — implementation of the C++ host mirror (viso.hpp) on top of
// the C-ABI (include/viso_hip.h).  No arithmetic of the hot path happens here:
// this file reshapes containers into the ABI's plain arrays and chains poses.
#include "viso.hpp"
#ifdef __linux__
#include <sched.h>
#endif
#include <cstdio>
#include <cstring>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <memory>
#include <thread>

namespace viso {

static void hip_check(int r, const char* where) {
    if (r < 0) throw std::runtime_error(std::string(where) + ": " + viso_last_error());
}

static std::vector<float> kp2mat(const KeyPoints& kp) {   // src/viso.cpp:246-256
    std::vector<float> m(kp.size() * 2);
    for (size_t i = 0; i < kp.size(); ++i) { m[2 * i] = kp[i].pt.x; m[2 * i + 1] = kp[i].pt.y; }
    return m;
}

static viso_match_params to_abi(const MatchParams& sp) {
    viso_match_params mp;
    std::memset(&mp, 0, sizeof(mp));
    mp.enforce_epipolar = sp.enforce_epipolar;
    mp.enforce_2nd_best = sp.enforce_2nd_best;
    mp.max_neighbors = sp.max_neighbors;
    mp.sampson_thresh = sp.sampson_thresh;
    mp.ratio_2nd_best = sp.ratio_2nd_best;
    mp.radius = sp.radius;
    if (sp.enforce_epipolar) {
        if (sp.F.rows != 3 || sp.F.cols != 3) throw std::invalid_argument("MatchParams::F must be 3x3 double");
        for (int i = 0; i < 9; ++i) mp.F[i] = sp.F.data[i];
    }
    return mp;
}

static viso_param to_abi(const param& p) {
    viso_param q;
    std::memset(&q, 0, sizeof(q));
    q.base = p.base; q.ransac_iter = p.ransac_iter; q.save_debug = p.save_debug;
    q.inlier_threshold = p.inlier_threshold; q.thresh = p.thresh;
    q.f = p.calib.f; q.cu = p.calib.cu; q.cv = p.calib.cv;
    return q;
}

void match_desc(const KeyPoints& kp1, const KeyPoints& kp2, const Descriptors& d1, const Descriptors& d2,
                Matches& match, const MatchParams& sp) {
    match.clear();                                             // :675
    if (d1.cols != d2.cols && d1.rows && d2.rows) throw std::invalid_argument("match_desc: d1.cols != d2.cols");   // :676
    if ((int)kp1.size() != d1.rows || (int)kp2.size() != d2.rows) throw std::invalid_argument("match_desc: keypoint/descriptor count mismatch");
    const int dlen = d1.rows ? d1.cols : d2.cols;
    if (kp1.empty()) return;
    std::vector<float> k1 = kp2mat(kp1), k2 = kp2mat(kp2);
    // Match (Vec3i) is a contiguous int triple: the rows are written straight into the list (the adapter pushes them back one by one
    // only because cv::Vec3i has no such guarantee on paper)
    static_assert(sizeof(Match) == 3 * sizeof(int32_t), "Match must be three contiguous ints");
    match.resize(kp1.size());
    int n = 0;
    viso_match_params mp = to_abi(sp);
    hip_check(viso_match_desc(k1.data(), (int)kp1.size(), k2.data(), (int)kp2.size(), d1.ptr(), d2.ptr(),
                              dlen > 0 ? dlen : 1, &mp, &match[0][0], &n), "match_desc");
    match.resize((size_t)n);
}

static const int32_t* rows_of(const Matches& m) { return m.empty() ? nullptr : &m[0][0]; }   // adapters/viso_hip_adapter.inc: &v[0][0]

void match_circle(const Matches& match_lr, const Matches& match_lr_prev, const Matches& match11,
                  const Matches& match22, std::vector<Vec4i>& circ_match, Matches& match_pcl) {
    int cap = (int)match_lr.size() + 16, n = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        std::unique_ptr<int32_t[]> circ(new int32_t[(size_t)cap * 4]), pcl(new int32_t[(size_t)cap * 2]);
        int r = viso_match_circle(rows_of(match_lr), (int)match_lr.size(), rows_of(match_lr_prev), (int)match_lr_prev.size(),
                                  rows_of(match11), (int)match11.size(), rows_of(match22), (int)match22.size(),
                                  circ.get(), pcl.get(), cap, &n);
        if (r == VISO_ERR_ARG && n > cap) { cap = n; continue; }   // duplicate keys produced more rows: retry
        hip_check(r, "match_circle");
        // the reference appends (push_back) to both outputs, :233-234
        circ_match.reserve(circ_match.size() + (size_t)n);
        match_pcl.reserve(match_pcl.size() + (size_t)n);
        for (int i = 0; i < n; ++i) {
            circ_match.push_back({circ[4 * i], circ[4 * i + 1], circ[4 * i + 2], circ[4 * i + 3]});
            match_pcl.push_back({pcl[2 * i], pcl[2 * i + 1], 0});
        }
        return;
    }
    throw std::runtime_error("match_circle: capacity retry failed");
}

void collect_matches(const KeyPoints& kp1, const KeyPoints& kp2, const Matches& match, Matd& x) {
    x.create(4, (int)match.size());
    if (match.empty()) return;
    std::vector<float> k1 = kp2mat(kp1), k2 = kp2mat(kp2);
    int r = viso_collect_matches(k1.data(), (int)kp1.size(), k2.data(), (int)kp2.size(), rows_of(match),
                                 (int)match.size(), x.ptr());
    if (r == VISO_ERR_ARG) throw std::out_of_range("collect_matches: match index out of range");   // vector::at
    hip_check(r, "collect_matches");
}

Matd triangulate_rectified(const Matd& x, const param& p) {
    if (x.rows != 4) throw std::invalid_argument("triangulate_rectified: x must be 4 x M");
    Matd X(3, x.cols);
    viso_param q = to_abi(p);
    hip_check(viso_triangulate_rectified(x.ptr(), x.cols, &q, X.ptr()), "triangulate_rectified");
    return X;
}

bool minimize_reproj(const Matd& X, const Matd& observe, std::vector<double>& tr, const param& p,
                     const std::vector<int>& active) {
    if (X.rows != 3 || observe.rows != 4 || X.cols != observe.cols || tr.size() != 6)
        throw std::invalid_argument("minimize_reproj: X 3xM, observe 4xM, tr[6] expected");
    viso_param q = to_abi(p);
    std::vector<int32_t> a(active.begin(), active.end());
    int r = viso_minimize_reproj(X.ptr(), observe.ptr(), X.cols, tr.data(), &q, a.data(), (int)a.size());
    hip_check(r, "minimize_reproj");
    return r == 1;
}

bool ransac_minimize_reproj(const Matd& X, const Matd& observe, std::vector<double>& best_tr,
                            std::vector<int>& best_inliers, const param& p) {
    if (X.rows != 3 || observe.rows != 4 || X.cols != observe.cols)
        throw std::invalid_argument("ransac_minimize_reproj: X 3xM, observe 4xM expected");
    if (best_tr.size() != 6) best_tr.assign(6, 0.0);
    best_inliers.clear();                                      // :1554
    viso_param q = to_abi(p);
    std::vector<int32_t> inl((size_t)std::max(1, X.cols));
    int n = 0;
    int r = viso_ransac_minimize_reproj(X.ptr(), observe.ptr(), X.cols, best_tr.data(), inl.data(), &n, &q,
                                        nullptr, p.ransac_seed, p.frame_index);
    hip_check(r, "ransac_minimize_reproj");
    best_inliers.assign(inl.begin(), inl.begin() + n);
    return r == 1;
}

void tr2mat(std::vector<double> tr, Matd& Tr) {
    if (tr.size() != 6) throw std::invalid_argument("tr2mat: tr must have 6 entries");
    if (Tr.rows != 4 || Tr.cols != 4) Tr.create(4, 4);
    viso_tr2mat(tr.data(), Tr.ptr());
}

Matd F_from_P(const Matd& P1, const Matd& P2) {
    if (P1.rows != 3 || P1.cols != 4 || P2.rows != 3 || P2.cols != 4) throw std::invalid_argument("F_from_P: 3x4 expected");
    Matd F(3, 3);
    viso_F_from_P(P1.ptr(), P2.ptr(), F.ptr());
    return F;
}

// ---- 3x3 SVD by one-sided Jacobi (host; used only by solveRigidMotion) -----
static void svd3(const double C[9], double U[9], double S[3], double V[9]) {
    double A[9];
    std::memcpy(A, C, sizeof(A));
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int k = 0; k < 3; ++k) { a += A[3 * k + p] * A[3 * k + p]; b += A[3 * k + q] * A[3 * k + q]; c += A[3 * k + p] * A[3 * k + q]; }
                off = std::max(off, std::fabs(c) / std::sqrt(std::max(a * b, 1e-300)));
                if (std::fabs(c) < 1e-300) continue;
                const double zeta = (b - a) / (2 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                const double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
                for (int k = 0; k < 3; ++k) {
                    const double x = A[3 * k + p], y = A[3 * k + q];
                    A[3 * k + p] = cs * x - sn * y; A[3 * k + q] = sn * x + cs * y;
                    const double vx = V[3 * k + p], vy = V[3 * k + q];
                    V[3 * k + p] = cs * vx - sn * vy; V[3 * k + q] = sn * vx + cs * vy;
                }
            }
        if (off < 1e-15) break;
    }
    for (int j = 0; j < 3; ++j) {
        double n = 0;
        for (int k = 0; k < 3; ++k) n += A[3 * k + j] * A[3 * k + j];
        S[j] = std::sqrt(n);
    }
    // order by decreasing singular value (Eigen's JacobiSVD convention)
    int idx[3] = {0, 1, 2};
    std::sort(idx, idx + 3, [&](int a, int b) { return S[a] > S[b]; });
    double A2[9], V2[9], S2[3];
    for (int j = 0; j < 3; ++j) { S2[j] = S[idx[j]]; for (int k = 0; k < 3; ++k) { A2[3 * k + j] = A[3 * k + idx[j]]; V2[3 * k + j] = V[3 * k + idx[j]]; } }
    std::memcpy(V, V2, sizeof(V2)); std::memcpy(S, S2, sizeof(S2));
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) U[3 * k + j] = S[j] > 1e-300 ? A2[3 * k + j] / S[j] : 0.0;
    // complete a rank-deficient U with cross products so that it stays orthonormal
    auto col = [&](int j, double out[3]) { for (int k = 0; k < 3; ++k) out[k] = U[3 * k + j]; };
    if (S[2] <= 1e-12 * std::max(S[0], 1e-300)) {
        double u0[3], u1[3];
        col(0, u0); col(1, u1);
        if (S[1] <= 1e-12 * std::max(S[0], 1e-300)) {   // rank 1: pick any vector orthogonal to u0
            double e[3] = {0, 0, 0};
            int m = std::fabs(u0[0]) < std::fabs(u0[1]) ? (std::fabs(u0[0]) < std::fabs(u0[2]) ? 0 : 2) : (std::fabs(u0[1]) < std::fabs(u0[2]) ? 1 : 2);
            e[m] = 1;
            u1[0] = u0[1] * e[2] - u0[2] * e[1]; u1[1] = u0[2] * e[0] - u0[0] * e[2]; u1[2] = u0[0] * e[1] - u0[1] * e[0];
            const double n = std::sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
            for (int k = 0; k < 3; ++k) { u1[k] /= n; U[3 * k + 1] = u1[k]; }
        }
        U[0 + 2] = u0[1] * u1[2] - u0[2] * u1[1];
        U[3 + 2] = u0[2] * u1[0] - u0[0] * u1[2];
        U[6 + 2] = u0[0] * u1[1] - u0[1] * u1[0];
    }
}

static double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// src/estimation.cpp:29-51
void solveRigidMotion(const Matf& A, const Matf& B, Matf& T) {
    if (A.cols <= 1) throw std::invalid_argument("solveRigidMotion: A.cols() > 1 required");                 // :32
    if (A.cols != B.cols || A.rows != B.rows) throw std::invalid_argument("solveRigidMotion: shape mismatch");   // :35-38
    if (A.rows != 3) throw std::invalid_argument("solveRigidMotion: A.rows() == 3 required");              // :39
    const int n = A.cols;
    double m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
    for (int r = 0; r < 3; ++r) { for (int i = 0; i < n; ++i) { m1[r] += A.at(r, i); m2[r] += B.at(r, i); } m1[r] /= n; m2[r] /= n; }
    double C[9] = {0};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int i = 0; i < n; ++i) s += (A.at(r, i) - m1[r]) * (B.at(c, i) - m2[c]); C[3 * r + c] = s; }
    double U[9], S[3], V[9], UVt[9], R[9];
    svd3(C, U, S, V);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += U[3 * r + k] * V[3 * c + k]; UVt[3 * r + c] = s; }
    const double d = det3(UVt);
    const double v[3] = {1, 1, d};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += U[3 * r + k] * v[k] * V[3 * c + k]; R[3 * r + c] = s; }
    T.create(4, 4);
    for (int r = 0; r < 3; ++r) {
        double t = m1[r];
        for (int c = 0; c < 3; ++c) { T.at(r, c) = (float)R[3 * r + c]; t -= R[3 * r + c] * m2[c]; }
        T.at(r, 3) = (float)t;
    }
    T.at(3, 3) = 1.f;
}

// Procrustes start for the GN solve (see viso.hpp).  Points whose disparity is not positive are left out.
std::vector<double> procrustes_tr(const Matd& X, const Matd& observe, const param& p, const std::vector<int>& active) {
    if (X.rows != 3 || observe.rows != 4 || X.cols != observe.cols)
        throw std::invalid_argument("procrustes_tr: X 3xM, observe 4xM expected");
    std::vector<int> use;
    for (int i : active) {
        if (i < 0 || i >= X.cols) throw std::invalid_argument("procrustes_tr: active index out of range");
        if (observe.at(0, i) - observe.at(2, i) > 0) use.push_back(i);
    }
    if (use.size() < 3) return std::vector<double>(6, 0.0);
    const int n = (int)use.size();
    Matf A(3, n), B(3, n), T;
    for (int k = 0; k < n; ++k) {
        const int i = use[k];
        const double d = observe.at(0, i) - observe.at(2, i);                      // triangulate_rectified, :1137-1162
        A.at(0, k) = (float)(p.base * (observe.at(0, i) - p.calib.cu) / d);
        A.at(1, k) = (float)(p.base * (observe.at(1, i) - p.calib.cv) / d);
        A.at(2, k) = (float)(p.calib.f * p.base / d);
        for (int r = 0; r < 3; ++r) B.at(r, k) = (float)X.at(r, i);
    }
    solveRigidMotion(A, B, T);                                                     // T * X_prev = X_cur
    // tr2mat's rotation (src/viso.cpp:109-133): r02 = sy, r12 = -sx cy, r22 = cx cy, r01 = -cy sz, r00 = cy cz
    const double r02 = std::min(1.0, std::max(-1.0, (double)T.at(0, 2)));
    std::vector<double> tr(6);
    tr[1] = std::asin(r02);
    tr[0] = std::atan2(-(double)T.at(1, 2), (double)T.at(2, 2));
    tr[2] = std::atan2(-(double)T.at(0, 1), (double)T.at(0, 0));
    for (int r = 0; r < 3; ++r) tr[3 + r] = T.at(r, 3);
    return tr;
}

bool minimize_reproj_from_procrustes(const Matd& X, const Matd& observe, std::vector<double>& tr, const param& p,
                                     const std::vector<int>& active) {
    tr = procrustes_tr(X, observe, p, active);
    return minimize_reproj(X, observe, tr, p, active);
}

// ---- sequence_odometry ------------------------------------------------------
namespace {
struct Ctx {
    viso_ctx* c;
    explicit Ctx(int device) : c(viso_ctx_create(device, nullptr)) { if (!c) throw std::runtime_error(std::string("viso_ctx_create: ") + viso_last_error()); }
    ~Ctx() { viso_ctx_destroy(c); }
};

// Two chunks in flight: while the GPU works on one chunk (its own context / HIP stream), the host packs and uploads
// the next one into the other slot.  A slot's batch is kept and reused while the chunk shape stays the same.  Results
// are drained strictly in chunk order, so the pose chain is the same as with one chunk at a time.
struct ChunkPipeline {
    struct Slot {
        Ctx ctx;
        viso_batch* b = nullptr;
        int nf = 0, cap = 0, dlen = 0;      // shape the batch was created for
        int global0 = 0;                    // global frame index of the chunk's first frame (its halo)
        bool busy = false;
        bool stamped = false;               // the chunk in flight carries time stamps (viso_batch_stamp)
        uint8_t* pin = nullptr;             // pinned staging buffer of the image-driven path: [frames][2][rows][cols]
        size_t pin_bytes = 0;
        explicit Slot(int device) : ctx(device) {}
        Slot(const Slot&) = delete;
        ~Slot() { if (b) viso_batch_destroy(b); if (pin) viso_host_free(pin); }
        uint8_t* pinned(size_t bytes) {
            if (pin_bytes < bytes) {
                if (pin) viso_host_free(pin);
                pin = static_cast<uint8_t*>(viso_host_alloc(bytes));
                pin_bytes = pin ? bytes : 0;
                if (!pin) throw std::runtime_error(std::string("viso_host_alloc: ") + viso_last_error());
            }
            return pin;
        }
    };
    Slot slot[2];
    int next = 0;
    OdometryResult& out;
    double pose[16];
    ChunkPipeline(OdometryResult& o, int device) : slot{Slot(device), Slot(device)}, out(o) {
        std::memcpy(pose, out.poses[0].ptr(), sizeof(pose));
    }

    // batch of the slot the next chunk goes to (drains the chunk that used it two steps ago first)
    viso_batch* acquire(int nf, int cap, int dlen, int global0) {
        Slot& s = slot[next];
        if (s.busy) drain(s);
        if (!s.b || s.nf != nf || s.cap != cap || s.dlen != dlen) {
            if (s.b) viso_batch_destroy(s.b);
            s.b = viso_batch_create(s.ctx.c, nf, cap, dlen);
            if (!s.b) throw std::runtime_error(std::string("viso_batch_create: ") + viso_last_error());
            s.nf = nf; s.cap = cap; s.dlen = dlen;
        }
        s.global0 = global0;
        return s.b;
    }
    void submitted() { slot[next].busy = true; next ^= 1; }
    Slot& current() { return slot[next]; }
    Slot& previous() { return slot[next ^ 1]; }
    void drain(Slot& s) {
        const int nf = s.nf;
        std::vector<double> tr((size_t)nf * 6);
        std::vector<int32_t> ok((size_t)nf), ninl((size_t)nf);
        const auto t0 = std::chrono::steady_clock::now();
        hip_check(viso_batch_get_poses(s.b, tr.data(), ok.data(), ninl.data()), "sequence_odometry");
        out.stats.drain_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (s.stamped) {
            double ms[2] = {0, 0};
            hip_check(viso_batch_stamp_ms(s.b, ms), "sequence_odometry");
            out.stats.upload_ms += ms[0];
            out.stats.gpu_ms += ms[1];
            s.stamped = false;
        }
        s.busy = false;
        for (int t = (s.global0 == 0 ? 0 : 1); t < nf; ++t) {
            out.ok.push_back(t == 0 ? 0 : ok[(size_t)t]);
            out.n_inliers.push_back(t == 0 ? 0 : ninl[(size_t)t]);
            std::array<double, 6> a{};
            if (t > 0) for (int j = 0; j < 6; ++j) a[(size_t)j] = tr[(size_t)t * 6 + j];
            out.tr.push_back(a);
            if (t > 0 && ok[(size_t)t]) {                              // :1313-1321
                viso_pose_update(pose, a.data(), pose);
                Matd P(4, 4);
                std::memcpy(P.ptr(), pose, sizeof(pose));
                out.poses.push_back(P);
                out.frame_of_pose.push_back(s.global0 + t);
            }
        }
    }
    void finish() {   // oldest first
        if (slot[next].busy) drain(slot[next]);
        if (slot[next ^ 1].busy) drain(slot[next ^ 1]);
    }
};
}

OdometryResult sequence_odometry(const Matd& P1, const Matd& P2, StereoFeatureGenerator frames, int chunk,
                                 uint64_t ransac_seed, uint64_t first_frame_index, int device) {
    if (P1.rows != 3 || P1.cols != 4 || P2.rows != 3 || P2.cols != 4) throw std::invalid_argument("sequence_odometry: P1,P2 must be 3x4");
    if (chunk < 1) chunk = 1;
    Matd F = F_from_P(P1, P2);                                        // :1176-1180
    param prm;
    prm.base = std::fabs(P2.at(0, 3) / P2.at(0, 0));                  // :1184
    prm.calib.f = P1.at(0, 0); prm.calib.cu = P1.at(0, 2); prm.calib.cv = P1.at(1, 2);   // :1185-1187
    viso_match_params st = to_abi(MatchParams(F)), tm = to_abi(MatchParams());
    viso_param vp = to_abi(prm);
    OdometryResult out;
    out.poses.push_back(Matd::eye(4));                                // :1189-1190
    out.frame_of_pose.push_back(0);
    ChunkPipeline pipe(out, device);
    std::vector<StereoFeatures> buf;                                  // frames of the current chunk (buf[0] = halo)
    int global0 = 0;                                                  // global index of buf[0]
    bool eos = false;
    while (!eos) {
        while ((int)buf.size() < chunk + 1) {
            std::optional<StereoFeatures> f = frames();
            if (!f) { eos = true; break; }
            buf.push_back(std::move(*f));
        }
        const int nf = (int)buf.size();
        if (nf == 0 || (nf == 1 && global0 > 0)) break;
        int cap = 1, dlen = VISO_DESC_LEN;
        for (auto& f : buf) {
            cap = std::max({cap, (int)f.kp1.size(), (int)f.kp2.size()});
            if (f.d1.rows) dlen = f.d1.cols; else if (f.d2.rows) dlen = f.d2.cols;
            if ((int)f.kp1.size() != f.d1.rows || (int)f.kp2.size() != f.d2.rows) throw std::invalid_argument("sequence_odometry: keypoint/descriptor count mismatch");
        }
        cap = (cap + 255) / 256 * 256;                                // coarse capacity classes: batches get reused
        std::vector<float> kp((size_t)nf * 2 * cap * 2, 0.f), desc((size_t)nf * 2 * cap * dlen, 0.f);
        std::vector<int32_t> n((size_t)nf * 2);
        for (int t = 0; t < nf; ++t)
            for (int side = 0; side < 2; ++side) {
                const KeyPoints& k = side ? buf[(size_t)t].kp2 : buf[(size_t)t].kp1;
                const Descriptors& d = side ? buf[(size_t)t].d2 : buf[(size_t)t].d1;
                if (d.rows && d.cols != dlen) throw std::invalid_argument("sequence_odometry: descriptor length changes between frames");
                n[(size_t)t * 2 + side] = (int)k.size();
                float* kd = kp.data() + ((size_t)t * 2 + side) * cap * 2;
                for (size_t i = 0; i < k.size(); ++i) { kd[2 * i] = k[i].pt.x; kd[2 * i + 1] = k[i].pt.y; }
                if (d.rows) std::memcpy(desc.data() + ((size_t)t * 2 + side) * cap * dlen, d.ptr(), sizeof(float) * (size_t)d.rows * dlen);
            }
        viso_batch* b = pipe.acquire(nf, cap, dlen, global0);
        int r = viso_batch_upload(b, 0, nf, kp.data(), desc.data(), n.data());
        if (r >= 0) r = viso_batch_set_params(b, &st, &tm, &vp, ransac_seed, first_frame_index + (uint64_t)global0);
        if (r >= 0) r = viso_batch_run(b);                            // asynchronous: the next chunk is packed meanwhile
        hip_check(r, "sequence_odometry");
        pipe.submitted();
        // keep the last frame as the next chunk's halo
        StereoFeatures last = std::move(buf.back());
        buf.clear();
        buf.push_back(std::move(last));
        global0 += nf - 1;
    }
    pipe.finish();
    return out;
}

// ---- the literal per-call loop (see viso.hpp) -------------------------------
OdometryResult sequence_odometry_per_call(const Matd& F, param prm, StereoFeatureGenerator frames,
                                          uint64_t first_frame_index, PerCallStats* per_call, PerCallTrace* trace) {
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
    PerCallStats local;
    PerCallStats& st = per_call ? *per_call : local;
    auto timed = [&](int fn, auto&& call) {
        const auto t0 = clk::now();
        call();
        st.us[fn] += us_since(t0);
        st.calls[fn] += 1;
    };
    OdometryResult out;
    out.poses.push_back(Matd::eye(4));                                // :1189-1190
    out.frame_of_pose.push_back(0);
    double pose[16];
    std::memcpy(pose, out.poses[0].ptr(), sizeof(pose));
    const MatchParams stereo(F), temporal;
    Matd X, X_prev;
    Descriptors d1, d2, d1_prev, d2_prev;
    KeyPoints kp1, kp2, kp1_prev, kp2_prev;
    Matches match_lr, match_lr_prev;
    bool first = true;
    double gen_s = 0;
    const auto t_loop = clk::now();
    for (int iter_num = 0;; ++iter_num) {
        const auto tg = clk::now();
        std::optional<StereoFeatures> f = frames();
        gen_s += us_since(tg) * 1e-6;
        if (!f) break;
        prm.frame_index = first_frame_index + (uint64_t)iter_num;   // adapters/libviso_hip.patch, :1207
        if (!first) {                                                 // :1208-1222
            const auto tc = clk::now();
            d1_prev = d1; d2_prev = d2;                               // copyTo: a copy, not a header
            kp1_prev = kp1; kp2_prev = kp2;
            match_lr_prev = match_lr;
            X_prev = X;
            kp1.clear(); kp2.clear(); match_lr.clear();
            st.carry_s += us_since(tc) * 1e-6;
        }
        kp1 = std::move(f->kp1); kp2 = std::move(f->kp2);             // detector.detect / extractor.compute, :1226-1231
        d1 = std::move(f->d1); d2 = std::move(f->d2);
        timed(VISO_PLAIN_MATCH_DESC, [&] { match_desc(kp1, kp2, d1, d2, match_lr, stereo); });                 // :1240
        Matd x;
        timed(VISO_PLAIN_COLLECT_MATCHES, [&] { collect_matches(kp1, kp2, match_lr, x); });                    // :1246
        timed(VISO_PLAIN_TRIANGULATE, [&] { X = triangulate_rectified(x, prm); });                             // :1247
        st.frames += 1;
        out.ok.push_back(0); out.n_inliers.push_back(0); out.tr.push_back(std::array<double, 6>{});
        if (trace) { trace->match_lr.push_back(match_lr); trace->match11.emplace_back(); trace->match22.emplace_back(); trace->n_circle.push_back(0); }
        if (first) { first = false; continue; }                      // :1256-1260
        Matches match11, match22;
        timed(VISO_PLAIN_MATCH_DESC, [&] { match_desc(kp1, kp1_prev, d1, d1_prev, match11, temporal); });       // :1264
        timed(VISO_PLAIN_MATCH_DESC, [&] { match_desc(kp2, kp2_prev, d2, d2_prev, match22, temporal); });       // :1275
        Matches match_pcl;
        std::vector<Vec4i> circ_match;
        timed(VISO_PLAIN_MATCH_CIRCLE, [&] { match_circle(match_lr, match_lr_prev, match11, match22, circ_match, match_pcl); });   // :1282
        if (trace) { trace->match11.back() = match11; trace->match22.back() = match22; trace->n_circle.back() = (int)circ_match.size(); }
        if (circ_match.size() < 3) continue;                         // :1283-1288
        const int mc = (int)circ_match.size();
        Matd Xp_c(3, mc), x_c(4, mc);                                 // :1292-1305
        for (int i = 0; i < mc; ++i) {
            for (int r = 0; r < 4; ++r) x_c.at(r, i) = x.at(r, match_pcl[(size_t)i][0]);
            for (int r = 0; r < 3; ++r) Xp_c.at(r, i) = X_prev.at(r, match_pcl[(size_t)i][1]);
        }
        std::vector<int> inliers;
        std::vector<double> tr(6, 0.0);
        bool ok = false;
        timed(VISO_PLAIN_RANSAC, [&] { ok = ransac_minimize_reproj(Xp_c, x_c, tr, inliers, prm); });            // :1313
        out.ok.back() = ok ? 1 : 0;
        out.n_inliers.back() = (int)inliers.size();
        for (int j = 0; j < 6; ++j) out.tr.back()[(size_t)j] = tr[(size_t)j];
        if (ok) {                                                     // :1315-1321
            viso_pose_update(pose, tr.data(), pose);
            Matd P(4, 4);
            std::memcpy(P.ptr(), pose, sizeof(pose));
            out.poses.push_back(P);
            out.frame_of_pose.push_back(iter_num);
        }
    }
    st.wall_s = us_since(t_loop) * 1e-6 - gen_s;
    return out;
}

}  // namespace viso

// ---------------------------------------------------------------------------
// Front-end mirror
#include <cstdio>

#include "png_read.hpp"

namespace viso {

Image imread_pgm(const std::string& file_name) {
    Image im;
    FILE* fp = std::fopen(file_name.c_str(), "rb");
    if (!fp) return im;
    auto token = [&](int& v) {   // next integer, skipping whitespace and '#' comments
        int c = std::fgetc(fp);
        for (;;) {
            while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = std::fgetc(fp);
            if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(fp); continue; }
            break;
        }
        if (c < '0' || c > '9') return false;
        v = 0;
        while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = std::fgetc(fp); }
        return true;   // the single whitespace after the token has been consumed
    };
    int w = 0, h = 0, maxv = 0;
    const bool magic = std::fgetc(fp) == 'P' && std::fgetc(fp) == '5';
    if (magic && token(w) && token(h) && token(maxv) && w > 0 && h > 0 && maxv > 0 && maxv <= 255) {
        im.data.resize((size_t)w * h);
        if (std::fread(im.data.data(), 1, im.data.size(), fp) == im.data.size()) { im.rows = h; im.cols = w; }
        else im.data.clear();
    }
    std::fclose(fp);
    return im;
}

Image imread_gray(const std::string& file_name) {
    const size_t n = file_name.size();
    if (n >= 4 && (file_name.compare(n - 4, 4, ".png") == 0 || file_name.compare(n - 4, 4, ".PNG") == 0)) {
        Image im;
        if (!read_png_gray(file_name, im.rows, im.cols, im.data)) { im.rows = im.cols = 0; im.data.clear(); }
        return im;
    }
    return imread_pgm(file_name);
}

HarrisBinnedFeatureDetector::HarrisBinnedFeatureDetector(int radius, int n, int nbinx, int nbiny, float k,
                                                         int block_size, int aperture_size)
    : m_radius(radius), m_nbinx(nbinx), m_nbiny(nbiny), m_block_size(block_size),
      m_aperture_size(aperture_size), m_n(n), m_k(k) {
    if (nbinx <= 0 || nbiny <= 0) throw std::invalid_argument("HarrisBinnedFeatureDetector: nbinx>0 && nbiny>0");   // :920
    if (block_size != 3 || aperture_size != 5) throw std::invalid_argument("HarrisBinnedFeatureDetector: only block_size 3 / aperture_size 5 (the reference's values, :915-916)");
}

void HarrisBinnedFeatureDetector::detect(const Image& image, KeyPoints& kp) const {
    if (image.empty()) throw std::invalid_argument("detect: empty image");
    std::vector<float> xy((size_t)std::max(1, m_n) * 2), resp((size_t)std::max(1, m_n));
    int n = 0;
    int r = viso_detect_harris_binned(image.data.data(), image.rows, image.cols, m_n, m_nbinx, m_nbiny, (double)m_k,
                                      xy.data(), resp.data(), &n);
    if (r == VISO_ERR_ARG) throw std::invalid_argument("detect: stridex>0 && stridey>0");   // :934
    hip_check(r, "detect");
    for (int i = 0; i < n; ++i) {   // :964-971 (the reference appends)
        KeyPoint k;
        k.pt.x = xy[(size_t)2 * i]; k.pt.y = xy[(size_t)2 * i + 1];
        k.response = resp[(size_t)i];
        k.size = (float)(2 * m_radius + 1);
        kp.push_back(k);
    }
}

void MyFeatureExtractor::compute(const Image& image, KeyPoints& kp, Descriptors& d) const {
    if (image.empty()) throw std::invalid_argument("compute: empty image");
    d.create((int)kp.size(), descriptorSize());
    if (kp.empty()) return;
    std::vector<float> xy = kp2mat(kp);
    hip_check(viso_extract_descriptors(image.data.data(), image.rows, image.cols, xy.data(), (int)kp.size(),
                                       m_descriptor_radius, d.ptr()), "compute");
}

static std::string format_mask(const std::string& mask, int index) {
    char buf[4096];
    std::snprintf(buf, sizeof(buf), mask.c_str(), index);   // boost::format(mask) % index, src/viso.h:90-91
    return buf;
}

bool StereoImageGenerator::read(int index, Image& left, Image& right) const {
    left = imread_gray(format_mask(m_mask.first, index));
    if (left.empty()) return false;
    right = imread_gray(format_mask(m_mask.second, index));
    return !right.empty();
}

int StereoImageGenerator::read_into(int index, int side, int rows, int cols, uint8_t* dst) const {
    const std::string name = format_mask(side ? m_mask.second : m_mask.first, index);
    const size_t n = name.size();
    if (n >= 4 && (name.compare(n - 4, 4, ".png") == 0 || name.compare(n - 4, 4, ".PNG") == 0)) {
        bool other = false;
        if (read_png_gray_to(name, rows, cols, dst, &other)) return 1;
        return other ? 2 : 0;
    }
    Image im = imread_pgm(name);
    if (im.empty()) return 0;
    if (im.rows != rows || im.cols != cols) return 2;
    std::memcpy(dst, im.data.data(), (size_t)rows * cols);
    return 1;
}

bool StereoImageGenerator::read_to(int index, int side, int rows, int cols, uint8_t* dst) const {
    return read_into(index, side, rows, cols, dst) == 1;
}

StereoImageGenerator::result_type StereoImageGenerator::operator()() {
    if (m_index > m_end) return std::nullopt;
    Image a, b;
    const bool ok = read(m_index, a, b);
    m_index++;
    if (!ok) return std::nullopt;
    return std::make_pair(std::move(a), std::move(b));
}

namespace {
// Worker threads for the image decoding of one rank.  Tasks of a chunk are independent (one image each); the calling
// thread waits for the whole chunk, the GPU meanwhile works on the previous one.
class DecodePool {
public:
    explicit DecodePool(int n) {
        for (int i = 0; i < n; ++i) th.emplace_back([this] { work(); });
    }
    ~DecodePool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv_work.notify_all();
        for (auto& t : th) t.join();
    }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> l(m); q.push_back(std::move(f)); ++pending; }
        cv_work.notify_one();
    }
    void wait_all() {
        std::unique_lock<std::mutex> l(m);
        cv_done.wait(l, [this] { return pending == 0; });
    }
    double busy_seconds() { std::lock_guard<std::mutex> l(m); return busy_s; }
    int size() const { return (int)th.size(); }
private:
    void work() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m);
                cv_work.wait(l, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            const auto t0 = std::chrono::steady_clock::now();
            f();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            { std::lock_guard<std::mutex> l(m); busy_s += dt; --pending; }
            cv_done.notify_all();
        }
    }
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<std::function<void()>> q;
    int pending = 0;
    bool stop = false;
    double busy_s = 0;
};

int default_decode_threads() {
    if (const char* e = std::getenv("VISO_DECODE_THREADS")) { const int v = std::atoi(e); if (v > 0) return std::min(v, 256); }
    return std::max(1, std::min(64, cpu_budget()));   // decoding is what bounds a KITTI run (17.6 s of thread time against 0.06 s of kernels)
}
}  // namespace

int cpu_budget() {
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
#ifdef __linux__
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
            char q[64] = "";
            long long period = 0;
            if (std::fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
                const long long quota = std::atoll(q);
                if (quota > 0) { const int c = (int)((quota + period - 1) / period); if (c > 0 && c < n) n = c; }
            }
            std::fclose(f);
        }
    }
#endif
    return n;
}

// Chunks of `chunk` frames (plus the one-frame halo), two in flight: while the GPU runs chunk c (its own context,
// asynchronous uploads from the slot's pinned buffer), the worker threads decode chunk c + 1 straight into the other
// slot's pinned buffer.  The halo frame is copied from the previous slot's buffer, not decoded twice.
OdometryResult sequence_odometry(const Matd& P1, const Matd& P2, StereoImageGenerator& images, int chunk,
                                 uint64_t ransac_seed, uint64_t first_frame_index, int device, int decode_threads) {
    using clock = std::chrono::steady_clock;
    auto since = [](clock::time_point t0) { return std::chrono::duration<double>(clock::now() - t0).count(); };
    const auto t_start = clock::now();
    if (P1.rows != 3 || P1.cols != 4 || P2.rows != 3 || P2.cols != 4) throw std::invalid_argument("sequence_odometry: P1,P2 must be 3x4");
    if (chunk < 1) chunk = 1;
    const int MAX_FEATURE_NUM = 1200;                                   // :1171
    HarrisBinnedFeatureDetector detector(5, MAX_FEATURE_NUM);           // :1172
    Matd F = F_from_P(P1, P2);
    param prm;
    prm.base = std::fabs(P2.at(0, 3) / P2.at(0, 0));
    prm.calib.f = P1.at(0, 0); prm.calib.cu = P1.at(0, 2); prm.calib.cv = P1.at(1, 2);
    viso_match_params st = to_abi(MatchParams(F)), tm = to_abi(MatchParams());
    viso_param vp = to_abi(prm);
    OdometryResult out;
    out.poses.push_back(Matd::eye(4));
    out.frame_of_pose.push_back(0);
    if (images.index() > images.end()) return out;
    // the first frame fixes the geometry of the sequence
    Image first_l, first_r;
    {
        const auto t0 = clock::now();
        const bool ok = images.read(images.index(), first_l, first_r);
        out.stats.decode_wait_s += since(t0);
        out.stats.decode_cpu_s += since(t0);
        if (!ok) { images.seek(images.index() + 1); return out; }      // the generator stops at an unreadable pair (src/viso.h:94-96)
    }
    const int rows = first_l.rows, cols = first_l.cols;
    if (first_r.rows != rows || first_r.cols != cols) throw std::invalid_argument("sequence_odometry: image size changes inside a sequence");
    const size_t per = (size_t)rows * cols;
    const int first_index = images.index();
    images.seek(first_index + 1);
    DecodePool pool(decode_threads > 0 ? decode_threads : default_decode_threads());
    out.stats.decode_threads = pool.size();
    ChunkPipeline pipe(out, device);
    int global0 = 0;                 // frame (relative to first_index) of the current chunk's halo
    int frames_read = 1;
    bool eos = false;
    bool have_halo_in_prev = false;  // false only for the first chunk (its frame 0 is first_l / first_r)
    std::vector<uint8_t> ok_flag;
    while (!eos) {
        // frames global0 + 1 .. global0 + want of this chunk still have to be decoded
        const long left = (long)images.end() - (long)(first_index + global0);
        const int want = (int)std::min<long>(chunk, std::max<long>(0, left));
        if (want == 0 && global0 > 0) break;
        ChunkPipeline::Slot& s = pipe.current();
        if (s.busy) pipe.drain(s);                                      // its previous chunk (two steps ago)
        uint8_t* pin = s.pinned((size_t)(chunk + 1) * 2 * per);
        if (!have_halo_in_prev) {
            std::memcpy(pin, first_l.data.data(), per);
            std::memcpy(pin + per, first_r.data.data(), per);
        } else {
            ChunkPipeline::Slot& p = pipe.previous();
            std::memcpy(pin, p.pin + (size_t)(p.nf - 1) * 2 * per, 2 * per);   // the previous chunk's last frame
        }
        ok_flag.assign((size_t)want * 2, 0);
        const auto t0 = clock::now();
        for (int j = 0; j < want; ++j)
            for (int side = 0; side < 2; ++side) {
                uint8_t* dst = pin + ((size_t)(j + 1) * 2 + side) * per;
                uint8_t* flag = &ok_flag[(size_t)j * 2 + side];
                const int index = first_index + global0 + 1 + j;
                pool.submit([&images, index, side, rows, cols, dst, flag] {
                    // a worker must not let an exception escape (std::terminate): a file that cannot be decoded -- whatever the
                    // decoder ran into, bad_alloc on a corrupt header included -- is an unreadable image
                    try { *flag = (uint8_t)images.read_into(index, side, rows, cols, dst); } catch (...) { *flag = 0; }
                });
            }
        pool.wait_all();
        out.stats.decode_wait_s += since(t0);
        int got = 0;
        while (got < want && ok_flag[(size_t)got * 2] == 1 && ok_flag[(size_t)got * 2 + 1] == 1) ++got;
        // The first frame that is not usable ends the sequence, as the reference's generator does for an unreadable pair
        // (src/viso.h:94-96) -- unless it DOES decode, with another size: that is not an end of stream but an input error,
        // and it is reported as one (the chunks in flight are drained by the pipeline's destructor)
        if (got < want && (ok_flag[(size_t)got * 2] == 2 || ok_flag[(size_t)got * 2 + 1] == 2))
            throw std::invalid_argument("sequence_odometry: image size changes inside a sequence (frame " +
                                        std::to_string(first_index + global0 + 1 + got) + ")");
        if (got < want || first_index + global0 + got >= images.end()) eos = true;
        images.seek(first_index + global0 + got + (got < want ? 2 : 1));   // where operator() would stand now
        frames_read += got;
        const int nf = got + 1;
        if (nf == 1 && global0 > 0) break;
        const auto t1 = clock::now();
        viso_batch* b = pipe.acquire(nf, MAX_FEATURE_NUM, VISO_DESC_LEN, global0);
        int r = viso_batch_upload_images(b, 0, 0, nullptr, rows, cols, nullptr, nullptr);          // device buffers for this geometry
        if (r >= 0) r = viso_batch_set_params(b, &st, &tm, &vp, ransac_seed, first_frame_index + (uint64_t)global0);
        if (r >= 0) r = viso_batch_stamp(b, 0);
        if (r >= 0) r = viso_batch_upload_images_async(b, 0, nf, pin, rows, cols, nullptr, nullptr);
        if (r >= 0) r = viso_batch_stamp(b, 1);
        if (r >= 0) r = viso_batch_detect(b, detector.n(), detector.nbinx(), detector.nbiny(), (double)detector.k());   // :1226-1227
        if (r >= 0) r = viso_batch_run_images(b, 0);                                                                    // :1230-1313, asynchronous
        hip_check(r, "sequence_odometry");
        s.stamped = true;
        out.stats.issue_s += since(t1);
        pipe.submitted();
        have_halo_in_prev = true;
        global0 += nf - 1;
    }
    pipe.finish();
    out.stats.frames = frames_read;
    out.stats.decode_cpu_s += pool.busy_seconds();
    out.stats.wall_s = since(t_start);
    return out;
}

}  // namespace viso
