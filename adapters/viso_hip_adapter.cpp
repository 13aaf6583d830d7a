// viso_hip_adapter.cpp — the translation unit a libviso maintainer adds next to src/viso.cpp to run the hot path on
// libviso_hip.so (MI355X).  It keeps the reference's own signatures (src/viso.h:74-79,162 and the file-local
// functions of src/viso.cpp cited per function) and forwards plain pointers to the C-ABI of include/viso_hip.h;
// sequence_odometry (src/viso.cpp:1167-1330) and kitti.cpp then compile unchanged.
//
// Build inside the reference tree: remove (or #ifndef VISO_USE_HIP) the reference's definitions of the functions
// below, add this file to add_library(viso ...) in src/CMakeLists.txt:17, link -lviso_hip.
// NOT compiled in the libviso_amd repository: it needs the reference's viso.h, OpenCV and Boost, none of which
// exist in that build image.  libviso_amd/host/viso.hpp is the same wiring over dependency-free
// stand-in types and IS compiled and tested there.
#include <cstring>
#include <stdexcept>
#include <vector>

#include "viso.h"            // the reference's header, unchanged
#include "viso_hip.h"

static viso_match_params abi(const MatchParams& sp) {          // src/viso.cpp:48-75
    viso_match_params mp{};
    mp.enforce_epipolar = sp.enforce_epipolar;  mp.enforce_2nd_best = sp.enforce_2nd_best;
    mp.max_neighbors = sp.max_neighbors;        mp.sampson_thresh = sp.sampson_thresh;
    mp.ratio_2nd_best = sp.ratio_2nd_best;      mp.radius = sp.radius;
    if (sp.enforce_epipolar) { CV_Assert(sp.F.type() == CV_64F && sp.F.isContinuous());
                               memcpy(mp.F, sp.F.ptr<double>(), 9 * sizeof(double)); }
    return mp;
}
static viso_param abi(const struct param& p) {                 // src/viso.h:58-72
    viso_param q{};  q.base = p.base; q.ransac_iter = p.ransac_iter; q.inlier_threshold = p.inlier_threshold;
    q.thresh = p.thresh; q.save_debug = p.save_debug; q.f = p.calib.f; q.cu = p.calib.cu; q.cv = p.calib.cv;
    return q;
}

void match_desc(const KeyPoints& kp1, const KeyPoints& kp2, const Descriptors& d1, const Descriptors& d2,
                Matches& match, const MatchParams& sp) {        // replaces src/viso.cpp:669-726
    match.clear();
    BOOST_ASSERT_MSG(d1.cols == d2.cols, "d1.cols!=d2.cols");
    Mat k1 = kp2mat(kp1), k2 = kp2mat(kp2);                     // N x 2 CV_32F, src/viso.cpp:246-256
    Mat a = d1.isContinuous() ? d1 : d1.clone(), b = d2.isContinuous() ? d2 : d2.clone();
    std::vector<int32_t> out(3 * kp1.size());  int n = 0;
    viso_match_params mp = abi(sp);
    int r = viso_match_desc(k1.ptr<float>(), k1.rows, k2.ptr<float>(), k2.rows,
                            a.ptr<float>(), b.ptr<float>(), d1.cols, &mp, out.data(), &n);
    if (r < 0) throw std::runtime_error(viso_last_error());
    for (int i = 0; i < n; ++i) match.push_back(Match(out[3*i], out[3*i+1], out[3*i+2]));
}

bool minimize_reproj(const Mat& X, const Mat& observe, vector<double>& tr, const struct param& param,
                     const vector<int>& active) {               // replaces src/viso.cpp:1583-1623
    viso_param q = abi(param);                                  // X: 3xM CV_64F, observe: 4xM CV_64F, continuous
    return viso_minimize_reproj(X.ptr<double>(), observe.ptr<double>(), X.cols, tr.data(), &q,
                                active.data(), (int)active.size()) == 1;
}

bool ransac_minimize_reproj(const Mat& X, const Mat& observe, vector<double>& best_tr,
                            vector<int>& best_inliers, const struct param& param) {   // :1543-1580
    viso_param q = abi(param);
    best_inliers.assign(X.cols, 0);  int n = 0;
    static uint64_t call = 0;                                   // stream key in place of random_device (:93)
    int r = viso_ransac_minimize_reproj(X.ptr<double>(), observe.ptr<double>(), X.cols, best_tr.data(),
                                        best_inliers.data(), &n, &q, /*samples*/nullptr, /*seed*/0, call++);
    best_inliers.resize(n);
    return r == 1;
}

void match_circle(const Matches& lr, const Matches& lrp, const Matches& m11, const Matches& m22,
                  vector<Vec4i>& circ, Matches& pcl) {          // replaces src/viso.cpp:207-243
    // Vec3i / Vec4i are contiguous int triples / quads: pass &v[0][0]
    int cap = (int)lr.size() + 16, n = 0;
    std::vector<int32_t> c(4 * cap), p(2 * cap);
    viso_match_circle(lr.empty() ? 0 : &lr[0][0], (int)lr.size(), lrp.empty() ? 0 : &lrp[0][0], (int)lrp.size(),
                      m11.empty() ? 0 : &m11[0][0], (int)m11.size(), m22.empty() ? 0 : &m22[0][0], (int)m22.size(),
                      c.data(), p.data(), cap, &n);
    for (int i = 0; i < n; ++i) { circ.push_back(Vec4i(c[4*i], c[4*i+1], c[4*i+2], c[4*i+3])); pcl.push_back(Match(p[2*i], p[2*i+1])); }
}

void tr2mat(vector<double> tr, Mat& Tr) {                       // replaces src/viso.cpp:109-133
    Tr.create(4, 4, CV_64F);
    viso_tr2mat(tr.data(), Tr.ptr<double>());
}

void collect_matches(const KeyPoints& kp1, const KeyPoints& kp2, const Matches& match, Mat& x) {   // :501-514
    Mat k1 = kp2mat(kp1), k2 = kp2mat(kp2);
    x.create(4, (int)match.size(), CV_64F);
    if (match.empty()) return;
    int r = viso_collect_matches(k1.ptr<float>(), k1.rows, k2.ptr<float>(), k2.rows, &match[0][0], (int)match.size(),
                                 x.ptr<double>());
    if (r < 0) throw std::runtime_error(viso_last_error());
}

template <> Mat triangulate_rectified<double>(const Mat& x, const struct param& param) {          // :1137-1162
    viso_param q = abi(param);
    Mat X(3, x.cols, CV_64F);
    Mat xc = x.isContinuous() ? x : x.clone();
    if (x.cols && viso_triangulate_rectified(xc.ptr<double>(), x.cols, &q, X.ptr<double>()) < 0)
        throw std::runtime_error(viso_last_error());
    return X;
}
