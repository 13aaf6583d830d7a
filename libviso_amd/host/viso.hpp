// viso.hpp — C++ host mirror of the reference's call surface for the hot path
// (alexkreimer/libviso src/viso.h:51-79,138-139,162 and the file-local
// match_desc / MatchParams / match_circle / collect_matches /
// triangulate_rectified of src/viso.cpp), forwarding to the C-ABI of
// include/viso_hip.h.  Same names, same argument meaning, same outputs.
//
// The reference's types come from OpenCV (cv::Mat, cv::KeyPoint, cv::Vec3i),
// which is not available to this build, so minimal stand-ins with the same
// member names live in namespace viso (Mat_<T> = dense row-major matrix with
// .rows/.cols/.at(r,c)); INTEGRATION.md shows the 20-line adapter from the
// real OpenCV types.  Error behaviour: where the reference asserts
// (BOOST_ASSERT_MSG / assert, e.g. src/viso.cpp:676) this mirror throws
// std::invalid_argument; HIP failures throw std::runtime_error.  Solver
// functions return the reference's bool.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/viso_hip.h"

namespace viso {
// Host threads this process may really use: the hardware's, cut down to the affinity mask and to the cgroup's CPU quota
// (/sys/fs/cgroup/cpu.max).  A container that shows 256 hardware threads and is given 16 cores' worth of time runs 64 decode
// threads no faster than 16 (KITTI rehearsal, profiles/r06_kitti_rehearsal.txt: 1.38 s with 16 threads, 1.41 s with 64, the
// thread time 17 s -> 41 s): the runners size their decode pools from this.
int cpu_budget();


template <class T>
struct Mat_ {
    int rows = 0, cols = 0;
    std::vector<T> data;
    Mat_() = default;
    Mat_(int r, int c, T v = T()) : rows(r), cols(c), data((size_t)r * c, v) {}
    void create(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, T()); }
    T& at(int r, int c) { return data[(size_t)r * cols + c]; }
    const T& at(int r, int c) const { return data[(size_t)r * cols + c]; }
    T* ptr(int r = 0) { return data.data() + (size_t)r * cols; }
    const T* ptr(int r = 0) const { return data.data() + (size_t)r * cols; }
    bool empty() const { return data.empty(); }
    static Mat_ eye(int n) { Mat_ m(n, n); for (int i = 0; i < n; ++i) m.at(i, i) = T(1); return m; }
};
using Matd = Mat_<double>;
using Matf = Mat_<float>;

struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 0, response = 0; };   // cv::KeyPoint fields the path reads
using KeyPoints = std::vector<KeyPoint>;
using Match = std::array<int, 3>;          // Vec3i: i1, i2, dist   (src/viso.h:52)
using Matches = std::vector<Match>;
using Vec4i = std::array<int, 4>;
using Descriptors = Matf;                  // N x 121 CV_32F        (src/viso.h:55)

// struct param, src/viso.h:58-72
struct param {
    param() : ransac_iter(50), inlier_threshold(2), thresh(1e-4), save_debug(true) { base = 0; calib.f = calib.cu = calib.cv = 0; }
    double base;
    int ransac_iter;
    double inlier_threshold;
    double thresh;
    bool save_debug;
    struct { double f, cu, cv; } calib;
    // deterministic replacement for randomsample's random_device (src/viso.cpp:93): stream key
    uint64_t ransac_seed = 0, frame_index = 0;
};

// struct MatchParams, src/viso.cpp:48-75
struct MatchParams {
    bool enforce_epipolar;
    Matd F;
    double alg_thresh = 0;
    double sampson_thresh = 0;
    bool enforce_2nd_best;
    double ratio_2nd_best;
    bool allow_ann = true;
    int max_neighbors;
    double radius;
    explicit MatchParams(const Matd& F_)
        : enforce_epipolar(true), F(F_), sampson_thresh(1), enforce_2nd_best(false), ratio_2nd_best(.8),
          max_neighbors(200), radius(80) {}
    MatchParams() : enforce_epipolar(false), enforce_2nd_best(true), ratio_2nd_best(.9), max_neighbors(250), radius(80) {}
};

// src/viso.cpp:669-726
void match_desc(const KeyPoints& kp1, const KeyPoints& kp2, const Descriptors& d1, const Descriptors& d2,
                Matches& match, const MatchParams& sp = MatchParams());
// src/viso.cpp:207-243
void match_circle(const Matches& match_lr, const Matches& match_lr_prev, const Matches& match11,
                  const Matches& match22, std::vector<Vec4i>& circ_match, Matches& match_pcl);
// src/viso.cpp:501-514
void collect_matches(const KeyPoints& kp1, const KeyPoints& kp2, const Matches& match, Matd& x);
// src/viso.cpp:1137-1162 (T = double is the only instantiation the stereo path uses, :1247)
Matd triangulate_rectified(const Matd& x, const param& param);
// src/viso.h:74-79
bool ransac_minimize_reproj(const Matd& X, const Matd& observe, std::vector<double>& best_tr,
                            std::vector<int>& best_inliers, const param& param);
bool minimize_reproj(const Matd& X, const Matd& observe, std::vector<double>& tr, const param& param,
                     const std::vector<int>& active);
// src/viso.h:162
void tr2mat(std::vector<double> tr, Matd& Tr);
// src/mvg.h:41-66 (+ the normalisation of src/viso.cpp:1177-1180)
Matd F_from_P(const Matd& P1, const Matd& P2);
// src/estimation.h:7-9 — Procrustes; dead on the stereo path, kept for the header surface.
// A, B: 3 x n.  T: 4 x 4 with [R|t] minimising sum |R b_i + t - a_i|^2 (the reference's argument order).
void solveRigidMotion(const Matf& A, const Matf& B, Matf& T);
// solveRigidMotion as a closed-form start for the Gauss-Newton solve (optional: the reference starts every solve
// from 0, src/viso.cpp:1557, and never calls its Procrustes).  procrustes_tr triangulates the current observations
// (:1137-1162), aligns them with the previous frame's points X over `active` and returns the motion in tr2mat's
// parametrisation (rx, ry, rz, tx, ty, tz); minimize_reproj_from_procrustes runs minimize_reproj from there.
std::vector<double> procrustes_tr(const Matd& X, const Matd& observe, const param& param, const std::vector<int>& active);
bool minimize_reproj_from_procrustes(const Matd& X, const Matd& observe, std::vector<double>& tr, const param& param,
                                     const std::vector<int>& active);

// One stereo frame as the front-end hands it to the hot path: what
// detector.detect + extractor.compute produce at src/viso.cpp:1226-1231.
struct StereoFeatures {
    KeyPoints kp1, kp2;
    Descriptors d1, d2;
};
// Generator in the style of StereoImageGenerator (src/viso.h:81-100): returns
// std::nullopt at end of stream.
using StereoFeatureGenerator = std::function<std::optional<StereoFeatures>()>;

// Where a run's wall time went (the image-driven sequence_odometry fills all of it; the feature-driven one the GPU
// side only).  A rank of the KITTI runner is bound by PNG decoding, not by the GPU: these are the numbers that say so.
struct OdometryStats {
    int frames = 0;               // frames read
    int decode_threads = 0;       // worker threads that decoded the images
    double wall_s = 0;            // whole call
    double decode_wait_s = 0;     // calling thread blocked until a chunk's images were decoded (the runner's critical path)
    double decode_cpu_s = 0;      // decode time summed over the worker threads (file read + inflate + unfilter)
    double issue_s = 0;           // calling thread inside upload / detect / run calls (asynchronous: issue cost)
    double drain_wait_s = 0;      // calling thread blocked in result read-back (GPU not done yet)
    double upload_ms = 0;         // GPU time stamps: host -> device copies of all chunks
    double gpu_ms = 0;            // GPU time stamps: detection + description + matching + solver of all chunks
};

struct OdometryResult {
    std::vector<Matd> poses;        // what the reference returns: poses[0] = I, then one per solved frame
    std::vector<int> frame_of_pose; // frame index of poses[i] (0 for the identity) — the reference drops failed frames silently (:1287,:1323)
    std::vector<int> ok, n_inliers; // per frame
    std::vector<std::array<double, 6>> tr;
    OdometryStats stats;
};

// sequence_odometry, src/viso.cpp:1167-1330, minus the front-end and the debug
// dumps: frames are pulled from `frames`, processed on the GPU in chunks of
// `chunk` frames (one-frame halo between chunks), poses chained on the host.
// first_frame_index: index of the generator's first frame in the whole sequence — the RANSAC stream of frame t
// is keyed on (ransac_seed, first_frame_index + t), so a sub-range gives the records of the full run
// (kitti_shard.hpp).  device: HIP device ordinal.
OdometryResult sequence_odometry(const Matd& P1, const Matd& P2, StereoFeatureGenerator frames,
                                 int chunk = 64, uint64_t ransac_seed = 0, uint64_t first_frame_index = 0,
                                 int device = 0);

// The LITERAL drop-in flow: the loop body of the reference's sequence_odometry (src/viso.cpp:1205-1327) over the plain
// family — one C-ABI call per reference function, in the reference's order (match_desc :1240, collect_matches :1246,
// triangulate_rectified :1247, match_desc :1264 and :1275, match_circle :1282, the gather of :1292-1305 on the host,
// ransac_minimize_reproj :1313), one frame at a time, with the copyTo carry-over of :1208-1222.  This is what an
// unchanged kitti.cpp gets from adapters/libviso_hip.patch; sequence_odometry above is the batched form of the same loop.
// `per_call` (may be null) receives, per VISO_PLAIN_* function, the number of calls and the wall time spent inside this
// mirror's wrapper of it (container reshaping included, as in the adapter).  `trace` (may be null) keeps every frame's
// match lists and circle size for comparisons with the batch family.
struct PerCallStats {
    long calls[VISO_PLAIN_N] = {0, 0, 0, 0, 0, 0, 0};
    double us[VISO_PLAIN_N] = {0, 0, 0, 0, 0, 0, 0};
    double wall_s = 0;            // the whole loop, generator excluded
    double carry_s = 0;           // the copyTo carry-over of :1208-1222 (the reference's own host cost)
    int frames = 0;
};
struct PerCallTrace {
    std::vector<Matches> match_lr, match11, match22;   // per frame (match11 / match22 empty for the first)
    std::vector<int> n_circle;
};
OdometryResult sequence_odometry_per_call(const Matd& F, param prm, StereoFeatureGenerator frames,
                                          uint64_t first_frame_index = 0, PerCallStats* per_call = nullptr,
                                          PerCallTrace* trace = nullptr);

}  // namespace viso

// ---------------------------------------------------------------------------
// Front-end mirror: the reference's detector /
// extractor classes and the image-driven sequence_odometry, on the GPU.
namespace viso {

// Grayscale image (what cv::imread(name, CV_LOAD_IMAGE_GRAYSCALE) returns, src/viso.h:92-93).
struct Image {
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
    bool empty() const { return data.empty(); }
};
// Binary PGM (P5, maxval <= 255).
Image imread_pgm(const std::string& file_name);
// cv::imread(name, CV_LOAD_IMAGE_GRAYSCALE) for the formats the path meets: 8-bit non-interlaced PNG
// (KITTI's image_0/%06d.png; own decoder in png_read.hpp) by extension, otherwise binary PGM.
Image imread_gray(const std::string& file_name);

// HarrisBinnedFeatureDetector, src/viso.cpp:911-979.  `k` is stored (the
// reference forgets to, :915-919); block_size 3 / aperture_size 5 are what the
// device kernel implements.
class HarrisBinnedFeatureDetector {
public:
    HarrisBinnedFeatureDetector(int radius, int n, int nbinx = 24, int nbiny = 5, float k = .04f,
                                int block_size = 3, int aperture_size = 5);
    void detect(const Image& image, KeyPoints& kp) const;
    int n() const { return m_n; }
    int nbinx() const { return m_nbinx; }
    int nbiny() const { return m_nbiny; }
    float k() const { return m_k; }
private:
    int m_radius, m_nbinx, m_nbiny, m_block_size, m_aperture_size, m_n;
    float m_k;
};

// MyFeatureExtractor, src/viso.cpp:981-1025.
class MyFeatureExtractor {
public:
    explicit MyFeatureExtractor(int descriptor_radius) : m_descriptor_radius(descriptor_radius) {}
    int descriptorSize() const { return (2 * m_descriptor_radius + 1) * (2 * m_descriptor_radius + 1); }
    void compute(const Image& image, KeyPoints& kp, Descriptors& d) const;
private:
    int m_descriptor_radius;
};

// StereoImageGenerator, src/viso.h:81-100: printf-style masks, begin/end frame, stops at the first unreadable pair.
class StereoImageGenerator {
public:
    typedef std::optional<std::pair<Image, Image>> result_type;
    StereoImageGenerator(const std::pair<std::string, std::string>& mask, int begin = 0, int end = 2147483647)
        : m_mask(mask), m_index(begin), m_end(end) {}
    result_type operator()();
    // Frame `index` alone — what operator() does for its current index.  const and thread safe: the image-driven
    // sequence_odometry decodes the frames of a chunk on worker threads through these.
    bool read(int index, Image& left, Image& right) const;
    // One image of frame `index` (side 0 = left) into caller memory of a known geometry (a pinned upload buffer);
    // false if it cannot be read or has another size.
    bool read_to(int index, int side, int rows, int cols, uint8_t* dst) const;
    // The same with the reason of a failure: 1 = read, 0 = cannot be opened / decoded (the generator's end of stream,
    // src/viso.h:94-96), 2 = decodes, but with another geometry than rows x cols.
    int read_into(int index, int side, int rows, int cols, uint8_t* dst) const;
    int index() const { return m_index; }
    int end() const { return m_end; }
    void seek(int index) { m_index = index; }
private:
    std::pair<std::string, std::string> m_mask;
    int m_index, m_end;
};

// sequence_odometry(P1, P2, images, dbg_dir), src/viso.h:138-139 / src/viso.cpp:1167-1330, without the
// debug dumps: detection (MAX_FEATURE_NUM 1200, radius 5, :1171-1174), description, matching and the
// solver all run on the device, `chunk` frames per batch.
// decode_threads: worker threads that decode a chunk's images while the GPU works on the previous chunk
// (0 = $VISO_DECODE_THREADS, else min(16, hardware threads)).  The frames are consumed in order and the sequence ends
// at the first pair that cannot be decoded, exactly like the reference's generator (src/viso.h:94-96).
OdometryResult sequence_odometry(const Matd& P1, const Matd& P2, StereoImageGenerator& images,
                                 int chunk = 64, uint64_t ransac_seed = 0, uint64_t first_frame_index = 0,
                                 int device = 0, int decode_threads = 0);

}  // namespace viso
