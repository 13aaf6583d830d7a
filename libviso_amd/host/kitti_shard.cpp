// kitti_shard.cpp — see kitti_shard.hpp.  Host code only: ranges, record files, the pose chain; every frame's
// arithmetic happens in viso::sequence_odometry -> libviso_hip.so.
#include "kitti_shard.hpp"

#include <climits>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>

#include "kitti_io.hpp"

namespace viso {

std::vector<std::pair<int, int>> partition(int n_frames, int world) {
    std::vector<std::pair<int, int>> out;
    if (world < 1) world = 1;
    const int n_pairs = n_frames > 1 ? n_frames - 1 : 0;
    const int base = n_pairs / world, rem = n_pairs % world;
    int t = 0;
    for (int r = 0; r < world; ++r) {
        const int k = base + (r < rem ? 1 : 0);
        out.emplace_back(t, t + k);
        t += k;
    }
    return out;
}

static std::string frame_file(const std::string& seq_base, int side, int index, const std::string& ext) {
    char buf[64];
    std::snprintf(buf, sizeof buf, "/image_%d/%06d", side, index);
    return seq_base + buf + ext;
}

static bool readable(const std::string& f) {
    FILE* fp = std::fopen(f.c_str(), "rb");
    if (fp) std::fclose(fp);
    return fp != nullptr;
}

std::string kitti_image_ext(const std::string& seq_base, int begin) {
    return readable(frame_file(seq_base, 0, begin, ".png")) ? ".png" : ".pgm";
}

int kitti_count_frames(const std::string& seq_base, int begin, int end) {
    const std::string ext = kitti_image_ext(seq_base, begin);
    int n = 0;
    for (long i = begin; i <= end; ++i, ++n)
        if (!readable(frame_file(seq_base, 0, (int)i, ext)) || !readable(frame_file(seq_base, 1, (int)i, ext))) break;
    return n;
}

std::vector<FrameRecord> kitti_run_range(const std::string& seq_base, const Matd& P1, const Matd& P2, int begin,
                                         int first, int last, int device, int chunk, uint64_t ransac_seed,
                                         int decode_threads, OdometryStats* stats) {
    std::vector<FrameRecord> rec;
    if (last <= first) return rec;
    const std::string ext = kitti_image_ext(seq_base, begin);
    StereoImageGenerator images({seq_base + "/image_0/%06d" + ext, seq_base + "/image_1/%06d" + ext},
                                begin + first, begin + last);
    OdometryResult res = sequence_odometry(P1, P2, images, chunk, ransac_seed, (uint64_t)(begin + first), device, decode_threads);
    if (stats) *stats = res.stats;
    // res.ok / res.tr / res.n_inliers: one entry per frame read, entry 0 = this range's first frame (no pose)
    for (size_t t = 1; t < res.ok.size(); ++t) {
        FrameRecord r;
        for (int j = 0; j < 6; ++j) r.tr[j] = res.tr[t][(size_t)j];
        r.ok = res.ok[t];
        r.n_inl = res.n_inliers[t];
        r.frame = begin + first + (int)t;
        r.reserved = 0;
        rec.push_back(r);
    }
    return rec;
}

std::vector<Matd> chain_records(const FrameRecord* rec, int n, bool reference_pose_list) {
    std::vector<Matd> poses;
    poses.push_back(Matd::eye(4));                                   // src/viso.cpp:1189-1190
    double pose[16];
    std::memcpy(pose, poses[0].ptr(), sizeof pose);
    for (int i = 0; i < n; ++i) {
        if (!rec[i].ok) continue;                                    // :1287, :1323: nothing is pushed
        viso_pose_update(pose, rec[i].tr, pose);                     // :1315-1321
        Matd P(4, 4);
        std::memcpy(P.ptr(), pose, sizeof pose);
        if (reference_pose_list) poses.back() = P;                   // :1317-1319: the product lands in poses.back()'s buffer
        poses.push_back(P);                                          // :1321: ... and its clone is pushed
    }
    return poses;
}

static const int32_t REC_MAGIC = 0x56534B52;   // "VSKR"

bool write_records(const std::string& file_name, int first, int last, const std::vector<FrameRecord>& rec) {
    // written under a temporary name and renamed: a reader never sees a partial file
    const std::string tmp = file_name + ".tmp";
    FILE* fp = std::fopen(tmp.c_str(), "wb");
    if (!fp) return false;
    const int32_t hdr[4] = {REC_MAGIC, first, last, (int32_t)rec.size()};
    bool ok = std::fwrite(hdr, sizeof hdr, 1, fp) == 1;
    if (ok && !rec.empty()) ok = std::fwrite(rec.data(), sizeof(FrameRecord), rec.size(), fp) == rec.size();
    ok = (std::fclose(fp) == 0) && ok;
    return ok && std::rename(tmp.c_str(), file_name.c_str()) == 0;
}

bool read_records(const std::string& file_name, int& first, int& last, std::vector<FrameRecord>& rec) {
    FILE* fp = std::fopen(file_name.c_str(), "rb");
    if (!fp) return false;
    int32_t hdr[4];
    bool ok = std::fread(hdr, sizeof hdr, 1, fp) == 1 && hdr[0] == REC_MAGIC && hdr[3] >= 0 && hdr[2] >= hdr[1] &&
              hdr[3] <= hdr[2] - hdr[1];
    if (ok) {
        first = hdr[1]; last = hdr[2];
        rec.resize((size_t)hdr[3]);
        if (hdr[3]) ok = std::fread(rec.data(), sizeof(FrameRecord), rec.size(), fp) == rec.size();
    }
    std::fclose(fp);
    return ok;
}

std::vector<FrameRecord> stitch_records(const std::vector<std::vector<FrameRecord>>& parts,
                                        const std::vector<std::pair<int, int>>& ranges) {
    std::vector<FrameRecord> all;
    for (size_t r = 0; r < parts.size() && r < ranges.size(); ++r) {
        all.insert(all.end(), parts[r].begin(), parts[r].end());
        if ((int)parts[r].size() < ranges[r].second - ranges[r].first) break;   // the sequence ends here for one process too
    }
    return all;
}

void mkdirs(const std::string& path) {
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') ::mkdir(path.substr(0, i).c_str(), 0777);
}

}  // namespace viso

// ---- C entry points ---------------------------------------------------------------------------------------
static thread_local std::string g_host_err;
static thread_local viso::OdometryStats g_last_stats;
static thread_local int g_decode_threads = 0;

extern "C" const char* viso_host_last_error(void) { return g_host_err.c_str(); }
namespace viso { void set_host_error(const std::string& s) { g_host_err = s; } }   // for the other C entry points (drop_in.cpp)

extern "C" int viso_kitti_count_frames(const char* seq_base, int begin, int end) {
    if (!seq_base || begin < 0 || end < begin) { g_host_err = "viso_kitti_count_frames: bad argument"; return VISO_ERR_ARG; }
    return viso::kitti_count_frames(seq_base, begin, end);
}

extern "C" int viso_kitti_run_range(const char* seq_base, int begin, int first, int last, int device, int chunk,
                                    uint64_t ransac_seed, double* rec8, int* n_done) {
    if (!seq_base || begin < 0 || first < 0 || last < first || !n_done || (last > first && !rec8)) {
        g_host_err = "viso_kitti_run_range: bad argument";
        return VISO_ERR_ARG;
    }
    *n_done = 0;
    try {
        viso::Matd P1, P2;
        if (!viso::loadCalib(std::string(seq_base) + "/calib.txt", P1, P2)) {
            g_host_err = std::string("cannot read ") + seq_base + "/calib.txt";
            return VISO_ERR_ARG;
        }
        g_last_stats = viso::OdometryStats();
        std::vector<viso::FrameRecord> rec = viso::kitti_run_range(seq_base, P1, P2, begin, first, last, device, chunk, ransac_seed,
                                                                   g_decode_threads, &g_last_stats);
        for (size_t i = 0; i < rec.size(); ++i) {
            for (int j = 0; j < 6; ++j) rec8[i * 8 + (size_t)j] = rec[i].tr[j];
            rec8[i * 8 + 6] = rec[i].ok;
            rec8[i * 8 + 7] = rec[i].n_inl;
        }
        *n_done = (int)rec.size();
        return VISO_OK;
    } catch (const std::exception& e) {
        g_host_err = e.what();
        return VISO_ERR_HIP;
    }
}

extern "C" void viso_kitti_last_stats(double out[9]) {
    const viso::OdometryStats& s = g_last_stats;
    const double v[9] = {(double)s.frames, (double)s.decode_threads, s.wall_s, s.decode_wait_s, s.decode_cpu_s, s.issue_s,
                         s.drain_wait_s, s.upload_ms, s.gpu_ms};
    for (int i = 0; i < 9; ++i) out[i] = v[i];
}

extern "C" void viso_kitti_set_decode_threads(int n) { g_decode_threads = n > 0 ? n : 0; }

extern "C" int viso_kitti_write_poses(const char* file_name, const double* rec8, int n, int* n_poses) {
    return viso_kitti_write_poses2(file_name, rec8, n, 0, n_poses);
}

extern "C" int viso_kitti_write_poses2(const char* file_name, const double* rec8, int n, int reference_pose_list, int* n_poses) {
    if (!file_name || n < 0 || (n > 0 && !rec8)) { g_host_err = "viso_kitti_write_poses: bad argument"; return VISO_ERR_ARG; }
    std::vector<viso::FrameRecord> rec((size_t)n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < 6; ++j) rec[(size_t)i].tr[j] = rec8[(size_t)i * 8 + (size_t)j];
        rec[(size_t)i].ok = (int32_t)rec8[(size_t)i * 8 + 6];
        rec[(size_t)i].n_inl = (int32_t)rec8[(size_t)i * 8 + 7];
        rec[(size_t)i].frame = rec[(size_t)i].reserved = 0;
    }
    std::vector<viso::Matd> poses = viso::chain_records(rec.data(), n, reference_pose_list != 0);
    const std::string f = file_name;
    const size_t slash = f.rfind('/');
    if (slash != std::string::npos && slash > 0) viso::mkdirs(f.substr(0, slash));
    if (!viso::savePoses(f, poses)) { g_host_err = "cannot write " + f; return VISO_ERR_ARG; }
    if (n_poses) *n_poses = (int)poses.size();
    return VISO_OK;
}
