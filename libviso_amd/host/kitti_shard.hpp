// kitti_shard.hpp — one KITTI sequence over W GPUs (BASELINE configs[3]).
//
// The reference's `kitti` driver (src/kitti.cpp:79-118) is one process over one sequence.  The only cross-frame
// dependency of its loop is the *_prev state (src/viso.cpp:1208-1222): the relative motion of frame t needs frames
// t-1 and t, and the trajectory is the prefix product pose_t = pose_{t-1} * inv(Tr_t) (:1315-1321).  So:
//   partition   rank r of W takes a contiguous range of frame pairs, with a one-frame halo (it re-does detection,
//               description and the stereo match of its first frame); same rule as libviso_amd/shard.py
//   keys        RANSAC triples are drawn from a stream keyed on the ABSOLUTE KITTI frame index (the file number),
//               so a frame's record does not depend on the partition, on `begin`, or on the chunking
//   records     per frame pair {tr[6], ok, n_inl} = 64 B: what crosses ranks, once (RCCL all-gather in
//               libviso_amd/kitti_shard.py, or rank files read by `viso_kitti --gather`)
//   chain       the host prefix product over the gathered records and the KITTI pose file (src/kitti.cpp:49-64)
// Any W gives the byte-identical pose file (tests/test_gpu_kitti_shard.py).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "viso.hpp"

namespace viso {

// one frame pair (t-1, t): what sequence_odometry's loop body leaves behind (src/viso.cpp:1313-1324)
struct FrameRecord {
    double tr[6];
    int32_t ok;
    int32_t n_inl;
    int32_t frame;      // absolute KITTI index of the pair's second frame (the file number)
    int32_t reserved;
};
static_assert(sizeof(FrameRecord) == 64, "FrameRecord is the 64-byte wire format");

// [(first, last)] per rank over frames 0..n_frames-1 (relative to `begin`): rank r solves the pairs (t-1, t) for
// t in (first, last]; frames first..last are read by it.  Sizes differ by at most one, larger ranges first.
std::vector<std::pair<int, int>> partition(int n_frames, int world);

// "<seq_base>/image_0/%06d" + ".png" if frame `begin` exists as PNG, ".pgm" otherwise (src/kitti.cpp:108-110)
std::string kitti_image_ext(const std::string& seq_base, int begin);
// number of consecutive frames begin, begin+1, ... <= end for which both image files can be opened
int kitti_count_frames(const std::string& seq_base, int begin, int end);

// Frames [begin + first, begin + last] of the sequence through the device pipeline (device ordinal `device`):
// rec[i] = record of the pair ending at frame first + 1 + i.  Fewer than last - first records come back when an
// image in the range cannot be decoded (the reference's generator stops there, src/viso.h:94-96).
// stats (may be null): where the range's wall time went (decode / upload / GPU), see OdometryStats.
std::vector<FrameRecord> kitti_run_range(const std::string& seq_base, const Matd& P1, const Matd& P2, int begin,
                                         int first, int last, int device, int chunk = 64, uint64_t ransac_seed = 0,
                                         int decode_threads = 0, OdometryStats* stats = nullptr);

// poses[0] = I, then pose <- pose * inv(tr2mat(tr)) per solved record (src/viso.cpp:1189-1190, 1315-1321): the list
// [I, P1, ..., Pn] the reference's code reads as.
// reference_pose_list = true: the list the reference actually WRITES.  `Mat pose = poses.back(); pose =
// pose*tr_mat.inv(); poses.push_back(pose.clone());` (src/viso.cpp:1317-1321) assigns the product into a header that
// shares poses.back()'s buffer (cv::Mat is reference counted; the GEMM result is copied into the existing buffer), so
// the previous entry is overwritten before the clone is pushed: [P1, P2, ..., Pn, Pn] — no identity line, the last
// pose twice, the same number of lines.
std::vector<Matd> chain_records(const FrameRecord* rec, int n, bool reference_pose_list = false);

// rank files of `viso_kitti --rank r --world W`: header {magic, first, last, n_done} + n_done records
bool write_records(const std::string& file_name, int first, int last, const std::vector<FrameRecord>& rec);
bool read_records(const std::string& file_name, int& first, int& last, std::vector<FrameRecord>& rec);
// Records of all ranks in rank order -> one list; stops at the first rank that came back short (see above).
std::vector<FrameRecord> stitch_records(const std::vector<std::vector<FrameRecord>>& parts,
                                        const std::vector<std::pair<int, int>>& ranges);

void mkdirs(const std::string& path);

}  // namespace viso

// C entry points for the Python launcher (libviso_amd/kitti_shard.py: one rank per GPU under torch.distributed,
// records all-gathered over RCCL).  Return 1, or a negative code with the text in viso_host_last_error().
extern "C" {
int viso_kitti_count_frames(const char* seq_base, int begin, int end);
// rec8: [(last - first)][8] doubles = tr[6], ok, n_inl per pair; *n_done = pairs actually solved or failed (not skipped)
int viso_kitti_run_range(const char* seq_base, int begin, int first, int last, int device, int chunk,
                         uint64_t ransac_seed, double* rec8, int* n_done);
// OdometryStats of this thread's last viso_kitti_run_range: frames, decode_threads, wall_s, decode_wait_s,
// decode_cpu_s, issue_s, drain_wait_s, upload_ms, gpu_ms
void viso_kitti_last_stats(double out[9]);
// worker threads the next viso_kitti_run_range calls of this thread decode with (0 = default)
void viso_kitti_set_decode_threads(int n);
// chain n records and write the KITTI pose file (directories are created); *n_poses = lines written
int viso_kitti_write_poses(const char* file_name, const double* rec8, int n, int* n_poses);
// the same with the pose list the reference writes ([P1..Pn, Pn], see chain_records) when reference_pose_list != 0
int viso_kitti_write_poses2(const char* file_name, const double* rec8, int n, int reference_pose_list, int* n_poses);
const char* viso_host_last_error(void);
}
