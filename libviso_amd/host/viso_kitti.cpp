// viso_kitti — the reference's `kitti` driver (src/kitti.cpp:79-118) on the GPU
// pipeline: $KITTI_HOME/sequences/<seq>/{calib.txt,image_0/%06d.png,image_1/%06d.png}
// in, $KITTI_HOME/results/<seq>/<result_sha>/data/<seq>.txt out.  Images: KITTI's 8-bit
// grayscale PNGs (image_0/%06d.png, own decoder) or binary PGM (image_0/%06d.pgm).
//
//   KITTI_HOME=... viso_kitti result_sha seq_name [begin [end]] [options]
//
// One sequence over W GPUs (BASELINE configs[3], kitti_shard.hpp): frames shard into W contiguous
// ranges with a one-frame halo, each range is one process on one GPU, 64-byte records per frame pair
// are gathered once and chained on the host.  Any W writes the byte-identical pose file.
//   --gpus W            fork W rank processes (rank r on device r), wait, gather their rank files, write the poses
//   --rank r --world W  run range r only and write <result_dir>/shards/<seq>.<r>of<W>.rec   (one per GPU / node)
//   --gather W          read the W rank files and write the pose file
//   --device d          HIP device ordinal of this process (default: the rank, 0 without --rank)
//   --same-device       with --gpus: every rank on device 0 (rehearsal on a one-GPU box)
//   --chunk n           frames per device batch (default 64)      --seed s   RANSAC stream seed (default 0)
//   --decode-threads n  PNG decoding threads per rank (default: min(64, usable host threads / ranks): decoding bounds the run; usable = hardware, affinity, cgroup quota)
//   --reference-pose-list   write the list the reference's code actually produces, [P1, ..., Pn, Pn] (src/viso.cpp:1317-1321
//                       overwrites poses.back() before pushing the clone), instead of [I, P1, ..., Pn]
// Every rank reports where its wall time went: decode (PNG inflate on the worker threads; the calling thread's wait for
// it is the runner's critical path), upload and GPU seconds from time stamps on the device.
// libviso_amd/kitti_shard.py is the same runner with the gather as an RCCL all-gather (torch.distributed).
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#include "kitti_io.hpp"
#include "kitti_shard.hpp"
#include "viso.hpp"

namespace {
struct Args {
    const char* result_sha = nullptr;
    std::string seq_name;
    int begin = 0, end = INT_MAX;
    int gpus = 0, rank = -1, world = 0, gather = 0, device = -1, chunk = 64, decode_threads = 0;
    bool same_device = false, reference_pose_list = false;
    unsigned long long seed = 0;
};

bool parse(int argc, char** argv, Args& a) {
    int pos = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        auto val = [&](int& dst) { if (i + 1 >= argc) return false; dst = std::atoi(argv[++i]); return true; };
        if (s == "--gpus") { if (!val(a.gpus)) return false; }
        else if (s == "--rank") { if (!val(a.rank)) return false; }
        else if (s == "--world") { if (!val(a.world)) return false; }
        else if (s == "--gather") { if (!val(a.gather)) return false; }
        else if (s == "--device") { if (!val(a.device)) return false; }
        else if (s == "--chunk") { if (!val(a.chunk)) return false; }
        else if (s == "--decode-threads") { if (!val(a.decode_threads)) return false; }
        else if (s == "--reference-pose-list") a.reference_pose_list = true;
        else if (s == "--seed") { if (i + 1 >= argc) return false; a.seed = std::strtoull(argv[++i], nullptr, 10); }
        else if (s == "--same-device") a.same_device = true;
        else if (s.rfind("--", 0) == 0) return false;
        else {
            if (pos == 0) a.result_sha = argv[i];
            else if (pos == 1) a.seq_name = s;
            else if (pos == 2) a.begin = std::atoi(argv[i]);                                   // src/kitti.cpp:86-94
            else if (pos == 3) a.end = std::atoi(argv[i]);
            else return false;
            ++pos;
        }
    }
    if (pos < 2) return false;
    if ((a.rank >= 0) != (a.world > 0)) return false;
    if (a.rank >= a.world && a.world > 0) return false;
    return true;
}

std::string rank_file(const std::string& result_dir, const std::string& seq, int r, int w) {
    return result_dir + "/shards/" + seq + "." + std::to_string(r) + "of" + std::to_string(w) + ".rec";
}

void print_stats(const char* who, const viso::OdometryStats& s) {
    std::printf("%s: %d frames in %.3f s (%.0f frames/s) | decode: %d threads, %.3f s of thread time, runner waited %.3f s | "
                "GPU stamps: upload %.3f s, kernels %.3f s | host: issue %.3f s, waiting for results %.3f s\n",
                who, s.frames, s.wall_s, s.wall_s > 0 ? s.frames / s.wall_s : 0.0, s.decode_threads, s.decode_cpu_s, s.decode_wait_s,
                s.upload_ms * 1e-3, s.gpu_ms * 1e-3, s.issue_s, s.drain_wait_s);
}

int threads_per_rank(int ranks) {
    const int share = viso::cpu_budget() / (ranks > 0 ? ranks : 1);   // hardware threads, affinity and cgroup quota
    return share < 1 ? 1 : (share > 64 ? 64 : share);
}
}  // namespace

int main(int argc, char** argv) {
    Args a;
    if (!parse(argc, argv, a)) {
        std::printf("usage: demo result_sha seq_name begin end [--gpus W | --rank r --world W | --gather W] "
                    "[--device d] [--same-device] [--chunk n] [--seed s] [--decode-threads n] [--reference-pose-list]\n");   // :81-85
        return 1;
    }
    const char* home = std::getenv("KITTI_HOME");                                              // :96
    if (!home) { std::fprintf(stderr, "KITTI_HOME is not set\n"); return 2; }
    const std::string seq_base = std::string(home) + "/sequences/" + a.seq_name;
    const std::string result_dir = std::string(home) + "/results/" + a.seq_name + "/" + a.result_sha;   // :100
    const std::string out = result_dir + "/data/" + a.seq_name + ".txt";                       // :114
    viso::Matd P1, P2;
    if (!viso::loadCalib(seq_base + "/calib.txt", P1, P2)) { std::fprintf(stderr, "cannot read %s/calib.txt\n", seq_base.c_str()); return 2; }
    const int n_frames = viso::kitti_count_frames(seq_base, a.begin, a.end);

    // ---- --gpus W: fork the ranks BEFORE this process touches the GPU (the parent never does: nothing above makes a
    // HIP call), each child carries on below as `--rank r --world W`; the parent waits and gathers.  fork without exec:
    // a child initialises its own HIP runtime on its first call ----
    const auto t_start = std::chrono::steady_clock::now();
    if (a.decode_threads <= 0 && !std::getenv("VISO_DECODE_THREADS"))
        a.decode_threads = threads_per_rank(a.gpus > 1 ? a.gpus : 1);
    if (a.gpus > 1) {
        std::fflush(nullptr);
        std::vector<pid_t> kids;
        bool child = false;
        for (int r = 0; r < a.gpus && !child; ++r) {
            const pid_t pid = ::fork();
            if (pid < 0) { std::perror("fork"); return 5; }
            if (pid == 0) {
                child = true;
                a.rank = r; a.world = a.gpus; a.gpus = 0;
                if (a.device < 0) a.device = a.same_device ? 0 : r;
            } else {
                kids.push_back(pid);
            }
        }
        if (!child) {
            int bad = 0;
            for (pid_t pid : kids) {
                int st = 0;
                if (::waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad;
            }
            if (bad) { std::fprintf(stderr, "%d of %d rank processes failed\n", bad, (int)kids.size()); return 6; }
            a.gather = (int)kids.size();
        }
    }

    try {
        // ---- --gather W: rank files -> one pose file ----
        if (a.gather > 0) {
            const auto ranges = viso::partition(n_frames, a.gather);
            std::vector<std::vector<viso::FrameRecord>> parts((size_t)a.gather);
            for (int r = 0; r < a.gather; ++r) {
                int first = 0, last = 0;
                const std::string f = rank_file(result_dir, a.seq_name, r, a.gather);
                if (!viso::read_records(f, first, last, parts[(size_t)r])) { std::fprintf(stderr, "cannot read %s\n", f.c_str()); return 3; }
                if (first != ranges[(size_t)r].first || last != ranges[(size_t)r].second) {
                    std::fprintf(stderr, "%s covers frames %d..%d, expected %d..%d (begin/end changed?)\n", f.c_str(), first, last,
                                 ranges[(size_t)r].first, ranges[(size_t)r].second);
                    return 3;
                }
            }
            const std::vector<viso::FrameRecord> all = viso::stitch_records(parts, ranges);
            const std::vector<viso::Matd> poses = viso::chain_records(all.data(), (int)all.size(), a.reference_pose_list);
            viso::mkdirs(result_dir + "/data");
            if (!viso::savePoses(out, poses)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 3; }
            int solved = 0;
            for (const auto& r : all) solved += r.ok;
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            std::printf("frames %zu solved %d poses %zu ranks %d -> %s (%.3f s, %.0f frames/s of the runner)\n", all.size() + (n_frames > 0),
                        solved, poses.size(), a.gather, out.c_str(), wall, wall > 0 ? (double)(all.size() + (n_frames > 0)) / wall : 0.0);
            return 0;
        }
        // ---- --rank r --world W: this rank's range -> its rank file ----
        if (a.world > 0) {
            const auto range = viso::partition(n_frames, a.world)[(size_t)a.rank];
            const int device = a.device >= 0 ? a.device : a.rank;
            viso::OdometryStats stats;
            std::vector<viso::FrameRecord> rec = viso::kitti_run_range(seq_base, P1, P2, a.begin, range.first, range.second,
                                                                       device, a.chunk, a.seed, a.decode_threads, &stats);
            viso::mkdirs(result_dir + "/shards");
            const std::string f = rank_file(result_dir, a.seq_name, a.rank, a.world);
            if (!viso::write_records(f, range.first, range.second, rec)) { std::fprintf(stderr, "cannot write %s\n", f.c_str()); return 3; }
            std::printf("rank %d/%d device %d frames %d..%d pairs %zu -> %s\n", a.rank, a.world, device, a.begin + range.first,
                        a.begin + range.second, rec.size(), f.c_str());
            print_stats(("rank " + std::to_string(a.rank)).c_str(), stats);
            return 0;
        }
        // ---- one process, one GPU: the reference's flow (:108-116) ----
        const std::string ext = viso::kitti_image_ext(seq_base, a.begin);
        viso::StereoImageGenerator images({seq_base + "/image_0/%06d" + ext, seq_base + "/image_1/%06d" + ext}, a.begin, a.end);
        viso::OdometryResult res = viso::sequence_odometry(P1, P2, images, a.chunk, a.seed, (uint64_t)a.begin,
                                                           a.device >= 0 ? a.device : 0, a.decode_threads);      // :111
        viso::mkdirs(result_dir + "/data");                                                    // :112-113
        if (a.reference_pose_list && res.poses.size() > 1) {                                   // [P1, ..., Pn, Pn], see kitti_shard.hpp
            res.poses.erase(res.poses.begin());
            res.poses.push_back(res.poses.back());
        }
        if (!viso::savePoses(out, res.poses)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 3; }
        int solved = 0;
        for (int v : res.ok) solved += v;
        std::printf("frames %zu solved %d poses %zu -> %s\n", res.ok.size(), solved, res.poses.size(), out.c_str());
        print_stats("one process", res.stats);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 4;
    }
    return 0;
}
