// viso_kitti — the reference's `kitti` driver (src/kitti.cpp:79-118) on the GPU
// pipeline: $KITTI_HOME/sequences/<seq>/{calib.txt,image_0/%06d.png,image_1/%06d.png}
// in, $KITTI_HOME/results/<seq>/<result_sha>/data/<seq>.txt out.  Images: KITTI's 8-bit
// grayscale PNGs (image_0/%06d.png, own decoder) or binary PGM (image_0/%06d.pgm).
//
//   KITTI_HOME=... viso_kitti result_sha seq_name [begin [end]]
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <sys/stat.h>

#include "kitti_io.hpp"
#include "viso.hpp"

static void mkdirs(const std::string& path) {
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') ::mkdir(path.substr(0, i).c_str(), 0777);
}

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: demo result_sha seq_name begin end\n"); return 1; }   // :81-85
    int begin = 0, end = INT_MAX;
    if (argc > 3) begin = std::atoi(argv[3]);
    if (argc > 4) end = std::atoi(argv[4]);
    const char* result_sha = argv[1];
    const char* home = std::getenv("KITTI_HOME");                                              // :96
    if (!home) { std::fprintf(stderr, "KITTI_HOME is not set\n"); return 2; }
    const std::string seq_name = argv[2];
    const std::string seq_base = std::string(home) + "/sequences/" + seq_name;
    const std::string result_dir = std::string(home) + "/results/" + seq_name + "/" + result_sha;   // :100
    viso::Matd P1, P2;
    if (!viso::loadCalib(seq_base + "/calib.txt", P1, P2)) { std::fprintf(stderr, "cannot read %s/calib.txt\n", seq_base.c_str()); return 2; }
    // image_0/%06d.png like the reference (:108-110); .pgm if the sequence was converted
    char first[4096];
    std::snprintf(first, sizeof first, (seq_base + "/image_0/%06d.png").c_str(), begin);
    FILE* probe = std::fopen(first, "rb");
    const std::string ext = probe ? ".png" : ".pgm";
    if (probe) std::fclose(probe);
    viso::StereoImageGenerator images({seq_base + "/image_0/%06d" + ext, seq_base + "/image_1/%06d" + ext}, begin, end);
    try {
        viso::OdometryResult res = viso::sequence_odometry(P1, P2, images);                    // :111
        mkdirs(result_dir + "/data");                                                          // :112-113
        const std::string out = result_dir + "/data/" + seq_name + ".txt";                     // :114
        if (!viso::savePoses(out, res.poses)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 3; }
        int solved = 0;
        for (int v : res.ok) solved += v;
        std::printf("frames %zu solved %d poses %zu -> %s\n", res.ok.size(), solved, res.poses.size(), out.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 4;
    }
    return 0;
}
