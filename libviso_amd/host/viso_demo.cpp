// viso_demo — the reference's `kitti` driver (src/kitti.cpp:79-118) with the
// image front-end replaced by a feature file: reads calib.txt, pulls stereo
// features frame by frame, runs sequence_odometry on the GPU and writes the
// pose file in the KITTI format.
//
//   viso_demo <features.bin> <calib.txt> <poses_out.txt> [chunk] [seed]
//
// features.bin: int32 magic 0x5653464D, nf, cap, dlen; int32 n[nf][2];
//               float kp[nf][2][cap][2]; float desc[nf][2][cap][dlen]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kitti_io.hpp"
#include "viso.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: viso_demo features.bin calib.txt poses_out.txt [chunk] [seed]\n"); return 1; }
    const int chunk = argc > 4 ? std::atoi(argv[4]) : 64;
    const uint64_t seed = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 0;
    viso::Matd P1, P2;
    if (!viso::loadCalib(argv[2], P1, P2)) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    FILE* fp = std::fopen(argv[1], "rb");
    if (!fp) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    int32_t hdr[4];
    if (std::fread(hdr, sizeof(int32_t), 4, fp) != 4 || hdr[0] != 0x5653464D) { std::fprintf(stderr, "bad feature file\n"); return 2; }
    const int nf = hdr[1], cap = hdr[2], dlen = hdr[3];
    std::vector<int32_t> n((size_t)nf * 2);
    std::vector<float> kp((size_t)nf * 2 * cap * 2), desc((size_t)nf * 2 * cap * dlen);
    if (std::fread(n.data(), sizeof(int32_t), n.size(), fp) != n.size() ||
        std::fread(kp.data(), sizeof(float), kp.size(), fp) != kp.size() ||
        std::fread(desc.data(), sizeof(float), desc.size(), fp) != desc.size()) { std::fprintf(stderr, "short feature file\n"); return 2; }
    std::fclose(fp);
    int t = 0;
    viso::StereoFeatureGenerator gen = [&]() -> std::optional<viso::StereoFeatures> {
        if (t >= nf) return std::nullopt;
        viso::StereoFeatures f;
        for (int side = 0; side < 2; ++side) {
            const int cnt = n[(size_t)t * 2 + side];
            viso::KeyPoints& k = side ? f.kp2 : f.kp1;
            viso::Descriptors& d = side ? f.d2 : f.d1;
            k.resize((size_t)cnt);
            d.create(cnt, dlen);
            const float* ks = kp.data() + ((size_t)t * 2 + side) * cap * 2;
            const float* ds = desc.data() + ((size_t)t * 2 + side) * cap * dlen;
            for (int i = 0; i < cnt; ++i) { k[(size_t)i].pt.x = ks[2 * i]; k[(size_t)i].pt.y = ks[2 * i + 1]; k[(size_t)i].size = 11; }
            std::copy(ds, ds + (size_t)cnt * dlen, d.data.begin());
        }
        ++t;
        return f;
    };
    try {
        viso::OdometryResult res = viso::sequence_odometry(P1, P2, gen, chunk, seed);
        if (!viso::savePoses(argv[3], res.poses)) { std::fprintf(stderr, "cannot write %s\n", argv[3]); return 3; }
        int solved = 0;
        for (int v : res.ok) solved += v;
        std::printf("frames %d solved %d poses %zu\n", nf, solved, res.poses.size());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 4;
    }
    return 0;
}
