// kitti_io.hpp — the two file formats the reference's kitti driver reads and
// writes (alexkreimer/libviso src/kitti.cpp:23-64), restated:
//   calib.txt : "P0: <12 doubles>\nP1: <12 doubles>" (3x4 row-major each; the
//               driver reads the first two lines, :31-44)
//   poses     : one line per pose, 12 x "%lf" = first three rows of the 4x4
//               pose, six decimals (:56-60)
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "viso.hpp"

namespace viso {

inline bool loadCalib(const std::string& file_name, Matd& p1, Matd& p2) {
    FILE* fp = std::fopen(file_name.c_str(), "r");
    if (!fp) return false;
    p1.create(3, 4); p2.create(3, 4);
    int n = 0;
    bool ok = std::fscanf(fp, "P%d:", &n) == 1;
    for (int i = 0; ok && i < 12; ++i) ok = std::fscanf(fp, "%lf", &p1.data[(size_t)i]) == 1;
    ok = ok && std::fscanf(fp, " P%d:", &n) == 1;
    for (int i = 0; ok && i < 12; ++i) ok = std::fscanf(fp, "%lf", &p2.data[(size_t)i]) == 1;
    std::fclose(fp);
    return ok;
}

inline bool savePoses(const std::string& file_name, const std::vector<Matd>& poses) {
    FILE* fp = std::fopen(file_name.c_str(), "w+");
    if (!fp) return false;
    for (const Matd& pose : poses) {
        std::fprintf(fp, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf\n",
                     pose.at(0, 0), pose.at(0, 1), pose.at(0, 2), pose.at(0, 3),
                     pose.at(1, 0), pose.at(1, 1), pose.at(1, 2), pose.at(1, 3),
                     pose.at(2, 0), pose.at(2, 1), pose.at(2, 2), pose.at(2, 3));
    }
    std::fclose(fp);
    return true;
}

}  // namespace viso
