// selftest — host-only checks of the C++ mirror (no GPU needed): the known
// answers of the reference's own (disabled) tests.
//   test_solveRigidMotion  reference test/test.cpp:171-205
//   test_F_from_P          reference src/mvg.cpp:73-89
//   loadCalib / savePoses  reference src/kitti.cpp:23-64 (format round trip)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "kitti_io.hpp"
#include "viso.hpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(int argc, char** argv) {
    using namespace viso;
    {   // 4 points, R = 90 deg about X, t = (1,2,3); ||T - T^||_F^2 < 1e-12 (float data: 1e-10)
        const float X[3][4] = {{0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
        Matf B(3, 4), A(3, 4), T;
        const float R[3][3] = {{1, 0, 0}, {0, 0, -1}, {0, 1, 0}};
        const float t[3] = {1, 2, 3};
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 3; ++r) {
                B.at(r, i) = X[r][i];
                A.at(r, i) = R[r][0] * X[0][i] + R[r][1] * X[1][i] + R[r][2] * X[2][i] + t[r];
            }
        solveRigidMotion(A, B, T);
        double e = 0;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) e += std::pow(T.at(r, c) - R[r][c], 2);
            e += std::pow(T.at(r, 3) - t[r], 2);
        }
        CHECK(e < 1e-10);
        // a reflection-prone configuration (coplanar points): det(R) must stay +1
        Matf A2(3, 4), B2(3, 4), T2;
        const float P[3][4] = {{0, 1, 0, 1}, {0, 0, 1, 1}, {0, 0, 0, 0}};
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 3; ++r) { B2.at(r, i) = P[r][i]; A2.at(r, i) = R[r][0] * P[0][i] + R[r][1] * P[1][i] + R[r][2] * P[2][i] + t[r]; }
        solveRigidMotion(A2, B2, T2);
        const double det = T2.at(0, 0) * (T2.at(1, 1) * T2.at(2, 2) - T2.at(1, 2) * T2.at(2, 1)) -
                           T2.at(0, 1) * (T2.at(1, 0) * T2.at(2, 2) - T2.at(1, 2) * T2.at(2, 0)) +
                           T2.at(0, 2) * (T2.at(1, 0) * T2.at(2, 1) - T2.at(1, 1) * T2.at(2, 0));
        CHECK(std::fabs(det - 1.0) < 1e-5);
        double e2 = 0;
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 3; ++r) {
                const double p = T2.at(r, 0) * P[0][i] + T2.at(r, 1) * P[1][i] + T2.at(r, 2) * P[2][i] + T2.at(r, 3);
                e2 += std::pow(p - A2.at(r, i), 2);
            }
        CHECK(e2 < 1e-9);
        bool threw = false;
        try { Matf a(3, 1), b(3, 1), tt; solveRigidMotion(a, b, tt); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);   // BOOST_ASSERT_MSG(A.cols() > 1), src/estimation.cpp:32
    }
    {   // F_from_P known answer
        Matd P1(3, 4), P2(3, 4);
        for (int i = 0; i < 3; ++i) { P1.at(i, i) = 1; P2.at(i, i) = 1; }
        P2.at(0, 3) = 1;
        Matd F = F_from_P(P1, P2);
        const double want[9] = {0, 0, 0, 0, 0, 1, 0, -1, 0};
        for (int i = 0; i < 9; ++i) CHECK(F.data[i] == want[i]);
    }
    {   // tr2mat: rotation block orthonormal, translation in the last column
        Matd T;
        tr2mat({0.1, -0.2, 0.3, 1, 2, 3}, T);
        double o = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += T.at(i, k) * T.at(j, k);
                o += std::fabs(s - (i == j));
            }
        CHECK(o < 1e-14 && T.at(0, 3) == 1 && T.at(1, 3) == 2 && T.at(2, 3) == 3 && T.at(3, 3) == 1);
    }
    if (argc > 1) {   // calib.txt / pose file round trip in the given directory
        std::string dir = argv[1];
        FILE* fp = std::fopen((dir + "/calib.txt").c_str(), "w");
        std::fprintf(fp, "P0: 7.188560000000e+02 0 6.071928000000e+02 0 0 7.188560000000e+02 1.852157000000e+02 0 0 0 1 0\n");
        std::fprintf(fp, "P1: 7.188560000000e+02 0 6.071928000000e+02 -3.861448000000e+02 0 7.188560000000e+02 1.852157000000e+02 0 0 0 1 0\n");
        std::fprintf(fp, "P2: 1 2 3 4 5 6 7 8 9 10 11 12\n");
        std::fclose(fp);
        Matd P1, P2;
        CHECK(loadCalib(dir + "/calib.txt", P1, P2));
        CHECK(P1.at(0, 0) == 718.856 && P2.at(0, 3) == -386.1448 && P1.at(1, 2) == 185.2157 && P2.at(2, 2) == 1);
        CHECK(!loadCalib(dir + "/missing.txt", P1, P2));
        std::vector<Matd> poses(2, Matd::eye(4));
        poses[1].at(0, 3) = 1.2345678;
        CHECK(savePoses(dir + "/poses.txt", poses));
        fp = std::fopen((dir + "/poses.txt").c_str(), "r");
        char line[512];
        CHECK(std::fgets(line, sizeof line, fp) && std::string(line) == "1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000\n");
        CHECK(std::fgets(line, sizeof line, fp) && std::string(line).substr(0, 37) == "1.000000 0.000000 0.000000 1.234568 0");
        std::fclose(fp);
    }
    int i = 2;
    for (; i + 1 < argc && std::string(argv[i]) != "--bad"; i += 2) {   // pairs (png, pgm) that must decode to the same pixels
        Image a = imread_gray(argv[i]), b = imread_gray(argv[i + 1]);
        CHECK(!a.empty() && !b.empty() && a.rows == b.rows && a.cols == b.cols && a.data == b.data);
    }
    for (++i; i < argc; ++i) {   // behind "--bad": malformed / hostile files that must be refused (and must not exhaust memory)
        Image a = imread_gray(argv[i]);
        if (!a.empty()) std::printf("accepted a malformed file: %s\n", argv[i]);
        CHECK(a.empty());
    }
    CHECK(imread_gray("/nonexistent/file.png").empty());
    std::printf(fails ? "selftest: %d failure(s)\n" : "selftest ok\n", fails);
    return fails ? 1 : 0;
}
