// cv_boost_stub.hpp — TYPE-CHECK STUB, test infrastructure only (tests/test_adapter_compiles.py).
//
// adapters/viso_hip_adapter.inc is compiled by a libviso maintainer inside the reference's src/viso.cpp, against
// OpenCV and Boost, which this build image does not have.  This header declares exactly the OpenCV / Boost names the
// adapter and the few reference declarations in front of it (struct param, MatchParams, kp2mat, triangulate_rectified)
// use, with the signatures of OpenCV 2.4 / 3.0 `core.hpp`, so that `g++ -std=c++11 -fsyntax-only` can check the
// adapter's types and overloads.  It is not OpenCV, implements nothing the product uses, is never linked into anything,
// and is not an oracle: nothing is executed.
#pragma once
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#define CV_32F 5
#define CV_64F 6
#define CV_Assert(expr) do { if (!(expr)) throw 0; } while (0)
#define BOOST_ASSERT_MSG(expr, msg) assert((expr) && (msg))

namespace cv {

template <typename T> struct DataType;
template <> struct DataType<float> { enum { type = CV_32F }; };
template <> struct DataType<double> { enum { type = CV_64F }; };

template <typename T, int N>
class Vec {                                            // opencv2/core/core.hpp: Vec<_Tp, cn>, contiguous val[cn]
public:
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(); }
    Vec(T v0, T v1) { assert(N >= 2); for (int i = 0; i < N; ++i) val[i] = T(); val[0] = v0; val[1] = v1; }
    Vec(T v0, T v1, T v2) { assert(N >= 3); for (int i = 0; i < N; ++i) val[i] = T(); val[0] = v0; val[1] = v1; val[2] = v2; }
    Vec(T v0, T v1, T v2, T v3) { assert(N >= 4); for (int i = 0; i < N; ++i) val[i] = T(); val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; }
    const T& operator[](int i) const { return val[i]; }
    T& operator[](int i) { return val[i]; }
};
typedef Vec<int, 2> Vec2i;
typedef Vec<int, 3> Vec3i;
typedef Vec<int, 4> Vec4i;
typedef Vec<float, 2> Vec2f;

template <typename T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;

class KeyPoint {                                       // opencv2/features2d: pt, size, angle, response, octave, class_id
public:
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};

class Mat {                                            // the members the adapter and the cut declarations touch
public:
    int flags, dims, rows, cols;
    unsigned char* data;
    Mat() : flags(0), dims(0), rows(0), cols(0), data(0) {}
    Mat(int r, int c, int type) : flags(type), dims(2), rows(r), cols(c), data(0) {}
    Mat(const Mat& m) : flags(m.flags), dims(m.dims), rows(m.rows), cols(m.cols), data(m.data) {}
    Mat& operator=(const Mat& m) { flags = m.flags; dims = m.dims; rows = m.rows; cols = m.cols; data = m.data; return *this; }
    void create(int r, int c, int type) { rows = r; cols = c; flags = type; }
    Mat clone() const { return *this; }
    void copyTo(Mat& m) const { m = *this; }
    bool isContinuous() const { return true; }
    int type() const { return flags; }
    bool empty() const { return data == 0; }
    template <typename T> T* ptr(int i0 = 0) { return reinterpret_cast<T*>(data) + (size_t)i0 * cols; }
    template <typename T> const T* ptr(int i0 = 0) const { return reinterpret_cast<const T*>(data) + (size_t)i0 * cols; }
    template <typename T> T& at(int i0, int i1) { return reinterpret_cast<T*>(data)[(size_t)i0 * cols + i1]; }
    template <typename T> const T& at(int i0, int i1) const { return reinterpret_cast<const T*>(data)[(size_t)i0 * cols + i1]; }
};

}  // namespace cv
