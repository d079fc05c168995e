// cvlite.h -- the handful of OpenCV value types that cross the hot-path boundary, as plain
// layout-compatible stand-ins (OpenCV is not a dependency of this build; with a real OpenCV on
// the include path define SFM_HAVE_OPENCV and these are skipped).  Only what
// include/Sfm.h:89,107-117 and include/BundleAdjustment.h:19-20 of the reference mention.
#pragma once
#ifndef SFM_HAVE_OPENCV
#include <cstddef>
#include <cstring>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_8UC3 16  /* CV_MAKETYPE(CV_8U, 3) */

namespace cv {

struct Point2d {
  double x, y;
  Point2d() : x(0), y(0) {}
  Point2d(double x_, double y_) : x(x_), y(y_) {}
};

struct Point3d {
  double x, y, z;
  Point3d() : x(0), y(0), z(0) {}
  Point3d(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};

struct DMatch {  // 16 bytes, as in OpenCV
  int queryIdx, trainIdx, imgIdx;
  float distance;
  DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(0), distance(d) {}
};

struct KeyPoint {
  struct { float x, y; } pt;
  float size, angle, response;
  int octave, class_id;
};

template <typename T, int M, int N>
struct Matx {
  T val[M * N];  // row-major, like cv::Matx
  Matx() { for (int i = 0; i < M * N; ++i) val[i] = T(0); }
  T& operator()(int r, int c) { return val[r * N + c]; }
  const T& operator()(int r, int c) const { return val[r * N + c]; }
};
typedef Matx<double, 3, 4> Matx34d;
typedef Matx<double, 3, 3> Matx33d;

// dense row-major matrix of double (cv::Mat_<double>): K (3x3) and distCoef (1x5)
template <typename T>
struct Mat_ {
  int rows, cols;
  std::vector<T> data;
  Mat_() : rows(0), cols(0) {}
  Mat_(int r, int c) : rows(r), cols(c), data((size_t)r * c, T(0)) {}
  template <typename U> T& at(int r, int c) { return data[(size_t)r * cols + c]; }
  template <typename U> const T& at(int r, int c) const { return data[(size_t)r * cols + c]; }
  T& operator()(int r, int c) { return data[(size_t)r * cols + c]; }
  const T& operator()(int r, int c) const { return data[(size_t)r * cols + c]; }
};

// descriptor matrix: rows = features, row-major contiguous, CV_32F (SIFT) or CV_8U (ORB/AKAZE);
// also the 8-bit images of the loader (CV_8UC1 gray, CV_8UC3 BGR: interleaved channels)
struct Mat {
  int rows, cols, depth, ch;
  std::vector<unsigned char> bytes;
  Mat() : rows(0), cols(0), depth(CV_32F), ch(1) {}
  Mat(int r, int c, int type, const void* src = nullptr) : rows(r), cols(c), depth(type & 7), ch((type >> 3) + 1) {
    bytes.resize((size_t)r * c * elemSize());
    if (src && !bytes.empty()) std::memcpy(bytes.data(), src, bytes.size());
  }
  size_t elemSize() const { return (depth == CV_32F ? 4 : 1) * (size_t)ch; }
  int type() const { return depth | ((ch - 1) << 3); }
  int channels() const { return ch; }
  bool empty() const { return bytes.empty(); }
  const unsigned char* ptr() const { return bytes.data(); }
  unsigned char* ptr() { return bytes.data(); }
};

}  // namespace cv
#endif
