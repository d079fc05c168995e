// ba_selftest.cpp -- BundleAdjustment::adjustBundle (the reference's signature, include/BundleAdjustment.h:19-20) on a problem of
// any size written by tests/test_gpu_host_cpp.py, in the reference's own containers: what the mirror writes back, and when.
// Reference policy under test (src/BundleAdjustment.cpp:118-129): 500 iterations / 10 s, results written back ONLY on
// CONVERGENCE, otherwise "Bundle adjustment failed." and the inputs untouched.
//   in:  i32 n_cam, n_pt, n_obs | n_cam x f64 pose[12] (row-major [R|t]) | n_pt x f64 xyz[3] | f64 K[9]
//        | n_obs x (i32 view, i32 point) | n_obs x f64 xy[2] (pixel coordinates, principal point included)
//   out: f64 K[9] | n_cam x f64 pose[12] | n_pt x f64 xyz[3]      (the containers after the call)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "BundleAdjustment.h"

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
  if (n && fread(p, sizeof(T), n, f) != n) {
    fprintf(stderr, "ba_selftest: short read\n");
    exit(2);
  }
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int hdr[3];
  rd(f, hdr, 3);
  const int n_cam = hdr[0], n_pt = hdr[1], n_obs = hdr[2];
  std::vector<cv::Matx34d> poses(n_cam);
  for (auto& P : poses) rd(f, P.val, 12);
  std::vector<Point3D> cloud(n_pt);
  for (auto& p : cloud) {
    double x[3];
    rd(f, x, 3);
    p.pt = cv::Point3d(x[0], x[1], x[2]);
  }
  Intrinsics K;
  K.K = cv::Mat_<double>(3, 3);
  K.distCoef = cv::Mat_<double>(1, 5);
  rd(f, K.K.data.data(), 9);
  std::vector<int> vp(2 * (size_t)n_obs);
  std::vector<double> xy(2 * (size_t)n_obs);
  rd(f, vp.data(), vp.size());
  rd(f, xy.data(), xy.size());
  fclose(f);
  std::vector<std::vector<cv::Point2d>> feats(n_cam);
  for (int o = 0; o < n_obs; ++o) {  // the observation becomes the view's next feature; the point's track points at it
    const int view = vp[2 * (size_t)o], pt = vp[2 * (size_t)o + 1];
    cloud[pt].idxImage[view] = (int)feats[view].size();
    feats[view].push_back(cv::Point2d(xy[2 * (size_t)o], xy[2 * (size_t)o + 1]));
  }
  BundleAdjustment::adjustBundle(cloud, poses, K, feats);
  FILE* o = fopen(argv[2], "wb");
  if (!o) return 2;
  fwrite(K.K.data.data(), sizeof(double), 9, o);
  for (const auto& P : poses) fwrite(P.val, sizeof(double), 12, o);
  for (const auto& p : cloud) {
    const double x[3] = {p.pt.x, p.pt.y, p.pt.z};
    fwrite(x, sizeof(double), 3, o);
  }
  fclose(o);
  return 0;
}
