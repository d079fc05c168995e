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
#include "ba_profile.h"
#include "hip_backend.h"

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
  // test switches of THIS executable (the mirror itself reads no environment): the reference's two limits made reachable,
  // and SFM_BA_SELFTEST_CALLS = n: the call repeated on fresh copies of the containers (what a caller that adjusts the same
  // structure again sees: the second call re-uses the first one's plan), a profile line per call on stdout
  {
    int mi = -1;
    double mt = -1.0;
    if (const char* e = getenv("SFM_BA_TEST_MAX_ITERATIONS")) {
      char* end = nullptr;
      const long v = strtol(e, &end, 10);
      if (end == e || *end || v < 0) {
        fprintf(stderr, "ba_selftest: SFM_BA_TEST_MAX_ITERATIONS=%s is not a count\n", e);
        return 2;
      }
      mi = (int)v;
    }
    if (const char* e = getenv("SFM_BA_TEST_MAX_TIME_S")) {
      char* end = nullptr;
      const double v = strtod(e, &end);
      if (end == e || *end || !(v >= 0)) {
        fprintf(stderr, "ba_selftest: SFM_BA_TEST_MAX_TIME_S=%s is not a duration\n", e);
        return 2;
      }
      mt = v;
    }
    sfm_ba_set_test_limits(mi, mt);
  }
  const int calls = getenv("SFM_BA_SELFTEST_CALLS") ? atoi(getenv("SFM_BA_SELFTEST_CALLS")) : 1;
  if (calls > 1) sfm_hip_context();  // (measurement: HIP's start-up -- 160 ms -- is the process's, not the first call's)
  const std::vector<Point3D> cloud0 = cloud;
  const std::vector<cv::Matx34d> poses0 = poses;
  const double f0 = K.K.at<double>(0, 0);
  // SFM_BA_SELFTEST_NEW_STRUCTURE = k: k more calls behind those, each on a structure the library has not seen (the track of
  // one more point loses its first view): what a caller that adds or drops a view between calls pays in a warm process
  const int fresh = getenv("SFM_BA_SELFTEST_NEW_STRUCTURE") ? atoi(getenv("SFM_BA_SELFTEST_NEW_STRUCTURE")) : 0;
  const int n_calls = (calls > 1 ? calls : 1) + (fresh > 0 ? fresh : 0);
  for (int c = 0; c < n_calls; ++c) {
    if (c) {
      cloud = cloud0;
      poses = poses0;
      K.K.at<double>(0, 0) = f0;
      K.K.at<double>(1, 1) = f0;
    }
    for (int k = 0; k <= c - (calls > 1 ? calls : 1) && k < n_pt; ++k) {
      auto& track = cloud[n_pt - 1 - k].idxImage;
      if (track.size() > 2) track.erase(track.begin());
    }
    BundleAdjustment::adjustBundle(cloud, poses, K, feats);
    const SfmBaCallProfile& p = sfm_ba_last_call_profile();
    printf("{\"call\": %d, \"n_cam\": %d, \"n_pt\": %d, \"n_obs\": %d, \"pack_ms\": %.3f, \"create_ms\": %.3f, \"set_params_ms\": %.3f, "
           "\"run_ms\": %.3f, \"get_params_ms\": %.3f, \"keep_ms\": %.3f, \"solve_ms\": %.3f, \"writeback_ms\": %.3f, \"total_ms\": %.3f, "
           "\"plan_reused\": %d, \"front_plan_reused\": %d}\n",
           c, p.n_cam, p.n_pt, p.n_obs, p.pack_ms, p.solve.create_ms, p.solve.set_params_ms, p.solve.run_ms, p.solve.get_params_ms,
           p.solve.keep_ms, p.solve_ms, p.writeback_ms, p.total_ms, p.solve.plan_reused, p.solve.front_plan_reused);
  }
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
