// BundleAdjustment.cpp -- adjustBundle over the sfmhip C ABI.  Mirrors the policies of the
// reference's src/BundleAdjustment.cpp:46-175; the solve itself runs on the MI355X.
#include "BundleAdjustment.h"
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include "ba_profile.h"
#include "hip_backend.h"
#include "sfmhip.h"

namespace {

thread_local SfmBaCallProfile g_profile;
int g_test_max_iterations = -1;
double g_test_max_time_s = -1.0;

// [0, n) cut over the library's host threads (the containers are read-only in the pack, every point is written by one block in
// the write-back): walking 10^5 std::map tracks is a chain of cache misses on one core -- 8 ms at cfg4, most of what the call
// costs beside the solve.  (The pool lives in libsfmhip: threads started per pass cost 0.3-0.5 ms each time.)
template <typename F>
void parallel_blocks(int n, F fn) {
  struct Call {
    F* f;
    static void run(int lo, int hi, void* u) { (*static_cast<Call*>(u)->f)(lo, hi); }
  } call{&fn};
  if (sfmhip_host_parallel_for(n, &Call::run, &call) != SFMHIP_OK) fn(0, n);
}

// ceres::RotationMatrixToAngleAxis of the rotation part of a pose (R(i,j) = pose(i,j))
void pose_to_angle_axis(const cv::Matx34d& P, double aa[3]) {
  double q[4];
  const double trace = P(0, 0) + P(1, 1) + P(2, 2);
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (P(2, 1) - P(1, 2)) * t;
    q[2] = (P(0, 2) - P(2, 0)) * t;
    q[3] = (P(1, 0) - P(0, 1)) * t;
  } else {
    int i = 0;
    if (P(1, 1) > P(0, 0)) i = 1;
    if (P(2, 2) > P(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(P(i, i) - P(j, j) - P(k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (P(k, j) - P(j, k)) * t;
    q[j + 1] = (P(j, i) + P(i, j)) * t;
    q[k + 1] = (P(k, i) + P(i, k)) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double k = 2.0;
  if (s2 > 0.0) {
    const double st = std::sqrt(s2), ct = q[0];
    k = 2.0 * ((ct < 0.0) ? std::atan2(-st, -ct) : std::atan2(st, ct)) / st;
  }
  aa[0] = q[1] * k;
  aa[1] = q[2] * k;
  aa[2] = q[3] * k;
}

// ceres::AngleAxisToRotationMatrix into the rotation part of a pose
void angle_axis_to_pose(const double aa[3], cv::Matx34d& P) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = std::cos(theta), s = std::sin(theta);
    P(0, 0) = c + wx * wx * (1.0 - c);
    P(1, 0) = wz * s + wx * wy * (1.0 - c);
    P(2, 0) = -wy * s + wx * wz * (1.0 - c);
    P(0, 1) = wx * wy * (1.0 - c) - wz * s;
    P(1, 1) = c + wy * wy * (1.0 - c);
    P(2, 1) = wx * s + wy * wz * (1.0 - c);
    P(0, 2) = wy * s + wx * wz * (1.0 - c);
    P(1, 2) = -wx * s + wy * wz * (1.0 - c);
    P(2, 2) = c + wz * wz * (1.0 - c);
  } else {
    P(0, 0) = 1.0;    P(0, 1) = -aa[2]; P(0, 2) = aa[1];
    P(1, 0) = aa[2];  P(1, 1) = 1.0;    P(1, 2) = -aa[0];
    P(2, 0) = -aa[1]; P(2, 1) = aa[0];  P(2, 2) = 1.0;
  }
}

}  // namespace

const SfmBaCallProfile& sfm_ba_last_call_profile() { return g_profile; }
void sfm_ba_set_test_limits(int max_iterations, double max_time_s) {
  g_test_max_iterations = max_iterations;
  g_test_max_time_s = max_time_s;
}

void BundleAdjustment::adjustBundle(std::vector<Point3D>& pointCloud, std::vector<cv::Matx34d>& cameraPoses,
                                    Intrinsics& intrinsics,
                                    const std::vector<std::vector<cv::Point2d>>& image2dFeatures) {
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
  const int n_cam = (int)cameraPoses.size(), n_pt = (int)pointCloud.size();
  std::vector<double> cams6(6 * (size_t)n_cam, 0.0);
  std::vector<char> empty(n_cam, 0);
  for (int i = 0; i < n_cam; ++i) {
    const cv::Matx34d& pose = cameraPoses[i];
    if (pose(0, 0) == 0 && pose(1, 1) == 0 && pose(2, 2) == 0) {  // unregistered view
      empty[i] = 1;
      continue;
    }
    pose_to_angle_axis(pose, &cams6[6 * (size_t)i]);
    for (int r = 0; r < 3; ++r) cams6[6 * (size_t)i + 3 + r] = pose(r, 3);
  }
  double focal = intrinsics.K.at<double>(0, 0);
  const double cx = intrinsics.K.at<double>(0, 2), cy = intrinsics.K.at<double>(1, 2);
  // one residual block per (view, feature) of every track, point-major in std::map order (src/BundleAdjustment.cpp:83-110):
  // the tracks' sizes give every point its place in the flat arrays, then the points are walked side by side
  // (the flat arrays are this thread's and keep their capacity between calls: 30 MB of fresh vectors per call cost their page
  // faults and their unmapping -- 3 ms of a cfg4 call)
  static thread_local std::vector<double> pts3, obs_xy;
  static thread_local std::vector<size_t> first;
  static thread_local std::vector<int32_t> obs_cam, obs_pt;
  pts3.resize(3 * (size_t)n_pt);
  first.assign((size_t)n_pt + 1, 0);
  for (int i = 0; i < n_pt; ++i) first[i + 1] = first[i] + pointCloud[i].idxImage.size();
  const size_t n_obs = first[n_pt];
  obs_cam.resize(n_obs);
  obs_pt.resize(n_obs);
  obs_xy.resize(2 * n_obs);
  {
    // (plain pointers for the workers: a thread_local named inside their lambda would be THEIR thread's vector, an empty one)
    double* const p3 = pts3.data();
    double* const oxy = obs_xy.data();
    int32_t* const oc = obs_cam.data();
    int32_t* const op = obs_pt.data();
    const size_t* const fst = first.data();
    parallel_blocks(n_pt, [&, p3, oxy, oc, op, fst](int lo, int hi) {
      for (int i = lo; i < hi; ++i) {
        const Point3D& p = pointCloud[i];
        p3[3 * (size_t)i] = p.pt.x;
        p3[3 * (size_t)i + 1] = p.pt.y;
        p3[3 * (size_t)i + 2] = p.pt.z;
        size_t w = fst[i];
        for (const auto& kv : p.idxImage) {  // (view, 2-D feature index)
          const cv::Point2d& f = image2dFeatures[kv.first][kv.second];
          oc[w] = kv.first;
          op[w] = i;
          oxy[2 * w] = f.x - cx;  // the optimiser does not know the principal point
          oxy[2 * w + 1] = f.y - cy;
          ++w;
        }
      }
    });
  }
  sfmhip_ba_opts opts;
  sfmhip_ba_default_opts(&opts);  // DENSE_SCHUR LM, 500 iterations, 10 s
  // (test hook, never set by the product -- ba_profile.h: the reference's two limits made reachable on problems that
  // converge in milliseconds; tests/test_gpu_host_cpp.py checks that nothing is written back behind them)
  if (g_test_max_iterations >= 0) opts.max_iterations = g_test_max_iterations;
  if (g_test_max_time_s >= 0.0) opts.max_time_s = g_test_max_time_s;
  g_profile = SfmBaCallProfile();
  g_profile.n_cam = n_cam, g_profile.n_pt = n_pt, g_profile.n_obs = (int)n_obs;
  g_profile.pack_ms = ms_since(t_start);
  const auto t_solve = clk::now();
  sfmhip_ba_summary summary;
  const int rc = sfmhip_ba_solve(sfm_hip_context(), n_cam, n_pt, (int)n_obs, cams6.data(), pts3.data(), &focal,
                                 obs_cam.data(), obs_pt.data(), obs_xy.data(), &opts, &summary);
  g_profile.solve_ms = ms_since(t_solve);
  sfmhip_ba_last_solve_profile(sfm_hip_context(), &g_profile.solve);
  g_profile.total_ms = ms_since(t_start);
  if (rc != SFMHIP_OK) {
    std::cerr << "Bundle adjustment failed: " << sfmhip_error_string(rc) << std::endl;
    return;
  }
  std::cout << "Bundle adjustment: iterations " << summary.iterations << ", cost " << summary.initial_cost << " -> "
            << summary.final_cost << ", " << summary.time_s << " s" << std::endl;
  if (summary.termination != SFMHIP_BA_CONVERGENCE) {
    std::cerr << "Bundle adjustment failed." << std::endl;  // results are discarded, inputs untouched
    return;
  }
  const auto t_wb = clk::now();
  intrinsics.K.at<double>(0, 0) = focal;
  intrinsics.K.at<double>(1, 1) = focal;
  for (int i = 0; i < n_cam; ++i) {
    if (empty[i]) continue;
    cv::Matx34d& pose = cameraPoses[i];
    angle_axis_to_pose(&cams6[6 * (size_t)i], pose);
    for (int r = 0; r < 3; ++r) pose(r, 3) = cams6[6 * (size_t)i + 3 + r];
  }
  {
    const double* const p3 = pts3.data();  // (not the thread_local's name: see above)
    parallel_blocks(n_pt, [&, p3](int lo, int hi) {
      for (int i = lo; i < hi; ++i) {
        pointCloud[i].pt.x = p3[3 * (size_t)i];
        pointCloud[i].pt.y = p3[3 * (size_t)i + 1];
        pointCloud[i].pt.z = p3[3 * (size_t)i + 2];
      }
    });
  }
  g_profile.writeback_ms = ms_since(t_wb);
  g_profile.total_ms = ms_since(t_start);
}
