// selftest.cpp -- drives the C++ host classes on inputs written by tests/test_gpu_host_cpp.py and
// dumps their outputs for comparison with the oracle.  Formats are flat little-endian binaries:
//   in:  i32 n_img | per image: i32 rows, i32 cols, i32 depth(0=u8,5=f32), payload
//        i32 n_views | per view: i32 n, n x (f64 x, f64 y)
//        f64 K[9] | f64 dist[5] | f64 P1[12] | f64 P2[12]
//        BA block: i32 n_cam, n_cam x f64 pose[12] | i32 n_pt, per point: f64 xyz[3], i32 n, n x (i32 view, i32 feat)
//   out: matches of pair (0,1): i32 n, n x (i32 q, i32 t, f32 d)
//        cloud of triangulateViews(0,1): i32 n, n x (f64 xyz[3], i32 q, i32 t)
//        matches of pair (0,1) again, served by the pair cache after matchAllPairs(): same layout
//        find2D3DMatches(NEW_VIEW=1) with done views {0} on that cloud: i32 DONEVIEW, i32 n,
//              n x (f64 xyz[3], f64 xy[2])
//        mergeNewPoints(cloud shifted by (0,0,0.004) ++ cloud shifted by (5,0,0) ++ the same again):
//              i32 cloud size before, i32 after, then (after-before) x f64 xyz[3] of the appended points
//        findBestPair() (SURVEY 8f-1 scoring half): i32 n, n x (f32 poseInliersRatio, i32 q, i32 t) in map order,
//              then i32 findHomographyInliers(0, 1, matches)
//        after adjustCurrentBundle: f64 K[9], n_cam x f64 pose[12], n_pt x f64 xyz[3]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "Sfm.h"

template <typename T>
static T rd(FILE* f) {
  T v;
  if (fread(&v, sizeof(T), 1, f) != 1) {
    fprintf(stderr, "selftest: short read\n");
    exit(2);
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  FILE* o = fopen(argv[2], "wb");
  if (!f || !o) return 2;
  StructFromMotion sfm;
  const int n_img = rd<int>(f);
  std::vector<cv::Mat> desc;
  for (int i = 0; i < n_img; ++i) {
    const int rows = rd<int>(f), cols = rd<int>(f), depth = rd<int>(f);
    cv::Mat m(rows, cols, depth);
    if (!m.bytes.empty() && fread(m.bytes.data(), 1, m.bytes.size(), f) != m.bytes.size()) return 2;
    desc.push_back(m);
  }
  sfm.setDescriptors(desc);
  const int n_views = rd<int>(f);
  std::vector<std::vector<cv::Point2d>> pts2d(n_views);
  for (int v = 0; v < n_views; ++v) {
    const int n = rd<int>(f);
    for (int i = 0; i < n; ++i) {
      const double x = rd<double>(f), y = rd<double>(f);
      pts2d[v].push_back(cv::Point2d(x, y));
    }
  }
  sfm.setPoints2D(pts2d);
  Intrinsics K;
  K.K = cv::Mat_<double>(3, 3);
  K.distCoef = cv::Mat_<double>(1, 5);
  for (int i = 0; i < 9; ++i) K.K.data[i] = rd<double>(f);
  for (int i = 0; i < 5; ++i) K.distCoef.data[i] = rd<double>(f);
  sfm.setCameraMatrix(K);
  cv::Matx34d P1, P2;
  for (int i = 0; i < 12; ++i) P1.val[i] = rd<double>(f);
  for (int i = 0; i < 12; ++i) P2.val[i] = rd<double>(f);

  // ---- getMatching(0, 1)
  Matching matches;
  sfm.getMatching(0, 1, &matches);
  int n = (int)matches.size();
  fwrite(&n, 4, 1, o);
  for (const cv::DMatch& m : matches) {
    fwrite(&m.queryIdx, 4, 1, o);
    fwrite(&m.trainIdx, 4, 1, o);
    fwrite(&m.distance, 4, 1, o);
  }
  // ---- triangulateViews on those matches
  std::vector<Point3D> cloud;
  sfm.triangulateViews(pts2d[0], pts2d[1], P1, P2, matches, K, std::make_pair(0, 1), cloud);
  n = (int)cloud.size();
  fwrite(&n, 4, 1, o);
  for (const Point3D& p : cloud) {
    fwrite(&p.pt.x, 8, 3, o);
    const int q = p.idxImage.at(0), t = p.idxImage.at(1);
    fwrite(&q, 4, 1, o);
    fwrite(&t, 4, 1, o);
  }
  // ---- pair cache: one batched launch, then getMatching is served from it
  sfm.matchAllPairs();
  Matching cached;
  sfm.getMatching(0, 1, &cached);
  n = (int)cached.size();
  fwrite(&n, 4, 1, o);
  for (const cv::DMatch& m : cached) {
    fwrite(&m.queryIdx, 4, 1, o);
    fwrite(&m.trainIdx, 4, 1, o);
    fwrite(&m.distance, 4, 1, o);
  }
  // ---- find2D3DMatches / mergeNewPoints on the triangulated cloud
  sfm.nReconstructionCloud = cloud;
  std::set<int> done;
  done.insert(0);
  sfm.setDoneViews(done);
  std::vector<cv::Point3d> p3;
  std::vector<cv::Point2d> p2;
  Matching best;
  int DONEVIEW = -1;
  sfm.find2D3DMatches(1, p3, p2, best, DONEVIEW);
  n = (int)p3.size();
  fwrite(&DONEVIEW, 4, 1, o);
  fwrite(&n, 4, 1, o);
  for (int i = 0; i < n; ++i) {
    fwrite(&p3[i].x, 8, 3, o);
    fwrite(&p2[i].x, 8, 2, o);
  }
  {
    std::vector<Point3D> fresh;
    for (int rep = 0; rep < 3; ++rep)
      for (const Point3D& p : cloud) {
        Point3D q = p;
        if (rep == 0) q.pt.z += 0.004;
        else q.pt.x += 5.0;
        fresh.push_back(q);
      }
    const int before = (int)sfm.nReconstructionCloud.size();
    sfm.mergeNewPoints(fresh);
    const int after = (int)sfm.nReconstructionCloud.size();
    fwrite(&before, 4, 1, o);
    fwrite(&after, 4, 1, o);
    for (int i = before; i < after; ++i) fwrite(&sfm.nReconstructionCloud[i].pt.x, 8, 3, o);
  }
  // ---- findBestPair: the pair cache, then the E-matrix RANSAC score of every pair with >= 120 matches
  {
    const std::map<float, std::pair<int, int>> best_pairs = sfm.findBestPair();
    n = (int)best_pairs.size();
    fwrite(&n, 4, 1, o);
    for (const auto& kv : best_pairs) {
      fwrite(&kv.first, 4, 1, o);
      fwrite(&kv.second.first, 4, 1, o);
      fwrite(&kv.second.second, 4, 1, o);
    }
    const int hom = sfm.findHomographyInliers(0, 1, cached);
    fwrite(&hom, 4, 1, o);
  }
  // ---- adjustCurrentBundle on the BA block
  const int n_cam = rd<int>(f);
  std::vector<cv::Matx34d> poses(n_cam);
  for (int c = 0; c < n_cam; ++c)
    for (int i = 0; i < 12; ++i) poses[c].val[i] = rd<double>(f);
  const int n_pt = rd<int>(f);
  sfm.nReconstructionCloud.clear();
  for (int i = 0; i < n_pt; ++i) {
    Point3D p;
    p.pt.x = rd<double>(f);
    p.pt.y = rd<double>(f);
    p.pt.z = rd<double>(f);
    const int k = rd<int>(f);
    for (int j = 0; j < k; ++j) {
      const int view = rd<int>(f), feat = rd<int>(f);
      p.idxImage[view] = feat;
    }
    sfm.nReconstructionCloud.push_back(p);
  }
  sfm.setCameraPoses(poses);
  sfm.adjustCurrentBundle();
  fwrite(sfm.intrinsics().K.data.data(), 8, 9, o);
  for (const cv::Matx34d& P : sfm.cameraPoses()) fwrite(P.val, 8, 12, o);
  for (const Point3D& p : sfm.nReconstructionCloud) fwrite(&p.pt.x, 8, 3, o);
  fclose(f);
  fclose(o);
  return 0;
}
