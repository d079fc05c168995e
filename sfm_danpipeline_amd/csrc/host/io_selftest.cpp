// io_selftest.cpp -- drives imagesLOAD / getCameraMatrix / PMVS2 of the C++ host mirror:
//   io_selftest <image dir> <calibration.xml> <out.bin>      (PMVS2 writes ./denseCloud: run it in a scratch directory)
//   io_selftest --ply2pcd <in.ply> <out.pcd>                 (prints the point count)
//   io_selftest --features <image dir> <out.bin>             (needs the GPU) imagesLOAD + extractFeature;
//        out.bin: i32 n; per image: i32 rows, cols, gray bytes, i32 nk, nk x (6 f32 keypoint, octave bit-copied), nk x 128 f32,
//        nk x (f64 x, f64 y) of imagesPts2D
//   io_selftest --cfg1 <image dir> <calibration.xml> <out.bin> [pose.bin]     (needs the GPU) BASELINE.json's cfg1 call
//        order on a directory of frames: imagesLOAD -> getCameraMatrix -> extractFeature -> matchAllPairs ->
//        findBestPair, and with pose.bin (i32 q, i32 t, f64 Pq[12], f64 Pt[12]: the pose step, getCameraPose, is out of
//        scope and comes from the caller) triangulateViews(q, t) -> adjustCurrentBundle.  out.bin:
//        i32 n; per image: i32 nk, nk x (5 f32 + i32 octave), nk x 128 f32            features
//        f64 K[9], dist[5]
//        i32 n_pairs; per pair (q < t, the loop order of src/Sfm.cpp:511-512): i32 q, t, n, n x (i32 q, i32 t, f32 d)
//        i32 n_map; n_map x (f32 ratio, i32 q, i32 t)                                 findBestPair's map, ascending
//        i32 n_scored; per pair with >= 120 matches: i32 q, t, n, E inliers, E iterations, H inliers, H iterations,
//              n bytes E mask, n bytes H mask; i32 flags (sfmhip_score_last_flags)    the same scores through the C ABI
//        with pose.bin: i32 n_cloud, n_cloud x (f64 xyz[3], i32 fq, i32 ft); then after adjustCurrentBundle:
//              f64 K[9], f64 Pq[12], f64 Pt[12], n_cloud x f64 xyz[3]
// out.bin: i32 ok_images, i32 ok_calib, i32 n; per image: i32 rows, cols, BGR bytes, gray bytes; f64 K[9], dist[5]
#include <algorithm>
#include <cstdio>
#include <string>
#include "Sfm.h"
#include "hip_backend.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  if (std::string(argv[1]) == "--features") {
    StructFromMotion sfm;
    if (!sfm.imagesLOAD(argv[2])) return 3;
    sfm.extractFeature();
    FILE* o = fopen(argv[3], "wb");
    if (!o) return 2;
    const int n = (int)sfm.grayImages().size();
    fwrite(&n, 4, 1, o);
    for (int i = 0; i < n; ++i) {
      const cv::Mat& g = sfm.grayImages()[i];
      fwrite(&g.rows, 4, 1, o);
      fwrite(&g.cols, 4, 1, o);
      fwrite(g.ptr(), 1, g.bytes.size(), o);
      const int nk = (int)sfm.keypoints()[i].size();
      fwrite(&nk, 4, 1, o);
      for (const cv::KeyPoint& k : sfm.keypoints()[i]) {
        const float f[5] = {k.pt.x, k.pt.y, k.size, k.angle, k.response};
        fwrite(f, 4, 5, o);
        fwrite(&k.octave, 4, 1, o);
      }
      fwrite(sfm.descriptors()[i].ptr(), 1, sfm.descriptors()[i].bytes.size(), o);
      for (const cv::Point2d& p : sfm.points2D()[i]) fwrite(&p.x, 8, 2, o);
    }
    fclose(o);
    return 0;
  }
  if (std::string(argv[1]) == "--cfg1") {
    if (argc < 5) return 2;
    StructFromMotion sfm;
    if (!sfm.imagesLOAD(argv[2])) return 3;
    if (!sfm.getCameraMatrix(argv[3])) return 4;
    sfm.extractFeature();
    FILE* o = fopen(argv[4], "wb");
    if (!o) return 2;
    const int n = (int)sfm.grayImages().size();
    fwrite(&n, 4, 1, o);
    for (int i = 0; i < n; ++i) {
      const int nk = (int)sfm.keypoints()[i].size();
      fwrite(&nk, 4, 1, o);
      for (const cv::KeyPoint& k : sfm.keypoints()[i]) {
        const float f[5] = {k.pt.x, k.pt.y, k.size, k.angle, k.response};
        fwrite(f, 4, 5, o);
        fwrite(&k.octave, 4, 1, o);
      }
      fwrite(sfm.descriptors()[i].ptr(), 1, sfm.descriptors()[i].bytes.size(), o);
    }
    fwrite(sfm.intrinsics().K.data.data(), 8, 9, o);
    fwrite(sfm.intrinsics().distCoef.data.data(), 8, 5, o);
    sfm.matchAllPairs();
    const int n_pairs = n * (n - 1) / 2;
    fwrite(&n_pairs, 4, 1, o);
    std::vector<std::pair<int, int>> scored;
    std::vector<Matching> scored_m;
    for (int q = 0; q < n - 1; ++q)
      for (int t = q + 1; t < n; ++t) {
        Matching m;
        sfm.getMatching(q, t, &m);
        const int nm = (int)m.size();
        fwrite(&q, 4, 1, o);
        fwrite(&t, 4, 1, o);
        fwrite(&nm, 4, 1, o);
        for (const cv::DMatch& d : m) {
          fwrite(&d.queryIdx, 4, 1, o);
          fwrite(&d.trainIdx, 4, 1, o);
          fwrite(&d.distance, 4, 1, o);
        }
        if (nm >= 120) {
          scored.push_back(std::make_pair(q, t));
          scored_m.push_back(m);
        }
      }
    const std::map<float, std::pair<int, int>> best = sfm.findBestPair();
    const int n_map = (int)best.size();
    fwrite(&n_map, 4, 1, o);
    for (const auto& kv : best) {
      fwrite(&kv.first, 4, 1, o);
      fwrite(&kv.second.first, 4, 1, o);
      fwrite(&kv.second.second, 4, 1, o);
    }
    {  // the same scores pair by pair through the C ABI, with masks and iteration counts
      const int ns = (int)scored.size();
      fwrite(&ns, 4, 1, o);
      int flags = 0;
      const cv::Mat_<double>& Km = sfm.intrinsics().K;
      for (int p = 0; p < ns; ++p) {
        Points2d l, r;
        sfm.AlignedPointsFromMatch(sfm.points2D()[scored[p].first], sfm.points2D()[scored[p].second], scored_m[p], l, r);
        const int nm = (int)l.size();
        const int32_t off[2] = {0, nm};
        int32_t e_inl = 0, e_it = 0, h_inl = 0, h_it = 0;
        std::vector<uint8_t> em(nm), hm(nm);
        double mx = -1.7976931348623157e308;
        for (const cv::Point2d& pt : l) mx = std::max(mx, std::max(pt.x, pt.y));
        const double thr = 0.004 * mx;
        if (sfmhip_score_essential(sfm_hip_context(), 1, off, &l[0].x, &r[0].x, Km(0, 0), Km(1, 1), Km(0, 2), Km(1, 2), 0.999, 1.0,
                                   &e_inl, em.data(), &e_it) != SFMHIP_OK)
          return 5;
        flags |= sfmhip_score_last_flags(sfm_hip_context());
        if (sfmhip_score_homography(sfm_hip_context(), 1, off, &l[0].x, &r[0].x, &thr, 0.995, 2000, &h_inl, hm.data(), &h_it) != SFMHIP_OK)
          return 5;
        const int hdr[7] = {scored[p].first, scored[p].second, nm, e_inl, e_it, h_inl, h_it};
        fwrite(hdr, 4, 7, o);
        fwrite(em.data(), 1, nm, o);
        fwrite(hm.data(), 1, nm, o);
      }
      fwrite(&flags, 4, 1, o);
    }
    if (argc > 5) {
      FILE* pf = fopen(argv[5], "rb");
      if (!pf) return 2;
      int q = 0, t = 0;
      cv::Matx34d Pq, Pt;
      if (fread(&q, 4, 1, pf) != 1 || fread(&t, 4, 1, pf) != 1 || fread(Pq.val, 8, 12, pf) != 12 || fread(Pt.val, 8, 12, pf) != 12) return 2;
      fclose(pf);
      Matching m;
      sfm.getMatching(q, t, &m);
      std::vector<Point3D> cloud;
      sfm.triangulateViews(sfm.points2D()[q], sfm.points2D()[t], Pq, Pt, m, sfm.intrinsics(), std::make_pair(q, t), cloud);
      const int nc = (int)cloud.size();
      fwrite(&nc, 4, 1, o);
      for (const Point3D& p : cloud) {
        fwrite(&p.pt.x, 8, 3, o);
        const int fq = p.idxImage.at(q), ft = p.idxImage.at(t);
        fwrite(&fq, 4, 1, o);
        fwrite(&ft, 4, 1, o);
      }
      // baseReconstruction's bookkeeping (src/Sfm.cpp:476-479), then the bundle adjustment it leaves commented out (:492)
      sfm.nReconstructionCloud = cloud;
      std::vector<cv::Matx34d> poses(n);
      poses[q] = Pq;
      poses[t] = Pt;
      sfm.setCameraPoses(poses);
      sfm.adjustCurrentBundle();
      fwrite(sfm.intrinsics().K.data.data(), 8, 9, o);
      fwrite(sfm.cameraPoses()[q].val, 8, 12, o);
      fwrite(sfm.cameraPoses()[t].val, 8, 12, o);
      for (const Point3D& p : sfm.nReconstructionCloud) fwrite(&p.pt.x, 8, 3, o);
    }
    fclose(o);
    return 0;
  }
  if (std::string(argv[1]) == "--ply2pcd") {
    std::printf("%zu\n", StructFromMotion::convertPLYtoPCD(argv[2], argv[3]));
    return 0;
  }
  StructFromMotion sfm;
  const int ok_img = sfm.imagesLOAD(argv[1]) ? 1 : 0;
  const int ok_cal = sfm.getCameraMatrix(argv[2]) ? 1 : 0;
  FILE* o = fopen(argv[3], "wb");
  if (!o) return 2;
  fwrite(&ok_img, 4, 1, o);
  fwrite(&ok_cal, 4, 1, o);
  const int n = ok_img ? (int)sfm.colorImages().size() : 0;
  fwrite(&n, 4, 1, o);
  for (int i = 0; i < n; ++i) {
    const cv::Mat& c = sfm.colorImages()[i];
    const cv::Mat& g = sfm.grayImages()[i];
    fwrite(&c.rows, 4, 1, o);
    fwrite(&c.cols, 4, 1, o);
    fwrite(c.ptr(), 1, c.bytes.size(), o);
    fwrite(g.ptr(), 1, g.bytes.size(), o);
  }
  if (ok_cal) {
    fwrite(sfm.intrinsics().K.data.data(), 8, 9, o);
    fwrite(sfm.intrinsics().distCoef.data.data(), 8, 5, o);
  }
  fclose(o);
  if (ok_img && ok_cal) {
    std::vector<cv::Matx34d> poses(n);
    for (int i = 0; i < n; ++i) {
      poses[i](0, 0) = poses[i](1, 1) = poses[i](2, 2) = 1.0;
      poses[i](0, 3) = 0.25 * i;
      poses[i](1, 3) = -0.5;
      poses[i](2, 3) = 1.0 / 3.0 * i;
    }
    sfm.setCameraPoses(poses);
    sfm.PMVS2();
  }
  return 0;
}
