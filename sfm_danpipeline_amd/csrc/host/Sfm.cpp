// Sfm.cpp -- hot-path members of StructFromMotion over the sfmhip C ABI.
#include "Sfm.h"
#include <cstdlib>
#include <iostream>
#include "hip_backend.h"

sfmhip_ctx* sfm_hip_context() {
  static sfmhip_ctx* ctx = nullptr;
  if (!ctx) {
    const char* dev = std::getenv("SFM_HIP_DEVICE");
    const int rc = sfmhip_init(dev ? std::atoi(dev) : 0, &ctx);
    if (rc != SFMHIP_OK) {
      std::cerr << "sfmhip_init: " << sfmhip_error_string(rc) << std::endl;
      std::exit(-1);
    }
  }
  return ctx;
}

void StructFromMotion::getMatching(const int& idx_query, const int& idx_train, Matching* goodMatches) {
  const cv::Mat& q = imagesDescriptors.at(idx_query);
  const cv::Mat& t = imagesDescriptors.at(idx_train);
  if (q.rows == 0) return;
  std::vector<int32_t> oq(q.rows), ot(q.rows);
  std::vector<float> od(q.rows);
  int32_t n = 0;
  // cv::NORM_L2 whatever the descriptor type, like the reference (src/Sfm.cpp:593)
  const int rc = sfmhip_match_knn2(sfm_hip_context(), q.ptr(), q.rows, t.ptr(), t.rows, q.cols,
                                   q.type() == CV_32F ? SFMHIP_F32 : SFMHIP_U8, SFMHIP_L2, NN_MATCH_RATIO, oq.data(),
                                   ot.data(), od.data(), &n);
  if (rc != SFMHIP_OK) {
    std::cerr << "getMatching: " << sfmhip_error_string(rc) << std::endl;
    return;
  }
  for (int i = 0; i < n; ++i) goodMatches->push_back(cv::DMatch(oq[i], ot[i], od[i]));  // appends
}

void StructFromMotion::AlignedPointsFromMatch(const Points2d& queryImg, const Points2d& trainImg, const Matching& matches,
                                              Points2d& alignedL, Points2d& alignedR) {
  std::vector<int> leftId, rightId;
  AlignedPoints(queryImg, trainImg, matches, alignedL, alignedR, leftId, rightId);
}

void StructFromMotion::AlignedPoints(const Points2d& queryImg, const Points2d& trainImg, const Matching& matches,
                                     Points2d& alignedL, Points2d& alignedR, std::vector<int>& idLeftOrigen,
                                     std::vector<int>& idRightOrigen) {
  for (const cv::DMatch& m : matches) {
    alignedL.push_back(queryImg[m.queryIdx]);
    alignedR.push_back(trainImg[m.trainIdx]);
    idLeftOrigen.push_back(m.queryIdx);
    idRightOrigen.push_back(m.trainIdx);
  }
}

bool StructFromMotion::triangulateViews(const Points2d& query, const Points2d& train, const cv::Matx34d& P1,
                                        const cv::Matx34d& P2, const Matching& matches, const Intrinsics& matrixK,
                                        const std::pair<int, int>& pair, std::vector<Point3D>& pointcloud) {
  pointcloud.clear();
  Points2d alignedQuery, alignedTrain;
  std::vector<int> leftBackReference, rightBackReference;
  AlignedPoints(query, train, matches, alignedQuery, alignedTrain, leftBackReference, rightBackReference);
  const int m = (int)alignedQuery.size();
  if (m == 0) return true;
  double dist[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 5 && i < (int)matrixK.distCoef.data.size(); ++i) dist[i] = matrixK.distCoef.data[i];
  std::vector<double> X(3 * (size_t)m);
  std::vector<uint8_t> keep(m);
  const float MIN_REPROJECTION_ERROR = 6.0;
  static_assert(sizeof(cv::Point2d) == 2 * sizeof(double), "Point2d must be two packed doubles");
  const int rc = sfmhip_triangulate(sfm_hip_context(), P1.val, P2.val, matrixK.K.data.data(), dist,
                                    &alignedQuery[0].x, &alignedTrain[0].x, m, MIN_REPROJECTION_ERROR, X.data(),
                                    nullptr, keep.data());
  if (rc != SFMHIP_OK) {
    std::cerr << "triangulateViews: " << sfmhip_error_string(rc) << std::endl;
    return true;  // the reference has no failure path here (src/Sfm.cpp:877)
  }
  for (int i = 0; i < m; ++i) {
    if (!keep[i]) continue;
    Point3D p;
    p.pt = cv::Point3d(X[3 * (size_t)i], X[3 * (size_t)i + 1], X[3 * (size_t)i + 2]);
    p.idxImage[pair.first] = leftBackReference[i];
    p.idxImage[pair.second] = rightBackReference[i];
    p.pt2D[pair.first] = imagesPts2D.at(pair.first).at(leftBackReference[i]);
    p.pt2D[pair.second] = imagesPts2D.at(pair.second).at(rightBackReference[i]);
    pointcloud.push_back(p);
  }
  return true;
}

void StructFromMotion::adjustCurrentBundle() {
  BundleAdjustment::adjustBundle(nReconstructionCloud, nCameraPoses, cameraMatrix, imagesPts2D);
}
