// Sfm.cpp -- hot-path members of StructFromMotion over the sfmhip C ABI.
#include "Sfm.h"
#include <algorithm>
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
  if (pairCacheOn) {
    auto it = pairCache.find(std::make_pair(idx_query, idx_train));
    if (it != pairCache.end()) {
      goodMatches->insert(goodMatches->end(), it->second.begin(), it->second.end());  // appends, like :605
      return;
    }
  }
  const cv::Mat& q = imagesDescriptors.at(idx_query);
  const cv::Mat& t = imagesDescriptors.at(idx_train);
  if (q.rows == 0) return;
  std::vector<int32_t> oq(q.rows), ot(q.rows);
  std::vector<float> od(q.rows);
  int32_t n = 0;
  int rc = SFMHIP_OK;
  // every image uploaded and prepared once; cv::NORM_L2 whatever the descriptor type, like the
  // reference (src/Sfm.cpp:593)
  bool uniform = true;
  for (const cv::Mat& d : imagesDescriptors) uniform = uniform && d.cols == q.cols && d.type() == q.type();
  if (uniform) {
    sfmhip_ctx* ctx = sfm_hip_context();
    if (!devSet) {
      std::vector<int32_t> rows;
      for (const cv::Mat& d : imagesDescriptors) rows.push_back(d.rows);
      rc = sfmhip_imageset_create(ctx, (int)rows.size(), rows.data(), q.cols, q.type() == CV_32F ? SFMHIP_F32 : SFMHIP_U8,
                                  SFMHIP_L2, &devSet);
      for (size_t i = 0; rc == SFMHIP_OK && i < rows.size(); ++i) rc = uploadOrAdopt(devSet, (int)i);
      if (rc == SFMHIP_OK) rc = sfmhip_imageset_prepare_async(devSet);
      const int32_t first[2] = {idx_query, idx_train};
      if (rc == SFMHIP_OK) rc = sfmhip_matchplan_create(devSet, first, 1, &devPlan);
    }
    const int32_t pr[2] = {idx_query, idx_train};
    if (rc == SFMHIP_OK) rc = sfmhip_matchplan_set_pairs(devPlan, pr, 1);
    if (rc == SFMHIP_OK) rc = sfmhip_matchplan_run_async(devPlan, NN_MATCH_RATIO);
    int64_t tot = 0;
    if (rc == SFMHIP_OK) rc = sfmhip_matchplan_fetch(devPlan, &n, oq.data(), ot.data(), od.data(), q.rows, &tot);
    if (rc != SFMHIP_OK) releaseDeviceSet();
  } else {
    descriptors();  // (host rows wanted: download what lives in HBM only)
    rc = sfmhip_match_knn2(sfm_hip_context(), q.ptr(), q.rows, t.ptr(), t.rows, q.cols,
                           q.type() == CV_32F ? SFMHIP_F32 : SFMHIP_U8, SFMHIP_L2, NN_MATCH_RATIO, oq.data(), ot.data(),
                           od.data(), &n);
  }
  if (rc != SFMHIP_OK) {
    std::cerr << "getMatching: " << sfmhip_error_string(rc) << std::endl;
    return;
  }
  for (int i = 0; i < n; ++i) goodMatches->push_back(cv::DMatch(oq[i], ot[i], od[i]));  // appends
}

void StructFromMotion::releaseDeviceDescriptors() {
  for (void* p : devDescriptors) sfmhip_device_free(p);
  devDescriptors.clear();
}

// descriptor rows of one image into an image set: in place when extractFeature left them in HBM, else one upload
int StructFromMotion::uploadOrAdopt(sfmhip_imageset* set, int image) {
  if (imagesDescriptors[image].rows == 0) return SFMHIP_OK;
  if ((size_t)image < devDescriptors.size() && devDescriptors[image]) return sfmhip_imageset_adopt_device(set, image, devDescriptors[image]);
  return sfmhip_imageset_upload(set, image, imagesDescriptors[image].ptr());
}

const std::vector<cv::Mat>& StructFromMotion::descriptors() {
  for (size_t i = 0; i < imagesDescriptors.size(); ++i) {
    cv::Mat& m = imagesDescriptors[i];
    if (m.rows > 0 && m.bytes.empty() && i < devDescriptors.size() && devDescriptors[i]) {
      m.bytes.resize((size_t)m.rows * m.cols * m.elemSize());
      const int rc = sfmhip_device_download(sfm_hip_context(), m.bytes.data(), devDescriptors[i], m.bytes.size());
      if (rc != SFMHIP_OK) std::cerr << "descriptors: " << sfmhip_error_string(rc) << std::endl;
    }
  }
  return imagesDescriptors;
}

void StructFromMotion::releaseDeviceSet() {
  if (devPlan) sfmhip_matchplan_destroy(devPlan);
  if (devSet) sfmhip_imageset_destroy(devSet);
  devPlan = nullptr;
  devSet = nullptr;
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

// cv::minMaxIdx over a vector<Point2d>: the largest of all x and y (the array is taken as single-channel)
static double max_coordinate(const Points2d& pts) {
  double m = -1.7976931348623157e308;
  for (const cv::Point2d& p : pts) m = std::max(m, std::max(p.x, p.y));
  return m;
}

// reference src/Sfm.cpp:667-689
int StructFromMotion::findHomographyInliers(const int& idx_query, const int& idx_train, const Matching& matches) {
  Points2d query_points, train_points;
  AlignedPointsFromMatch(imagesPts2D.at(idx_query), imagesPts2D.at(idx_train), matches, query_points, train_points);
  if (matches.size() < 4) return 0;
  const int32_t offsets[2] = {0, (int32_t)query_points.size()};
  std::vector<double> l, r;
  for (size_t i = 0; i < query_points.size(); ++i) {
    l.push_back(query_points[i].x);
    l.push_back(query_points[i].y);
    r.push_back(train_points[i].x);
    r.push_back(train_points[i].y);
  }
  const double thr = 0.004 * max_coordinate(query_points);
  int32_t inl = 0;
  const int rc = sfmhip_score_homography(sfm_hip_context(), 1, offsets, l.data(), r.data(), &thr, 0.995, 2000, &inl, nullptr, nullptr);
  if (rc != SFMHIP_OK) {
    std::cerr << "findHomographyInliers: " << sfmhip_error_string(rc) << std::endl;
    return 0;
  }
  return inl;
}

// reference src/Sfm.cpp:499-585
std::map<float, std::pair<int, int>> StructFromMotion::findBestPair() {
  std::cout << "Getting best two views for baseline..." << std::endl;
  std::map<float, std::pair<int, int>> numInliers;
  const int numImg = (int)imagesDescriptors.size();
  if (!pairCacheOn) matchAllPairs();
  std::vector<std::pair<int, int>> ids;
  std::vector<size_t> nmatch;
  std::vector<int32_t> offsets(1, 0);
  std::vector<double> left, right, hthr;
  for (int queryImage = 0; queryImage < numImg - 1; queryImage++)
    for (int trainImage = queryImage + 1; trainImage < numImg; trainImage++) {
      Matching correspondences;
      getMatching(queryImage, trainImage, &correspondences);
      if (correspondences.size() < 120) continue;  // :533
      Points2d alignedLeft, alignedRight;
      AlignedPointsFromMatch(imagesPts2D.at(queryImage), imagesPts2D.at(trainImage), correspondences, alignedLeft, alignedRight);
      for (size_t i = 0; i < alignedLeft.size(); ++i) {
        left.push_back(alignedLeft[i].x);
        left.push_back(alignedLeft[i].y);
        right.push_back(alignedRight[i].x);
        right.push_back(alignedRight[i].y);
      }
      offsets.push_back((int32_t)(left.size() / 2));
      hthr.push_back(0.004 * max_coordinate(alignedLeft));  // :674,681
      ids.push_back(std::make_pair(queryImage, trainImage));
      nmatch.push_back(correspondences.size());
    }
  std::vector<int32_t> inliers(ids.size() + 1, 0);
  const cv::Mat_<double>& Km = cameraMatrix.K;
  const int rc = sfmhip_score_essential(sfm_hip_context(), (int)ids.size(), offsets.data(), left.data(), right.data(), Km(0, 0),
                                        Km(1, 1), Km(0, 2), Km(1, 2), 0.999, 1.0, inliers.data(), nullptr, nullptr);  // :542-543
  std::vector<int32_t> hom(ids.size() + 1, 0);
  int rc2 = rc;
  if (rc == SFMHIP_OK)
    rc2 = sfmhip_score_homography(sfm_hip_context(), (int)ids.size(), offsets.data(), left.data(), right.data(), hthr.data(), 0.995,
                                  2000, hom.data(), nullptr, nullptr);  // :545
  if (rc != SFMHIP_OK || rc2 != SFMHIP_OK) {
    std::cerr << "findBestPair: " << sfmhip_error_string(rc != SFMHIP_OK ? rc : rc2) << std::endl;
    return numInliers;
  }
  for (size_t p = 0; p < ids.size(); ++p) {
    const float poseInliersRatio = (float)inliers[p] / (float)nmatch[p];  // :563
    std::cout << "pair:" << "[" << ids[p].first << "," << ids[p].second << "]" << " has:" << nmatch[p] << " matches " << hom[p]
              << " inliers and " << poseInliersRatio << " pose inliers ratio." << std::endl;  // :567
    numInliers[poseInliersRatio] = ids[p];  // :569
  }
  return numInliers;
}

void StructFromMotion::matchAllPairs() {
  const int n = (int)imagesDescriptors.size();
  pairCache.clear();
  pairCacheOn = false;
  if (n < 2) return;
  const cv::Mat& d0 = imagesDescriptors[0];
  for (const cv::Mat& d : imagesDescriptors)
    if (d.cols != d0.cols || d.type() != d0.type()) return;  // mixed descriptor kinds: per-pair calls only
  std::vector<int32_t> rows(n), pairs;
  for (int i = 0; i < n; ++i) rows[i] = imagesDescriptors[i].rows;
  for (int q = 0; q < n - 1; ++q)
    for (int t = q + 1; t < n; ++t) {  // the loop order of src/Sfm.cpp:511-512
      pairs.push_back(q);
      pairs.push_back(t);
    }
  sfmhip_ctx* ctx = sfm_hip_context();
  sfmhip_imageset* set = nullptr;
  int rc = sfmhip_imageset_create(ctx, n, rows.data(), d0.cols, d0.type() == CV_32F ? SFMHIP_F32 : SFMHIP_U8, SFMHIP_L2, &set);
  for (int i = 0; rc == SFMHIP_OK && i < n; ++i) rc = uploadOrAdopt(set, i);
  if (rc == SFMHIP_OK) rc = sfmhip_imageset_prepare_async(set);
  // The pair list goes to the device in batches, two plans taking turns: while the device sweeps batch b, the lists of
  // batch b-1 -- packed into pinned host memory by the plans' second stream (sfmhip_matchplan_pipeline) -- become the
  // cache's vector<DMatch>s on the host.  Results do not depend on the batching (pairs are independent).
  const int n_pairs = (int)pairs.size() / 2;
  const int batch = std::max(64, (n_pairs + 7) / 8);
  const int n_batches = (n_pairs + batch - 1) / batch;
  sfmhip_matchplan* plans[2] = {nullptr, nullptr};
  // Pinned memory is bounded: each of the four buffers (two plans, two slots) holds at most kPipeBytes of {q, t, dist}
  // records, sized for a quarter of the batch's query rows surviving the ratio test.  A batch with more matches than that,
  // or a host that will not pin the buffers at all, takes the copying fetch for that batch -- slower, never wrong.
  const int64_t kPipeBytes = (int64_t)256 << 20;
  int maxRows = 0;
  for (int r : rows) maxRows = std::max(maxRows, r);
  const int64_t pipeCapacity = std::max<int64_t>(4096, std::min<int64_t>((int64_t)batch * maxRows / 4, kPipeBytes / 12));
  bool piped = true;
  auto collectCopying = [&](int b) -> int {
    const int first = b * batch, np = std::min(batch, n_pairs - first);
    std::vector<int32_t> c(np);
    int64_t total = 0;
    int r = sfmhip_matchplan_fetch(plans[b & 1], c.data(), nullptr, nullptr, nullptr, 0, &total);  // (counts first: the sizes)
    if (r != SFMHIP_OK) return r;
    std::vector<int32_t> q((size_t)total + 1), t((size_t)total + 1);
    std::vector<float> d((size_t)total + 1);
    r = sfmhip_matchplan_fetch(plans[b & 1], c.data(), q.data(), t.data(), d.data(), total, &total);
    size_t off = 0;
    for (int p = 0; r == SFMHIP_OK && p < np; ++p) {
      Matching& m = pairCache[std::make_pair((int)pairs[2 * (first + p)], (int)pairs[2 * (first + p) + 1])];
      m.reserve((size_t)c[p]);
      for (int i = 0; i < c[p]; ++i, ++off) m.push_back(cv::DMatch(q[off], t[off], d[off]));
    }
    return r;
  };
  auto collect = [&](int b) -> int {  // batch b's lists -> pairCache
    if (!piped) return collectCopying(b);
    const int32_t *cnt = nullptr, *oq = nullptr, *ot = nullptr;
    const float* od = nullptr;
    int64_t total = 0;
    int r = sfmhip_matchplan_fetch_wait(plans[b & 1], 0, &cnt, &oq, &ot, &od, &total);
    if (r == SFMHIP_ERR_ALLOC) return collectCopying(b);  // more matches than a pinned buffer holds
    if (r != SFMHIP_OK) return r;
    const int first = b * batch, np = std::min(batch, n_pairs - first);
    size_t off = 0;
    for (int p = 0; p < np; ++p) {
      Matching& m = pairCache[std::make_pair((int)pairs[2 * (first + p)], (int)pairs[2 * (first + p) + 1])];
      m.reserve((size_t)cnt[p]);
      for (int i = 0; i < cnt[p]; ++i, ++off) m.push_back(cv::DMatch(oq[off], ot[off], od[off]));
    }
    return SFMHIP_OK;
  };
  for (int b = 0; rc == SFMHIP_OK && b < n_batches; ++b) {
    const int first = b * batch, np = std::min(batch, n_pairs - first);
    sfmhip_matchplan*& pl = plans[b & 1];
    if (!pl) {
      rc = sfmhip_matchplan_create(set, pairs.data() + 2 * first, np, &pl);
      if (rc == SFMHIP_OK && piped && sfmhip_matchplan_pipeline(pl, pipeCapacity) != SFMHIP_OK) {
        // no pinned memory to be had: the pass goes on without the overlap
        std::cerr << "matchAllPairs: pinned buffers refused (" << pipeCapacity * 12 << " bytes each); copying fetch" << std::endl;
        piped = false;
        for (sfmhip_matchplan* q : plans)
          if (q) sfmhip_matchplan_pipeline(q, -1);
      }
    } else {
      rc = sfmhip_matchplan_set_pairs(pl, pairs.data() + 2 * first, np);  // (np <= the batch it was created with)
    }
    if (rc == SFMHIP_OK) rc = sfmhip_matchplan_run_async(pl, NN_MATCH_RATIO);
    if (rc == SFMHIP_OK && b > 0) rc = collect(b - 1);
  }
  if (rc == SFMHIP_OK && n_batches > 0) rc = collect(n_batches - 1);
  if (rc == SFMHIP_OK) {
    pairCacheOn = true;
  } else {
    pairCache.clear();
    std::cerr << "matchAllPairs: " << sfmhip_error_string(rc) << std::endl;
  }
  sfmhip_matchplan_destroy(plans[0]);
  sfmhip_matchplan_destroy(plans[1]);
  sfmhip_imageset_destroy(set);
}

void StructFromMotion::find2D3DMatches(const int& NEW_VIEW, std::vector<cv::Point3d>& points3D,
                                       std::vector<cv::Point2d>& points2D, Matching& bestMatches, int& DONEVIEW) {
  points3D.clear();
  points2D.clear();
  int bestNumMatches = 0;
  Matching bestMatch;
  for (int doneView : nDoneViews) {  // src/Sfm.cpp:1020-1042
    const int queryImage = NEW_VIEW < doneView ? NEW_VIEW : doneView;
    const int trainImage = NEW_VIEW < doneView ? doneView : NEW_VIEW;
    Matching match;
    getMatching(queryImage, trainImage, &match);
    const int numMatches = (int)match.size();
    if (numMatches > bestNumMatches) {
      bestMatch = match;
      bestNumMatches = numMatches;
      DONEVIEW = doneView;
    }
  }
  bestMatches = bestMatch;
  if (bestMatch.empty() || nReconstructionCloud.empty()) return;
  // cloud tracks as CSR (std::map order), matches as two index arrays
  std::vector<int32_t> ptr(nReconstructionCloud.size() + 1, 0), views, feats, mq, mt;
  for (size_t i = 0; i < nReconstructionCloud.size(); ++i) {
    for (const auto& kv : nReconstructionCloud[i].idxImage) {
      views.push_back(kv.first);
      feats.push_back(kv.second);
    }
    ptr[i + 1] = (int32_t)views.size();
  }
  for (const cv::DMatch& m : bestMatch) {
    mq.push_back(m.queryIdx);
    mt.push_back(m.trainIdx);
  }
  std::vector<int32_t> oc(nReconstructionCloud.size()), of(nReconstructionCloud.size());
  int32_t n = 0;
  const int rc = sfmhip_find_2d3d(sfm_hip_context(), ptr.data(), views.data(), feats.data(), (int)nReconstructionCloud.size(),
                                  DONEVIEW, NEW_VIEW, mq.data(), mt.data(), (int)mq.size(), oc.data(), of.data(), &n);
  if (rc != SFMHIP_OK) {
    std::cerr << "find2D3DMatches: " << sfmhip_error_string(rc) << std::endl;
    return;
  }
  const std::vector<cv::Point2d>& newViewFeatures = imagesPts2D.at(NEW_VIEW);
  for (int i = 0; i < n; ++i) {
    points2D.push_back(newViewFeatures.at(of[i]));  // :1078-1079
    points3D.push_back(nReconstructionCloud[oc[i]].pt);
  }
}

void StructFromMotion::mergeNewPoints(const std::vector<Point3D>& newPointCloud) {
  if (newPointCloud.empty()) return;
  const float MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE = 0.01;  // src/Sfm.cpp:1216
  static_assert(sizeof(cv::Point3d) == 3 * sizeof(double), "Point3d must be three packed doubles");
  std::vector<double> cloud(3 * nReconstructionCloud.size()), fresh(3 * newPointCloud.size());
  for (size_t i = 0; i < nReconstructionCloud.size(); ++i) {
    cloud[3 * i] = nReconstructionCloud[i].pt.x;
    cloud[3 * i + 1] = nReconstructionCloud[i].pt.y;
    cloud[3 * i + 2] = nReconstructionCloud[i].pt.z;
  }
  for (size_t i = 0; i < newPointCloud.size(); ++i) {
    fresh[3 * i] = newPointCloud[i].pt.x;
    fresh[3 * i + 1] = newPointCloud[i].pt.y;
    fresh[3 * i + 2] = newPointCloud[i].pt.z;
  }
  std::vector<uint8_t> accept(newPointCloud.size());
  int32_t n = 0;
  const int rc = sfmhip_merge_new_points(sfm_hip_context(), cloud.data(), (int)nReconstructionCloud.size(), fresh.data(),
                                         (int)newPointCloud.size(), MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE, accept.data(), &n);
  if (rc != SFMHIP_OK) {
    std::cerr << "mergeNewPoints: " << sfmhip_error_string(rc) << std::endl;
    return;
  }
  for (size_t i = 0; i < newPointCloud.size(); ++i)
    if (accept[i]) nReconstructionCloud.push_back(newPointCloud[i]);  // no track merging: :1225-1240
}
