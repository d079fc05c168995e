// Sfm.h -- the hot-path part of the reference's StructFromMotion (include/Sfm.h:15-35,89,
// 107-117): same member names, same method signatures.  Everything above the hot path
// (image loading, feature extraction, RANSAC pose, PMVS/PCL post-processing) is out of scope
//  and is replaced by the loader methods at the bottom.
#pragma once
#include <map>
#include <set>
#include <string>
#include "BundleAdjustment.h"
#include "sfmhip.h"

class StructFromMotion {
 private:
  std::vector<cv::Matx34d> nCameraPoses;
  Intrinsics cameraMatrix;
  float NN_MATCH_RATIO;
  std::vector<std::vector<cv::KeyPoint>> imagesKeypoints;
  std::vector<cv::Mat> imagesDescriptors;
  std::vector<std::vector<cv::Point2d>> imagesPts2D;
  int detector;
  std::set<int> nDoneViews;  // reference include/Sfm.h:24-25
  std::set<int> nGoodViews;
  // image I/O state, reference include/Sfm.h:18-23
  std::vector<cv::Mat> mColorImages;
  std::vector<cv::Mat> mGrayImages;
  std::vector<cv::Mat> nImages;
  std::vector<std::string> nImagesPath;
  std::string pathImages;
  // pair-match cache: getMatching is a pure function of two descriptor
  // matrices and the reference calls it ~1.5 N^2 times for N(N-1)/2 distinct pairs
  std::map<std::pair<int, int>, Matching> pairCache;
  bool pairCacheOn;
  // descriptors resident in HBM + a reusable one-pair plan: a getMatching call that misses the
  // cache costs one kernel sequence instead of two uploads and a dozen allocations
  sfmhip_imageset* devSet;
  sfmhip_matchplan* devPlan;
  void releaseDeviceSet();
  // descriptor rows that extractFeature left in HBM (sfmhip_sift_batch), image by image (nullptr: host rows only).
  // The matcher adopts them in place; imagesDescriptors[i] then carries the shape only until descriptors() is asked.
  std::vector<void*> devDescriptors;
  void releaseDeviceDescriptors();
  int uploadOrAdopt(sfmhip_imageset* set, int image);

 public:
  std::vector<Point3D> nReconstructionCloud;

  StructFromMotion() : NN_MATCH_RATIO(0.8f), detector(1), pairCacheOn(false), devSet(nullptr), devPlan(nullptr) {}
  ~StructFromMotion() {
    releaseDeviceSet();
    releaseDeviceDescriptors();
  }
  StructFromMotion(const StructFromMotion&) = delete;
  StructFromMotion& operator=(const StructFromMotion&) = delete;

  // reference include/Sfm.h:89, src/Sfm.cpp:590-608
  void getMatching(const int& queryImage, const int& trainImage, Matching* goodMatches);
  // reference include/Sfm.h:107-111, src/Sfm.cpp:694-711
  void AlignedPointsFromMatch(const Points2d& queryImg, const Points2d& trainImg, const Matching& matches,
                              Points2d& alignedL, Points2d& alignedR);
  void AlignedPoints(const Points2d& queryImg, const Points2d& trainImg, const Matching& matches, Points2d& alignedL,
                     Points2d& alignedR, std::vector<int>& idLeftOrigen, std::vector<int>& idRightOrigen);
  // reference include/Sfm.h:115-117, src/Sfm.cpp:804-878
  bool triangulateViews(const Points2d& left, const Points2d& right, const cv::Matx34d& P1, const cv::Matx34d& P2,
                        const Matching& matches, const Intrinsics& matrixK, const std::pair<int, int>& imagePair,
                        std::vector<Point3D>& pointcloud);
  // reference include/Sfm.h:135-137, src/Sfm.cpp:1011-1095: best done view by match count, then the
  // 2D-3D association of the cloud against that view's matches
  void find2D3DMatches(const int& NEW_VIEW, std::vector<cv::Point3d>& points3D, std::vector<cv::Point2d>& points2D,
                       Matching& bestMatches, int& DONEVIEW);
  // reference include/Sfm.h:150, src/Sfm.cpp:1212-1244
  void mergeNewPoints(const std::vector<Point3D>& newPointCloud);
  // The all-pairs loop of findBestPair (src/Sfm.cpp:511-515) as ONE batched device launch; fills
  // the pair cache that getMatching then serves from (identical results, pair order q<t).
  void matchAllPairs();
  // reference include/Sfm.h:83, src/Sfm.cpp:499-585: the all-pairs matching (matchAllPairs: one batched launch),
  // then for every pair with >= 120 matches the pose-inlier ratio of cv::findEssentialMat(RANSAC, 0.999, 1.0)
  // (sfmhip_score_essential, all pairs in one call), collected in the map keyed by that float: ascending, equal keys
  // overwrite; the homography inlier count the reference prints next to it (:545,567) comes from
  // sfmhip_score_homography, all pairs in one call.  Not mirrored: the drawMatches / imshow / waitKey(100) per pair.
  std::map<float, std::pair<int, int>> findBestPair();
  // reference include/Sfm.h:137, src/Sfm.cpp:667-689: cv::findHomography(RANSAC, 0.004 * maxVal) inlier count of one pair
  int findHomographyInliers(const int& idx_query, const int& idx_train, const Matching& matches);
  // reference src/Sfm.cpp:883-888 is a stub whose call names a member that no longer exists;
  // wired here with imagesPts2D, the member of the required type
  void adjustCurrentBundle();

  // ---- detector / descriptor front end (csrc/host/SfmIO.cpp)
  // reference include/Sfm.h:85, src/Sfm.cpp:257-296: every gray image through getFeature (the imshow / waitKey(100) per
  // image are not mirrored)
  void extractFeature();
  // reference include/Sfm.h:81, src/Sfm.cpp:300-403: detector 1 = SIFT(0, 3, 0.04, 10, 1.6)->detectAndCompute on the
  // device (sfmhip_sift_detect_and_compute); detectors 2 (AKAZE) and 3 (ORB) are not built and leave the image's
  // containers empty with a message
  void getFeature(const cv::Mat& image, const int& numImage);
  // reference include/Sfm.h:95, src/Sfm.cpp:397-403
  void keypointstoPoints(std::vector<cv::KeyPoint>& keypoints, Points2d& points2D);
  void setKeypoints(int numImage, const float* kp, int n);
  const std::vector<std::vector<cv::KeyPoint>>& keypoints() const { return imagesKeypoints; }
  // (downloads the rows of images whose descriptors live in HBM only)
  const std::vector<cv::Mat>& descriptors();
  const std::vector<std::vector<cv::Point2d>>& points2D() const { return imagesPts2D; }

  // ---- I/O and interchange formats (csrc/host/SfmIO.cpp)
  // reference include/Sfm.h:77, src/Sfm.cpp:118-198: scan a directory for .jpg/.png, sort, decode to BGR,
  // x0.6 bilinear iff rows > 480 and cols > 640, colour + gray copies.  PNG and Huffman-coded 8-bit JPEG -- sequential and
  // progressive -- are decoded here (no OpenCV, no libpng / libjpeg; the JPEG path follows libjpeg's integer IDCT, fancy
  // upsampling and colour tables byte for byte, interleaved or one scan per component, any of the 1x1 / 2x1 / 1x2 / 2x2
  // samplings; a JPEG is turned as its EXIF orientation says, like cv::imread's default; PNG: Adam7 too); an arithmetic-coded,
  // lossless or 12-bit JPEG is reported and fails the load.
  bool imagesLOAD(const std::string& directoryPath);
  // reference include/Sfm.h:99, src/Sfm.cpp:203-252: the OpenCV FileStorage XML with Camera_Matrix and
  // Distortion_Coefficients.  Values are read as the numbers the file holds (the reference reads a `dt f`
  // matrix through at<double>); coefficient slots kept in file order.
  bool getCameraMatrix(const std::string str);
  // reference include/Sfm.h:179, src/Sfm.cpp:1246-1303: denseCloud/{visualize,txt,models}, options.txt,
  // visualize/%04d.jpg = a byte copy of the input file (what the reference's `cp -f` does; its imwrite
  // targets the command string and writes nothing), txt/%04d.txt = "CONTOUR" + K*P.  Does NOT run pmvs2.
  void PMVS2();
  // reference src/Sfm.cpp:69-81 (step 8 of map3D): denseCloud/models/options.txt.ply -> MAP3D.pcd.  The reference
  // does this with pcl::PLYReader + pcl::io::savePCDFile (ASCII, PointXYZRGB); here a PLY reader (ascii and
  // binary_little_endian; x y z [nx ny nz] [diffuse_]red green blue) and a PCD v0.7 ASCII writer in the layout PCL
  // 1.8 documents (FIELDS x y z rgb, rgb = the packed 0x00RRGGBB word printed as a float).  Returns the point count
  // (0 = "ply file is empty", the reference's failure case).  Not pinned against PCL: it is absent from the image.
  static size_t convertPLYtoPCD(const std::string& plyPath, const std::string& pcdPath);
  const std::vector<cv::Mat>& colorImages() const { return mColorImages; }
  const std::vector<cv::Mat>& grayImages() const { return mGrayImages; }
  const std::vector<std::string>& imagePaths() const { return nImagesPath; }

  // ---- stand-ins for the out-of-scope front end: hand the pipeline state in directly
  void setDescriptors(const std::vector<cv::Mat>& d) {
    imagesDescriptors = d;
    releaseDeviceSet();
    releaseDeviceDescriptors();
    clearPairCache();
  }
  void setPoints2D(const std::vector<std::vector<cv::Point2d>>& p) { imagesPts2D = p; }
  void setCameraMatrix(const Intrinsics& k) { cameraMatrix = k; }
  void setCameraPoses(const std::vector<cv::Matx34d>& p) { nCameraPoses = p; }
  const std::vector<cv::Matx34d>& cameraPoses() const { return nCameraPoses; }
  const Intrinsics& intrinsics() const { return cameraMatrix; }
  void setMatchRatio(float r) { NN_MATCH_RATIO = r; }
  void setDoneViews(const std::set<int>& v) { nDoneViews = v; }
  void setGoodViews(const std::set<int>& v) { nGoodViews = v; }
  void clearPairCache() { pairCache.clear(); pairCacheOn = false; }
  size_t pairCacheSize() const { return pairCache.size(); }
};
