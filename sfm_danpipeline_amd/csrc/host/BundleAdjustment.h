// BundleAdjustment.h -- call surface of the reference's include/BundleAdjustment.h:6-22.
#pragma once
#include "Utilities.h"

class BundleAdjustment {
 public:
  BundleAdjustment() {}
  ~BundleAdjustment() {}

  // reference include/BundleAdjustment.h:19-20 / src/BundleAdjustment.cpp:46-175
  static void adjustBundle(std::vector<Point3D>& pointCloud, std::vector<cv::Matx34d>& cameraPoses,
                           Intrinsics& intrinsics, const std::vector<std::vector<cv::Point2d>>& image2dFeatures);
};
