// Utilities.h -- shared types of the pipeline, same names and fields as the reference's
// include/Utilities.h:26-43 (the OpenCV / PCL header dump of :4-21 is not carried over).
#pragma once
#include <map>
#include <vector>
#include "cvlite.h"

using Keypoints = std::vector<cv::KeyPoint>;
using Matching = std::vector<cv::DMatch>;
using Points2d = std::vector<cv::Point2d>;

struct Intrinsics {
  cv::Mat_<double> K;
  cv::Mat_<double> distCoef;
};

struct Point3D {
  cv::Point3d pt;
  std::map<const int, int> idxImage;          // view -> feature index: the track / visibility
  std::map<const int, cv::Point2d> pt2D;
};
