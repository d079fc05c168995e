// io_selftest.cpp -- drives imagesLOAD / getCameraMatrix / PMVS2 of the C++ host mirror:
//   io_selftest <image dir> <calibration.xml> <out.bin>      (PMVS2 writes ./denseCloud: run it in a scratch directory)
//   io_selftest --ply2pcd <in.ply> <out.pcd>                 (prints the point count)
//   io_selftest --features <image dir> <out.bin>             (needs the GPU) imagesLOAD + extractFeature;
//        out.bin: i32 n; per image: i32 rows, cols, gray bytes, i32 nk, nk x (6 f32 keypoint, octave bit-copied), nk x 128 f32,
//        nk x (f64 x, f64 y) of imagesPts2D
// out.bin: i32 ok_images, i32 ok_calib, i32 n; per image: i32 rows, cols, BGR bytes, gray bytes; f64 K[9], dist[5]
#include <cstdio>
#include <string>
#include "Sfm.h"

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
