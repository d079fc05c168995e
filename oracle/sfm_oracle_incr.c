/*
 * sfm_oracle_incr.c -- CPU restatement of the incremental-loop glue next to the hot path
 * (SURVEY.md section 8f-2).  TEST INFRASTRUCTURE ONLY (see sfm_oracle.h).
 *
 * Literal loops of the reference:
 *   orc_find_2d3d         src/Sfm.cpp:1047-1090  (2D-3D association inside find2D3DMatches)
 *   orc_merge_new_points  src/Sfm.cpp:1212-1244  (mergeNewPoints)
 * PARITY UNPINNED: the reference has no tests or fixtures for them; both are pure index /
 * compare logic over the containers of include/Utilities.h:38-43.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include "sfm_oracle.h"

/* The cloud's tracks (Point3D::idxImage, a std::map ordered by view) come as CSR:
 * entries trk_ptr[p] .. trk_ptr[p+1]-1 of (trk_view, trk_feat), ascending view.  For every cloud
 * point, in cloud order: the first entry whose view is done_view, then the first match (in
 * match order) whose queryIdx (done_view < new_view: "originating view is left") or trainIdx
 * (else) equals that feature index; emits (cloud index, feature index in the new view). */
int orc_find_2d3d(const int32_t* trk_ptr, const int32_t* trk_view, const int32_t* trk_feat,
                  int n_cloud, int done_view, int new_view, const int32_t* match_q,
                  const int32_t* match_t, int n_match, int32_t* out_cloud, int32_t* out_feat,
                  int32_t* n_out) {
  int n = 0;
  for (int p = 0; p < n_cloud; ++p) {                          /* :1048 */
    int found = 0;
    for (int e = trk_ptr[p]; e < trk_ptr[p + 1]; ++e) {        /* :1052 */
      const int view = trk_view[e], feat = trk_feat[e];
      if (view != done_view) continue;                         /* :1058 */
      for (int m = 0; m < n_match; ++m) {                      /* :1061 */
        int matched = -1;
        if (view < new_view) {                                 /* :1064 originating view is 'left' */
          if (match_q[m] == feat) matched = match_t[m];
        } else {
          if (match_t[m] == feat) matched = match_q[m];
        }
        if (matched >= 0) {                                    /* :1076 */
          out_cloud[n] = p;
          out_feat[n] = matched;
          ++n;
          found = 1;
          break;
        }
      }
      if (found) break;                                        /* :1086 */
    }
  }
  *n_out = n;
  return 0;
}

/* cv::norm(Point3d) = sqrt(x*x + y*y + z*z) in double (core/types.hpp), compared with the
 * float literal 0.01 promoted to double (src/Sfm.cpp:1216,1227).  The cloud grows inside the
 * loop: a new point accepted earlier can block a later one (src/Sfm.cpp:1226,1237). */
int orc_merge_new_points(const double* cloud_xyz, int n_cloud, const double* new_xyz, int n_new,
                         float min_dist, uint8_t* accept, int32_t* n_accepted) {
  const double r = (double)min_dist;
  int n_acc = 0;
  for (int i = 0; i < n_new; ++i) {
    const double* q = new_xyz + 3 * (size_t)i;
    int found = 0;
    for (int j = 0; j < n_cloud && !found; ++j) {
      const double dx = cloud_xyz[3 * (size_t)j] - q[0], dy = cloud_xyz[3 * (size_t)j + 1] - q[1],
                   dz = cloud_xyz[3 * (size_t)j + 2] - q[2];
      if (sqrt(dx * dx + dy * dy + dz * dz) < r) found = 1;
    }
    for (int j = 0; j < i && !found; ++j) {                    /* points appended by this call */
      if (!accept[j]) continue;
      const double dx = new_xyz[3 * (size_t)j] - q[0], dy = new_xyz[3 * (size_t)j + 1] - q[1],
                   dz = new_xyz[3 * (size_t)j + 2] - q[2];
      if (sqrt(dx * dx + dy * dy + dz * dz) < r) found = 1;
    }
    accept[i] = found ? 0 : 1;
    n_acc += accept[i];
  }
  *n_accepted = n_acc;
  return 0;
}
