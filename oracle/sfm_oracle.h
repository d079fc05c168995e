/*
 * sfm_oracle.h -- CPU restatement of the reference's feature-matching /
 * triangulation / bundle-adjustment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (sfm_danpipeline_amd/,
 * include/) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (codebydant/sfM_danPipeline) ships no tests,
 * fixtures or golden vectors, and its arithmetic lives in OpenCV 3.4.1 / Ceres 1.13.0,
 * neither of which is vendored under /root/reference nor installed here.  This file
 * restates the published algorithms of those pinned versions at the reference's
 * call sites (cited per function) and is cross-checked against independent
 * numpy / scipy restatements (oracle/np_check.py, tests/golden/).
 */
#ifndef SFM_ORACLE_H
#define SFM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DTYPE_F32 = 0, ORC_DTYPE_U8 = 1 };
enum { ORC_NORM_L2 = 0, ORC_NORM_HAMMING = 1 };

/* reference: src/Sfm.cpp:590-608 (getMatching) -> cv::BFMatcher(NORM_L2,false).knnMatch(k=2)
 * + ratio test `d0 <= ratio*d1` (src/Sfm.cpp:603-607).  knn_idx/knn_dist (nq*2, may be NULL)
 * receive the raw k=2 lists (-1 / FLT_MAX padded).  Outputs are in ascending queryIdx.
 * dim = elements per row (f32) or bytes per row (u8).  Nt<2: no match is emitted (the
 * reference would read knn[i][1] out of bounds, src/Sfm.cpp:604).  threads<=1: serial. */
int orc_match_knn2(const void* q, int nq, const void* t, int nt, int dim, int dtype, int norm,
                   float ratio, int32_t* out_q, int32_t* out_t, float* out_dist, int32_t* out_n,
                   int32_t* knn_idx, float* knn_dist, int threads);

/* the same for a list of pairs, parallel over (pair, query row); counts[p] = #matches of pair p.
 * CPU-baseline leg of bench.py. */
int orc_match_many(const void* const* imgs, const int32_t* n_rows, int dim, int dtype, int norm,
                   const int32_t* pairs, int n_pairs, float ratio, int threads, int32_t* counts);

/* reference: src/Sfm.cpp:804-878 (triangulateViews): undistortPoints (dist==0 path and the
 * 5-iteration path), cv::triangulatePoints 4x4 DLT, convertPointsFromHomogeneous,
 * projectPoints in both views, float 6 px filter.  xy1/xy2 are the already-gathered pixel
 * coordinates (AlignedPoints, src/Sfm.cpp:694-711).  X: 3*m doubles, err: 2*m floats (may be
 * NULL), keep: m bytes. */
/* orc_match_many plus a checksum per pair: checksums[2p] = sum, [2p+1] = xor of orc_match_mix over the
 * pair's matches (queryIdx, trainIdx, bit pattern of the float distance). */
int orc_match_many_checksum(const void* const* imgs, const int32_t* n_rows, int dim, int dtype, int norm,
                            const int32_t* pairs, int n_pairs, float ratio, int threads, int32_t* counts,
                            uint64_t* checksums);
uint64_t orc_match_mix(uint32_t q, uint32_t t, uint32_t dist_bits);
/* bench.py's CPU-baseline matcher (L2, f32 rows): the same lists, organised as cv::batchDistance under parallel_for_ --
 * blocks of query rows in parallel, train rows in cache-sized tiles, a SIMD sum of squared differences, per-thread
 * counts / checksums merged once.  See sfm_oracle_match.c. */
int orc_match_many_blocked(const void* const* imgs, const int32_t* n_rows, int dim, const int32_t* pairs, int n_pairs,
                           float ratio, int threads, int32_t* counts, uint64_t* checksums);

int orc_triangulate(const double P1[12], const double P2[12], const double K[9],
                    const double dist[5], const double* xy1, const double* xy2, int m,
                    float max_err, double* X, float* err, uint8_t* keep);

/* ceres/rotation.h helpers used at src/BundleAdjustment.cpp:16,67,150 */
void orc_rotmat_colmajor_to_angleaxis(const double R[9], double aa[3]);
void orc_angleaxis_to_rotmat_colmajor(const double aa[3], double R[9]);
void orc_angleaxis_rotate_point(const double aa[3], const double X[3], double out[3]);

/* residual + analytic Jacobian of SimpleReprojectionError (src/BundleAdjustment.cpp:10-35).
 * Jc 2x6, Jp 2x3, Jf 2x1 row-major; any J pointer may be NULL. */
void orc_ba_residual(const double cam[6], const double X[3], double focal, const double obs[2],
                     double r[2], double* Jc, double* Jp, double* Jf);

typedef struct {
  int max_iterations;           /* 500   src/BundleAdjustment.cpp:118 */
  double max_time_s;            /* 10    src/BundleAdjustment.cpp:120 ; <=0 disables */
  double function_tolerance;    /* 1e-6  Ceres 1.13 default */
  double gradient_tolerance;    /* 1e-10 */
  double parameter_tolerance;   /* 1e-8  */
  double initial_radius;        /* 1e4   */
  double max_radius;            /* 1e16  */
  double min_radius;            /* 1e-32 */
  double min_relative_decrease; /* 1e-3  */
  double min_lm_diagonal;       /* 1e-6  */
  double max_lm_diagonal;       /* 1e32  */
  int jacobi_scaling;           /* 1     */
  int max_consecutive_invalid;  /* 5     */
  int verbose;
} orc_ba_opts;

enum { ORC_BA_CONVERGENCE = 0, ORC_BA_NO_CONVERGENCE = 1, ORC_BA_FAILURE = 2 };

typedef struct {
  int termination;       /* ORC_BA_* */
  int iterations;        /* LM iterations attempted (successful + unsuccessful) */
  int successful_steps;
  double initial_cost;
  double final_cost;
  double final_radius;
  double gradient_max_norm;
  double time_s;
} orc_ba_summary;

void orc_ba_default_opts(orc_ba_opts* o);
/* threads of the per-point passes (what Ceres' num_threads does); 1 = the serial reference order (default) */
void orc_ba_set_threads(int n);
/* bench.py's timed cpu_baseline leg only: the reduced solve by a blocked, vectorised right-looking Cholesky (what Eigen's LLT
 * inside Ceres 1.13 is) instead of the checker's row-by-row factorisation; stats: {flops, seconds} of the factorisations since
 * the last reset.  orc_chol_solve: either factorisation on a caller's system (S n x n row-major, lower triangle read, overwritten). */
void orc_ba_set_blocked_cholesky(int on);
void orc_ba_cholesky_stats(double out[2], int reset);
int orc_chol_solve(double* S, int n, const double* rhs, double* x, int blocked);

/* ceres::Solve(DENSE_SCHUR, LM) as configured at src/BundleAdjustment.cpp:115-123.
 * Parameters are updated in place with the best accepted iterate whatever the termination
 * type; the caller applies the reference's write-back-only-on-CONVERGENCE rule
 * (src/BundleAdjustment.cpp:126-129).  Observations may come in any order. */
int orc_ba_solve(int n_cam, int n_pt, int n_obs, double* cams6, double* pts3, double* focal,
                 const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                 const orc_ba_opts* opts, orc_ba_summary* summary);

/* One linearisation at x: cost, the Jacobi scale vector it implies, and (for a given radius)
 * the reduced system S (dim x dim, row-major, full symmetric), rhs g (dim), with
 * dim = 6*n_cam+1 and column order [cam0(6) ... camN-1(6) focal].  scale_in==NULL: compute
 * scale from this Jacobian (Ceres iteration 0), else use it.  Used by tests of the sharded
 * path (sum of per-shard S equals the unsharded S) and by the iterate-level checks. */
int orc_ba_reduced_system(int n_cam, int n_pt, int n_obs, const double* cams6, const double* pts3,
                          double focal, const int32_t* obs_cam, const int32_t* obs_pt,
                          const double* obs_xy, double radius, const double* scale_in,
                          double* scale_out, double* S, double* g, double* cost);

/* Time `iters` LM iterations (linearise + eliminate + reduced solve + back-substitute +
 * candidate cost) from the given start without convergence checks; returns seconds. */
double orc_ba_time_iterations(int n_cam, int n_pt, int n_obs, const double* cams6,
                              const double* pts3, double focal, const int32_t* obs_cam,
                              const int32_t* obs_pt, const double* obs_xy, int iters,
                              double* final_cost);

/* find2D3DMatches association (src/Sfm.cpp:1047-1090) and mergeNewPoints (src/Sfm.cpp:1212-1244):
 * see sfm_oracle_incr.c */
int orc_find_2d3d(const int32_t* trk_ptr, const int32_t* trk_view, const int32_t* trk_feat,
                  int n_cloud, int done_view, int new_view, const int32_t* match_q,
                  const int32_t* match_t, int n_match, int32_t* out_cloud, int32_t* out_feat,
                  int32_t* n_out);
int orc_merge_new_points(const double* cloud_xyz, int n_cloud, const double* new_xyz, int n_new,
                         float min_dist, uint8_t* accept, int32_t* n_accepted);

/* findBestPair's scoring (src/Sfm.cpp:543-546): cv::findEssentialMat(RANSAC) as OpenCV 3.4.1 runs it -- see
 * sfm_oracle_score.c.  flags: bit 0 / bit 1 = a sample reached one of the two solvePoly corners that are not restated. */
int orc_five_point(const double* q1, const double* q2, double* E, int* flags);
void orc_em_normalize(const double* xy, int n, const double K[9], double* out);
int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters);
int orc_find_essential_mat(const double* pts1, const double* pts2, int count, const double K[9], double prob,
                           double threshold, int max_iters, uint8_t* mask, double* E_out, int* iters, int* flags);
/* findHomographyInliers (src/Sfm.cpp:667-689): cv::findHomography(RANSAC) as OpenCV 3.4.1 runs it -- see sfm_oracle_score.c */
int orc_homography_kernel(const float* M, const float* m, int count, double* H);
int orc_find_homography(const double* pts1, const double* pts2, int count, double threshold, double confidence, int max_iters,
                        uint8_t* mask, int* iters);
int orc_score_essential_many(int n_pairs, const int32_t* offsets, const double* left_xy, const double* right_xy,
                             const double K[9], double prob, double threshold, int32_t* counts, int32_t* iters,
                             uint8_t* masks, int threads, int32_t* flags_any);

#ifdef __cplusplus
}
#endif
#endif
