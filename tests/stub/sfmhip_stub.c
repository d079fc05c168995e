/* sfmhip_stub.c -- a CPU stand-in for the part of the C ABI (include/sfmhip.h) that the C++ host mirror calls,
 * answered by the CPU oracle.  TEST INFRASTRUCTURE ONLY: it exists so that csrc/host/{Sfm,BundleAdjustment}.cpp and
 * selftest.cpp can run under AddressSanitizer / UBSan on a machine without a GPU (sanitizers are not available on
 * the GPU pool).  Nothing in the product links or loads it. */
#include <stdlib.h>
#include <string.h>
#include "../../include/sfmhip.h"
#include "../../oracle/sfm_oracle.h"

struct sfmhip_ctx { int dummy; };
struct sfmhip_imageset {
  int n_images, dim, dtype, norm;
  int* n_rows;
  void** rows;
};
struct sfmhip_matchplan {
  struct sfmhip_imageset* set;
  int n_pairs, cap_pairs;
  int32_t* pairs;
  int32_t* counts;
  int32_t **q, **t;
  float** d;
  int pipe_on;         /* sfmhip_matchplan_pipeline: fetch_wait hands out pointers into `packed` */
  int pipe_small;      /* (test switch: every batch overflows the pinned buffers) */
  int32_t* packed;
};

int sfmhip_device_count(void) { return 1; }
int sfmhip_init(int device, sfmhip_ctx** out) {
  (void)device;
  *out = (sfmhip_ctx*)calloc(1, sizeof(sfmhip_ctx));
  return *out ? SFMHIP_OK : SFMHIP_ERR_ALLOC;
}
void sfmhip_shutdown(sfmhip_ctx* ctx) { free(ctx); }
const char* sfmhip_error_string(int status) { return status == 0 ? "ok" : "error (stub)"; }

int sfmhip_match_knn2(sfmhip_ctx* ctx, const void* q, int nq, const void* t, int nt, int dim, int dtype, int norm, float ratio,
                      int32_t* out_q, int32_t* out_t, float* out_dist, int32_t* out_n) {
  (void)ctx;
  return orc_match_knn2(q, nq, t, nt, dim, dtype, norm, ratio, out_q, out_t, out_dist, out_n, NULL, NULL, 1) ? SFMHIP_ERR_ARG : SFMHIP_OK;
}

int sfmhip_imageset_create(sfmhip_ctx* ctx, int n_images, const int32_t* n_rows, int dim, int dtype, int norm, sfmhip_imageset** out) {
  (void)ctx;
  sfmhip_imageset* s = (sfmhip_imageset*)calloc(1, sizeof *s);
  s->n_images = n_images, s->dim = dim, s->dtype = dtype, s->norm = norm;
  s->n_rows = (int*)malloc(sizeof(int) * (size_t)n_images);
  s->rows = (void**)calloc((size_t)n_images, sizeof(void*));
  memcpy(s->n_rows, n_rows, sizeof(int) * (size_t)n_images);
  *out = s;
  return SFMHIP_OK;
}
int sfmhip_imageset_upload(sfmhip_imageset* s, int image, const void* host_rows) {
  const size_t bytes = (size_t)s->n_rows[image] * (size_t)s->dim * (s->dtype == SFMHIP_F32 ? 4 : 1);
  free(s->rows[image]);
  s->rows[image] = malloc(bytes ? bytes : 1);
  if (bytes) memcpy(s->rows[image], host_rows, bytes);
  return SFMHIP_OK;
}
/* (the stand-in's "device" is host memory) */
int sfmhip_imageset_adopt_device(sfmhip_imageset* s, int image, const void* device_rows) { return sfmhip_imageset_upload(s, image, device_rows); }
void sfmhip_device_free(void* p) { free(p); }
void sfmhip_host_free(void* p) { free(p); }
int sfmhip_device_download(sfmhip_ctx* ctx, void* host_dst, const void* device_src, size_t bytes) {
  (void)ctx;
  if (bytes) memcpy(host_dst, device_src, bytes);
  return SFMHIP_OK;
}
int sfmhip_imageset_prepare_async(sfmhip_imageset* s) { (void)s; return SFMHIP_OK; }
void sfmhip_imageset_destroy(sfmhip_imageset* s) {
  if (!s) return;
  for (int i = 0; i < s->n_images; ++i) free(s->rows[i]);
  free(s->rows);
  free(s->n_rows);
  free(s);
}

static void plan_free_lists(sfmhip_matchplan* pl) {
  for (int p = 0; p < pl->cap_pairs; ++p) {
    free(pl->q[p]);
    free(pl->t[p]);
    free(pl->d[p]);
    pl->q[p] = pl->t[p] = NULL;
    pl->d[p] = NULL;
  }
}
int sfmhip_matchplan_create(sfmhip_imageset* s, const int32_t* pairs, int n_pairs, sfmhip_matchplan** out) {
  sfmhip_matchplan* pl = (sfmhip_matchplan*)calloc(1, sizeof *pl);
  pl->set = s, pl->n_pairs = n_pairs, pl->cap_pairs = n_pairs > 0 ? n_pairs : 1;
  pl->pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)pl->cap_pairs);
  if (n_pairs) memcpy(pl->pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs);
  pl->counts = (int32_t*)calloc((size_t)pl->cap_pairs, sizeof(int32_t));
  pl->q = (int32_t**)calloc((size_t)pl->cap_pairs, sizeof(int32_t*));
  pl->t = (int32_t**)calloc((size_t)pl->cap_pairs, sizeof(int32_t*));
  pl->d = (float**)calloc((size_t)pl->cap_pairs, sizeof(float*));
  *out = pl;
  return SFMHIP_OK;
}
int sfmhip_matchplan_set_pairs(sfmhip_matchplan* pl, const int32_t* pairs, int n_pairs) {
  if (n_pairs > pl->cap_pairs) return SFMHIP_ERR_ARG;
  pl->n_pairs = n_pairs;
  if (n_pairs) memcpy(pl->pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs);
  return SFMHIP_OK;
}
int sfmhip_matchplan_run_async(sfmhip_matchplan* pl, float ratio) {
  sfmhip_imageset* s = pl->set;
  plan_free_lists(pl);
  for (int p = 0; p < pl->n_pairs; ++p) {
    const int a = pl->pairs[2 * p], b = pl->pairs[2 * p + 1];
    const int nq = s->n_rows[a];
    pl->q[p] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1));
    pl->t[p] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1));
    pl->d[p] = (float*)malloc(sizeof(float) * (size_t)(nq ? nq : 1));
    int32_t n = 0;
    if (nq && s->n_rows[b] >= 2)
      orc_match_knn2(s->rows[a], nq, s->rows[b], s->n_rows[b], s->dim, s->dtype, s->norm, ratio, pl->q[p], pl->t[p], pl->d[p], &n,
                     NULL, NULL, 1);
    pl->counts[p] = n;
  }
  return SFMHIP_OK;
}
int sfmhip_matchplan_fetch(sfmhip_matchplan* pl, int32_t* counts, int32_t* out_q, int32_t* out_t, float* out_dist, int64_t capacity,
                           int64_t* total) {
  int64_t tot = 0;
  for (int p = 0; p < pl->n_pairs; ++p) {
    counts[p] = pl->counts[p];
    tot += counts[p];
  }
  if (total) *total = tot;
  if (!out_q && !out_t && !out_dist) return SFMHIP_OK;
  if (tot > capacity) return SFMHIP_ERR_ARG;
  int64_t off = 0;
  for (int p = 0; p < pl->n_pairs; ++p) {
    const size_t n = (size_t)pl->counts[p];
    if (out_q) memcpy(out_q + off, pl->q[p], n * 4);
    if (out_t) memcpy(out_t + off, pl->t[p], n * 4);
    if (out_dist) memcpy(out_dist + off, pl->d[p], n * 4);
    off += (int64_t)n;
  }
  return SFMHIP_OK;
}
int sfmhip_matchplan_pipeline(sfmhip_matchplan* pl, int64_t capacity) {
  /* SFMHIP_STUB_PIPELINE=refuse: no pinned memory to be had; =small: buffers that overflow on the first batch -- the two
   * ways matchAllPairs must fall back to the copying fetch (csrc/host/Sfm.cpp) */
  const char* mode = getenv("SFMHIP_STUB_PIPELINE");
  if (capacity < 0) {
    pl->pipe_on = 0;
    return SFMHIP_OK;
  }
  if (mode && !strcmp(mode, "refuse")) return SFMHIP_ERR_ALLOC;
  pl->pipe_small = mode && !strcmp(mode, "small");
  pl->pipe_on = 1;
  return SFMHIP_OK;
}
int sfmhip_matchplan_fetch_wait(sfmhip_matchplan* pl, int back, const int32_t** counts, const int32_t** out_q, const int32_t** out_t,
                                const float** out_dist, int64_t* total) {
  if (!pl->pipe_on || back != 0) return SFMHIP_ERR_STATE; /* (the stand-in keeps the latest run only) */
  int64_t tot = 0;
  for (int p = 0; p < pl->n_pairs; ++p) tot += pl->counts[p];
  if (pl->pipe_small && tot > 0) {
    if (total) *total = tot;
    if (counts) *counts = pl->counts;
    return SFMHIP_ERR_ALLOC;
  }
  free(pl->packed);
  pl->packed = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(tot ? tot : 1));
  int64_t off = 0;
  for (int p = 0; p < pl->n_pairs; ++p) {
    const size_t n = (size_t)pl->counts[p];
    memcpy(pl->packed + off, pl->q[p], n * 4);
    memcpy(pl->packed + tot + off, pl->t[p], n * 4);
    memcpy(pl->packed + 2 * tot + off, pl->d[p], n * 4);
    off += (int64_t)n;
  }
  if (total) *total = tot;
  if (counts) *counts = pl->counts;
  if (out_q) *out_q = pl->packed;
  if (out_t) *out_t = pl->packed + tot;
  if (out_dist) *out_dist = (const float*)(pl->packed + 2 * tot);
  return SFMHIP_OK;
}
void sfmhip_matchplan_destroy(sfmhip_matchplan* pl) {
  if (!pl) return;
  plan_free_lists(pl);
  free(pl->packed);
  free(pl->q);
  free(pl->t);
  free(pl->d);
  free(pl->pairs);
  free(pl->counts);
  free(pl);
}

int sfmhip_triangulate(sfmhip_ctx* ctx, const double P1[12], const double P2[12], const double K[9], const double dist[5],
                       const double* xy1, const double* xy2, int m, float max_err, double* X, float* err, uint8_t* keep) {
  (void)ctx;
  float* e = err ? err : (float*)malloc(sizeof(float) * 2 * (size_t)(m ? m : 1));
  const int rc = orc_triangulate(P1, P2, K, dist, xy1, xy2, m, max_err, X, e, keep);
  if (!err) free(e);
  return rc ? SFMHIP_ERR_ARG : SFMHIP_OK;
}
/* the scoring half of findBestPair, answered by the C restatement of cv::findEssentialMat(RANSAC) */
static int g_score_flags;
int sfmhip_score_essential(sfmhip_ctx* ctx, int n_pairs, const int32_t* offsets, const double* left_xy, const double* right_xy,
                           double fx, double fy, double cx, double cy, double prob, double threshold, int32_t* inliers,
                           uint8_t* mask, int32_t* iterations) {
  (void)ctx;
  const double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
  int32_t fl = 0;
  orc_score_essential_many(n_pairs, offsets, left_xy, right_xy, K, prob, threshold, inliers, iterations, mask, 1, &fl);
  g_score_flags = fl;
  return SFMHIP_OK;
}
int sfmhip_score_last_flags(sfmhip_ctx* ctx) { (void)ctx; return g_score_flags; }
int sfmhip_score_homography_kernel(sfmhip_ctx* ctx, int n_samples, const float* M, const float* m, double* H, int32_t* ok) {
  (void)ctx;
  for (int i = 0; i < n_samples; ++i) ok[i] = orc_homography_kernel(M + 8 * (size_t)i, m + 8 * (size_t)i, 4, H + 9 * (size_t)i);
  return SFMHIP_OK;
}
int sfmhip_score_five_point(sfmhip_ctx* ctx, int n_samples, const double* q1, const double* q2, double* models, int32_t* n_models) {
  (void)ctx;
  for (int i = 0; i < n_samples; ++i) {
    int fl = 0;
    const int n = orc_five_point(q1 + 10 * (size_t)i, q2 + 10 * (size_t)i, models + 90 * (size_t)i, &fl);
    n_models[i] = n | (fl << 8);
  }
  return SFMHIP_OK;
}

/* the SIFT front end is device code (its checker is numpy): the stand-in finds no keypoints */
int sfmhip_sift_detect_and_compute(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int n_octave_layers,
                                   double contrast_threshold, double edge_threshold, double sigma, int capacity, float* keypoints,
                                   float* descriptors, int32_t* n_keypoints) {
  (void)ctx; (void)gray; (void)rows; (void)cols; (void)n_octave_layers; (void)contrast_threshold; (void)edge_threshold; (void)sigma;
  (void)capacity; (void)keypoints; (void)descriptors;
  *n_keypoints = 0;
  return SFMHIP_OK;
}

int sfmhip_sift_batch(sfmhip_ctx* ctx, int n_images, const uint8_t* const* gray, const int32_t* rows, const int32_t* cols,
                      int n_octave_layers, double contrast_threshold, double edge_threshold, double sigma, float** keypoints,
                      void** d_descriptors, int32_t* n_keypoints) {
  (void)ctx; (void)gray; (void)rows; (void)cols; (void)n_octave_layers; (void)contrast_threshold; (void)edge_threshold; (void)sigma;
  for (int i = 0; i < n_images; ++i) {
    keypoints[i] = (float*)malloc(6 * sizeof(float));
    d_descriptors[i] = NULL;
    n_keypoints[i] = 0;
  }
  return SFMHIP_OK;
}

int sfmhip_score_homography(sfmhip_ctx* ctx, int n_pairs, const int32_t* offsets, const double* left_xy, const double* right_xy,
                            const double* thresholds, double confidence, int max_iters, int32_t* inliers, uint8_t* mask,
                            int32_t* iterations) {
  (void)ctx; (void)left_xy; (void)right_xy; (void)thresholds; (void)confidence; (void)max_iters;
  for (int p = 0; p < n_pairs; ++p) {
    inliers[p] = offsets[p + 1] - offsets[p];
    if (iterations) iterations[p] = 1;
    if (mask)
      for (int i = offsets[p]; i < offsets[p + 1]; ++i) mask[i] = 1;
  }
  return SFMHIP_OK;
}

int sfmhip_find_2d3d(sfmhip_ctx* ctx, const int32_t* trk_ptr, const int32_t* trk_view, const int32_t* trk_feat, int n_cloud,
                     int done_view, int new_view, const int32_t* match_q, const int32_t* match_t, int n_match, int32_t* out_cloud,
                     int32_t* out_feat, int32_t* n_out) {
  (void)ctx;
  return orc_find_2d3d(trk_ptr, trk_view, trk_feat, n_cloud, done_view, new_view, match_q, match_t, n_match, out_cloud, out_feat, n_out)
             ? SFMHIP_ERR_ARG : SFMHIP_OK;
}
int sfmhip_host_parallel_for(int n, void (*fn)(int lo, int hi, void* user), void* user) {
  if (n < 0 || !fn) return SFMHIP_ERR_ARG;
  if (n) fn(0, n, user);
  return SFMHIP_OK;
}
int sfmhip_ba_last_solve_profile(sfmhip_ctx* ctx, sfmhip_ba_solve_profile* out) {
  (void)ctx;
  if (!out) return SFMHIP_ERR_ARG;
  memset(out, 0, sizeof *out); /* (the stand-in measures nothing) */
  return SFMHIP_OK;
}
int sfmhip_merge_new_points(sfmhip_ctx* ctx, const double* cloud_xyz, int n_cloud, const double* new_xyz, int n_new, float min_dist,
                            uint8_t* accept, int32_t* n_accepted) {
  (void)ctx;
  return orc_merge_new_points(cloud_xyz, n_cloud, new_xyz, n_new, min_dist, accept, n_accepted) ? SFMHIP_ERR_ARG : SFMHIP_OK;
}
void sfmhip_ba_default_opts(sfmhip_ba_opts* o) { orc_ba_default_opts((orc_ba_opts*)o); }
int sfmhip_ba_solve(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, double* cams6, double* pts3, double* focal, const int32_t* obs_cam,
                    const int32_t* obs_pt, const double* obs_xy, const sfmhip_ba_opts* opts, sfmhip_ba_summary* summary) {
  (void)ctx;
  if (summary) summary->spin_timeouts = 0; /* (the checker's summary is a prefix of the C ABI's) */
  return orc_ba_solve(n_cam, n_pt, n_obs, cams6, pts3, focal, obs_cam, obs_pt, obs_xy, (const orc_ba_opts*)opts, (orc_ba_summary*)summary)
             ? SFMHIP_ERR_ARG : SFMHIP_OK;
}
