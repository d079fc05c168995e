/*
 * sfmhip.h -- C ABI of the MI355X (gfx950) implementation of the feature-matching +
 * triangulation + bundle-adjustment hot path of codebydant/sfM_danPipeline.
 *
 * This is the drop-in boundary.  The reference has no plugin / FFI
 * layer: its boundary is three C++ member functions, which the host mirror in
 * sfm_danpipeline_amd/csrc/host/ (Sfm.h, BundleAdjustment.h) keeps verbatim and forwards here:
 *
 *   StructFromMotion::getMatching      reference include/Sfm.h:89,  src/Sfm.cpp:590-608
 *   StructFromMotion::triangulateViews reference include/Sfm.h:115-117, src/Sfm.cpp:804-878
 *   BundleAdjustment::adjustBundle     reference include/BundleAdjustment.h:19-20,
 *                                      src/BundleAdjustment.cpp:46-175
 *
 * Conventions: plain C types only; every function returns an int status (0 = ok, <0 = error,
 * see sfmhip_error_string); no exceptions cross this boundary; the caller owns every buffer
 * it passes; calls on one context are synchronous to the caller unless the name says
 * `_async`; a context is bound to one HIP device and is not thread-safe (the reference is
 * single-threaded, src/Sfm.cpp:9-109).  There is NO CPU fallback behind this ABI: without a
 * gfx950 device sfmhip_init fails.
 */
#ifndef SFMHIP_H
#define SFMHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFMHIP_VERSION 1

/* status codes */
enum {
  SFMHIP_OK = 0,
  SFMHIP_ERR_NO_DEVICE = -1,
  SFMHIP_ERR_HIP = -2,       /* a HIP runtime call failed; sfmhip_last_hip_error() has the code */
  SFMHIP_ERR_ARG = -3,
  SFMHIP_ERR_ALLOC = -4,
  SFMHIP_ERR_UNSUPPORTED = -5,
  SFMHIP_ERR_STATE = -6,
  SFMHIP_ERR_COMM = -7,
  SFMHIP_ERR_TIMEOUT = -8    /* a bounded spin inside a kernel ran out and the level-by-level fallback did too: a scheduling
                                fault or a bug, never a property of the data (a matrix that is not positive definite is an
                                invalid LM step, not an error) */
};

/* descriptor element type of a cv::Mat row (reference include/Sfm.h:29, src/Sfm.cpp:326) */
enum { SFMHIP_F32 = 0, SFMHIP_U8 = 1 };
/* distance.  L2 is what the reference always uses (cv::NORM_L2, src/Sfm.cpp:593) -- also on
 * binary ORB/AKAZE rows; HAMMING is cv::NORM_HAMMING as asked for by BASELINE.json cfg5. */
enum { SFMHIP_L2 = 0, SFMHIP_HAMMING = 1 };

typedef struct sfmhip_ctx sfmhip_ctx;
typedef struct sfmhip_imageset sfmhip_imageset;
typedef struct sfmhip_matchplan sfmhip_matchplan;
typedef struct sfmhip_ba sfmhip_ba;

/* ---- context ---- */
/* gfx950 devices visible to this process (>= 0), or a negative SFMHIP_ERR_*: what a one-process-per-GPU host program
 * maps its rank onto (device = local rank % count) */
int sfmhip_device_count(void);
int sfmhip_init(int device, sfmhip_ctx** out);
/* same, but all work is enqueued on an existing hipStream_t (e.g. torch's current stream) */
int sfmhip_init_on_stream(int device, void* hip_stream, sfmhip_ctx** out);
void sfmhip_shutdown(sfmhip_ctx* ctx);
int sfmhip_synchronize(sfmhip_ctx* ctx);
/* the context's HIP device and hipStream_t (for code that enqueues next to it: RCCL, see sfmhip_rccl.h) */
int sfmhip_device(sfmhip_ctx* ctx);
void* sfmhip_stream(sfmhip_ctx* ctx);
/* Stage timing (sfmhip_matchplan_last_timing, sfmhip_ba_last_timing) is opt-in: the hipEvents it
 * records between the stages of a run cost ~5-10 us of stream bubble each (3-4 % of a cfg2 sweep
 * or a cfg4 LM iteration).  Off by default; while off the *_last_timing calls report zeros. */
int sfmhip_set_timing(sfmhip_ctx* ctx, int enable);
/* Measurement probes (csrc/probe.hip; bench.py runs them on the box it benches, nothing on the product path calls them).
 * sfmhip_probe_i8_mfma_peak: what the chip sustains for bare v_mfma_i32_32x32x32_i8 on random operands, every SIMD busy, for
 * `seconds` (one launch): multiply-add operations per second (2 per MAC) and the shader clock that launch held -- the ceiling
 * to read the k-NN sweep's roofline fraction against, next to the nominal peak.  sfmhip_probe_clock_start / _read: a
 * one-wave kernel on THIS context's stream that watches the shader and the 100 MHz counters for `seconds` while the caller runs
 * the load to be qualified on another stream; _read waits for it and returns the shader clock in GHz.  Qualify the matcher
 * behind cv::BFMatcher::knnMatch, reference src/Sfm.cpp:593-599. */
int sfmhip_probe_i8_mfma_peak(sfmhip_ctx* ctx, double seconds, double* ops_per_s, double* shader_ghz);
int sfmhip_probe_clock_start(sfmhip_ctx* ctx, double seconds);
int sfmhip_probe_clock_read(sfmhip_ctx* ctx, double* shader_ghz);
const char* sfmhip_error_string(int status);
int sfmhip_last_hip_error(void);
int sfmhip_version(void);

/* ---- getMatching: one pair, host buffers (reference src/Sfm.cpp:590-608) ----
 * q,t: row-major descriptor matrices (nq x dim, nt x dim; dim = elements per row).
 * Emits, in ascending queryIdx, knn[i][0] of every query with d0 <= ratio*d1 (float compare),
 * k-NN ties broken towards the lower trainIdx exactly as cv::batchDistance does.
 * out_q/out_t/out_dist: caller-allocated, nq entries each.  nt < 2 emits nothing. */
int sfmhip_match_knn2(sfmhip_ctx* ctx, const void* q, int nq, const void* t, int nt, int dim,
                      int dtype, int norm, float ratio, int32_t* out_q, int32_t* out_t,
                      float* out_dist, int32_t* out_n);

/* ---- all-pairs matching with descriptors resident in HBM (the findBestPair loop,
 *      reference src/Sfm.cpp:511-515, as one batched launch) ---- */
int sfmhip_imageset_create(sfmhip_ctx* ctx, int n_images, const int32_t* n_rows, int dim,
                           int dtype, int norm, sfmhip_imageset** out);
/* copy one image's descriptor matrix host -> HBM */
int sfmhip_imageset_upload(sfmhip_imageset* set, int image, const void* host_rows);
/* or adopt rows that already live in HBM (no copy; must stay valid while the set is used) */
int sfmhip_imageset_adopt_device(sfmhip_imageset* set, int image, const void* device_rows);
/* device pass over every image: integrality check, centring to i8 / bit expansion, row norms,
 * tie-break key bases.  Asynchronous on the context's stream. */
int sfmhip_imageset_prepare_async(sfmhip_imageset* set);
void sfmhip_imageset_destroy(sfmhip_imageset* set);

/* pairs: n_pairs x (queryImage, trainImage) int32, host memory */
int sfmhip_matchplan_create(sfmhip_imageset* set, const int32_t* pairs, int n_pairs,
                            sfmhip_matchplan** out);
/* point an existing plan at another pair list (n_pairs <= the count it was created with): the
 * device buffers are reused, e.g. a one-pair plan serving every getMatching(q,t) call over a
 * resident image set (reference src/Sfm.cpp:426,977,1031) */
int sfmhip_matchplan_set_pairs(sfmhip_matchplan* plan, const int32_t* pairs, int n_pairs);
/* k-NN + ratio test + ordered compaction for every pair of the plan; results stay in HBM */
int sfmhip_matchplan_run_async(sfmhip_matchplan* plan, float ratio);
/* counts[n_pairs]; optional concatenated lists (capacity entries each, pair-major, ascending
 * queryIdx inside a pair); *total receives the number of matches over all pairs. */
int sfmhip_matchplan_fetch(sfmhip_matchplan* plan, int32_t* counts, int32_t* out_q,
                           int32_t* out_t, float* out_dist, int64_t capacity, int64_t* total);
/* Pipelined fetch -- what keeps getMatching's host-visible contract (a host Matching*, reference include/Sfm.h:89) from
 * costing a stall per sweep when sweeps follow each other (the all-pairs loop of findBestPair, src/Sfm.cpp:511-515, over
 * batches of pairs): once switched on, every sfmhip_matchplan_run_async is followed, on a second HIP stream, by a pass
 * that packs counts and {queryIdx | trainIdx | distance} lists straight into one of two pinned host buffers while the
 * first stream goes on with the next run.  capacity = matches a buffer holds (0: a quarter of n_pairs x max rows;
 * negative: switch the pipeline off again).
 * sfmhip_matchplan_fetch_wait(plan, back, ...) waits for the lists of the latest run (back = 0) or of the run before it
 * (back = 1) and hands out pointers INTO the pinned buffer: counts[n_pairs], then total entries each of q, t, dist,
 * pair-major -- valid until two more runs have been enqueued.  SFMHIP_ERR_ALLOC with *total set when a run found more
 * matches than a buffer holds (switch the pipeline on again with that capacity; sfmhip_matchplan_fetch still works). */
int sfmhip_matchplan_pipeline(sfmhip_matchplan* plan, int64_t capacity);
int sfmhip_matchplan_fetch_wait(sfmhip_matchplan* plan, int back, const int32_t** counts, const int32_t** out_q,
                                const int32_t** out_t, const float** out_dist, int64_t* total);
/* raw k=2 lists of pair `pair`: idx[nq*2] (-1 padded), dist[nq*2] */
int sfmhip_matchplan_fetch_knn(sfmhip_matchplan* plan, int pair, int32_t* idx, float* dist);
/* seconds of device time of the last run's kernels, by stage (hipEvents on the ctx stream):
 * [0]=prepare (last prepare_async) [1]=knn kernels [2]=compaction; and the knn-kernel count */
int sfmhip_matchplan_last_timing(sfmhip_matchplan* plan, double seconds[3]);
/* seconds of the last run's k-NN kernel alone -- the one launch the MFMA roofline is priced on (stage [1]
 * above also holds the exact / fix-up kernels that follow it) */
int sfmhip_matchplan_last_knn_kernel_time(sfmhip_matchplan* plan, double* seconds);
void sfmhip_matchplan_destroy(sfmhip_matchplan* plan);

/* ---- triangulateViews numerics (reference src/Sfm.cpp:812-860) ----
 * P1,P2: cv::Matx34d row-major; K 3x3 row-major; dist k1,k2,p1,p2,k3; xy1/xy2: m gathered
 * pixel pairs (AlignedPoints, src/Sfm.cpp:694-711).  X: 3*m, keep: m (1 = both reprojection
 * errors <= max_err as float), err: 2*m floats or NULL. */
int sfmhip_triangulate(sfmhip_ctx* ctx, const double P1[12], const double P2[12],
                       const double K[9], const double dist[5], const double* xy1,
                       const double* xy2, int m, float max_err, double* X, float* err,
                       uint8_t* keep);

/* ---- incremental-loop glue next to the hot path ----
 * find2D3DMatches, the 2D-3D association (reference src/Sfm.cpp:1047-1090).  The cloud's tracks
 * (Point3D::idxImage, a std::map ordered by view) come as CSR: entries trk_ptr[p]..trk_ptr[p+1]-1
 * of (trk_view, trk_feat), ascending view.  For every cloud point, in cloud order: its feature
 * in done_view, then the FIRST match (match order) whose queryIdx (done_view < new_view) or
 * trainIdx (otherwise) is that feature; emits (cloud index, feature index in the new view).
 * out_cloud/out_feat: n_cloud entries each. */
int sfmhip_find_2d3d(sfmhip_ctx* ctx, const int32_t* trk_ptr, const int32_t* trk_view,
                     const int32_t* trk_feat, int n_cloud, int done_view, int new_view,
                     const int32_t* match_q, const int32_t* match_t, int n_match,
                     int32_t* out_cloud, int32_t* out_feat, int32_t* n_out);
/* mergeNewPoints (reference src/Sfm.cpp:1212-1244): new point i is appended iff no point
 * already in the cloud -- the existing ones and the new points appended before it -- lies closer
 * than min_dist (cv::norm of the difference in double, against the float literal promoted to
 * double).  accept: n_new bytes (1 = appended). */
int sfmhip_merge_new_points(sfmhip_ctx* ctx, const double* cloud_xyz, int n_cloud,
                            const double* new_xyz, int n_new, float min_dist, uint8_t* accept,
                            int32_t* n_accepted);

/* ---- the detector / descriptor front end of getFeature (reference src/Sfm.cpp:300-330) ----
 * cv::xfeatures2d::SIFT::create(nfeatures = 0, nOctaveLayers, contrastThreshold, edgeThreshold, sigma)
 *     ->detectAndCompute(gray, noArray(), keypoints, descriptors)
 * as OpenCV 3.4.1's float pipeline computes it (doubled base image, Gaussian / DoG pyramids, refined extrema,
 * orientation peaks, removeDuplicatedSorted, 4x4x8 descriptors x512 saturated to 8 bit and stored as float).
 * gray: rows x cols 8-bit, host memory.  keypoints: 6 floats each -- pt.x, pt.y, size, angle, response, and the
 * int32 `octave` field bit-copied into the sixth float -- in OpenCV's sorted order; descriptors: 128 floats each
 * (integer values 0..255: the input layout of the matcher).  capacity: keypoints the output arrays hold; 0 with null
 * arrays = count only.  *n_keypoints receives the count; more than `capacity` returns SFMHIP_ERR_ARG.
 * Parity unpinned (OpenCV is not available here): see DESIGN.md section 1, row f-3. */
int sfmhip_sift_detect_and_compute(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int n_octave_layers,
                                   double contrast_threshold, double edge_threshold, double sigma, int capacity,
                                   float* keypoints, float* descriptors, int32_t* n_keypoints);
/* The same with the descriptors LEFT IN HBM: *d_descriptors receives a device array of n_keypoints x 128 f32 rows that
 * the caller owns (sfmhip_device_free) and can hand to sfmhip_imageset_adopt_device -- the reference keeps keypoints and
 * descriptors of an image in one container (src/Sfm.cpp:326) and matches them right away; here the rows never leave the
 * device between extraction and matching.  keypoints: host, capacity x 6 floats as above (capacity 0: count only). */
int sfmhip_sift_detect_and_compute_device(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int n_octave_layers,
                                          double contrast_threshold, double edge_threshold, double sigma, int capacity,
                                          float* keypoints, void** d_descriptors, int32_t* n_keypoints);
/* extractFeature's loop (reference src/Sfm.cpp:283-290) as one call: n_images gray images, several in flight on worker
 * streams of the context (an image's front end is ~100 small launches and two read-backs: latency the next image hides).
 * keypoints[i]: a host array of n_keypoints[i] x 6 floats allocated by the library (sfmhip_host_free); d_descriptors[i]: a
 * device array of n_keypoints[i] x 128 f32 (sfmhip_device_free).  Image by image the results of the one-image entry. */
int sfmhip_sift_batch(sfmhip_ctx* ctx, int n_images, const uint8_t* const* gray, const int32_t* rows, const int32_t* cols,
                      int n_octave_layers, double contrast_threshold, double edge_threshold, double sigma, float** keypoints,
                      void** d_descriptors, int32_t* n_keypoints);
void sfmhip_device_free(void* device_ptr);
void sfmhip_host_free(void* host_ptr);
/* device -> host copy of bytes the library left in HBM (e.g. descriptor rows), on the context's stream, synchronous */
int sfmhip_device_download(sfmhip_ctx* ctx, void* host_dst, const void* device_src, size_t bytes);

/* ---- the scoring half of findBestPair (reference src/Sfm.cpp:536-563) ----
 * For every pair of a batch the inlier count of
 *   cv::findEssentialMat(alignedLeft, alignedRight, K, CV_RANSAC, prob, threshold, mask)       (src/Sfm.cpp:542-543)
 * as OpenCV 3.4.1 computes it: points normalised as p * (1 / f) + (-c / f), threshold / ((fx + fy) / 2), cv::RNG
 * restarted at (uint64)-1, five distinct sample indices, the models of a sample in turn, goodCount > max(best, 4)
 * updates the best and the iteration limit (RANSACUpdateNumIters, at most 1000), error = squared epipolar residual
 * over the four squared line coefficients as a float <= (float)(t*t).  The five-point solver follows the library's
 * runKernel step by step (Jacobi-SVD null space, 10 x 20 elimination, tenth-degree polynomial, solvePoly's
 * Durand-Kerner iteration, real iff |imag| <= 1e-10, SVD::solveZ, models in root order); OpenCV is not available to
 * pin it against.
 * offsets: n_pairs + 1 prefix sums of the match counts; left_xy / right_xy: 2 doubles per match (pixels), pair after
 * pair (host memory); inliers: n_pairs; mask (optional): one byte per match; iterations (optional): RANSAC iterations
 * run per pair.  Pairs with fewer than 5 matches score 0 (findEssentialMat returns an empty matrix). */
int sfmhip_score_essential(sfmhip_ctx* ctx, int n_pairs, const int32_t* offsets, const double* left_xy,
                           const double* right_xy, double fx, double fy, double cx, double cy, double prob,
                           double threshold, int32_t* inliers, uint8_t* mask, int32_t* iterations);

/* OR over the five-point samples of the context's last sfmhip_score_essential call: bit 0 = two Durand-Kerner
 * iterates coincided bit for bit, bit 1 = the polynomial's leading coefficient was <= DBL_EPSILON -- the two corners
 * of cv::solvePoly whose library behaviour (a cube-root branch; part of a work buffer returned as roots) is not
 * reproduced.  0 = every sample went the documented way. */
int sfmhip_score_last_flags(sfmhip_ctx* ctx);
/* EMEstimatorCallback::runKernel (OpenCV 3.4.1 calib3d/five-point.cpp) for explicit samples: q1 / q2 = n_samples x five
 * normalised points (x, y) each; models: n_samples x 10 row-major 3 x 3 matrices (unit Frobenius norm, the library's
 * order: the order of solvePoly's roots); n_models[i] = count | flags << 8 (flags as sfmhip_score_last_flags).  What the
 * RANSAC above runs per iteration, exposed for sample-level parity checks. */
int sfmhip_score_five_point(sfmhip_ctx* ctx, int n_samples, const double* q1, const double* q2, double* models,
                            int32_t* n_models);

/* The homography side of the same loop: findHomographyInliers (reference src/Sfm.cpp:667-689) =
 *   cv::countNonZero(mask) of cv::findHomography(query_points, train_points, CV_RANSAC, 0.004 * maxVal, mask)
 * as OpenCV 3.4.1 (calib3d/fundam.cpp) runs it: points converted to float, 4-point samples (drawn again when
 * checkSubset rejects them: a collinear triple, or the orientation of a triple not preserved), normalised DLT
 * (smallest eigenvector of L^T L, H[2][2] = 1), the reprojection error and the threshold test in float arithmetic,
 * confidence / max_iters as given (findHomography's defaults: 0.995, 2000); the mask is the RANSAC mask.
 * thresholds: one per pair (the reference: 0.004 * the largest coordinate among the pair's query points; <= 0: 3).
 * Pairs with fewer than 4 matches score 0.  Parity unpinned, like sfmhip_score_essential. */
/* HomographyEstimatorCallback::runKernel (OpenCV 3.4.1 calib3d/fundam.cpp) for explicit samples: M / m = n_samples x four
 * float points (x, y); H: n_samples x 9 doubles (scaled by 1 / H[8], as the library does); ok[i] = 0 for a degenerate sample.  For sample-level parity checks. */
int sfmhip_score_homography_kernel(sfmhip_ctx* ctx, int n_samples, const float* M, const float* m, double* H, int32_t* ok);
int sfmhip_score_homography(sfmhip_ctx* ctx, int n_pairs, const int32_t* offsets, const double* left_xy,
                            const double* right_xy, const double* thresholds, double confidence, int max_iters,
                            int32_t* inliers, uint8_t* mask, int32_t* iterations);

/* ---- adjustBundle solver core (reference src/BundleAdjustment.cpp:46-175) ---- */
typedef struct {
  int max_iterations;           /* 500   src/BundleAdjustment.cpp:118 */
  double max_time_s;            /* 10    src/BundleAdjustment.cpp:120 ; <=0 disables */
  double function_tolerance;    /* 1e-6  Ceres 1.13 defaults from here on */
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
} sfmhip_ba_opts;

enum { SFMHIP_BA_CONVERGENCE = 0, SFMHIP_BA_NO_CONVERGENCE = 1, SFMHIP_BA_FAILURE = 2 };

typedef struct {
  int termination; /* SFMHIP_BA_*; the C++ wrapper writes results back only on CONVERGENCE
                      (src/BundleAdjustment.cpp:126-129) */
  int iterations;
  int successful_steps;
  double initial_cost;
  double final_cost;
  double final_radius;
  double gradient_max_norm;
  double time_s;
  int spin_timeouts; /* reduced solves whose hand-off between fronts timed out (a busy device) and were repeated level by level;
                        0 in every run this build has measured.  They never change the LM trajectory. */
} sfmhip_ba_summary;

void sfmhip_ba_default_opts(sfmhip_ba_opts* o);

/* One-shot: host buffers in, optimised parameters out (in place).  cams6: n_cam x
 * (angle-axis 3, translation 3); pts3: n_pt x 3; one shared focal; one residual block per
 * (obs_cam[o], obs_pt[o], obs_xy[2o..2o+1]) with the principal point already subtracted. */
int sfmhip_ba_solve(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, double* cams6, double* pts3,
                    double* focal, const int32_t* obs_cam, const int32_t* obs_pt,
                    const double* obs_xy, const sfmhip_ba_opts* opts, sfmhip_ba_summary* summary);

/* Where the last sfmhip_ba_solve call on this context spent its time (milliseconds of host wall clock, stage by stage), and
 * whether it reused the problem the call before it had set up.  The one-shot entry point is what
 * BundleAdjustment::adjustBundle (reference include/BundleAdjustment.h:19-20, src/BundleAdjustment.cpp:46-123) maps to: the
 * reference builds its ceres::Problem from the containers on every call, and so does this -- create_ms is that cost --
 * unless the observation structure (n_cam, n_pt, obs_cam[], obs_pt[]) equals the previous call's, in which case the kept
 * problem takes the new measurements and parameters and create_ms is the comparison + the re-upload. */
typedef struct sfmhip_ba_solve_profile {
  double create_ms;     /* sfmhip_ba_create (or, plan_reused: structure comparison + new measurements) */
  double set_params_ms; /* parameters to the device */
  double run_ms;        /* the LM loop (the first call on a structure also plans the reduced system's front tree here) */
  double get_params_ms; /* parameters back */
  double keep_ms;       /* keeping the problem for the next call (a copy of obs_cam / obs_pt) or destroying it */
  double total_ms;
  int plan_reused;
  int front_plan_reused; /* a NEW structure whose camera graph equals the last one's: the front tree was not planned again */
} sfmhip_ba_solve_profile;
int sfmhip_ba_last_solve_profile(sfmhip_ctx* ctx, sfmhip_ba_solve_profile* out);
/* The library's host threads (a pool of up to 16 that lives as long as the process: the set-up's passes run on it) for a
 * caller's own pass over its containers -- fn(lo, hi, user) on disjoint blocks of [0, n), the calling thread taking one of
 * them; returns when all are done.  Below 20 000 items, or when the pool is busy, fn(0, n, user) runs on the caller alone /
 * on threads started for the call.  What BundleAdjustment::adjustBundle's mirror packs and writes back with (the reference
 * walks its std::map tracks on one thread, src/BundleAdjustment.cpp:83-110: 8 ms at cfg4). */
int sfmhip_host_parallel_for(int n, void (*fn)(int lo, int hi, void* user), void* user);

/* Persistent problem object (multi-GPU: every rank holds all cameras + focal and its own
 * block of points with their observations; the per-iteration sum of the reduced camera
 * system goes through `allreduce`, e.g. RCCL via torch.distributed). */
typedef int (*sfmhip_allreduce_fn)(void* device_f64_buffer, size_t count, void* user);

int sfmhip_ba_create(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, const int32_t* obs_cam,
                     const int32_t* obs_pt, const double* obs_xy, sfmhip_ba** out);
/* rank/world of the calling process; world==1 (the default) never calls fn.  The callback sums
 * `count` doubles in place across ranks (ncclAllReduce(sum, ncclDouble) over xGMI) and must be
 * ordered after the work already enqueued on the context's stream. */
int sfmhip_ba_set_allreduce(sfmhip_ba* ba, sfmhip_allreduce_fn fn, void* user, int rank, int world);
int sfmhip_ba_set_params(sfmhip_ba* ba, const double* cams6, const double* pts3, double focal);
int sfmhip_ba_get_params(sfmhip_ba* ba, double* cams6, double* pts3, double* focal);
int sfmhip_ba_run(sfmhip_ba* ba, const sfmhip_ba_opts* opts, sfmhip_ba_summary* summary);
/* `iters` LM iterations (linearise + Schur eliminate + all-reduce + reduced solve +
 * back-substitute + candidate cost, accept/reject as usual) without convergence tests.
 * Resumable: the first call after set_params linearises and computes the Jacobi scaling,
 * later calls continue the same trust-region state. */
int sfmhip_ba_iterate(sfmhip_ba* ba, int iters, sfmhip_ba_summary* summary);
/* One linearisation at the current parameters: reduced system of this rank's points,
 * dim = 6*n_cam+1, S row-major full symmetric, before any all-reduce.  Test hook. */
int sfmhip_ba_reduced_system(sfmhip_ba* ba, double radius, double* S, double* g, double* cost);
/* Residual and Jacobian of n single observations as the solver linearises them (SimpleReprojectionError,
 * reference src/BundleAdjustment.cpp:10-35; analytic derivative of the theta^2 branch autodiff takes).
 * cams6: n x 6, pts3: n x 3, obs_xy: n x 2; r: n x 2, Jc: n x (2x6 row-major), Jp: n x (2x3), Jf: n x 2.
 * Test hook. */
int sfmhip_ba_linearize_obs(sfmhip_ctx* ctx, int n, const double* cams6, const double* pts3, double focal,
                            const double* obs_xy, double* r, double* Jc, double* Jp, double* Jf);
/* Test hook: the solution z of the damped reduced system (S + D/radius) z = g at the current parameters, as the
 * solver's own factorisation (dense or dissected, see sfmhip_ba_reduced_layout) computes it; z: 6*n_cam + 1
 * doubles in the solver's scaled coordinates (the system sfmhip_ba_reduced_system returns); *chol_failed != 0 when
 * a pivot was not positive.  Single rank only. */
int sfmhip_ba_reduced_step(sfmhip_ba* ba, double radius, double* z, int* chol_failed);
/* How the reduced camera system of this problem is factored (decided at the first run / iterate; zeros before):
 * layout[0] = independent interior chains of the dissected camera graph (0: one dense factorisation),
 * layout[1] = 32-column tiles of the longest chain, layout[2] = tiles of the separator (with the focal),
 * layout[3] = tiles of the dense matrix.  The factorisation's dependency chain is layout[1] + layout[2] tiles
 * instead of layout[3].  Replaces nothing in the reference (Eigen's dense LLT, src/BundleAdjustment.cpp:116,
 * has no such choice); SFMHIP_BA_ND=0 in the environment keeps the dense factorisation. */
int sfmhip_ba_reduced_layout(sfmhip_ba* ba, int32_t layout[4]);
/* The front tree, when the camera graph dissects recursively into fronts that fit one compute unit each (the default
 * where it exists; SFMHIP_BA_ND=1 keeps the chains + separator plan, =0 the dense factorisation, =2 tree or dense):
 * tree[0] = fronts (0: no tree), tree[1] = levels, tree[2] = 32-column tile steps on the longest leaf-to-root path (the
 * dependency chain), tree[3] = tiles (own + border) of the largest front.  sfmhip_ba_reduced_layout then reports no
 * chains.  Behind Eigen's LLT, reference src/BundleAdjustment.cpp:116. */
int sfmhip_ba_reduced_tree(sfmhip_ba* ba, int32_t tree[4]);
/* Test hook: ONE trust-region decision (TrustRegionMinimizer + LevenbergMarquardtStrategy of Ceres 1.13 behind reference
 * src/BundleAdjustment.cpp:115-123), taken on the HOST by the very function the device runs at the end of every step evaluation
 * (lm_decide in csrc/ba.hip; compiled without contraction on both sides, so the two agree bit for bit).  Needs no GPU.
 * state: options in, trust-region state in and out; `accepted` is set when this decision took the candidate; `stop` is -1 while
 * the loop runs, an SFMHIP_BA_* termination type once a rule has fired (further calls change nothing), 100 when `solve_info` < 0
 * (a bounded spin ran out: neither an iteration nor a step). */
typedef struct {
  double gradient_tolerance, parameter_tolerance, function_tolerance, min_relative_decrease, max_radius, min_radius;
  int max_consecutive_invalid, max_iterations, timing_only /* sfmhip_ba_iterate: no convergence tests */, pad;
  double radius, decrease_factor, cost, gradient_max_norm, x_norm;
  int iterations, successful_steps, invalid_steps, lin_unread /* the last accepted step's linearisation has not been read */;
  int accepted, stop;
} sfmhip_lm_state;
typedef struct {
  double lin_cost, lin_failed_blocks, lin_gradient_max;                          /* of the linearisation at x */
  double candidate_cost, model_cost_change, step_norm2, candidate_norm2;         /* of the step evaluation */
  int solve_info, pad;                                                           /* > 0: a pivot was not positive; < 0: time-out */
} sfmhip_lm_inputs;
int sfmhip_ba_lm_decide(sfmhip_lm_state* state, const sfmhip_lm_inputs* in);
/* device seconds of the last run/iterate by kernel group:
 * [0]=linearise+eliminate [1]=allreduce [2]=reduced solve [3]=back-substitute+cost */
int sfmhip_ba_last_timing(sfmhip_ba* ba, double seconds[4], int* launches);
void sfmhip_ba_destroy(sfmhip_ba* ba);

#ifdef __cplusplus
}
#endif
#endif
