/*
 * sfm_oracle_match.c -- CPU restatement of the matcher and the two-view triangulation.
 * TEST INFRASTRUCTURE ONLY (see sfm_oracle.h).  PARITY UNPINNED (no reference goldens exist).
 *
 * Follows:  reference src/Sfm.cpp:590-608 (getMatching), :694-711 (AlignedPoints),
 *           :804-878 (triangulateViews)
 * and the published algorithms of the libraries those lines call, at the versions pinned by
 * the reference's CMakeLists.txt:46,58 (OpenCV 3.4.1): core/batch_distance.cpp (knn insertion
 * list), calib3d/undistort.cpp (undistortPoints), calib3d/triangulate.cpp (4x4 DLT),
 * core/lapack.cpp (JacobiSVDImpl_), calib3d/calibration.cpp (projectPoints).
 */
#include "sfm_oracle.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- distances (OpenCV core/stat.cpp normL2Sqr_/normHamming as used by batchDistance) ---- */

/* f32 rows: float accumulation in 8 SIMD-style partial sums, fixed combine order.  OpenCV leaves
 * the lane order unspecified; for integer-valued SIFT rows (every partial < 2^24) any order
 * gives the same exact value. */
static float l2_f32(const float* a, const float* b, int n) {
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int k = 0;
  for (; k + 8 <= n; k += 8)
    for (int l = 0; l < 8; ++l) {
      float d = a[k + l] - b[k + l];
      acc[l] += d * d;
    }
  for (int l = 0; k < n; ++k, ++l) {
    float d = a[k] - b[k];
    acc[l] += d * d;
  }
  float s = ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
  return sqrtf(s);
}

static float l2_u8(const uint8_t* a, const uint8_t* b, int n) {
  int s = 0;
  for (int k = 0; k < n; ++k) {
    int d = (int)a[k] - (int)b[k];
    s += d * d;
  }
  return sqrtf((float)s); /* batchDistL2_8u32f: std::sqrt((float)normL2Sqr) */
}

static int hamming_u8(const uint8_t* a, const uint8_t* b, int n) {
  int s = 0, k = 0;
  for (; k + 8 <= n; k += 8) {
    uint64_t x, y;
    memcpy(&x, a + k, 8);
    memcpy(&y, b + k, 8);
    s += __builtin_popcountll(x ^ y);
  }
  for (; k < n; ++k) s += __builtin_popcount((unsigned)(a[k] ^ b[k]));
  return s;
}

/* K=2 insertion list of cv::batchDistance: strict '<' to enter, strict '>' while bubbling, so
 * equal distances keep the lower train index in front and a candidate equal to the current
 * 2nd-best does not replace it. */
#define KNN2_INSERT(d, j, d0, i0, d1, i1) \
  do {                                    \
    if ((d) < (d1)) {                     \
      if ((d0) > (d)) {                   \
        (d1) = (d0);                      \
        (i1) = (i0);                      \
        (d0) = (d);                       \
        (i0) = (j);                       \
      } else {                            \
        (d1) = (d);                       \
        (i1) = (j);                       \
      }                                   \
    }                                     \
  } while (0)

int orc_match_knn2(const void* q, int nq, const void* t, int nt, int dim, int dtype, int norm,
                   float ratio, int32_t* out_q, int32_t* out_t, float* out_dist, int32_t* out_n,
                   int32_t* knn_idx, float* knn_dist, int threads) {
  if (nq < 0 || nt < 0 || dim <= 0) return -1;
  if (norm == ORC_NORM_HAMMING && dtype != ORC_DTYPE_U8) return -2;
  int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(nq > 0 ? nq : 1));
  float* bd = (float*)malloc(sizeof(float) * 2 * (size_t)(nq > 0 ? nq : 1));
  if (!bi || !bd) return -3;
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int i = 0; i < nq; ++i) {
    int32_t i0 = -1, i1 = -1;
    if (norm == ORC_NORM_HAMMING) {
      int d0 = INT_MAX, d1 = INT_MAX;
      const uint8_t* a = (const uint8_t*)q + (size_t)i * dim;
      for (int j = 0; j < nt; ++j) {
        int d = hamming_u8(a, (const uint8_t*)t + (size_t)j * dim, dim);
        KNN2_INSERT(d, j, d0, i0, d1, i1);
      }
      bd[2 * i] = i0 >= 0 ? (float)d0 : FLT_MAX;
      bd[2 * i + 1] = i1 >= 0 ? (float)d1 : FLT_MAX;
    } else {
      float d0 = FLT_MAX, d1 = FLT_MAX;
      if (dtype == ORC_DTYPE_F32) {
        const float* a = (const float*)q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
          float d = l2_f32(a, (const float*)t + (size_t)j * dim, dim);
          KNN2_INSERT(d, j, d0, i0, d1, i1);
        }
      } else {
        const uint8_t* a = (const uint8_t*)q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
          float d = l2_u8(a, (const uint8_t*)t + (size_t)j * dim, dim);
          KNN2_INSERT(d, j, d0, i0, d1, i1);
        }
      }
      bd[2 * i] = d0;
      bd[2 * i + 1] = d1;
    }
    bi[2 * i] = i0;
    bi[2 * i + 1] = i1;
  }
  /* ratio test, src/Sfm.cpp:603-607; float multiply + float compare */
  int n = 0;
  for (int i = 0; i < nq; ++i) {
    if (bi[2 * i + 1] < 0) continue; /* Nt < 2: reference reads out of bounds; emit nothing */
    if (bd[2 * i] <= ratio * bd[2 * i + 1]) {
      if (out_q) out_q[n] = i;
      if (out_t) out_t[n] = bi[2 * i];
      if (out_dist) out_dist[n] = bd[2 * i];
      ++n;
    }
  }
  if (out_n) *out_n = n;
  if (knn_idx) memcpy(knn_idx, bi, sizeof(int32_t) * 2 * (size_t)nq);
  if (knn_dist) memcpy(knn_dist, bd, sizeof(float) * 2 * (size_t)nq);
  free(bi);
  free(bd);
  return 0;
}

/* Many pairs in one call, parallel over (pair, query row) so that every host core has work:
 * the CPU-baseline leg of bench.py.  counts[p] = matches of pair p (ratio-filtered). */
static int match_many_impl(const void* const* imgs, const int32_t* n_rows, int dim, int dtype, int norm,
                           const int32_t* pairs, int n_pairs, float ratio, int threads, int32_t* counts,
                           uint64_t* checksums);

int orc_match_many(const void* const* imgs, const int32_t* n_rows, int dim, int dtype, int norm,
                   const int32_t* pairs, int n_pairs, float ratio, int threads, int32_t* counts) {
  return match_many_impl(imgs, n_rows, dim, dtype, norm, pairs, n_pairs, ratio, threads, counts, NULL);
}

/* The same with a checksum of every pair's match list: checksums[2p] = sum, [2p+1] = xor over the
 * pair's matches of orc_match_mix(queryIdx, trainIdx, distance bits) -- what full-size parity
 * tests and bench.py compare with the GPU lists (all 1225 pairs of cfg2, not a sample). */
int orc_match_many_checksum(const void* const* imgs, const int32_t* n_rows, int dim, int dtype, int norm,
                            const int32_t* pairs, int n_pairs, float ratio, int threads, int32_t* counts,
                            uint64_t* checksums) {
  return match_many_impl(imgs, n_rows, dim, dtype, norm, pairs, n_pairs, ratio, threads, counts, checksums);
}

uint64_t orc_match_mix(uint32_t q, uint32_t t, uint32_t dist_bits) {
  uint64_t x = (uint64_t)q * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xC2B2AE3D27D4EB4Full +
               (uint64_t)dist_bits * 0x165667B19E3779F9ull;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 32;
  return x;
}

static int match_many_impl(const void* const* imgs, const int32_t* n_rows, int dim, int dtype, int norm,
                           const int32_t* pairs, int n_pairs, float ratio, int threads, int32_t* counts,
                           uint64_t* checksums) {
  if (n_pairs < 0 || dim <= 0) return -1;
  if (threads < 1) threads = 1;
  int64_t* off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_pairs + 1));
  if (!off) return -3;
  off[0] = 0;
  for (int p = 0; p < n_pairs; ++p) off[p + 1] = off[p] + n_rows[pairs[2 * p]];
  const int64_t total = off[n_pairs];
  for (int p = 0; p < n_pairs; ++p) counts[p] = 0;
  if (checksums) memset(checksums, 0, sizeof(uint64_t) * 2 * (size_t)n_pairs);
  const size_t esz = dtype == ORC_DTYPE_F32 ? 4 : 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
  for (int64_t w = 0; w < total; ++w) {
    int lo = 0, hi = n_pairs - 1; /* pair of work item w */
    while (lo < hi) {
      int mid = (lo + hi + 1) / 2;
      if (off[mid] <= w) lo = mid; else hi = mid - 1;
    }
    const int p = lo, i = (int)(w - off[p]);
    const int qi = pairs[2 * p], ti = pairs[2 * p + 1], nt = n_rows[ti];
    const unsigned char* q = (const unsigned char*)imgs[qi] + (size_t)i * dim * esz;
    const unsigned char* t = (const unsigned char*)imgs[ti];
    int32_t i0 = -1, i1 = -1;
    int pass = 0;
    float dist0 = 0.f;
    if (norm == ORC_NORM_HAMMING) {
      int d0 = INT_MAX, d1 = INT_MAX;
      for (int j = 0; j < nt; ++j) {
        int d = hamming_u8(q, t + (size_t)j * dim, dim);
        KNN2_INSERT(d, j, d0, i0, d1, i1);
      }
      pass = i1 >= 0 && (float)d0 <= ratio * (float)d1;
      dist0 = (float)d0;
    } else {
      float d0 = FLT_MAX, d1 = FLT_MAX;
      for (int j = 0; j < nt; ++j) {
        float d = dtype == ORC_DTYPE_F32 ? l2_f32((const float*)q, (const float*)(t + (size_t)j * dim * 4), dim)
                                         : l2_u8(q, t + (size_t)j * dim, dim);
        KNN2_INSERT(d, j, d0, i0, d1, i1);
      }
      pass = i1 >= 0 && d0 <= ratio * d1;
      dist0 = d0;
    }
    if (pass) {
#pragma omp atomic
      counts[p]++;
      if (checksums) {
        uint32_t bits;
        memcpy(&bits, &dist0, 4);
        const uint64_t x = orc_match_mix((uint32_t)i, (uint32_t)i0, bits);
#pragma omp atomic
        checksums[2 * p] += x;
#pragma omp atomic
        checksums[2 * p + 1] ^= x;
      }
    }
  }
  free(off);
  return 0;
}

/* ---- the CPU-baseline matcher of bench.py: the same lists as match_many_impl, organised the way
 * cv::BFMatcher::knnMatch runs on a multi-core host (core/batch_distance.cpp: BatchDistInvoker under parallel_for_):
 * parallel over blocks of QUERY rows; a block's queries meet the train rows tile by tile (a tile of train rows stays
 * in L1/L2 while the block's queries go over it); the sum of squared differences is a SIMD loop with independent
 * accumulators (hal::normL2Sqr_; here four train rows per query at a time); every thread keeps its own counts / checksums, merged once at the end -- no atomics
 * on the data path.  L2 / f32 rows only (cfg2); integer-valued rows make every summation order exact, so the lists and
 * checksums equal those of the row-by-row restatement above (bench.py asserts it).  FMA contraction is allowed here:
 * it is a timing leg, and on integer-valued rows FMAs change nothing. */
#define MB_QB 8    /* query rows per block */
#define MB_TB 64   /* train rows per tile: 64 x 128 floats = 32 KB */
typedef float v16f __attribute__((vector_size(64), aligned(4)));
static inline float hsum16(v16f v) {
  float s = 0.f;
  for (int l = 0; l < 16; ++l) s += v[l];
  return s;
}
/* squared distances of one query row to four train rows: two independent accumulators per train row (eight FMA chains) */
__attribute__((optimize("fp-contract=fast"))) static inline void ssd4_f32(const float* restrict a, const float* restrict b0,
                                                                            const float* restrict b1, const float* restrict b2,
                                                                            const float* restrict b3, int n, float out[4]) {
  v16f s00 = {0}, s01 = {0}, s10 = {0}, s11 = {0}, s20 = {0}, s21 = {0}, s30 = {0}, s31 = {0};
  int k = 0;
  for (; k + 32 <= n; k += 32) {
    const v16f a0 = *(const v16f*)(a + k), a1 = *(const v16f*)(a + k + 16);
    v16f d;
    d = a0 - *(const v16f*)(b0 + k); s00 += d * d;
    d = a1 - *(const v16f*)(b0 + k + 16); s01 += d * d;
    d = a0 - *(const v16f*)(b1 + k); s10 += d * d;
    d = a1 - *(const v16f*)(b1 + k + 16); s11 += d * d;
    d = a0 - *(const v16f*)(b2 + k); s20 += d * d;
    d = a1 - *(const v16f*)(b2 + k + 16); s21 += d * d;
    d = a0 - *(const v16f*)(b3 + k); s30 += d * d;
    d = a1 - *(const v16f*)(b3 + k + 16); s31 += d * d;
  }
  out[0] = hsum16(s00 + s01);
  out[1] = hsum16(s10 + s11);
  out[2] = hsum16(s20 + s21);
  out[3] = hsum16(s30 + s31);
  for (; k < n; ++k) {
    const float d0 = a[k] - b0[k], d1 = a[k] - b1[k], d2 = a[k] - b2[k], d3 = a[k] - b3[k];
    out[0] += d0 * d0; out[1] += d1 * d1; out[2] += d2 * d2; out[3] += d3 * d3;
  }
}

int orc_match_many_blocked(const void* const* imgs, const int32_t* n_rows, int dim, const int32_t* pairs, int n_pairs,
                           float ratio, int threads, int32_t* counts, uint64_t* checksums) {
  if (n_pairs < 0 || dim <= 0) return -1;
  if (threads < 1) threads = 1;
  int64_t* off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_pairs + 1)); /* blocks of query rows before pair p */
  int32_t* cnt_t = (int32_t*)calloc((size_t)threads * (size_t)(n_pairs > 0 ? n_pairs : 1), sizeof(int32_t));
  uint64_t* cs_t = (uint64_t*)calloc((size_t)threads * 2 * (size_t)(n_pairs > 0 ? n_pairs : 1), sizeof(uint64_t));
  if (!off || !cnt_t || !cs_t) return -3;
  off[0] = 0;
  for (int p = 0; p < n_pairs; ++p) off[p + 1] = off[p] + (n_rows[pairs[2 * p]] + MB_QB - 1) / MB_QB;
  const int64_t total = off[n_pairs];
#pragma omp parallel num_threads(threads)
  {
    int tid = 0;
#ifdef _OPENMP
    extern int omp_get_thread_num(void);
    tid = omp_get_thread_num();
#endif
    int32_t* my_cnt = cnt_t + (size_t)tid * n_pairs;
    uint64_t* my_cs = cs_t + (size_t)tid * 2 * n_pairs;
#pragma omp for schedule(dynamic, 4)
    for (int64_t w = 0; w < total; ++w) {
      int lo = 0, hi = n_pairs - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (off[mid] <= w) lo = mid; else hi = mid - 1;
      }
      const int p = lo, qi = pairs[2 * p], ti = pairs[2 * p + 1], nq = n_rows[qi], nt = n_rows[ti];
      const int q0 = (int)(w - off[p]) * MB_QB, q1 = q0 + MB_QB < nq ? q0 + MB_QB : nq;
      const float* Q = (const float*)imgs[qi];
      const float* T = (const float*)imgs[ti];
      float d0[MB_QB], d1[MB_QB];
      int32_t i0[MB_QB], i1[MB_QB];
      for (int r = 0; r < MB_QB; ++r) d0[r] = d1[r] = FLT_MAX, i0[r] = i1[r] = -1;
      for (int t0 = 0; t0 < nt; t0 += MB_TB) {
        const int t1 = t0 + MB_TB < nt ? t0 + MB_TB : nt;
        for (int i = q0; i < q1; ++i) {
          const float* a = Q + (size_t)i * dim;
          const int r = i - q0;
          float buf[MB_TB + 4];
          for (int j = t0; j < t1; j += 4) { /* (the last group may repeat the tile's last row: its result is not read) */
            const int j1 = j + 1 < t1 ? j + 1 : t1 - 1, j2 = j + 2 < t1 ? j + 2 : t1 - 1, j3 = j + 3 < t1 ? j + 3 : t1 - 1;
            ssd4_f32(a, T + (size_t)j * dim, T + (size_t)j1 * dim, T + (size_t)j2 * dim, T + (size_t)j3 * dim, dim, buf + (j - t0));
          }
          for (int j = t0; j < t1; ++j) { /* batchDistL2_32f: sqrt of every entry, then the K-list insertion in j order */
            const float d = sqrtf(buf[j - t0]);
            KNN2_INSERT(d, j, d0[r], i0[r], d1[r], i1[r]);
          }
        }
      }
      for (int i = q0; i < q1; ++i) {
        const int r = i - q0;
        if (i1[r] >= 0 && d0[r] <= ratio * d1[r]) {
          my_cnt[p]++;
          uint32_t bits;
          memcpy(&bits, &d0[r], 4);
          const uint64_t x = orc_match_mix((uint32_t)i, (uint32_t)i0[r], bits);
          my_cs[2 * p] += x;
          my_cs[2 * p + 1] ^= x;
        }
      }
    }
  }
  for (int p = 0; p < n_pairs; ++p) {
    int32_t c = 0;
    uint64_t a = 0, x = 0;
    for (int t = 0; t < threads; ++t) {
      c += cnt_t[(size_t)t * n_pairs + p];
      a += cs_t[((size_t)t * n_pairs + p) * 2];
      x ^= cs_t[((size_t)t * n_pairs + p) * 2 + 1];
    }
    counts[p] = c;
    if (checksums) checksums[2 * p] = a, checksums[2 * p + 1] = x;
  }
  free(off);
  free(cnt_t);
  free(cs_t);
  return 0;
}

/* ---- triangulation ---- */

/* OpenCV core/lapack.cpp JacobiSVDImpl_<double> on At (n rows of length m), here m=n=4, giving
 * Vt sorted by descending singular value; the DLT solution is Vt row 3. */
static void jacobi_svd4_vt(double At[4][4], double W[4], double Vt[4][4]) {
  const int n = 4, m = 4;
  const double eps = DBL_EPSILON * 10;
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) sd += At[i][k] * At[i][k];
    W[i] = sd;
    for (int k = 0; k < n; ++k) Vt[i][k] = (i == k) ? 1.0 : 0.0;
  }
  const int max_iter = 30;
  for (int iter = 0; iter < max_iter; ++iter) {
    int changed = 0;
    for (int i = 0; i < n - 1; ++i)
      for (int j = i + 1; j < n; ++j) {
        double* Ai = At[i];
        double* Aj = At[j];
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; ++k) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = hypot(p, beta), c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; ++k) {
          double t0 = c * Ai[k] + s * Aj[k];
          double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = 1;
        double* Vi = Vt[i];
        double* Vj = Vt[j];
        for (int k = 0; k < n; ++k) {
          double t0 = c * Vi[k] + s * Vj[k];
          double t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0;
          Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) sd += At[i][k] * At[i][k];
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < n - 1; ++i) {
    int j = i;
    for (int k = i + 1; k < n; ++k)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double tw = W[i];
      W[i] = W[j];
      W[j] = tw;
      for (int k = 0; k < m; ++k) {
        double ta = At[i][k];
        At[i][k] = At[j][k];
        At[j][k] = ta;
      }
      for (int k = 0; k < n; ++k) {
        double tv = Vt[i][k];
        Vt[i][k] = Vt[j][k];
        Vt[j][k] = tv;
      }
    }
  }
}

/* calib3d undistortPoints without R/P: x=(u-cx)*(1/fx), then 5 fixed-point iterations
 * (k1,k2,p1,p2,k3) -- exact no-op when every coefficient is zero (reference calibration,
 * data/temple/camera_calibration_template.xml:17-19). */
static void undistort_point(const double K[9], const double d[5], double u, double v, double* xo,
                            double* yo) {
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const double ifx = 1. / fx, ify = 1. / fy;
  double x = (u - cx) * ifx, y = (v - cy) * ify;
  const double x0 = x, y0 = y;
  const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
  for (int j = 0; j < 5; ++j) {
    double r2 = x * x + y * y;
    double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
    double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  *xo = x;
  *yo = y;
}

/* calib3d projectPoints with R (the R->rvec->R round trip of src/Sfm.cpp:836,843 is dropped:
 * it perturbs R by ~1e-16), t, K, dist. */
static void project_point(const double P[12], const double K[9], const double d[5],
                          const double X[3], double* u, double* v) {
  double x = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
  double y = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
  double z = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
  z = z ? 1. / z : 1;
  x *= z;
  y *= z;
  const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
  double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
  double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
  double xd = x * cdist + p1 * a1 + p2 * a2;
  double yd = y * cdist + p1 * a3 + p2 * a1;
  *u = xd * K[0] + K[2];
  *v = yd * K[4] + K[5];
}

int orc_triangulate(const double P1[12], const double P2[12], const double K[9],
                    const double dist[5], const double* xy1, const double* xy2, int m,
                    float max_err, double* X, float* err, uint8_t* keep) {
  if (m < 0) return -1;
  for (int i = 0; i < m; ++i) {
    double x1, y1, x2, y2;
    undistort_point(K, dist, xy1[2 * i], xy1[2 * i + 1], &x1, &y1); /* src/Sfm.cpp:820 */
    undistort_point(K, dist, xy2[2 * i], xy2[2 * i + 1], &x2, &y2); /* src/Sfm.cpp:821 */
    /* cv::triangulatePoints, src/Sfm.cpp:826 : A is 4x4, SVD works on At */
    double A[4][4], At[4][4], W[4], Vt[4][4];
    for (int k = 0; k < 4; ++k) {
      A[0][k] = x1 * P1[8 + k] - P1[0 + k];
      A[1][k] = y1 * P1[8 + k] - P1[4 + k];
      A[2][k] = x2 * P2[8 + k] - P2[0 + k];
      A[3][k] = y2 * P2[8 + k] - P2[4 + k];
    }
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) At[c][r] = A[r][c];
    jacobi_svd4_vt(At, W, Vt);
    /* convertPointsFromHomogeneous, src/Sfm.cpp:833 */
    double w = Vt[3][3];
    double scale = w != 0 ? 1. / w : 1.;
    double Xi[3] = {Vt[3][0] * scale, Vt[3][1] * scale, Vt[3][2] * scale};
    X[3 * i] = Xi[0];
    X[3 * i + 1] = Xi[1];
    X[3 * i + 2] = Xi[2];
    double u1, v1, u2, v2;
    project_point(P1, K, dist, Xi, &u1, &v1); /* src/Sfm.cpp:840 */
    project_point(P2, K, dist, Xi, &u2, &v2); /* src/Sfm.cpp:847 */
    double dx1 = u1 - xy1[2 * i], dy1 = v1 - xy1[2 * i + 1];
    double dx2 = u2 - xy2[2 * i], dy2 = v2 - xy2[2 * i + 1];
    const float e1 = (float)sqrt(dx1 * dx1 + dy1 * dy1); /* src/Sfm.cpp:856 */
    const float e2 = (float)sqrt(dx2 * dx2 + dy2 * dy2); /* src/Sfm.cpp:857 */
    if (err) {
      err[2 * i] = e1;
      err[2 * i + 1] = e2;
    }
    keep[i] = !(max_err < e1 || max_err < e2); /* src/Sfm.cpp:859-860 */
  }
  return 0;
}
