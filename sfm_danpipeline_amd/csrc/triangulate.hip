// triangulate.hip -- two-view DLT triangulation + reprojection filter on gfx950.
//
// Replaces the numeric body of StructFromMotion::triangulateViews (reference
// src/Sfm.cpp:812-860): cv::undistortPoints -> cv::triangulatePoints (4x4 homogeneous DLT,
// smallest right singular vector by one-sided Jacobi) -> convertPointsFromHomogeneous ->
// cv::projectPoints in both views -> float 6 px test.  One lane per match, f64 throughout,
// contraction off so the arithmetic is operation-for-operation the restated OpenCV sequence.
// HBM-bound by construction (2 x 16 B in, 24 B + 1 B (+8 B) out per match); the track /
// visibility bookkeeping (Point3D::idxImage, src/Sfm.cpp:862-873) is the host mirror's job.
#include "common.h"
#include "hypot_glibc.h"
#include <float.h>

namespace {

struct TriParams {
  double P1[12], P2[12], K[9], dist[5];
  float max_err;
};

__device__ __forceinline__ void undistort_point(const TriParams& p, double u, double v, double& xo, double& yo) {
  const double ifx = 1. / p.K[0], ify = 1. / p.K[4];
  double x = (u - p.K[2]) * ifx, y = (v - p.K[5]) * ify;
  const double x0 = x, y0 = y;
  const double k1 = p.dist[0], k2 = p.dist[1], p1 = p.dist[2], p2 = p.dist[3], k3 = p.dist[4];
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
    const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  xo = x;
  yo = y;
}

__device__ __forceinline__ void project_point(const double* P, const TriParams& p, const double X[3], double& u,
                                              double& v) {
  double x = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
  double y = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
  double z = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
  z = z ? 1. / z : 1;
  x *= z;
  y *= z;
  const double k1 = p.dist[0], k2 = p.dist[1], p1 = p.dist[2], p2 = p.dist[3], k3 = p.dist[4];
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
  const double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
  const double xd = x * cdist + p1 * a1 + p2 * a2;
  const double yd = y * cdist + p1 * a3 + p2 * a1;
  u = xd * p.K[0] + p.K[2];
  v = yd * p.K[4] + p.K[5];
}

// One-sided Jacobi on At (rows = columns of A), as OpenCV's JacobiSVDImpl_<double> runs it for a
// 4x4: rotations until every row pair is orthogonal to 10*eps, singular values = row norms,
// selection sort descending; returns Vt row 3.  All indices are compile-time so the 32 doubles
// stay in registers.
__device__ __forceinline__ void dlt_null_vector(double At[4][4], double out[4]) {
  const double eps = DBL_EPSILON * 10;
  double W[4], Vt[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) sd += At[i][k] * At[i][k];
    W[i] = sd;
#pragma unroll
    for (int k = 0; k < 4; ++k) Vt[i][k] = (i == k) ? 1.0 : 0.0;
  }
#pragma unroll 1
  for (int iter = 0; iter < 30; ++iter) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i + 1; j < 4; ++j) {
        double a = W[i], p = 0, b = W[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) p += At[i][k] * At[j][k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = sfm_hypot(p, beta);  // (the host libm's hypot, bit for bit: hypot_glibc.h)
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double t0 = c * At[i][k] + s * At[j][k];
          const double t1 = -s * At[i][k] + c * At[j][k];
          At[i][k] = t0;
          At[j][k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double t0 = c * Vt[i][k] + s * Vt[j][k];
          const double t1 = -s * Vt[i][k] + c * Vt[j][k];
          Vt[i][k] = t0;
          Vt[j][k] = t1;
        }
      }
    if (!__any(changed)) break;  // wave-uniform exit; converged lanes see no further rotation
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) sd += At[i][k] * At[i][k];
    W[i] = sqrt(sd);
  }
  // row of the smallest singular value under OpenCV's descending selection sort = the LAST
  // position; among equal values the sort keeps the earlier row earlier, so take the last
  // index attaining the minimum... except that selection sort swaps can reorder equal values;
  // replay the sort on (W, row id) to land on exactly the row OpenCV leaves in position 3.
  int id[4] = {0, 1, 2, 3};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int j = i;
#pragma unroll
    for (int k = i + 1; k < 4; ++k)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      const double tw = W[i];
      W[i] = W[j];
      W[j] = tw;
      const int ti = id[i];
      id[i] = id[j];
      id[j] = ti;
    }
  }
  const int sel = id[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = sel == 0 ? Vt[0][k] : sel == 1 ? Vt[1][k] : sel == 2 ? Vt[2][k] : Vt[3][k];
}

__global__ __launch_bounds__(256) void triangulate_kernel(TriParams p, const double2* __restrict__ xy1,
                                                          const double2* __restrict__ xy2, int m,
                                                          double* __restrict__ X, float* __restrict__ err,
                                                          unsigned char* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < m;
  const double2 a = live ? xy1[i] : make_double2(0, 0);
  const double2 b = live ? xy2[i] : make_double2(0, 0);
  double x1, y1, x2, y2;
  undistort_point(p, a.x, a.y, x1, y1);
  undistort_point(p, b.x, b.y, x2, y2);
  double At[4][4];  // At[c][r] = A[r][c]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    At[k][0] = x1 * p.P1[8 + k] - p.P1[0 + k];
    At[k][1] = y1 * p.P1[8 + k] - p.P1[4 + k];
    At[k][2] = x2 * p.P2[8 + k] - p.P2[0 + k];
    At[k][3] = y2 * p.P2[8 + k] - p.P2[4 + k];
  }
  double v[4];
  dlt_null_vector(At, v);
  const double scale = v[3] != 0 ? 1. / v[3] : 1.;
  const double Xi[3] = {v[0] * scale, v[1] * scale, v[2] * scale};
  double u1, v1, u2, v2;
  project_point(p.P1, p, Xi, u1, v1);
  project_point(p.P2, p, Xi, u2, v2);
  const double dx1 = u1 - a.x, dy1 = v1 - a.y, dx2 = u2 - b.x, dy2 = v2 - b.y;
  const float e1 = (float)sqrt(dx1 * dx1 + dy1 * dy1);
  const float e2 = (float)sqrt(dx2 * dx2 + dy2 * dy2);
  if (live) {
    X[3 * i] = Xi[0];
    X[3 * i + 1] = Xi[1];
    X[3 * i + 2] = Xi[2];
    if (err) {
      err[2 * i] = e1;
      err[2 * i + 1] = e2;
    }
    keep[i] = !(p.max_err < e1 || p.max_err < e2);
  }
}

}  // namespace

extern "C" int sfmhip_triangulate(sfmhip_ctx* ctx, const double P1[12], const double P2[12], const double K[9],
                                  const double dist[5], const double* xy1, const double* xy2, int m, float max_err,
                                  double* X, float* err, uint8_t* keep) {
  if (!ctx || !P1 || !P2 || !K || !dist || m < 0) return SFMHIP_ERR_ARG;
  if (m == 0) return SFMHIP_OK;
  if (!xy1 || !xy2 || !X || !keep) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  TriParams p;
  for (int i = 0; i < 12; ++i) {
    p.P1[i] = P1[i];
    p.P2[i] = P2[i];
  }
  for (int i = 0; i < 9; ++i) p.K[i] = K[i];
  for (int i = 0; i < 5; ++i) p.dist[i] = dist[i];
  p.max_err = max_err;
  const size_t n = (size_t)m;
  // one device slab: xy1 | xy2 | X | err | keep
  const size_t off_xy2 = n * 16, off_X = off_xy2 + n * 16, off_err = off_X + n * 24, off_keep = off_err + n * 8;
  const size_t bytes = off_keep + n;
  unsigned char* d = nullptr;
  SFM_HIP_TRY(hipMalloc((void**)&d, bytes));
  hipStream_t st = ctx->stream;
  int rc = SFMHIP_OK;
  auto fail = [&](hipError_t e) {
    if (e != hipSuccess) {
      g_sfmhip_last_hip_error = (int)e;
      rc = SFMHIP_ERR_HIP;
    }
    return e != hipSuccess;
  };
  do {
    if (fail(hipMemcpyAsync(d, xy1, n * 16, hipMemcpyHostToDevice, st))) break;
    if (fail(hipMemcpyAsync(d + off_xy2, xy2, n * 16, hipMemcpyHostToDevice, st))) break;
    hipLaunchKernelGGL(triangulate_kernel, dim3((m + 255) / 256), dim3(256), 0, st, p, (const double2*)d,
                       (const double2*)(d + off_xy2), m, (double*)(d + off_X), (float*)(d + off_err), d + off_keep);
    if (fail(hipGetLastError())) break;
    if (fail(hipMemcpyAsync(X, d + off_X, n * 24, hipMemcpyDeviceToHost, st))) break;
    if (err && fail(hipMemcpyAsync(err, d + off_err, n * 8, hipMemcpyDeviceToHost, st))) break;
    if (fail(hipMemcpyAsync(keep, d + off_keep, n, hipMemcpyDeviceToHost, st))) break;
    if (fail(hipStreamSynchronize(st))) break;
  } while (0);
  hipFree(d);
  return rc;
}
