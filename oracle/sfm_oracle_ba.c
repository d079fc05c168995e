/*
 * sfm_oracle_ba.c -- CPU restatement of the bundle adjustment the reference configures.
 * TEST INFRASTRUCTURE ONLY (see sfm_oracle.h).  PARITY UNPINNED (no reference goldens exist).
 *
 * Follows:  reference src/BundleAdjustment.cpp:5-44 (SimpleReprojectionError),
 *           :46-175 (adjustBundle: parameterisation, solver options, write-back policy)
 * and the published algorithms of Ceres 1.13.0 (pinned by the reference's CMakeLists.txt:58;
 * not vendored): ceres/rotation.h (AngleAxisRotatePoint, RotationMatrixToAngleAxis,
 * AngleAxisToRotationMatrix), trust_region_minimizer.cc (loop, tolerances),
 * levenberg_marquardt_strategy.cc (diagonal clamp, radius update), schur_eliminator_impl.h
 * (point blocks eliminated, cameras + the shared focal kept), dense Cholesky of the reduced
 * system (Eigen LLT inside DENSE_SCHUR).
 */
#include "sfm_oracle.h"
#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* Threads of the evaluation / elimination / back-substitution passes (OpenMP over points), for the "all cores"
 * leg of the CPU baseline: what Ceres does with num_threads > 1.  The dense Cholesky stays on one thread (Eigen
 * LLT in Ceres 1.13).  1 (the default, Ceres' default: src/BundleAdjustment.cpp:115-121 sets no thread count)
 * runs the loops serially in their written order: the parity tests use that. */
static int g_threads = 1;
void orc_ba_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------- ceres/rotation.h ---------------- */

void orc_angleaxis_rotate_point(const double aa[3], const double X[3], double out[3]) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
    const double w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
    const double wx[3] = {w[1] * X[2] - w[2] * X[1], w[2] * X[0] - w[0] * X[2],
                          w[0] * X[1] - w[1] * X[0]};
    const double tmp = (w[0] * X[0] + w[1] * X[1] + w[2] * X[2]) * (1.0 - c);
    for (int i = 0; i < 3; ++i) out[i] = X[i] * c + wx[i] * s + w[i] * tmp;
  } else {
    const double wx[3] = {aa[1] * X[2] - aa[2] * X[1], aa[2] * X[0] - aa[0] * X[2],
                          aa[0] * X[1] - aa[1] * X[0]};
    for (int i = 0; i < 3; ++i) out[i] = X[i] + wx[i];
  }
}

#define RM(i, j) R[(i) + 3 * (j)] /* column-major */
void orc_rotmat_colmajor_to_angleaxis(const double R[9], double aa[3]) {
  double q[4];
  const double trace = RM(0, 0) + RM(1, 1) + RM(2, 2);
  if (trace >= 0.0) {
    double t = sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (RM(2, 1) - RM(1, 2)) * t;
    q[2] = (RM(0, 2) - RM(2, 0)) * t;
    q[3] = (RM(1, 0) - RM(0, 1)) * t;
  } else {
    int i = 0;
    if (RM(1, 1) > RM(0, 0)) i = 1;
    if (RM(2, 2) > RM(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(RM(i, i) - RM(j, j) - RM(k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (RM(k, j) - RM(j, k)) * t;
    q[j + 1] = (RM(j, i) + RM(i, j)) * t;
    q[k + 1] = (RM(k, i) + RM(i, k)) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double st = sqrt(s2), ct = q[0];
    const double two_theta = 2.0 * ((ct < 0.0) ? atan2(-st, -ct) : atan2(st, ct));
    const double k = two_theta / st;
    aa[0] = q[1] * k;
    aa[1] = q[2] * k;
    aa[2] = q[3] * k;
  } else {
    aa[0] = q[1] * 2.0;
    aa[1] = q[2] * 2.0;
    aa[2] = q[3] * 2.0;
  }
}

void orc_angleaxis_to_rotmat_colmajor(const double aa[3], double R[9]) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = cos(theta), s = sin(theta);
    RM(0, 0) = c + wx * wx * (1.0 - c);
    RM(1, 0) = wz * s + wx * wy * (1.0 - c);
    RM(2, 0) = -wy * s + wx * wz * (1.0 - c);
    RM(0, 1) = wx * wy * (1.0 - c) - wz * s;
    RM(1, 1) = c + wy * wy * (1.0 - c);
    RM(2, 1) = wx * s + wy * wz * (1.0 - c);
    RM(0, 2) = wy * s + wx * wz * (1.0 - c);
    RM(1, 2) = -wx * s + wy * wz * (1.0 - c);
    RM(2, 2) = c + wz * wz * (1.0 - c);
  } else {
    RM(0, 0) = 1.0;
    RM(1, 0) = aa[2];
    RM(2, 0) = -aa[1];
    RM(0, 1) = -aa[2];
    RM(1, 1) = 1.0;
    RM(2, 1) = aa[0];
    RM(0, 2) = aa[1];
    RM(1, 2) = -aa[0];
    RM(2, 2) = 1.0;
  }
}
#undef RM

/* ---------------- residual + analytic Jacobian ---------------- */
/* Differentiates exactly the expression Ceres' autodiff sees (same theta^2 branch). */
void orc_ba_residual(const double cam[6], const double X[3], double focal, const double obs[2],
                     double r[2], double* Jc, double* Jp, double* Jf) {
  const double* aa = cam;
  double p[3], Rm[3][3] = {{0}}, dpdw[3][3] = {{0}};
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  const int need_j = (Jc || Jp);
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
    const double w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
    const double wx[3] = {w[1] * X[2] - w[2] * X[1], w[2] * X[0] - w[0] * X[2],
                          w[0] * X[1] - w[1] * X[0]};
    const double wd = w[0] * X[0] + w[1] * X[1] + w[2] * X[2];
    const double tmp = wd * (1.0 - c);
    for (int i = 0; i < 3; ++i) p[i] = X[i] * c + wx[i] * s + w[i] * tmp;
    if (need_j) {
      /* dp/dX = c I + s [w]x + (1-c) w w^T */
      const double oc = 1.0 - c;
      Rm[0][0] = c + oc * w[0] * w[0];
      Rm[0][1] = -s * w[2] + oc * w[0] * w[1];
      Rm[0][2] = s * w[1] + oc * w[0] * w[2];
      Rm[1][0] = s * w[2] + oc * w[1] * w[0];
      Rm[1][1] = c + oc * w[1] * w[1];
      Rm[1][2] = -s * w[0] + oc * w[1] * w[2];
      Rm[2][0] = -s * w[1] + oc * w[2] * w[0];
      Rm[2][1] = s * w[0] + oc * w[2] * w[1];
      Rm[2][2] = c + oc * w[2] * w[2];
      for (int j = 0; j < 3; ++j) {
        double dw[3];
        for (int i = 0; i < 3; ++i) dw[i] = ((i == j ? 1.0 : 0.0) - w[i] * w[j]) * ti;
        const double dwx[3] = {dw[1] * X[2] - dw[2] * X[1], dw[2] * X[0] - dw[0] * X[2],
                               dw[0] * X[1] - dw[1] * X[0]};
        const double dwd = dw[0] * X[0] + dw[1] * X[1] + dw[2] * X[2];
        const double dtmp = dwd * oc + wd * s * w[j];
        for (int i = 0; i < 3; ++i)
          dpdw[i][j] = -X[i] * s * w[j] + dwx[i] * s + wx[i] * c * w[j] + dw[i] * tmp + w[i] * dtmp;
      }
    }
  } else {
    const double wx[3] = {aa[1] * X[2] - aa[2] * X[1], aa[2] * X[0] - aa[0] * X[2],
                          aa[0] * X[1] - aa[1] * X[0]};
    for (int i = 0; i < 3; ++i) p[i] = X[i] + wx[i];
    if (need_j) {
      /* p = X + aa x X : dp/dX = I + [aa]x ; dp/daa = -[X]x */
      Rm[0][0] = 1;
      Rm[0][1] = -aa[2];
      Rm[0][2] = aa[1];
      Rm[1][0] = aa[2];
      Rm[1][1] = 1;
      Rm[1][2] = -aa[0];
      Rm[2][0] = -aa[1];
      Rm[2][1] = aa[0];
      Rm[2][2] = 1;
      dpdw[0][0] = 0;
      dpdw[0][1] = X[2];
      dpdw[0][2] = -X[1];
      dpdw[1][0] = -X[2];
      dpdw[1][1] = 0;
      dpdw[1][2] = X[0];
      dpdw[2][0] = X[1];
      dpdw[2][1] = -X[0];
      dpdw[2][2] = 0;
    }
  }
  p[0] += cam[3];
  p[1] += cam[4];
  p[2] += cam[5];
  const double xp = p[0] / p[2], yp = p[1] / p[2];
  r[0] = focal * xp - obs[0];
  r[1] = focal * yp - obs[1];
  if (Jf) {
    Jf[0] = xp;
    Jf[1] = yp;
  }
  if (need_j) {
    const double iz = 1.0 / p[2];
    const double drdp[2][3] = {{focal * iz, 0.0, -focal * xp * iz},
                               {0.0, focal * iz, -focal * yp * iz}};
    for (int a = 0; a < 2; ++a) {
      if (Jc) {
        for (int j = 0; j < 3; ++j)
          Jc[a * 6 + j] =
              drdp[a][0] * dpdw[0][j] + drdp[a][1] * dpdw[1][j] + drdp[a][2] * dpdw[2][j];
        for (int j = 0; j < 3; ++j) Jc[a * 6 + 3 + j] = drdp[a][j];
      }
      if (Jp)
        for (int j = 0; j < 3; ++j)
          Jp[a * 3 + j] = drdp[a][0] * Rm[0][j] + drdp[a][1] * Rm[1][j] + drdp[a][2] * Rm[2][j];
    }
  }
}

/* ---------------- problem container ---------------- */

typedef struct {
  int nc, np, no, dim;
  int* pt_ptr;   /* np+1 : observations grouped by point, input order kept inside a point */
  int* o_cam;    /* no */
  double* o_xy;  /* 2*no */
  char* cam_used;
  char* pt_used;
  /* state */
  double *cams, *pts, focal;         /* current x */
  double *cams_c, *pts_c, focal_c;   /* candidate */
  double *scale_c, *scale_p, scale_f; /* Jacobi scaling */
  double *r, *Jc, *Jp, *Jf;          /* per observation, scaled J */
  double *grad_c, *grad_p, grad_f;   /* unscaled gradient J^T r */
  double *diag_c, *diag_p, diag_f;   /* LM diagonal source (clamped column sq-norms) */
  double *S, *g, *z;                 /* reduced system */
  double *step_c, *step_p, step_f;   /* scaled-space step */
} ba_t;

static void* xcalloc(size_t n, size_t s) {
  void* p = calloc(n ? n : 1, s);
  if (!p) {
    fprintf(stderr, "oracle: out of memory\n");
    abort();
  }
  return p;
}

static ba_t* ba_new(int nc, int np, int no, const double* cams6, const double* pts3, double focal,
                    const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy) {
  ba_t* b = (ba_t*)xcalloc(1, sizeof(ba_t));
  b->nc = nc;
  b->np = np;
  b->no = no;
  b->dim = 6 * nc + 1;
  b->pt_ptr = (int*)xcalloc((size_t)np + 1, sizeof(int));
  b->o_cam = (int*)xcalloc((size_t)no, sizeof(int));
  b->o_xy = (double*)xcalloc((size_t)2 * no, sizeof(double));
  b->cam_used = (char*)xcalloc((size_t)nc, 1);
  b->pt_used = (char*)xcalloc((size_t)np, 1);
  for (int o = 0; o < no; ++o) b->pt_ptr[obs_pt[o] + 1]++;
  for (int p = 0; p < np; ++p) b->pt_ptr[p + 1] += b->pt_ptr[p];
  int* fill = (int*)xcalloc((size_t)np, sizeof(int));
  for (int o = 0; o < no; ++o) {
    int p = obs_pt[o], k = b->pt_ptr[p] + fill[p]++;
    b->o_cam[k] = obs_cam[o];
    b->o_xy[2 * k] = obs_xy[2 * o];
    b->o_xy[2 * k + 1] = obs_xy[2 * o + 1];
    b->cam_used[obs_cam[o]] = 1;
    b->pt_used[p] = 1;
  }
  free(fill);
  b->cams = (double*)xcalloc((size_t)6 * nc, sizeof(double));
  b->pts = (double*)xcalloc((size_t)3 * np, sizeof(double));
  b->cams_c = (double*)xcalloc((size_t)6 * nc, sizeof(double));
  b->pts_c = (double*)xcalloc((size_t)3 * np, sizeof(double));
  memcpy(b->cams, cams6, sizeof(double) * 6 * (size_t)nc);
  memcpy(b->pts, pts3, sizeof(double) * 3 * (size_t)np);
  b->focal = focal;
  b->scale_c = (double*)xcalloc((size_t)6 * nc, sizeof(double));
  b->scale_p = (double*)xcalloc((size_t)3 * np, sizeof(double));
  for (int i = 0; i < 6 * nc; ++i) b->scale_c[i] = 1.0;
  for (int i = 0; i < 3 * np; ++i) b->scale_p[i] = 1.0;
  b->scale_f = 1.0;
  b->r = (double*)xcalloc((size_t)2 * no, sizeof(double));
  b->Jc = (double*)xcalloc((size_t)12 * no, sizeof(double));
  b->Jp = (double*)xcalloc((size_t)6 * no, sizeof(double));
  b->Jf = (double*)xcalloc((size_t)2 * no, sizeof(double));
  b->grad_c = (double*)xcalloc((size_t)6 * nc, sizeof(double));
  b->grad_p = (double*)xcalloc((size_t)3 * np, sizeof(double));
  b->diag_c = (double*)xcalloc((size_t)6 * nc, sizeof(double));
  b->diag_p = (double*)xcalloc((size_t)3 * np, sizeof(double));
  b->S = (double*)xcalloc((size_t)b->dim * b->dim, sizeof(double));
  b->g = (double*)xcalloc((size_t)b->dim, sizeof(double));
  b->z = (double*)xcalloc((size_t)b->dim, sizeof(double));
  b->step_c = (double*)xcalloc((size_t)6 * nc, sizeof(double));
  b->step_p = (double*)xcalloc((size_t)3 * np, sizeof(double));
  return b;
}

static void ba_free(ba_t* b) {
  free(b->pt_ptr); free(b->o_cam); free(b->o_xy); free(b->cam_used); free(b->pt_used);
  free(b->cams); free(b->pts); free(b->cams_c); free(b->pts_c);
  free(b->scale_c); free(b->scale_p);
  free(b->r); free(b->Jc); free(b->Jp); free(b->Jf);
  free(b->grad_c); free(b->grad_p); free(b->diag_c); free(b->diag_p);
  free(b->S); free(b->g); free(b->z); free(b->step_c); free(b->step_p);
  free(b);
}

/* cost only, at (cams, pts, focal) */
static double ba_cost(const ba_t* b, const double* cams, const double* pts, double focal) {
  double cost = 0;
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads) reduction(+ : cost) schedule(static)
  for (int p = 0; p < b->np; ++p)
    for (int k = b->pt_ptr[p]; k < b->pt_ptr[p + 1]; ++k) {
      double r[2];
      orc_ba_residual(cams + 6 * b->o_cam[k], pts + 3 * p, focal, b->o_xy + 2 * k, r, 0, 0, 0);
      cost += r[0] * r[0] + r[1] * r[1];
    }
  return 0.5 * cost;
}

/* Evaluator::Evaluate at x: residuals, UNSCALED gradient, Jacobian (then column-scaled).
 * first!=0: estimate the Jacobi scale 1/(1+||col||) from this Jacobian (iteration 0). */
static double ba_linearize(ba_t* b, int first, int jacobi_scaling) {
  double cost = 0;
  memset(b->grad_c, 0, sizeof(double) * 6 * (size_t)b->nc);
  memset(b->grad_p, 0, sizeof(double) * 3 * (size_t)b->np);
  double grad_f = 0;
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads) reduction(+ : cost, grad_f) schedule(static)
  for (int p = 0; p < b->np; ++p)
    for (int k = b->pt_ptr[p]; k < b->pt_ptr[p + 1]; ++k) {
      const int c = b->o_cam[k];
      double* r = b->r + 2 * k;
      double *Jc = b->Jc + 12 * k, *Jp = b->Jp + 6 * k, *Jf = b->Jf + 2 * k;
      orc_ba_residual(b->cams + 6 * c, b->pts + 3 * p, b->focal, b->o_xy + 2 * k, r, Jc, Jp, Jf);
      cost += r[0] * r[0] + r[1] * r[1];
      for (int j = 0; j < 6; ++j) {
        const double v = Jc[j] * r[0] + Jc[6 + j] * r[1];
#pragma omp atomic
        b->grad_c[6 * c + j] += v;
      }
      for (int j = 0; j < 3; ++j) b->grad_p[3 * p + j] += Jp[j] * r[0] + Jp[3 + j] * r[1];
      grad_f += Jf[0] * r[0] + Jf[1] * r[1];
    }
  b->grad_f = grad_f;
  if (first && jacobi_scaling) {
    double* nc2 = (double*)xcalloc((size_t)6 * b->nc, sizeof(double));
    double* np2 = (double*)xcalloc((size_t)3 * b->np, sizeof(double));
    double nf2 = 0;
    for (int p = 0; p < b->np; ++p)
      for (int k = b->pt_ptr[p]; k < b->pt_ptr[p + 1]; ++k) {
        const int c = b->o_cam[k];
        const double *Jc = b->Jc + 12 * k, *Jp = b->Jp + 6 * k, *Jf = b->Jf + 2 * k;
        for (int j = 0; j < 6; ++j) nc2[6 * c + j] += Jc[j] * Jc[j] + Jc[6 + j] * Jc[6 + j];
        for (int j = 0; j < 3; ++j) np2[3 * p + j] += Jp[j] * Jp[j] + Jp[3 + j] * Jp[3 + j];
        nf2 += Jf[0] * Jf[0] + Jf[1] * Jf[1];
      }
    for (int i = 0; i < 6 * b->nc; ++i) b->scale_c[i] = 1.0 / (1.0 + sqrt(nc2[i]));
    for (int i = 0; i < 3 * b->np; ++i) b->scale_p[i] = 1.0 / (1.0 + sqrt(np2[i]));
    b->scale_f = 1.0 / (1.0 + sqrt(nf2));
    free(nc2);
    free(np2);
  }
  /* scale columns */
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads) schedule(static)
  for (int p = 0; p < b->np; ++p)
    for (int k = b->pt_ptr[p]; k < b->pt_ptr[p + 1]; ++k) {
      const int c = b->o_cam[k];
      double *Jc = b->Jc + 12 * k, *Jp = b->Jp + 6 * k, *Jf = b->Jf + 2 * k;
      for (int j = 0; j < 6; ++j) {
        Jc[j] *= b->scale_c[6 * c + j];
        Jc[6 + j] *= b->scale_c[6 * c + j];
      }
      for (int j = 0; j < 3; ++j) {
        Jp[j] *= b->scale_p[3 * p + j];
        Jp[3 + j] *= b->scale_p[3 * p + j];
      }
      Jf[0] *= b->scale_f;
      Jf[1] *= b->scale_f;
    }
  return 0.5 * cost;
}

/* LevenbergMarquardtStrategy: diagonal_ = clamp(squared column norms of the scaled J) */
static void ba_lm_diagonal(ba_t* b, double lo, double hi) {
  memset(b->diag_c, 0, sizeof(double) * 6 * (size_t)b->nc);
  memset(b->diag_p, 0, sizeof(double) * 3 * (size_t)b->np);
  b->diag_f = 0;
  for (int p = 0; p < b->np; ++p)
    for (int k = b->pt_ptr[p]; k < b->pt_ptr[p + 1]; ++k) {
      const int c = b->o_cam[k];
      const double *Jc = b->Jc + 12 * k, *Jp = b->Jp + 6 * k, *Jf = b->Jf + 2 * k;
      for (int j = 0; j < 6; ++j) b->diag_c[6 * c + j] += Jc[j] * Jc[j] + Jc[6 + j] * Jc[6 + j];
      for (int j = 0; j < 3; ++j) b->diag_p[3 * p + j] += Jp[j] * Jp[j] + Jp[3 + j] * Jp[3 + j];
      b->diag_f += Jf[0] * Jf[0] + Jf[1] * Jf[1];
    }
#define CLAMP(v) ((v) < lo ? lo : ((v) > hi ? hi : (v)))
  for (int i = 0; i < 6 * b->nc; ++i) b->diag_c[i] = CLAMP(b->diag_c[i]);
  for (int i = 0; i < 3 * b->np; ++i) b->diag_p[i] = CLAMP(b->diag_p[i]);
  b->diag_f = CLAMP(b->diag_f);
#undef CLAMP
}

/* 3x3 SPD inverse via LLT solve (InvertPSDMatrix<3>, full-rank path) */
static int inv3_spd(const double C[9], double Ci[9]) {
  double l00 = C[0];
  if (!(l00 > 0)) return -1;
  l00 = sqrt(l00);
  const double l10 = C[3] / l00, l20 = C[6] / l00;
  double l11 = C[4] - l10 * l10;
  if (!(l11 > 0)) return -1;
  l11 = sqrt(l11);
  const double l21 = (C[7] - l20 * l10) / l11;
  double l22 = C[8] - l20 * l20 - l21 * l21;
  if (!(l22 > 0)) return -1;
  l22 = sqrt(l22);
  /* inverse of L (lower) */
  const double i00 = 1 / l00, i11 = 1 / l11, i22 = 1 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  /* C^-1 = Li^T Li */
  Ci[0] = i00 * i00 + i10 * i10 + i20 * i20;
  Ci[1] = Ci[3] = i10 * i11 + i20 * i21;
  Ci[2] = Ci[6] = i20 * i22;
  Ci[4] = i11 * i11 + i21 * i21;
  Ci[5] = Ci[7] = i21 * i22;
  Ci[8] = i22 * i22;
  return 0;
}

/* SchurEliminator::Eliminate: upper block triangle of S and the reduced rhs.
 * S, g over [cam0..camN-1 | focal]; D^2 = diag/radius. */
static int ba_eliminate(ba_t* b, double radius, double* S, double* g) {
  const int dim = b->dim, fo = 6 * b->nc;
  memset(S, 0, sizeof(double) * (size_t)dim * dim);
  memset(g, 0, sizeof(double) * (size_t)dim);
  for (int i = 0; i < fo; ++i) S[(size_t)i * dim + i] = b->diag_c[i] / radius;
  S[(size_t)fo * dim + fo] = b->diag_f / radius;
  enum { MAXN = 4096 };
  int rc = 0;
  /* one thread: straight into S, g.  More: every thread sums its points into a private copy, added up below */
  const int T = g_threads > 1 ? (g_threads < 32 ? g_threads : 32) : 1;
  double* Sall = T > 1 ? (double*)xcalloc((size_t)T * ((size_t)dim * dim + dim), sizeof(double)) : NULL;
  double* const S_out = S;
  double* const g_out = g;
#pragma omp parallel num_threads(T) if (T > 1)
  {
  const int tid = T > 1 ? omp_get_thread_num() : 0;
  double* S = T > 1 ? Sall + (size_t)tid * ((size_t)dim * dim + dim) : S_out;
  double* g = T > 1 ? S + (size_t)dim * dim : g_out;
  double* W = (double*)xcalloc((size_t)MAXN * 18, sizeof(double));  /* Jc^T Jp per obs */
  double* WC = (double*)xcalloc((size_t)MAXN * 18, sizeof(double)); /* W * Cinv */
#pragma omp for schedule(static)
  for (int p = 0; p < b->np; ++p) {
    const int k0 = b->pt_ptr[p], n = b->pt_ptr[p + 1] - k0;
    if (n == 0 || rc) continue;
    if (n > MAXN) {
      rc = -4;
      continue;
    }
    double C[9] = {0}, gp[3] = {0}, wf[3] = {0};
    for (int j = 0; j < 3; ++j) C[4 * j] = b->diag_p[3 * p + j] / radius;
    for (int e = 0; e < n; ++e) {
      const int k = k0 + e, c = b->o_cam[k];
      const double *Jc = b->Jc + 12 * k, *Jp = b->Jp + 6 * k, *Jf = b->Jf + 2 * k, *r = b->r + 2 * k;
      for (int a = 0; a < 3; ++a)
        for (int bb = 0; bb < 3; ++bb) C[3 * a + bb] += Jp[a] * Jp[bb] + Jp[3 + a] * Jp[3 + bb];
      for (int a = 0; a < 3; ++a) {
        gp[a] += Jp[a] * r[0] + Jp[3 + a] * r[1];
        wf[a] += Jf[0] * Jp[a] + Jf[1] * Jp[3 + a];
      }
      double* We = W + 18 * e;
      for (int i = 0; i < 6; ++i)
        for (int a = 0; a < 3; ++a) We[3 * i + a] = Jc[i] * Jp[a] + Jc[6 + i] * Jp[3 + a];
      /* F^T F and F^T b */
      double* Scc = S + (size_t)(6 * c) * dim + 6 * c;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) Scc[(size_t)i * dim + j] += Jc[i] * Jc[j] + Jc[6 + i] * Jc[6 + j];
      for (int i = 0; i < 6; ++i) {
        S[(size_t)(6 * c + i) * dim + fo] += Jc[i] * Jf[0] + Jc[6 + i] * Jf[1];
        g[6 * c + i] += Jc[i] * r[0] + Jc[6 + i] * r[1];
      }
      S[(size_t)fo * dim + fo] += Jf[0] * Jf[0] + Jf[1] * Jf[1];
      g[fo] += Jf[0] * r[0] + Jf[1] * r[1];
    }
    double Ci[9];
    if (inv3_spd(C, Ci)) {
      rc = -5;
      continue;
    }
    double Cg[3], Cwf[3];
    for (int a = 0; a < 3; ++a) {
      Cg[a] = Ci[3 * a] * gp[0] + Ci[3 * a + 1] * gp[1] + Ci[3 * a + 2] * gp[2];
      Cwf[a] = Ci[3 * a] * wf[0] + Ci[3 * a + 1] * wf[1] + Ci[3 * a + 2] * wf[2];
    }
    for (int e = 0; e < n; ++e) {
      const double* We = W + 18 * e;
      double* WCe = WC + 18 * e;
      for (int i = 0; i < 6; ++i)
        for (int a = 0; a < 3; ++a)
          WCe[3 * i + a] =
              We[3 * i] * Ci[a] + We[3 * i + 1] * Ci[3 + a] + We[3 * i + 2] * Ci[6 + a];
    }
    for (int e1 = 0; e1 < n; ++e1) {
      const int c1 = b->o_cam[k0 + e1];
      const double* WC1 = WC + 18 * e1;
      const double* W1 = W + 18 * e1;
      for (int i = 0; i < 6; ++i) {
        g[6 * c1 + i] -= W1[3 * i] * Cg[0] + W1[3 * i + 1] * Cg[1] + W1[3 * i + 2] * Cg[2];
        S[(size_t)(6 * c1 + i) * dim + fo] -=
            W1[3 * i] * Cwf[0] + W1[3 * i + 1] * Cwf[1] + W1[3 * i + 2] * Cwf[2];
      }
      for (int e2 = 0; e2 < n; ++e2) {
        const int c2 = b->o_cam[k0 + e2];
        if (c2 < c1 || (c2 == c1 && e2 < e1)) continue; /* upper block triangle only */
        const double* W2 = W + 18 * e2;
        double* Sb = S + (size_t)(6 * c1) * dim + 6 * c2;
        if (c1 == c2 && e1 == e2) {
          for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j)
              Sb[(size_t)i * dim + j] -=
                  WC1[3 * i] * W2[3 * j] + WC1[3 * i + 1] * W2[3 * j + 1] + WC1[3 * i + 2] * W2[3 * j + 2];
        } else if (c1 == c2) { /* two observations of one camera in one point: both orders */
          for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
              double v = WC1[3 * i] * W2[3 * j] + WC1[3 * i + 1] * W2[3 * j + 1] +
                         WC1[3 * i + 2] * W2[3 * j + 2];
              if (j >= i) Sb[(size_t)i * dim + j] -= v;
              if (i >= j) Sb[(size_t)j * dim + i] -= v;
            }
        } else {
          for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j)
              Sb[(size_t)i * dim + j] -=
                  WC1[3 * i] * W2[3 * j] + WC1[3 * i + 1] * W2[3 * j + 1] + WC1[3 * i + 2] * W2[3 * j + 2];
        }
      }
    }
    S[(size_t)fo * dim + fo] -= wf[0] * Cwf[0] + wf[1] * Cwf[1] + wf[2] * Cwf[2];
    g[fo] -= wf[0] * Cg[0] + wf[1] * Cg[1] + wf[2] * Cg[2];
  }
  free(W);
  free(WC);
  }
  if (T > 1) {
    for (int t = 0; t < T; ++t) {
      const double* St = Sall + (size_t)t * ((size_t)dim * dim + dim);
#pragma omp parallel for num_threads(T) schedule(static)
      for (int i = 0; i < dim; ++i) {
        for (int j = i; j < dim; ++j) S[(size_t)i * dim + j] += St[(size_t)i * dim + j];
        g[i] += St[(size_t)dim * dim + i];
      }
    }
    free(Sall);
  }
  if (rc) return rc;
  for (int i = 0; i < dim; ++i) /* mirror to a full symmetric matrix */
    for (int j = i + 1; j < dim; ++j) S[(size_t)j * dim + i] = S[(size_t)i * dim + j];
  return 0;
}

/* dense LLT (row-oriented Cholesky-Banachiewicz) + two triangular solves; S is overwritten */
static int chol_solve(double* S, int n, const double* rhs, double* x) {
  for (int i = 0; i < n; ++i) {
    double* Li = S + (size_t)i * n;
    for (int j = 0; j <= i; ++j) {
      const double* Lj = S + (size_t)j * n;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      int k = 0;
      for (; k + 4 <= j; k += 4) {
        s0 += Li[k] * Lj[k];
        s1 += Li[k + 1] * Lj[k + 1];
        s2 += Li[k + 2] * Lj[k + 2];
        s3 += Li[k + 3] * Lj[k + 3];
      }
      for (; k < j; ++k) s0 += Li[k] * Lj[k];
      double v = Li[j] - ((s0 + s1) + (s2 + s3));
      if (i == j) {
        if (!(v > 0) || !isfinite(v)) return -1;
        Li[j] = sqrt(v);
      } else {
        Li[j] = v / Lj[j];
      }
    }
  }
  for (int i = 0; i < n; ++i) {
    const double* Li = S + (size_t)i * n;
    double v = rhs[i];
    for (int k = 0; k < i; ++k) v -= Li[k] * x[k];
    x[i] = v / Li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = x[i];
    for (int k = i + 1; k < n; ++k) v -= S[(size_t)k * n + i] * x[k];
    x[i] = v / S[(size_t)i * n + i];
  }
  return 0;
}

/* ---- the reduced solve as a competent CPU library does it (bench.py's TIMED cpu_baseline leg only) --------------------
 * Ceres 1.13's DENSE_SCHUR hands the reduced system to Eigen's LLT: a BLOCKED, vectorised right-looking Cholesky.  The
 * row-by-row factorisation above is the checker's arithmetic and stays what the parity tests compare with; timing IT as
 * "the CPU" understates the CPU several times over.  This one is what the baseline leg runs:
 * 64-column panels; the diagonal block factored in place, the panel below it solved row by row against the block, the
 * trailing lower triangle updated by a register-tiled rank-64 product (4 rows x 16 columns of accumulators, the panel
 * transposed once so that the inner loop is broadcast x contiguous vector -- GCC vector extensions: AVX-512 FMAs with
 * -march=native on the GPU boxes' hosts, AVX2 in the portable build).  Same answer to rounding (tests/test_oracle_geometry.py). */
typedef double v8d __attribute__((vector_size(64), aligned(8), may_alias));  /* (aligned(8): *(v8d*)p is an unaligned load) */
#define CHB 64
static int g_blocked_chol = 0;
static double g_chol_flops = 0.0, g_chol_seconds = 0.0;
void orc_ba_set_blocked_cholesky(int on) { g_blocked_chol = on ? 1 : 0; }
void orc_ba_cholesky_stats(double out[2], int reset) {
  out[0] = g_chol_flops, out[1] = g_chol_seconds;
  if (reset) g_chol_flops = g_chol_seconds = 0.0;
}

__attribute__((optimize("fp-contract=fast"))) static int chol_factor_blocked(double* S, int n) {
  double* Lt = (double*)malloc(sizeof(double) * CHB * (size_t)(n + 16));  /* the panel transposed: Lt[k][row] */
  if (!Lt) return -1;
#ifdef CHOL_PROFILE
  double tp[5] = {0, 0, 0, 0, 0}, tq;
#define TP(i) tp[i] += now_s() - tq, tq = now_s()
#else
#define TP(i)
#endif
  for (int k0 = 0; k0 < n; k0 += CHB) {
    const int kb = n - k0 < CHB ? n - k0 : CHB, k1 = k0 + kb;
#ifdef CHOL_PROFILE
    tq = now_s();
#endif
    /* diagonal block (the earlier panels are already folded into it) */
    for (int i = k0; i < k1; ++i) {
      double* Li = S + (size_t)i * n;
      for (int j = k0; j <= i; ++j) {
        const double* Lj = S + (size_t)j * n;
        double s = 0;
        for (int k = k0; k < j; ++k) s += Li[k] * Lj[k];
        const double v = Li[j] - s;
        if (i == j) {
          if (!(v > 0) || !isfinite(v)) {
            free(Lt);
            return -1;
          }
          Li[j] = sqrt(v);
        } else {
          Li[j] = v / Lj[j];
        }
      }
    }
    TP(0);
    if (k1 >= n) break;
    /* the panel below it, X L_kk^T = A: transposed first (Lt[k][row]: the layout the trailing update wants anyway), then
     * solved 64 rows at a time -- column j of X for 64 rows is eight vectors, x_j = (a_j - sum_{k<j} x_k L[j][k]) / L[j][j] */
    const int m = n - k1, mp = (m + 15) & ~15, ldt = n + 16;
    for (int k = 0; k < kb; ++k) {
      double* row = Lt + (size_t)k * ldt;
      for (int r = 0; r < m; ++r) row[r] = S[(size_t)(k1 + r) * n + k0 + k];
      for (int r = m; r < mp; ++r) row[r] = 0.0;
    }
    TP(1);
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads) schedule(static)
    for (int r0 = 0; r0 < mp; r0 += 64) {  /* 64 rows = eight independent vectors per column: the k-sum is a latency chain per vector */
      const int nv = mp - r0 < 64 ? (mp - r0) / 8 : 8;
      for (int j = 0; j < kb; ++j) {
        const double* Lj = S + (size_t)(k0 + j) * n + k0;
        double* xj = Lt + (size_t)j * ldt + r0;
        v8d x[8];
#pragma GCC unroll 8
        for (int v = 0; v < 8; ++v) x[v] = *(const v8d*)(xj + 8 * (v < nv ? v : 0));
        for (int k = 0; k < j; ++k) {
          const double* xk = Lt + (size_t)k * ldt + r0;
          const double l = Lj[k];
#pragma GCC unroll 8
          for (int v = 0; v < 8; ++v) {
            v8d y;
            y = *(const v8d*)(xk + 8 * (v < nv ? v : 0));
            x[v] -= y * l;
          }
        }
        const double dj = Lj[j];
#pragma GCC unroll 8
        for (int v = 0; v < 8; ++v) {
          x[v] /= dj;
          if (v < nv) *(v8d*)(xj + 8 * v) = x[v];
        }
      }
    }
    TP(2);
    for (int r = 0; r < m; ++r) {  /* the solved panel back into L's rows */
      double* Li = S + (size_t)(k1 + r) * n + k0;
      for (int k = 0; k < kb; ++k) Li[k] = Lt[(size_t)k * ldt + r];
    }
    TP(3);
    /* trailing update of the lower triangle: A[i][j] -= sum_k L[i][k] L[j][k], i >= j >= k1; 8 rows x 16 columns of accumulators */
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads) schedule(dynamic, 2)
    for (int ib = 0; ib < m; ib += 8) {
      const int nr = m - ib < 8 ? m - ib : 8;
      double* A[8];
      const double* Lr[8];
      for (int r = 0; r < 8; ++r) {
        const int ii = k1 + ib + (r < nr ? r : 0);
        A[r] = S + (size_t)ii * n + k1;
        Lr[r] = S + (size_t)ii * n + k0;
      }
      const int jmax = ib + nr;  /* columns j - k1 < jmax */
      for (int jb = 0; jb < jmax; jb += 16) {
        v8d acc[8][2];
        for (int r = 0; r < 8; ++r) acc[r][0] = acc[r][1] = (v8d){0};
        for (int k = 0; k < kb; ++k) {
          const double* row = Lt + (size_t)k * ldt + jb;
          v8d b0, b1;
          b0 = *(const v8d*)(row);
          b1 = *(const v8d*)(row + 8);
#pragma GCC unroll 8
          for (int r = 0; r < 8; ++r) {
            const double l = Lr[r][k];
            acc[r][0] += l * b0, acc[r][1] += l * b1;
          }
        }
        double tmp[8][16];  /* (the accumulators leave their registers only here: a variable index into them would keep them in memory) */
#pragma GCC unroll 8
        for (int r = 0; r < 8; ++r) {
          *(v8d*)&tmp[r][0] = acc[r][0];
          *(v8d*)&tmp[r][8] = acc[r][1];
        }
        for (int r = 0; r < nr; ++r) {
          const int lim = ib + r + 1 - jb;  /* columns of this chunk inside the lower triangle of row ib + r */
          const int w = lim < 16 ? lim : 16;
          for (int c = 0; c < w; ++c) A[r][jb + c] -= tmp[r][c];
        }
      }
    }
    TP(4);
  }
#ifdef CHOL_PROFILE
  fprintf(stderr, "chol phases ms: diag %.2f transpose %.2f trsm %.2f writeback %.2f trailing %.2f\n", tp[0] * 1e3, tp[1] * 1e3, tp[2] * 1e3, tp[3] * 1e3, tp[4] * 1e3);
#endif
  free(Lt);
  return 0;
}

static int chol_solve_blocked(double* S, int n, const double* rhs, double* x) {
  const double t0 = now_s();
  if (chol_factor_blocked(S, n)) return -1;
  g_chol_seconds += now_s() - t0;
  g_chol_flops += (double)n * n * n / 3.0;
  for (int i = 0; i < n; ++i) {
    const double* Li = S + (size_t)i * n;
    double v = rhs[i];
    for (int k = 0; k < i; ++k) v -= Li[k] * x[k];
    x[i] = v / Li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = x[i];
    for (int k = i + 1; k < n; ++k) v -= S[(size_t)k * n + i] * x[k];
    x[i] = v / S[(size_t)i * n + i];
  }
  return 0;
}

/* test / bench hook: solve S x = rhs (S: n x n row-major, lower triangle read, overwritten) with either factorisation */
int orc_chol_solve(double* S, int n, const double* rhs, double* x, int blocked) {
  return blocked ? chol_solve_blocked(S, n, rhs, x) : chol_solve(S, n, rhs, x);
}

/* SchurEliminator::BackSubstitute, then step = -solution (LM strategy) */
static int ba_backsub(ba_t* b, double radius, const double* z) {
  const int fo = 6 * b->nc;
  for (int i = 0; i < fo; ++i) b->step_c[i] = -z[i];
  b->step_f = -z[fo];
  int bad = 0;
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads) schedule(static)
  for (int p = 0; p < b->np; ++p) {
    const int k0 = b->pt_ptr[p], n = b->pt_ptr[p + 1] - k0;
    if (n == 0) {
      b->step_p[3 * p] = b->step_p[3 * p + 1] = b->step_p[3 * p + 2] = 0;
      continue;
    }
    double C[9] = {0}, e[3] = {0};
    for (int j = 0; j < 3; ++j) C[4 * j] = b->diag_p[3 * p + j] / radius;
    for (int q = 0; q < n; ++q) {
      const int k = k0 + q, c = b->o_cam[k];
      const double *Jc = b->Jc + 12 * k, *Jp = b->Jp + 6 * k, *Jf = b->Jf + 2 * k, *r = b->r + 2 * k;
      double s0 = r[0] - Jf[0] * z[fo], s1 = r[1] - Jf[1] * z[fo];
      for (int j = 0; j < 6; ++j) {
        s0 -= Jc[j] * z[6 * c + j];
        s1 -= Jc[6 + j] * z[6 * c + j];
      }
      for (int a = 0; a < 3; ++a) {
        e[a] += Jp[a] * s0 + Jp[3 + a] * s1;
        for (int bb = 0; bb < 3; ++bb) C[3 * a + bb] += Jp[a] * Jp[bb] + Jp[3 + a] * Jp[3 + bb];
      }
    }
    double Ci[9];
    if (inv3_spd(C, Ci)) {
      bad = 1;
      continue;
    }
    for (int a = 0; a < 3; ++a)
      b->step_p[3 * p + a] = -(Ci[3 * a] * e[0] + Ci[3 * a + 1] * e[1] + Ci[3 * a + 2] * e[2]);
  }
  return bad ? -5 : 0;
}

/* model_cost_change = -(J d).(r + J d / 2) with the scaled J and scaled step */
static double ba_model_cost_change(const ba_t* b) {
  double acc = 0;
#pragma omp parallel for if (g_threads > 1) num_threads(g_threads) reduction(+ : acc) schedule(static)
  for (int p = 0; p < b->np; ++p)
    for (int k = b->pt_ptr[p]; k < b->pt_ptr[p + 1]; ++k) {
      const int c = b->o_cam[k];
      const double *Jc = b->Jc + 12 * k, *Jp = b->Jp + 6 * k, *Jf = b->Jf + 2 * k, *r = b->r + 2 * k;
      double m0 = Jf[0] * b->step_f, m1 = Jf[1] * b->step_f;
      for (int j = 0; j < 6; ++j) {
        m0 += Jc[j] * b->step_c[6 * c + j];
        m1 += Jc[6 + j] * b->step_c[6 * c + j];
      }
      for (int j = 0; j < 3; ++j) {
        m0 += Jp[j] * b->step_p[3 * p + j];
        m1 += Jp[3 + j] * b->step_p[3 * p + j];
      }
      acc += m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
    }
  return -acc;
}

static double ba_x_norm(const ba_t* b) {
  double s = b->focal * b->focal;
  for (int c = 0; c < b->nc; ++c)
    if (b->cam_used[c])
      for (int j = 0; j < 6; ++j) s += b->cams[6 * c + j] * b->cams[6 * c + j];
  for (int p = 0; p < b->np; ++p)
    if (b->pt_used[p])
      for (int j = 0; j < 3; ++j) s += b->pts[3 * p + j] * b->pts[3 * p + j];
  return sqrt(s);
}

static double ba_grad_max(const ba_t* b) {
  double m = fabs(b->grad_f);
  for (int i = 0; i < 6 * b->nc; ++i)
    if (fabs(b->grad_c[i]) > m) m = fabs(b->grad_c[i]);
  for (int i = 0; i < 3 * b->np; ++i)
    if (fabs(b->grad_p[i]) > m) m = fabs(b->grad_p[i]);
  return m;
}

/* candidate = x + step*scale; returns ||delta|| */
static double ba_candidate(ba_t* b) {
  double s = 0;
  for (int i = 0; i < 6 * b->nc; ++i) {
    double d = b->step_c[i] * b->scale_c[i];
    b->cams_c[i] = b->cams[i] + d;
    if (b->cam_used[i / 6]) s += d * d;
  }
  for (int i = 0; i < 3 * b->np; ++i) {
    double d = b->step_p[i] * b->scale_p[i];
    b->pts_c[i] = b->pts[i] + d;
    if (b->pt_used[i / 3]) s += d * d;
  }
  double d = b->step_f * b->scale_f;
  b->focal_c = b->focal + d;
  s += d * d;
  return sqrt(s);
}

static int step_finite(const ba_t* b) {
  if (!isfinite(b->step_f)) return 0;
  for (int i = 0; i < 6 * b->nc; ++i)
    if (!isfinite(b->step_c[i])) return 0;
  for (int i = 0; i < 3 * b->np; ++i)
    if (!isfinite(b->step_p[i])) return 0;
  return 1;
}

void orc_ba_default_opts(orc_ba_opts* o) {
  o->max_iterations = 500;
  o->max_time_s = 10.0;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4;
  o->max_radius = 1e16;
  o->min_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1;
  o->max_consecutive_invalid = 5;
  o->verbose = 0;
}

/* TrustRegionMinimizer::Minimize (Ceres 1.13) with LevenbergMarquardtStrategy and DENSE_SCHUR */
static int ba_minimize(ba_t* b, const orc_ba_opts* o, orc_ba_summary* sum, int timing_only,
                       int timing_iters) {
  const double t0 = now_s();
  double radius = o->initial_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0, invalid = 0, iter = 0, nsucc = 0;
  int term = ORC_BA_NO_CONVERGENCE;
  double x_norm = ba_x_norm(b);
  double cost = ba_linearize(b, 1, o->jacobi_scaling);
  double gmax = ba_grad_max(b);
  sum->initial_cost = cost;
  if (!timing_only && gmax <= o->gradient_tolerance) {
    term = ORC_BA_CONVERGENCE;
    goto done;
  }
  for (;;) {
    if (timing_only) {
      if (iter >= timing_iters) break;
    } else {
      if (iter >= o->max_iterations) break;                              /* NO_CONVERGENCE */
      if (o->max_time_s > 0 && now_s() - t0 >= o->max_time_s) break;     /* NO_CONVERGENCE */
      if (radius < o->min_radius) {
        term = ORC_BA_CONVERGENCE;
        break;
      }
    }
    ++iter;
    /* ---- LevenbergMarquardtStrategy::ComputeStep ---- */
    if (!reuse_diagonal) ba_lm_diagonal(b, o->min_lm_diagonal, o->max_lm_diagonal);
    int bad = ba_eliminate(b, radius, b->S, b->g);
    if (!bad) bad = (g_blocked_chol ? chol_solve_blocked : chol_solve)(b->S, b->dim, b->g, b->z);
    if (!bad) bad = ba_backsub(b, radius, b->z);
    if (!bad && !step_finite(b)) bad = 1;
    double mcc = bad ? 0.0 : ba_model_cost_change(b);
    if (bad || !(mcc > 0.0)) { /* HandleInvalidStep */
      if (++invalid >= o->max_consecutive_invalid && !timing_only) {
        term = ORC_BA_FAILURE;
        break;
      }
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      if (o->verbose) fprintf(stderr, "[orc-ba] it %d invalid step, radius %.3e\n", iter, radius);
      continue;
    }
    invalid = 0;
    const double step_norm = ba_candidate(b);
    const double cost_c = ba_cost(b, b->cams_c, b->pts_c, b->focal_c);
    if (!timing_only) {
      if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) {
        term = ORC_BA_CONVERGENCE; /* ParameterToleranceReached: candidate not taken */
        break;
      }
      if (fabs(cost - cost_c) <= o->function_tolerance * cost) {
        term = ORC_BA_CONVERGENCE; /* FunctionToleranceReached: candidate not taken */
        break;
      }
    }
    const double rho = (cost - cost_c) / mcc;
    if (o->verbose)
      fprintf(stderr, "[orc-ba] it %d cost %.9e -> %.9e rho %.3e radius %.3e |step| %.3e\n", iter,
              cost, cost_c, rho, radius, step_norm);
    if (rho > o->min_relative_decrease) { /* HandleSuccessfulStep */
      memcpy(b->cams, b->cams_c, sizeof(double) * 6 * (size_t)b->nc);
      memcpy(b->pts, b->pts_c, sizeof(double) * 3 * (size_t)b->np);
      b->focal = b->focal_c;
      x_norm = ba_x_norm(b);
      cost = ba_linearize(b, 0, o->jacobi_scaling);
      gmax = ba_grad_max(b);
      ++nsucc;
      const double q = 2.0 * rho - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
      radius = fmin(o->max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
      if (!timing_only && gmax <= o->gradient_tolerance) {
        term = ORC_BA_CONVERGENCE;
        break;
      }
    } else { /* HandleUnsuccessfulStep */
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
    }
  }
done:
  sum->termination = term;
  sum->iterations = iter;
  sum->successful_steps = nsucc;
  sum->final_cost = cost;
  sum->final_radius = radius;
  sum->gradient_max_norm = gmax;
  sum->time_s = now_s() - t0;
  return 0;
}

int orc_ba_solve(int n_cam, int n_pt, int n_obs, double* cams6, double* pts3, double* focal,
                 const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                 const orc_ba_opts* opts, orc_ba_summary* summary) {
  if (n_cam < 0 || n_pt < 0 || n_obs < 0) return -1;
  for (int o = 0; o < n_obs; ++o)
    if (obs_cam[o] < 0 || obs_cam[o] >= n_cam || obs_pt[o] < 0 || obs_pt[o] >= n_pt) return -2;
  orc_ba_opts od;
  if (!opts) {
    orc_ba_default_opts(&od);
    opts = &od;
  }
  orc_ba_summary s;
  memset(&s, 0, sizeof s);
  ba_t* b = ba_new(n_cam, n_pt, n_obs, cams6, pts3, *focal, obs_cam, obs_pt, obs_xy);
  ba_minimize(b, opts, &s, 0, 0);
  memcpy(cams6, b->cams, sizeof(double) * 6 * (size_t)n_cam);
  memcpy(pts3, b->pts, sizeof(double) * 3 * (size_t)n_pt);
  *focal = b->focal;
  ba_free(b);
  if (summary) *summary = s;
  return 0;
}

int orc_ba_reduced_system(int n_cam, int n_pt, int n_obs, const double* cams6, const double* pts3,
                          double focal, const int32_t* obs_cam, const int32_t* obs_pt,
                          const double* obs_xy, double radius, const double* scale_in,
                          double* scale_out, double* S, double* g, double* cost) {
  ba_t* b = ba_new(n_cam, n_pt, n_obs, cams6, pts3, focal, obs_cam, obs_pt, obs_xy);
  if (scale_in) {
    memcpy(b->scale_c, scale_in, sizeof(double) * 6 * (size_t)n_cam);
    memcpy(b->scale_p, scale_in + 6 * n_cam, sizeof(double) * 3 * (size_t)n_pt);
    b->scale_f = scale_in[6 * n_cam + 3 * n_pt];
  }
  double c = ba_linearize(b, scale_in ? 0 : 1, 1);
  if (cost) *cost = c;
  if (scale_out) {
    memcpy(scale_out, b->scale_c, sizeof(double) * 6 * (size_t)n_cam);
    memcpy(scale_out + 6 * n_cam, b->scale_p, sizeof(double) * 3 * (size_t)n_pt);
    scale_out[6 * n_cam + 3 * n_pt] = b->scale_f;
  }
  ba_lm_diagonal(b, 1e-6, 1e32);
  int rc = ba_eliminate(b, radius, S, g);
  ba_free(b);
  return rc;
}

double orc_ba_time_iterations(int n_cam, int n_pt, int n_obs, const double* cams6,
                              const double* pts3, double focal, const int32_t* obs_cam,
                              const int32_t* obs_pt, const double* obs_xy, int iters,
                              double* final_cost) {
  orc_ba_opts o;
  orc_ba_default_opts(&o);
  orc_ba_summary s;
  memset(&s, 0, sizeof s);
  ba_t* b = ba_new(n_cam, n_pt, n_obs, cams6, pts3, focal, obs_cam, obs_pt, obs_xy);
  ba_minimize(b, &o, &s, 1, iters);
  if (final_cost) *final_cost = s.final_cost;
  ba_free(b);
  return s.time_s;
}
