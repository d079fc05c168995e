/*
 * sfm_oracle_score.c -- ORACLE (test infrastructure, not product): cv::findEssentialMat(p1, p2, K, RANSAC, 0.999, 1.0,
 * mask) as StructFromMotion::findBestPair calls it (reference src/Sfm.cpp:543-546), restated from OpenCV 3.4.1 as
 * published -- ONE route, the library's:
 *
 *   calib3d/five-point.cpp  findEssentialMat: points * (1/f) + (-c * (1/f)) (the MatExpr `(col - c) / f` evaluates
 *                           through convertTo(alpha, beta)), threshold / ((fx + fy) / 2);
 *                           EMEstimatorCallback::runKernel: Q (5 x 9) -> SVD::compute(FULL_UV), rows 5..8 of Vt as the
 *                           null space X, Y, Z, W; the 10 x 20 constraint matrix (getCoeffMat; Nister's monomial
 *                           order); A = A[:, :10].inv() * A[:, 10:]; the 3 x 13 matrix B from rows 4..9; the 11
 *                           coefficients of det B(z); cv::solvePoly; a root is real iff |imag| <= 1e-10; (x, y, 1)
 *                           from SVD::solveZ(B(z)), dropped when |w| < 1e-10; E = X x + Y y + Z z + W, divided by
 *                           its Frobenius norm; models in the order of the roots;
 *                           EMEstimatorCallback::computeError (float per match);
 *   calib3d/ptsetreg.cpp    RANSACPointSetRegistrator::run / getSubset / findInliers, RANSACUpdateNumIters;
 *   core/lapack.cpp         JacobiSVDImpl_<double> (one-sided Jacobi on the rows of A^T, minval DBL_MIN, eps
 *                           10 DBL_EPSILON; singular vectors of zero singular values filled from RNG(0x12345678) sign
 *                           vectors, orthogonalised twice against the rows before them), hal::LU (partial pivoting,
 *                           eps 100 DBL_EPSILON), Mat::inv(DECOMP_LU) = LU against the identity;
 *   core/mathfuncs.cpp      cv::solvePoly (Durand-Kerner in sequence, start (1+i)^k, 300 iterations or until no root
 *                           moves), cv::RNG (multiply with carry, 4164903690).
 *
 * PARITY UNPINNED: OpenCV is neither under /root/reference nor in this image and the reference holds no vector for this
 * call; everything above is restated from the published sources from memory.  What is knowingly NOT reproduced, all of it
 * at the level of the last bit: (1) getCoeffMat and the eleven determinant coefficients are machine-generated sums of
 * products in OpenCV; here the same polynomials are expanded in a fixed order of our own, so individual coefficients
 * may round differently; (2) the row order of the constraint matrix (it only steers LU pivoting); (3) cv::gemm's and
 * cv::norm's internal summation orders where not stated below; (4) solvePoly's branch for two iterates that coincide
 * bit for bit (num_same_root > 1, which calls cv::solveCubic) is not restated: a sample that reaches it sets bit 0 of
 * `flags` and is handled by skipping the zero factor only; (5) a leading coefficient <= DBL_EPSILON makes solvePoly
 * lower the degree to n and return part of its work buffer as the missing roots: restated for n <= 4, where that part
 * is the coefficient array itself (the singular-sample case, n = 1, among them: it yields the model E = W, which
 * RANSAC does score); for 5 <= n <= 9 the buffer is uninitialised memory: such a sample sets bit 1 and gets no model
 * from the tail.  Tests assert that no sample of theirs sets either bit.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "sfm_oracle.h"

/* ---------------------------------------------------------------- cv::RNG */
typedef struct { uint64_t state; } cvrng;
static inline unsigned cvrng_next(cvrng* r) {
  r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
  return (unsigned)r->state;
}
static inline int cvrng_uniform(cvrng* r, int a, int b) { return a == b ? a : (int)(cvrng_next(r) % (unsigned)(b - a) + a); }

/* ---------------------------------------------------------------- core/lapack.cpp JacobiSVDImpl_<double>
 * At: n rows of length m (row stride lda), orthogonalised in place; W: n; Vt: n x n (stride n) or NULL; rows n..n1-1 of
 * At (and rows whose singular value is <= minval) are filled as the library does. */
static void jacobi_svd(double* At, int lda, double* W, double* Vt, int m, int n, int n1) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  const int max_iter = m > 30 ? m : 30;
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) {
      const double t = At[i * lda + k];
      sd += t * t;
    }
    W[i] = sd;
    if (Vt) {
      for (int k = 0; k < n; ++k) Vt[i * n + k] = 0;
      Vt[i * n + i] = 1;
    }
  }
  for (int iter = 0; iter < max_iter; ++iter) {
    int changed = 0;
    for (int i = 0; i < n - 1; ++i)
      for (int j = i + 1; j < n; ++j) {
        double *Ai = At + i * lda, *Aj = At + j * lda;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; ++k) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = hypot(p, beta);
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
        for (int k = 0; k < m; ++k) {
          const double t0 = c * Ai[k] + s * Aj[k];
          const double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = 1;
        if (Vt) {
          double *Vi = Vt + i * n, *Vj = Vt + j * n;
          for (int k = 0; k < n; ++k) {
            const double t0 = c * Vi[k] + s * Vj[k];
            const double t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0;
            Vj[k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) {
      const double t = At[i * lda + k];
      sd += t * t;
    }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < n - 1; ++i) {
    int j = i;
    for (int k = i + 1; k < n; ++k)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i];
      W[i] = W[j];
      W[j] = t;
      if (Vt) {
        for (int k = 0; k < m; ++k) {
          t = At[i * lda + k];
          At[i * lda + k] = At[j * lda + k];
          At[j * lda + k] = t;
        }
        for (int k = 0; k < n; ++k) {
          t = Vt[i * n + k];
          Vt[i * n + k] = Vt[j * n + k];
          Vt[j * n + k] = t;
        }
      }
    }
  }
  if (!Vt) return;
  cvrng rng = {0x12345678};
  for (int i = 0; i < n1; ++i) {
    double sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
      /* a zero singular value: a vector of +-1/m, projected off the rows before it (twice), L1-normalised on the way */
      const double val0 = 1. / m;
      for (int k = 0; k < m; ++k) At[i * lda + k] = (cvrng_next(&rng) & 256) != 0 ? val0 : -val0;
      for (int iter = 0; iter < 2; ++iter)
        for (int j = 0; j < i; ++j) {
          sd = 0;
          for (int k = 0; k < m; ++k) sd += At[i * lda + k] * At[j * lda + k];
          double asum = 0;
          for (int k = 0; k < m; ++k) {
            const double t = At[i * lda + k] - sd * At[j * lda + k];
            At[i * lda + k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < m; ++k) At[i * lda + k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < m; ++k) {
        const double t = At[i * lda + k];
        sd += t * t;
      }
      sd = sqrt(sd);
    }
    const double s = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < m; ++k) At[i * lda + k] *= s;
  }
}

/* SVD::solveZ for a 3 x 3 matrix (row-major): the row of Vt of the smallest singular value.  SVD::compute hands
 * JacobiSVD the transpose (its rows = the columns of B); Vt accumulates the rotations. */
static void solve_z3(const double B[9], double z[3]) {
  double At[9], W[3], Vt[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) At[i * 3 + k] = B[k * 3 + i];
  jacobi_svd(At, 3, W, Vt, 3, 3, 3);
  z[0] = Vt[6];
  z[1] = Vt[7];
  z[2] = Vt[8];
}

/* hal::LU64f on A (m x m) with right-hand sides b (m x n): 0 when a pivot is below eps (then Mat::inv() gives zeros) */
static int lu_solve(double* A, int m, double* b, int n) {
  const double eps = DBL_EPSILON * 100;
  for (int i = 0; i < m; ++i) {
    int k = i;
    for (int j = i + 1; j < m; ++j)
      if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
    if (fabs(A[k * m + i]) < eps) return 0;
    if (k != i) {
      for (int j = i; j < m; ++j) {
        const double t = A[i * m + j];
        A[i * m + j] = A[k * m + j];
        A[k * m + j] = t;
      }
      for (int j = 0; j < n; ++j) {
        const double t = b[i * n + j];
        b[i * n + j] = b[k * n + j];
        b[k * n + j] = t;
      }
    }
    const double d = -1 / A[i * m + i];
    for (int j = i + 1; j < m; ++j) {
      const double alpha = A[j * m + i] * d;
      for (int kk = i + 1; kk < m; ++kk) A[j * m + kk] += alpha * A[i * m + kk];
      for (int kk = 0; kk < n; ++kk) b[j * n + kk] += alpha * b[i * n + kk];
    }
  }
  for (int i = m - 1; i >= 0; --i)
    for (int j = 0; j < n; ++j) {
      double s = b[i * n + j];
      for (int k = i + 1; k < m; ++k) s -= A[i * m + k] * b[k * n + j];
      b[i * n + j] = s / A[i * m + i];
    }
  return 1;
}

/* ---------------------------------------------------------------- the ten cubic constraints (getCoeffMat's polynomials)
 * columns (Nister): x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1
 * degree 1: [x y z 1]; degree 2: [x2 xy xz y2 yz z2 x y z 1] */
static const int T11[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
static const int T21[10][4] = {{0, 2, 4, 5}, {2, 3, 8, 9}, {4, 8, 10, 11}, {3, 1, 6, 7}, {8, 6, 13, 14},
                               {10, 13, 16, 17}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};
static void mul11(const double* a, const double* b, double* out, double s) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[T11[i][j]] += s * a[i] * b[j];
}
static void mul21(const double* a, const double* b, double* out, double s) {
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 4; ++j) out[T21[i][j]] += s * a[i] * b[j];
}
/* e: the 4 x 9 null-space basis (rows X, Y, Z, W, each a row-major 3 x 3); A: 10 x 20 */
static void coeff_mat(const double* e, double* A) {
  double E1[9][4];
  for (int c = 0; c < 9; ++c)
    for (int k = 0; k < 4; ++k) E1[c][k] = e[k * 9 + c];
  memset(A, 0, 200 * sizeof(double));
  static const int minors[3][5] = {{0, 4, 8, 5, 7}, {1, 5, 6, 3, 8}, {2, 3, 7, 4, 6}};
  for (int t = 0; t < 3; ++t) { /* det(E) along the first row */
    double m2[10] = {0};
    mul11(E1[minors[t][1]], E1[minors[t][2]], m2, 1.0);
    mul11(E1[minors[t][3]], E1[minors[t][4]], m2, -1.0);
    mul21(m2, E1[minors[t][0]], A, 1.0);
  }
  double EEt[3][3][10], tr[10];
  for (int a = 0; a < 3; ++a)
    for (int b = a; b < 3; ++b) {
      memset(EEt[a][b], 0, sizeof(EEt[a][b]));
      for (int k = 0; k < 3; ++k) mul11(E1[3 * a + k], E1[3 * b + k], EEt[a][b], 1.0);
      if (b != a) memcpy(EEt[b][a], EEt[a][b], sizeof(EEt[a][b]));
    }
  for (int c = 0; c < 10; ++c) tr[c] = EEt[0][0][c] + EEt[1][1][c] + EEt[2][2][c];
  for (int a = 0; a < 3; ++a) /* 2 E E^T E - tr(E E^T) E, entry (a, b) */
    for (int b = 0; b < 3; ++b) {
      double* row = A + 20 * (1 + 3 * a + b);
      for (int k = 0; k < 3; ++k) mul21(EEt[a][k], E1[3 * k + b], row, 2.0);
      mul21(tr, E1[3 * a + b], row, -1.0);
    }
}

/* ---------------------------------------------------------------- core/mathfuncs.cpp cv::solvePoly, real coefficients
 * c[0..n0] ascending; roots: n0 complex numbers (re, im).  Returns the degree actually solved; *same_root is set when
 * the library's num_same_root branch would have run. */
typedef struct { double re, im; } cplx;
static inline cplx cmul(cplx a, cplx b) { return (cplx){a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static inline cplx cadd(cplx a, cplx b) { return (cplx){a.re + b.re, a.im + b.im}; }
static inline cplx csub(cplx a, cplx b) { return (cplx){a.re - b.re, a.im - b.im}; }
static inline cplx cdiv(cplx a, cplx b) {
  const double t = 1. / (b.re * b.re + b.im * b.im);
  return (cplx){(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}
static int solve_poly(const double* c, int n0, cplx* roots, int max_iters, int* same_root) {
  cplx coeffs[16];
  int n = n0;
  for (int i = 0; i <= n; ++i) coeffs[i] = (cplx){c[i], 0};
  for (; n > 1; --n)
    if (fabs(coeffs[n].re) + fabs(coeffs[n].im) > DBL_EPSILON) break;
  cplx p = {1, 0}, r = {1, 1};
  for (int i = 0; i < n; ++i) {
    roots[i] = p;
    p = cmul(p, r);
  }
  max_iters = max_iters <= 0 ? 1000 : max_iters;
  for (int iter = 0; iter < max_iters; ++iter) {
    double max_diff = 0;
    for (int i = 0; i < n; ++i) {
      p = roots[i];
      cplx num = coeffs[n], denom = coeffs[n];
      for (int j = 0; j < n; ++j) {
        num = cadd(cmul(num, p), coeffs[n - j - 1]);
        if (j != i) {
          const cplx d = csub(p, roots[j]);
          if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
          else *same_root = 1;
        }
      }
      num = cdiv(num, denom);
      roots[i] = csub(p, num);
      const double a = sqrt(num.re * num.re + num.im * num.im);
      if (a > max_diff) max_diff = a;
    }
    if (max_diff <= 0) break;
  }
  for (int i = 0; i < n; ++i)
    if (fabs(roots[i].im) < 1e-100) roots[i].im = 0;
  /* A leading coefficient <= DBL_EPSILON lowers the degree to n < n0; the library then fills the tail of the root
   * array with `roots[n+1] = roots[n]`, and roots[n] is never written: it is what its work buffer held, which is the
   * real coefficient array converted in place -- roots[n] = (c[2n], c[2n+1]) for 2n+1 <= n0.  (A sample whose 10 x 10
   * block is singular has Mat::inv() = 0, an all-zero polynomial, n = 1, root 0 = 0/0 = NaN and nine roots (0, 0):
   * nine times the model E = W.)  Beyond that the buffer is uninitialised: not restated, flagged by the caller. */
  for (int k = n; k < n0; ++k) roots[k] = 2 * n + 1 <= n0 ? (cplx){c[2 * n], c[2 * n + 1]} : (cplx){0, 1};
  return n;
}

/* ---------------------------------------------------------------- EMEstimatorCallback::runKernel
 * q1, q2: five normalised points each (x, y).  E: up to 10 row-major 3 x 3 models.  Returns the count. */
int orc_five_point(const double* q1, const double* q2, double* E, int* flags) {
  double Q[9 * 9];
  memset(Q, 0, sizeof(Q));
  for (int i = 0; i < 5; ++i) {
    const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
    double* r = Q + 9 * i;
    r[0] = x1 * x2; r[1] = y1 * x2; r[2] = x2 * 1.0; r[3] = x1 * y2; r[4] = y1 * y2; r[5] = y2 * 1.0;
    r[6] = x1 * 1.0; r[7] = y1 * 1.0; r[8] = 1.0;
  }
  /* SVD::compute(Q, W, U, Vt, MODIFY_A | FULL_UV) with rows < cols: JacobiSVD runs on Q itself (n = 5 rows of length
   * m = 9, n1 = 9) and Vt = the nine rows it leaves */
  double W[5], V5[25];
  jacobi_svd(Q, 9, W, V5, 9, 5, 9);
  const double* e = Q + 9 * 5; /* rows 5..8: X, Y, Z, W */
  double A[200];
  coeff_mat(e, A);
  /* A = A.colRange(0, 10).inv() * A.colRange(10, 20) */
  double L[100], inv[100], R[100];
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) {
      L[i * 10 + j] = A[i * 20 + j];
      inv[i * 10 + j] = i == j ? 1.0 : 0.0;
    }
  if (!lu_solve(L, 10, inv, 10)) memset(inv, 0, sizeof(inv));
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) {
      double s = 0;
      for (int k = 0; k < 10; ++k) s += inv[i * 10 + k] * A[k * 20 + 10 + j];
      R[i * 10 + j] = s;
    }
  double b[39];
  for (int i = 0; i < 3; ++i) {
    const double* a1 = R + 10 * (2 * i + 4);
    const double* a2 = R + 10 * (2 * i + 5);
    double row1[13] = {0}, row2[13] = {0};
    for (int k = 0; k < 3; ++k) row1[1 + k] = a1[k], row1[5 + k] = a1[3 + k], row2[k] = a2[k], row2[4 + k] = a2[3 + k];
    for (int k = 0; k < 4; ++k) row1[9 + k] = a1[6 + k], row2[8 + k] = a2[6 + k];
    for (int k = 0; k < 13; ++k) b[13 * i + k] = row1[k] - row2[k];
  }
  /* det B(z): entries of row i are P_i (deg 3, b[13i .. +3]), Q_i (deg 3, +4..+7), R_i (deg 4, +8..+12), highest power
   * first; c[k] = coefficient of z^k */
  double c[11] = {0};
  static const int perm[6][3] = {{0, 1, 2}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}, {0, 2, 1}, {1, 0, 2}}; /* (row of P, of Q, of R) */
  static const double sign[6] = {1, 1, 1, -1, -1, -1};
  for (int t = 0; t < 6; ++t) {
    const double* P = b + 13 * perm[t][0];
    const double* Qq = b + 13 * perm[t][1] + 4;
    const double* Rr = b + 13 * perm[t][2] + 8;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        const double pq = sign[t] * P[i] * Qq[j];
        for (int k = 0; k < 5; ++k) c[(3 - i) + (3 - j) + (4 - k)] += pq * Rr[k];
      }
  }
  cplx roots[10];
  int same = 0;
  const int deg = solve_poly(c, 10, roots, 300, &same);
  if (same && flags) *flags |= 1;
  if (deg < 10 && 2 * deg + 1 > 10 && flags) *flags |= 2;
  int count = 0;
  for (int i = 0; i < 10; ++i) {
    if (fabs(roots[i].im) > 1e-10) continue;
    const double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
    double bz[9];
    for (int j = 0; j < 3; ++j) {
      const double* br = b + j * 13;
      bz[3 * j + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
      bz[3 * j + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
      bz[3 * j + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
    }
    double xy1[3];
    solve_z3(bz, xy1);
    if (fabs(xy1[2]) < 1e-10) continue;
    const double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2];
    double* Ev = E + 9 * count;
    for (int k = 0; k < 9; ++k) Ev[k] = ((e[k] * x + e[9 + k] * y) + e[18 + k] * z1) + e[27 + k];
    double s = 0;
    s += Ev[0] * Ev[0] + Ev[1] * Ev[1] + Ev[2] * Ev[2] + Ev[3] * Ev[3];
    s += Ev[4] * Ev[4] + Ev[5] * Ev[5] + Ev[6] * Ev[6] + Ev[7] * Ev[7];
    s += Ev[8] * Ev[8];
    const double inv_n = 1. / sqrt(s);
    for (int k = 0; k < 9; ++k) Ev[k] *= inv_n;
    ++count;
  }
  return count;
}

/* EMEstimatorCallback::computeError + findInliers' comparison for one match */
static inline float em_error(const double* E, double x1, double y1, double x2, double y2) {
  const double ex0 = E[0] * x1 + E[1] * y1 + E[2] * 1.0;
  const double ex1 = E[3] * x1 + E[4] * y1 + E[5] * 1.0;
  const double ex2 = E[6] * x1 + E[7] * y1 + E[8] * 1.0;
  const double et0 = E[0] * x2 + E[3] * y2 + E[6] * 1.0;
  const double et1 = E[1] * x2 + E[4] * y2 + E[7] * 1.0;
  const double x2tEx1 = x2 * ex0 + y2 * ex1 + 1.0 * ex2;
  const double a = ex0 * ex0, b = ex1 * ex1, c = et0 * et0, d = et1 * et1;
  return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = p > 0 ? p : 0;
  p = p < 1 ? p : 1;
  ep = ep > 0 ? ep : 0;
  ep = ep < 1 ? ep : 1;
  double num = 1 - p > DBL_MIN ? 1 - p : DBL_MIN;
  double denom = 1 - pow(1 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)nearbyint(num / denom); /* cvRound */
}

/* findEssentialMat's normalisation of one coordinate */
void orc_em_normalize(const double* xy, int n, const double K[9], double* out) {
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const double ax = 1. / fx, bx = -cx * ax, ay = 1. / fy, by = -cy * ay;
  for (int i = 0; i < n; ++i) {
    out[2 * i] = xy[2 * i] * ax + bx;
    out[2 * i + 1] = xy[2 * i + 1] * ay + by;
  }
}

/* cv::findEssentialMat(pts1, pts2, K, RANSAC, prob, threshold, mask).  Returns the inlier count of the mask (0 when
 * no model was found: the library leaves the mask uninitialised then; here it is zeroed).  E (9, may be NULL), iters
 * (iterations run, may be NULL), flags (see the header; may be NULL). */
int orc_find_essential_mat(const double* pts1, const double* pts2, int count, const double K[9], double prob,
                           double threshold, int max_iters, uint8_t* mask, double* E_out, int* iters, int* flags) {
  if (iters) *iters = 0;
  if (flags) *flags = 0;
  if (mask) memset(mask, 0, (size_t)(count > 0 ? count : 0));
  if (count < 5) return 0;
  double* p1 = (double*)malloc(sizeof(double) * 2 * (size_t)count);
  double* p2 = (double*)malloc(sizeof(double) * 2 * (size_t)count);
  uint8_t* cur = (uint8_t*)malloc((size_t)count);
  uint8_t* best = (uint8_t*)malloc((size_t)count);
  orc_em_normalize(pts1, count, K, p1);
  orc_em_normalize(pts2, count, K, p2);
  threshold /= (K[0] + K[4]) / 2;
  const float t = (float)(threshold * threshold);
  double models[90], bestE[9] = {0};
  int max_good = 0;
  if (count == 5) {
    const int nm = orc_five_point(p1, p2, models, flags);
    if (nm > 0) {
      max_good = 5;
      memcpy(bestE, models, sizeof(bestE));
      if (mask) memset(mask, 1, 5);
    }
  } else {
    cvrng rng = {~(uint64_t)0};
    int niters = max_iters > 1 ? max_iters : 1, iter;
    memset(best, 0, (size_t)count);
    for (iter = 0; iter < niters; ++iter) {
      int idx[5];
      double s1[10], s2[10];
      for (int i = 0; i < 5; ++i) {
        for (;;) {
          const int v = idx[i] = cvrng_uniform(&rng, 0, count);
          int j = 0;
          for (; j < i; ++j)
            if (v == idx[j]) break;
          if (j == i) break;
        }
        s1[2 * i] = p1[2 * idx[i]], s1[2 * i + 1] = p1[2 * idx[i] + 1];
        s2[2 * i] = p2[2 * idx[i]], s2[2 * i + 1] = p2[2 * idx[i] + 1];
      }
      const int nm = orc_five_point(s1, s2, models, flags);
      for (int m = 0; m < nm; ++m) {
        int good = 0;
        for (int i = 0; i < count; ++i) {
          const int f = em_error(models + 9 * m, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= t;
          cur[i] = (uint8_t)f;
          good += f;
        }
        if (good > (max_good > 4 ? max_good : 4)) {
          uint8_t* sw = cur;
          cur = best;
          best = sw;
          memcpy(bestE, models + 9 * m, sizeof(bestE));
          max_good = good;
          niters = orc_ransac_update_num_iters(prob, (double)(count - good) / count, 5, niters);
        }
      }
    }
    if (iters) *iters = iter;
    if (max_good > 0 && mask) memcpy(mask, best, (size_t)count);
  }
  if (E_out) memcpy(E_out, bestE, sizeof(bestE));
  free(p1);
  free(p2);
  free(cur);
  free(best);
  return max_good;
}

/* the scoring of many pairs (findBestPair's loop body, src/Sfm.cpp:536-563): offsets[n_pairs + 1] into the point
 * arrays; counts[p] = inliers, iters[p] = iterations run; OpenMP over pairs when threads > 1 (bench.py's CPU leg) */
int orc_score_essential_many(int n_pairs, const int32_t* offsets, const double* left_xy, const double* right_xy,
                             const double K[9], double prob, double threshold, int32_t* counts, int32_t* iters,
                             uint8_t* masks, int threads, int32_t* flags_any) {
  int any = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1) reduction(| : any)
  for (int p = 0; p < n_pairs; ++p) {
    const int o = offsets[p], n = offsets[p + 1] - o;
    int it = 0, fl = 0;
    counts[p] = orc_find_essential_mat(left_xy + 2 * (size_t)o, right_xy + 2 * (size_t)o, n, K, prob, threshold, 1000,
                                       masks ? masks + o : NULL, NULL, &it, &fl);
    if (iters) iters[p] = it;
    any |= fl;
  }
  if (flags_any) *flags_any = any;
  return 0;
}

/* ================================================================ findHomographyInliers (reference src/Sfm.cpp:667-689)
 * cv::findHomography(query, train, CV_RANSAC, 0.004 * maxVal, mask) as OpenCV 3.4.1 (calib3d/fundam.cpp, ptsetreg.cpp,
 * core/lapack.cpp) runs it: points converted to float; RANSAC with 4-point samples (getSubset redraws a sample that
 * checkSubset rejects: the last point collinear with two others -- float differences, double products --, or the
 * orientation of one of the four triples not preserved), confidence 0.995, at most 2000 iterations; runKernel =
 * normalised DLT: L^T L (upper triangle accumulated, completeSymm), cv::eigen = JacobiImpl_<double> (pivot = the largest
 * off-diagonal element, found through the row / column maximum indices; hypot-based rotation; eps = DBL_EPSILON,
 * 30 n^2 rotations at most; eigenvalues sorted descending with their vector rows), H0 = the last row, H = invHnorm H0
 * Hnorm2 by the 3 x 3 products' left-to-right sums, scaled by 1 / H[2][2]; the error and the threshold test in FLOAT
 * arithmetic; the mask returned is the RANSAC mask (the refit on the inliers and the LM refinement change H only).
 * PARITY UNPINNED like the rest of this file; one more unknown here: an OpenCV built WITH Eigen routes cv::eigen through
 * Eigen::SelfAdjointEigenSolver instead of its own Jacobi. */
static void jacobi_eigen9(double A[9][9], double W[9], double V[9][9]) {
  const int n = 9;
  const double eps = DBL_EPSILON;
  int indR[9], indC[9];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int k = 0; k < n; ++k) {
    W[k] = A[k][k];
    if (k < n - 1) {
      int m = k + 1;
      double mv = fabs(A[k][m]);
      for (int i = k + 2; i < n; ++i) {
        const double val = fabs(A[k][i]);
        if (mv < val) mv = val, m = i;
      }
      indR[k] = m;
    }
    if (k > 0) {
      int m = 0;
      double mv = fabs(A[0][k]);
      for (int i = 1; i < k; ++i) {
        const double val = fabs(A[i][k]);
        if (mv < val) mv = val, m = i;
      }
      indC[k] = m;
    }
  }
  const int max_iters = n * n * 30;
  for (int iters = 0; iters < max_iters; ++iters) {
    int k = 0;
    double mv = fabs(A[0][indR[0]]);
    for (int i = 1; i < n - 1; ++i) {
      const double val = fabs(A[i][indR[i]]);
      if (mv < val) mv = val, k = i;
    }
    int l = indR[k];
    for (int i = 1; i < n; ++i) {
      const double val = fabs(A[indC[i]][i]);
      if (mv < val) mv = val, k = indC[i], l = i;
    }
    const double p = A[k][l];
    if (fabs(p) <= eps) break;
    const double y = (W[l] - W[k]) * 0.5;
    double t = fabs(y) + hypot(p, y);
    double s = hypot(p, t);
    const double c = t / s;
    s = p / s;
    t = (p / t) * p;
    if (y < 0) s = -s, t = -t;
    A[k][l] = 0;
    W[k] -= t;
    W[l] += t;
    double a0, b0;
#define ORC_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
    for (int i = 0; i < k; ++i) ORC_ROT(A[i][k], A[i][l]);
    for (int i = k + 1; i < l; ++i) ORC_ROT(A[k][i], A[i][l]);
    for (int i = l + 1; i < n; ++i) ORC_ROT(A[k][i], A[l][i]);
    for (int i = 0; i < n; ++i) ORC_ROT(V[k][i], V[l][i]);
#undef ORC_ROT
    for (int j = 0; j < 2; ++j) {
      const int idx = j == 0 ? k : l;
      if (idx < n - 1) {
        int m = idx + 1;
        double mx = fabs(A[idx][m]);
        for (int i = idx + 2; i < n; ++i) {
          const double val = fabs(A[idx][i]);
          if (mx < val) mx = val, m = i;
        }
        indR[idx] = m;
      }
      if (idx > 0) {
        int m = 0;
        double mx = fabs(A[0][idx]);
        for (int i = 1; i < idx; ++i) {
          const double val = fabs(A[i][idx]);
          if (mx < val) mx = val, m = i;
        }
        indC[idx] = m;
      }
    }
  }
  for (int k = 0; k < n - 1; ++k) {
    int m = k;
    for (int i = k + 1; i < n; ++i)
      if (W[m] < W[i]) m = i;
    if (k != m) {
      double t = W[m];
      W[m] = W[k];
      W[k] = t;
      for (int i = 0; i < n; ++i) {
        t = V[m][i];
        V[m][i] = V[k][i];
        V[k][i] = t;
      }
    }
  }
}

/* HomographyEstimatorCallback::runKernel: M -> m, `count` float correspondences; H: 9 doubles.  Returns 0 when degenerate. */
int orc_homography_kernel(const float* M, const float* m, int count, double* H) {
  double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
  for (int i = 0; i < count; ++i) {
    cmx += m[2 * i]; cmy += m[2 * i + 1];
    cMx += M[2 * i]; cMy += M[2 * i + 1];
  }
  cmx /= count; cmy /= count; cMx /= count; cMy /= count;
  for (int i = 0; i < count; ++i) {
    smx += fabs(m[2 * i] - cmx); smy += fabs(m[2 * i + 1] - cmy);
    sMx += fabs(M[2 * i] - cMx); sMy += fabs(M[2 * i + 1] - cMy);
  }
  if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
  smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
  double A[9][9], W[9], V[9][9];
  memset(A, 0, sizeof A);
  for (int i = 0; i < count; ++i) {
    const double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
    const double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
    const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
    const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
    for (int j = 0; j < 9; ++j)
      for (int k = j; k < 9; ++k) A[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
  }
  for (int j = 0; j < 9; ++j)
    for (int k = 0; k < j; ++k) A[j][k] = A[k][j];
  jacobi_eigen9(A, W, V);
  const double* h0 = V[8];
  const double inv[9] = {1.0 / smx, 0, cmx, 0, 1.0 / smy, cmy, 0, 0, 1};
  const double hn2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
  double t[9], r[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) t[3 * a + b] = inv[3 * a] * h0[b] + inv[3 * a + 1] * h0[3 + b] + inv[3 * a + 2] * h0[6 + b];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) r[3 * a + b] = t[3 * a] * hn2[b] + t[3 * a + 1] * hn2[3 + b] + t[3 * a + 2] * hn2[6 + b];
  const double sc = 1.0 / r[8];
  for (int k = 0; k < 9; ++k) H[k] = r[k] * sc;
  return 1;
}

static int h_have_collinear(const float (*p)[2], int count) {
  const int i = count - 1;
  for (int j = 0; j < i; ++j) {
    const double dx1 = p[j][0] - p[i][0], dy1 = p[j][1] - p[i][1]; /* (float differences, widened) */
    for (int k = 0; k < j; ++k) {
      const double dx2 = p[k][0] - p[i][0], dy2 = p[k][1] - p[i][1];
      if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
    }
  }
  return 0;
}
static double h_det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static int h_check_subset(const float (*s1)[2], const float (*s2)[2]) {
  if (h_have_collinear(s1, 4) || h_have_collinear(s2, 4)) return 0;
  static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
  int negative = 0;
  for (int i = 0; i < 4; ++i) {
    const int* t = tt[i];
    const double A[9] = {s1[t[0]][0], s1[t[0]][1], 1., s1[t[1]][0], s1[t[1]][1], 1., s1[t[2]][0], s1[t[2]][1], 1.};
    const double B[9] = {s2[t[0]][0], s2[t[0]][1], 1., s2[t[1]][0], s2[t[1]][1], 1., s2[t[2]][0], s2[t[2]][1], 1.};
    negative += h_det3(A) * h_det3(B) < 0;
  }
  return negative == 0 || negative == 4;
}
static inline float h_error(const float* Hf, float Mx, float My, float mx, float my) {
  const float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
  const float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - mx;
  const float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - my;
  return dx * dx + dy * dy;
}

/* cv::findHomography(pts1, pts2, RANSAC, threshold, mask, max_iters, confidence): the RANSAC inlier count, the mask and
 * the iterations run.  pts: doubles (the reference hands over Point2d), converted to float as the library does. */
int orc_find_homography(const double* pts1, const double* pts2, int count, double threshold, double confidence, int max_iters,
                        uint8_t* mask, int* iters) {
  if (iters) *iters = 0;
  if (mask) memset(mask, 0, (size_t)(count > 0 ? count : 0));
  if (threshold <= 0) threshold = 3;
  if (count < 4) return 0;
  float* m1 = (float*)malloc(sizeof(float) * 2 * (size_t)count);
  float* m2 = (float*)malloc(sizeof(float) * 2 * (size_t)count);
  uint8_t* cur = (uint8_t*)malloc((size_t)count);
  uint8_t* best = (uint8_t*)calloc((size_t)count, 1);
  for (int i = 0; i < 2 * count; ++i) m1[i] = (float)pts1[i], m2[i] = (float)pts2[i];
  int max_good = 0;
  double H[9];
  if (count == 4) {
    if (orc_homography_kernel(m1, m2, 4, H)) {
      max_good = 4;
      if (mask) memset(mask, 1, 4);
    }
  } else {
    const float t = (float)(threshold * threshold);
    cvrng rng = {~(uint64_t)0};
    int niters = max_iters > 1 ? max_iters : 1, iter;
    for (iter = 0; iter < niters; ++iter) {
      int idx[4], found = 0;
      float s1[4][2], s2[4][2];
      for (int attempt = 0; attempt < 10000 && !found; ++attempt) {
        for (int i = 0; i < 4;) {
          const int v = cvrng_uniform(&rng, 0, count);
          int j = 0;
          for (; j < i; ++j)
            if (idx[j] == v) break;
          if (j < i) continue;
          idx[i++] = v;
        }
        for (int k = 0; k < 4; ++k) {
          s1[k][0] = m1[2 * idx[k]], s1[k][1] = m1[2 * idx[k] + 1];
          s2[k][0] = m2[2 * idx[k]], s2[k][1] = m2[2 * idx[k] + 1];
        }
        found = h_check_subset(s1, s2);
      }
      if (!found) break; /* (iteration 0: run() returns false; later: the loop ends) */
      if (!orc_homography_kernel(&s1[0][0], &s2[0][0], 4, H)) continue;
      float Hf[9];
      for (int k = 0; k < 9; ++k) Hf[k] = (float)H[k];
      int good = 0;
      for (int i = 0; i < count; ++i) {
        const int f = h_error(Hf, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= t;
        cur[i] = (uint8_t)f;
        good += f;
      }
      if (good > (max_good > 3 ? max_good : 3)) {
        uint8_t* sw = cur;
        cur = best;
        best = sw;
        max_good = good;
        niters = orc_ransac_update_num_iters(confidence, (double)(count - good) / count, 4, niters);
      }
    }
    if (iters) *iters = iter;
    if (max_good > 0 && mask) memcpy(mask, best, (size_t)count);
  }
  free(m1);
  free(m2);
  free(cur);
  free(best);
  return max_good;
}

