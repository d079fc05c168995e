// score.hip -- the scoring half of StructFromMotion::findBestPair (reference src/Sfm.cpp:536-563) on gfx950:
// per image pair the inlier count (and mask) of cv::findEssentialMat(left, right, K, RANSAC, prob, threshold).
//
// OpenCV 3.4.1's RANSAC (calib3d/ptsetreg.cpp) is sequential only in its bookkeeping: the sample of iteration k
// depends on the match count alone (cv::RNG restarts at (uint64)-1 in every call), the models of a sample and their
// inlier counts on nothing else, and the adaptive iteration limit on the counts in order.  So the host generates the
// sample table per distinct match count, the device solves every (pair, iteration) sample (five-point problem, one
// thread each) and counts the inliers of every model (one workgroup per (pair, iteration)), in chunks of iterations,
// and the host replays ptsetreg's update rule over the counts in order -- with the host's libm, as the reference
// does -- until every pair has reached its own iteration limit.
//
// The five-point solver follows OpenCV 3.4.1's EMEstimatorCallback::runKernel step by step (round 3; rounds 1-2 took
// another route to the same matrices): the null space from the library's one-sided Jacobi SVD with its RNG-filled
// singular vectors, the 10 x 20 constraint matrix in Nister's monomial order, Mat::inv() (LU against the identity)
// times the right half, the 3 x 13 matrix B, the tenth-degree polynomial, cv::solvePoly's Durand-Kerner iteration, the
// library's |imag| <= 1e-10 test, (x, y) from SVD::solveZ, E scaled to unit norm, models in root order.  What of the
// library cannot be reproduced to the last bit (its generated coefficient sums) is listed in the header of the
// checker (the CPU restatement of the same route: test infrastructure, never linked here); parity is unpinned (OpenCV is not in
// the image).  sfmhip_score_last_flags reports samples that reached a corner of solvePoly that is not restated.
#include "common.h"
#include "hypot_glibc.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <map>
#include <type_traits>
#include <vector>

namespace {

constexpr int MAX_MODELS = 10;

// ---------------------------------------------------------------- polynomials in (x, y, z)
// degree 1: [x, y, z, 1]; degree 2: [x2, xy, xz, y2, yz, z2, x, y, z, 1];
// degree 3 (Nister's order): [x3, y3, x2y, xy2, x2z, x2, y2z, y2, xyz, xy | xz2, xz, x, yz2, yz, y, z3, z2, z, 1]
__constant__ int T11[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
__constant__ int T21[10][4] = {{0, 2, 4, 5}, {2, 3, 8, 9}, {4, 8, 10, 11}, {3, 1, 6, 7}, {8, 6, 13, 14},
                               {10, 13, 16, 17}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};

__device__ void mul11(const double* a, const double* b, double* out /*10, accumulated*/, double s) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[T11[i][j]] += s * a[i] * b[j];
}
__device__ void mul21(const double* a /*10*/, const double* b /*4*/, double* out /*20, accumulated*/, double s) {
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 4; ++j) out[T21[i][j]] += s * a[i] * b[j];
}
// ---- cv::RNG (multiply with carry), as JacobiSVDImpl_ seeds it
struct DevRng {
  unsigned long long state;
  __device__ unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
};

// ---- core/lapack.cpp JacobiSVDImpl_<double>: one-sided Jacobi on the N rows (length M, stride LDA) of At; rows N..N1-1
// (and rows whose singular value is <= DBL_MIN) filled from RNG(0x12345678) sign vectors, orthogonalised twice against
// the rows before them.  Vt (N x N) accumulates the rotations.  Operation for operation the restatement in
// the CPU restatement used as checker (test infrastructure), which cites the library.
template <int M, int N, int N1, int LDA>
__device__ void jacobi_svd(double* At, double* W, double* Vt) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  const int max_iter = M > 30 ? M : 30;
  for (int i = 0; i < N; ++i) {
    double sd = 0;
    for (int k = 0; k < M; ++k) {
      const double t = At[i * LDA + k];
      sd += t * t;
    }
    W[i] = sd;
    for (int k = 0; k < N; ++k) Vt[i * N + k] = 0;
    Vt[i * N + i] = 1;
  }
  for (int iter = 0; iter < max_iter; ++iter) {
    bool changed = false;
    for (int i = 0; i < N - 1; ++i)
      for (int j = i + 1; j < N; ++j) {
        double *Ai = At + i * LDA, *Aj = At + j * LDA;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < M; ++k) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = sfm_hypot(p, beta);  // (the host libm's hypot, bit for bit: hypot_glibc.h)
        double c, sn;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          sn = sqrt(delta / gamma);
          c = p / (gamma * sn * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          sn = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < M; ++k) {
          const double t0 = c * Ai[k] + sn * Aj[k];
          const double t1 = -sn * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        double *Vi = Vt + i * N, *Vj = Vt + j * N;
        for (int k = 0; k < N; ++k) {
          const double t0 = c * Vi[k] + sn * Vj[k];
          const double t1 = -sn * Vi[k] + c * Vj[k];
          Vi[k] = t0;
          Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < N; ++i) {
    double sd = 0;
    for (int k = 0; k < M; ++k) {
      const double t = At[i * LDA + k];
      sd += t * t;
    }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < N - 1; ++i) {
    int j = i;
    for (int k = i + 1; k < N; ++k)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double t = W[i];
      W[i] = W[j];
      W[j] = t;
      for (int k = 0; k < M; ++k) {
        t = At[i * LDA + k];
        At[i * LDA + k] = At[j * LDA + k];
        At[j * LDA + k] = t;
      }
      for (int k = 0; k < N; ++k) {
        t = Vt[i * N + k];
        Vt[i * N + k] = Vt[j * N + k];
        Vt[j * N + k] = t;
      }
    }
  }
  DevRng rng{0x12345678ull};
  for (int i = 0; i < N1; ++i) {
    double sd = i < N ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
      const double val0 = 1. / M;
      for (int k = 0; k < M; ++k) At[i * LDA + k] = (rng.next() & 256) != 0 ? val0 : -val0;
      for (int it2 = 0; it2 < 2; ++it2)
        for (int j = 0; j < i; ++j) {
          sd = 0;
          for (int k = 0; k < M; ++k) sd += At[i * LDA + k] * At[j * LDA + k];
          double asum = 0;
          for (int k = 0; k < M; ++k) {
            const double t = At[i * LDA + k] - sd * At[j * LDA + k];
            At[i * LDA + k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < M; ++k) At[i * LDA + k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < M; ++k) {
        const double t = At[i * LDA + k];
        sd += t * t;
      }
      sd = sqrt(sd);
    }
    const double sc = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < M; ++k) At[i * LDA + k] *= sc;
  }
}

// ---- hal::LU64f: A (10 x 10) against 10 right-hand sides b; false when a pivot is below 100 DBL_EPSILON
__device__ bool lu_solve10(double* A, double* b) {
  constexpr int m = 10, n = 10;
  const double eps = DBL_EPSILON * 100;
  for (int i = 0; i < m; ++i) {
    int k = i;
    for (int j = i + 1; j < m; ++j)
      if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
    if (fabs(A[k * m + i]) < eps) return false;
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
      double sum = b[i * n + j];
      for (int k = i + 1; k < m; ++k) sum -= A[i * m + k] * b[k * n + j];
      b[i * n + j] = sum / A[i * m + i];
    }
  return true;
}

// ---- cv::solvePoly for real coefficients c[0..10] ascending: Durand-Kerner, the roots updated in sequence, start
// (1 + i)^k, 300 iterations or until no root moves.  flags: bit 0 = two iterates coincided (the library's
// num_same_root branch, not restated: the zero factor is skipped), bit 1 = leading coefficients <= DBL_EPSILON down to
// a degree n of 5..9 (the library then returns uninitialised memory as the missing roots: here they are marked
// non-real).  For n <= 4 the missing roots are what the library's work buffer holds there, the coefficient pair
// (c[2n], c[2n+1]) -- a singular sample (Mat::inv() = 0, all coefficients 0, n = 1) so gets root 0 = NaN and nine
// roots (0, 0), i.e. nine times the model E = W, which RANSAC scores.
struct Cplx {
  double re, im;
};
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return Cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cplx cdiv(Cplx a, Cplx b) {
  const double t = 1. / (b.re * b.re + b.im * b.im);
  return Cplx{(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}
__device__ void solve_poly10(const double* c, Cplx* roots, int* flags) {
  int n = 10;
  for (; n > 1; --n)
    if (fabs(c[n]) + 0.0 > DBL_EPSILON) break;
  if (n < 10 && 2 * n + 1 > 10) *flags |= 2;
  Cplx p{1, 0};
  const Cplx r{1, 1};
  for (int i = 0; i < n; ++i) {
    roots[i] = p;
    p = cmul(p, r);
  }
  // one pass over the roots R[0 .. nn) (nn = N when N > 0: compile-time, everything unrolled and in registers)
  auto sweep = [&](auto NN, Cplx* R) {
    constexpr int N = decltype(NN)::value;
    const int nn = N ? N : n;
    double max_diff = 0;
#pragma unroll
    for (int i = 0; i < nn; ++i) {
      const Cplx pi = R[i];
      Cplx num{c[nn], 0}, denom{c[nn], 0};
#pragma unroll
      for (int j = 0; j < nn; ++j) {
        num = cmul(num, pi);
        num.re += c[nn - j - 1];
        num.im += 0.0;
        if (j != i) {
          const Cplx d{pi.re - R[j].re, pi.im - R[j].im};
          if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
          else *flags |= 1;
        }
      }
      num = cdiv(num, denom);
      R[i] = Cplx{pi.re - num.re, pi.im - num.im};
      max_diff = fmax(max_diff, sqrt(num.re * num.re + num.im * num.im));
    }
    return max_diff;
  };
  if (n == 10) {
    // (the degree is ten unless leading coefficients vanish.  A copy of the roots indexed by compile-time constants
    // only: it lives in registers -- the caller's array, indexed by the runtime n elsewhere, sits in scratch memory,
    // and 300 x 10 x 10 reads of it per sample were the whole kernel's time)
    Cplx rr[10];
    {
      Cplx q{1, 0};
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        rr[i] = q;
        q = cmul(q, r);
      }
    }
    for (int iter = 0; iter < 300; ++iter)
      if (sweep(std::integral_constant<int, 10>{}, rr) <= 0) break;
#pragma unroll
    for (int i = 0; i < 10; ++i) roots[i] = rr[i];
  } else {
    for (int iter = 0; iter < 300; ++iter)
      if (sweep(std::integral_constant<int, 0>{}, roots) <= 0) break;
  }
  for (int i = 0; i < n; ++i)
    if (fabs(roots[i].im) < 1e-100) roots[i].im = 0;
  for (int k = n; k < 10; ++k) roots[k] = 2 * n + 1 <= 10 ? Cplx{c[2 * n], c[2 * n + 1]} : Cplx{0, 1};
}

// EMEstimatorCallback::runKernel (OpenCV 3.4.1 calib3d/five-point.cpp) for five normalised correspondences
// (q2^T E q1 = 0): up to ten row-major 3 x 3 models, unit Frobenius norm, in the order of solvePoly's roots.  The
// checker (a CPU restatement, test infrastructure) restates the same steps with the host's libm; see its header for what of the
// library is reproduced and what is not.
// part 1 (register- and scratch-heavy): the null space, the constraints, the elimination, B and the polynomial.
// work: e[36] (X, Y, Z, W row-major) | b[39] | c[11]
constexpr int FP_WORK = 86;
__device__ void five_point_setup(const double (*q1)[2], const double (*q2)[2], double* __restrict__ work) {
  double Q[9 * 9];
  for (int k = 0; k < 81; ++k) Q[k] = 0.0;
  for (int i = 0; i < 5; ++i) {
    const double x1 = q1[i][0], y1 = q1[i][1], x2 = q2[i][0], y2 = q2[i][1];
    double* r = Q + 9 * i;
    r[0] = x1 * x2; r[1] = y1 * x2; r[2] = x2 * 1.0; r[3] = x1 * y2; r[4] = y1 * y2; r[5] = y2 * 1.0;
    r[6] = x1 * 1.0; r[7] = y1 * 1.0; r[8] = 1.0;
  }
  double W5[5], V5[25];
  jacobi_svd<9, 5, 9, 9>(Q, W5, V5);
  const double* e = Q + 45;  // rows 5..8 of Vt: X, Y, Z, W
  // ---- the ten cubic constraints: det(E) = 0, 2 E E^T E - tr(E E^T) E = 0, E = x X + y Y + z Z + W
  double E1[9][4];
  for (int c = 0; c < 9; ++c)
    for (int k = 0; k < 4; ++k) E1[c][k] = e[k * 9 + c];
  double M[10][20];
  for (int r = 0; r < 10; ++r)
    for (int c = 0; c < 20; ++c) M[r][c] = 0.0;
  {
    double m2[10];
    const int minors[3][5] = {{0, 4, 8, 5, 7}, {1, 5, 6, 3, 8}, {2, 3, 7, 4, 6}};  // E0k * (Ea Eb - Ec Ed)
    for (int t = 0; t < 3; ++t) {
      for (int c = 0; c < 10; ++c) m2[c] = 0.0;
      mul11(E1[minors[t][1]], E1[minors[t][2]], m2, 1.0);
      mul11(E1[minors[t][3]], E1[minors[t][4]], m2, -1.0);
      mul21(m2, E1[minors[t][0]], M[0], 1.0);
    }
  }
  {
    double EEt[3][3][10], tr[10];
    for (int a = 0; a < 3; ++a)
      for (int b = a; b < 3; ++b) {
        for (int c = 0; c < 10; ++c) EEt[a][b][c] = 0.0;
        for (int k = 0; k < 3; ++k) mul11(E1[3 * a + k], E1[3 * b + k], EEt[a][b], 1.0);
        if (b != a)
          for (int c = 0; c < 10; ++c) EEt[b][a][c] = EEt[a][b][c];
      }
    for (int c = 0; c < 10; ++c) tr[c] = EEt[0][0][c] + EEt[1][1][c] + EEt[2][2][c];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double* row = M[1 + 3 * a + b];
        for (int k = 0; k < 3; ++k) mul21(EEt[a][k], E1[3 * k + b], row, 2.0);
        mul21(tr, E1[3 * a + b], row, -1.0);
      }
  }
  // ---- A = A.colRange(0, 10).inv() * A.colRange(10, 20): LU against the identity, then the product
  double L[100], inv[100];
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) {
      L[i * 10 + j] = M[i][j];
      inv[i * 10 + j] = i == j ? 1.0 : 0.0;
    }
  if (!lu_solve10(L, inv))
    for (int k = 0; k < 100; ++k) inv[k] = 0.0;
  // rows 4..9 of the product are all the 3 x 13 matrix B takes
  double b[39];
  for (int i = 0; i < 3; ++i) {
    double a1[10], a2[10];
    for (int j = 0; j < 10; ++j) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k < 10; ++k) {
        s1 += inv[(2 * i + 4) * 10 + k] * M[k][10 + j];
        s2 += inv[(2 * i + 5) * 10 + k] * M[k][10 + j];
      }
      a1[j] = s1;
      a2[j] = s2;
    }
    double row1[13], row2[13];
    for (int k = 0; k < 13; ++k) row1[k] = row2[k] = 0.0;
    for (int k = 0; k < 3; ++k) row1[1 + k] = a1[k], row1[5 + k] = a1[3 + k], row2[k] = a2[k], row2[4 + k] = a2[3 + k];
    for (int k = 0; k < 4; ++k) row1[9 + k] = a1[6 + k], row2[8 + k] = a2[6 + k];
    for (int k = 0; k < 13; ++k) b[13 * i + k] = row1[k] - row2[k];
  }
  // ---- det B(z): row i = [P_i (deg 3) | Q_i (deg 3) | R_i (deg 4)], highest power first; c[k] = coefficient of z^k
  double c[11];
  for (int k = 0; k < 11; ++k) c[k] = 0.0;
  {
    const int perm[6][3] = {{0, 1, 2}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}, {0, 2, 1}, {1, 0, 2}};
    const double sign[6] = {1, 1, 1, -1, -1, -1};
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
  }
  for (int k = 0; k < 36; ++k) work[k] = e[k];
  for (int k = 0; k < 39; ++k) work[36 + k] = b[k];
  for (int k = 0; k < 11; ++k) work[75 + k] = c[k];
}

// part 2 (few registers: runs at four times part 1's occupancy, and solvePoly's 300 Durand-Kerner iterations are most
// of a sample's time): the roots, (x, y) per real root, the models
// WS = stride of work's entries (1: a thread's own array; 64: the sample-major blocks of score_setup); the models go
// straight to Eout (9 doubles each, contiguous)
template <int WS>
__device__ int five_point_models(const double* __restrict__ work, double* __restrict__ Eout, int* flags) {
  double c[11];
  for (int k = 0; k < 11; ++k) c[k] = work[(size_t)(75 + k) * WS];
  Cplx roots[10];
  solve_poly10(c, roots, flags);
  int n = 0;
  for (int i = 0; i < 10; ++i) {
    if (fabs(roots[i].im) > 1e-10) continue;
    const double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
    // SVD::solveZ(B(z)): JacobiSVD on the transpose, the row of Vt of the smallest singular value
    double At[9], W3[3], Vt[9];
    for (int j = 0; j < 3; ++j) {
      const double* br = work + (size_t)(36 + j * 13) * WS;
      At[0 * 3 + j] = br[0 * WS] * z3 + br[1 * WS] * z2 + br[2 * WS] * z1 + br[3 * WS];
      At[1 * 3 + j] = br[4 * WS] * z3 + br[5 * WS] * z2 + br[6 * WS] * z1 + br[7 * WS];
      At[2 * 3 + j] = br[8 * WS] * z4 + br[9 * WS] * z3 + br[10 * WS] * z2 + br[11 * WS] * z1 + br[12 * WS];
    }
    jacobi_svd<3, 3, 3, 3>(At, W3, Vt);
    if (fabs(Vt[8]) < 1e-10) continue;
    const double x = Vt[6] / Vt[8], y = Vt[7] / Vt[8];
    double Ev[9];
    for (int k = 0; k < 9; ++k)
      Ev[k] = ((work[(size_t)k * WS] * x + work[(size_t)(9 + k) * WS] * y) + work[(size_t)(18 + k) * WS] * z1) + work[(size_t)(27 + k) * WS];
    double sq = 0;
    sq += Ev[0] * Ev[0] + Ev[1] * Ev[1] + Ev[2] * Ev[2] + Ev[3] * Ev[3];
    sq += Ev[4] * Ev[4] + Ev[5] * Ev[5] + Ev[6] * Ev[6] + Ev[7] * Ev[7];
    sq += Ev[8] * Ev[8];
    const double inv_n = 1. / sqrt(sq);
    for (int k = 0; k < 9; ++k) Eout[9 * n + k] = Ev[k] * inv_n;
    ++n;
  }
  return n;
}

__device__ int five_point(const double (*q1)[2], const double (*q2)[2], double (*Eout)[9], int* flags) {
  double work[FP_WORK];
  five_point_setup(q1, q2, work);
  return five_point_models<1>(work, &Eout[0][0], flags);
}

// ---------------------------------------------------------------- kernels
// findEssentialMat's normalisation: the MatExpr (col - c) / f is evaluated as col * (1 / f) + (-c * (1 / f))
__global__ void score_normalize(const double* __restrict__ xy, double* __restrict__ out, long long n2, double ax, double bx,
                                double ay, double by) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  out[i] = (i & 1) ? xy[i] * ay + by : xy[i] * ax + bx;
}

struct ScoreJob {  // one active pair of a chunk
  int off;         // first match of the pair in the point arrays
  int count;       // matches
  int samp;        // first row of its sample table (5 indices per iteration) for this chunk
};

// thread (job, iteration of the chunk): part 1 of the sample's solve -> work (FP_WORK doubles per sample, sample-major
// in blocks of 64 so that a wave's stores and loads are coalesced: work[(t / 64) * 64 * FP_WORK + k * 64 + t % 64])
__global__ __launch_bounds__(64) void score_setup(const ScoreJob* __restrict__ jobs, int n_jobs, int chunk,
                                                  const int* __restrict__ samples, const double* __restrict__ p1,
                                                  const double* __restrict__ p2, double* __restrict__ work) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_jobs * chunk) return;
  const int j = t / chunk, it = t - j * chunk;
  const ScoreJob jb = jobs[j];
  const int* s = samples + ((size_t)jb.samp + it) * 5;
  double q1[5][2], q2[5][2];
  for (int k = 0; k < 5; ++k) {
    const size_t m = (size_t)jb.off + s[k];
    q1[k][0] = p1[2 * m];
    q1[k][1] = p1[2 * m + 1];
    q2[k][0] = p2[2 * m];
    q2[k][1] = p2[2 * m + 1];
  }
  double w[FP_WORK];
  five_point_setup(q1, q2, w);
  double* out = work + (size_t)(t >> 6) * (64 * FP_WORK) + (t & 63);
  for (int k = 0; k < FP_WORK; ++k) out[(size_t)k * 64] = w[k];
}

// thread (job, iteration of the chunk): part 2 -> the sample's models
__global__ __launch_bounds__(256, 4) void score_roots(int n_samples, const double* __restrict__ work, double* __restrict__ models,
                                                   int* __restrict__ n_models) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_samples) return;
  int flags = 0;
  const int n = five_point_models<64>(work + (size_t)(t >> 6) * (64 * FP_WORK) + (t & 63), models + (size_t)t * (MAX_MODELS * 9), &flags);
  n_models[t] = n | (flags << 8);
}

// EMEstimatorCallback::computeError + findInliers for one correspondence
__device__ __forceinline__ bool is_inlier(const double* E, double x1, double y1, double x2, double y2, float t) {
  const double ex0 = E[0] * x1 + E[1] * y1 + E[2] * 1.0;
  const double ex1 = E[3] * x1 + E[4] * y1 + E[5] * 1.0;
  const double ex2 = E[6] * x1 + E[7] * y1 + E[8] * 1.0;
  const double et0 = E[0] * x2 + E[3] * y2 + E[6] * 1.0;
  const double et1 = E[1] * x2 + E[4] * y2 + E[7] * 1.0;
  const double x2tEx1 = x2 * ex0 + y2 * ex1 + 1.0 * ex2;
  const double a = ex0 * ex0, b = ex1 * ex1, c = et0 * et0, d = et1 * et1;
  const float err = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
  return err <= t;
}

// workgroup (job, iteration): inlier count of every model of the sample
__global__ __launch_bounds__(256) void score_count(const ScoreJob* __restrict__ jobs, int chunk, const double* __restrict__ p1,
                                                   const double* __restrict__ p2, const double* __restrict__ models,
                                                   const int* __restrict__ n_models, float t, int* __restrict__ counts) {
  __shared__ double sE[MAX_MODELS * 9];
  __shared__ int s_cnt[MAX_MODELS];
  const int j = blockIdx.x / chunk;
  const int slot = blockIdx.x;
  const ScoreJob jb = jobs[j];
  const int nm = n_models[slot] & 0xff;
  if (threadIdx.x < MAX_MODELS) s_cnt[threadIdx.x] = 0;
  if ((int)threadIdx.x < nm * 9) sE[threadIdx.x] = models[(size_t)slot * (MAX_MODELS * 9) + threadIdx.x];
  __syncthreads();
  if (nm > 0) {
    int cnt[MAX_MODELS];
    for (int m = 0; m < MAX_MODELS; ++m) cnt[m] = 0;
    for (int i = threadIdx.x; i < jb.count; i += 256) {
      const size_t k = (size_t)jb.off + i;
      const double x1 = p1[2 * k], y1 = p1[2 * k + 1], x2 = p2[2 * k], y2 = p2[2 * k + 1];
      for (int m = 0; m < nm; ++m) cnt[m] += is_inlier(sE + 9 * m, x1, y1, x2, y2, t) ? 1 : 0;
    }
    for (int m = 0; m < nm; ++m) {
      int v = cnt[m];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[m], v);
    }
  }
  __syncthreads();
  if (threadIdx.x < MAX_MODELS) counts[(size_t)slot * MAX_MODELS + threadIdx.x] = (int)threadIdx.x < nm ? s_cnt[threadIdx.x] : 0;
}

// the mask of each pair's best model: workgroup per pair
__global__ __launch_bounds__(256) void score_mask(const int* __restrict__ offsets, const double* __restrict__ p1,
                                                  const double* __restrict__ p2, const double* __restrict__ best_E,
                                                  const unsigned char* __restrict__ has, float t, unsigned char* __restrict__ mask) {
  const int pr = blockIdx.x;
  const int o = offsets[pr], n = offsets[pr + 1] - o;
  __shared__ double sE[9];
  if (threadIdx.x < 9) sE[threadIdx.x] = best_E[(size_t)pr * 9 + threadIdx.x];
  __syncthreads();
  const int h = has[pr];  // 0: no model (all zero); 1: a RANSAC model; 2: exactly five matches (all one)
  for (int i = threadIdx.x; i < n; i += 256) {
    const size_t k = (size_t)o + i;
    mask[k] = h == 2 ? 1 : h == 1 ? (is_inlier(sE, p1[2 * k], p1[2 * k + 1], p2[2 * k], p2[2 * k + 1], t) ? 1 : 0) : 0;
  }
}

// ---------------------------------------------------------------- host: cv::RNG, the sample tables, the update rule
struct CvRng {
  unsigned long long state = 0xFFFFFFFFFFFFFFFFull;  // RNG rng((uint64)-1)
  unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
struct SampleStream {  // the samples of one match count, generated on demand
  CvRng rng;
  std::vector<int> idx;  // 5 per iteration
  void extend(int count, int n_iters) {
    while ((int)idx.size() < 5 * n_iters) {
      int s[5];
      for (int i = 0; i < 5;) {
        const int v = rng.uniform(0, count);
        int j = 0;
        for (; j < i; ++j)
          if (s[j] == v) break;
        if (j < i) continue;  // drawn before: again
        s[i++] = v;
      }
      idx.insert(idx.end(), s, s + 5);
    }
  }
};
int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = std::max(p, 0.0);
  p = std::min(p, 1.0);
  ep = std::max(ep, 0.0);
  ep = std::min(ep, 1.0);
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::nearbyint(num / denom);  // cvRound
}


// ================================================================ homography (findHomographyInliers, src/Sfm.cpp:667-689)
// cv::findHomography(query, train, RANSAC, threshold, mask) as OpenCV 3.4.1 (calib3d/fundam.cpp) runs it: FLOAT
// points; 4-point samples, a sample drawn again when checkSubset rejects it (host, below); model = normalised DLT,
// the eigenvector of the smallest eigenvalue of L^T L (cv::eigen's Jacobi, restated) scaled to H[2][2] = 1; the error and the
// threshold test in float arithmetic; the mask is the RANSAC mask (the refit + LM refinement change H only).
__device__ int homography_kernel(const float (*M)[2], const float (*m)[2], double* H) {
  const int count = 4;
  double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
  for (int i = 0; i < count; ++i) {
    cmx += m[i][0]; cmy += m[i][1];
    cMx += M[i][0]; cMy += M[i][1];
  }
  cmx /= count; cmy /= count; cMx /= count; cMy /= count;
  for (int i = 0; i < count; ++i) {
    smx += fabs(m[i][0] - cmx); smy += fabs(m[i][1] - cmy);
    sMx += fabs(M[i][0] - cMx); sMy += fabs(M[i][1] - cMy);
  }
  if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
  smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
  double A[9][9], V[9][9];
  for (int j = 0; j < 9; ++j)
    for (int k = 0; k < 9; ++k) A[j][k] = 0.0, V[j][k] = j == k ? 1.0 : 0.0;
  for (int i = 0; i < count; ++i) {
    const double x = (m[i][0] - cmx) * smx, y = (m[i][1] - cmy) * smy;
    const double X = (M[i][0] - cMx) * sMx, Y = (M[i][1] - cMy) * sMy;
    const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
    const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
    for (int j = 0; j < 9; ++j)
      for (int k = j; k < 9; ++k) A[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
  }
  for (int j = 0; j < 9; ++j)
    for (int k = 0; k < j; ++k) A[j][k] = A[k][j];
  // cv::eigen = JacobiImpl_<double> (core/lapack.cpp), operation for operation the CPU restatement used as checker:
  // the pivot is the largest off-diagonal element (found through the row / column maximum indices indR / indC), the
  // rotation comes from two hypots (the host libm's, hypot_glibc.h), only the upper triangle is touched, the
  // eigenvector ROWS rotate along, at most 30 n^2 rotations; eigenvalues sorted descending with their rows
  double Wv[9];
  {
    constexpr int n = 9;
    const double eps = DBL_EPSILON;
    int indR[9], indC[9];
    for (int k = 0; k < n; ++k) {
      Wv[k] = A[k][k];
      if (k < n - 1) {
        int mm = k + 1;
        double mv = fabs(A[k][mm]);
        for (int i = k + 2; i < n; ++i) {
          const double val = fabs(A[k][i]);
          if (mv < val) mv = val, mm = i;
        }
        indR[k] = mm;
      }
      if (k > 0) {
        int mm = 0;
        double mv = fabs(A[0][k]);
        for (int i = 1; i < k; ++i) {
          const double val = fabs(A[i][k]);
          if (mv < val) mv = val, mm = i;
        }
        indC[k] = mm;
      }
    }
    for (int iters = 0; iters < n * n * 30; ++iters) {
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
      const double y = (Wv[l] - Wv[k]) * 0.5;
      double t = fabs(y) + sfm_hypot(p, y);
      double sn = sfm_hypot(p, t);
      const double c = t / sn;
      sn = p / sn;
      t = (p / t) * p;
      if (y < 0) sn = -sn, t = -t;
      A[k][l] = 0;
      Wv[k] -= t;
      Wv[l] += t;
      double a0, b0;
#define SFM_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * sn, v1 = a0 * sn + b0 * c
      for (int i = 0; i < k; ++i) SFM_ROT(A[i][k], A[i][l]);
      for (int i = k + 1; i < l; ++i) SFM_ROT(A[k][i], A[i][l]);
      for (int i = l + 1; i < n; ++i) SFM_ROT(A[k][i], A[l][i]);
      for (int i = 0; i < n; ++i) SFM_ROT(V[k][i], V[l][i]);
#undef SFM_ROT
      for (int j = 0; j < 2; ++j) {
        const int idx = j == 0 ? k : l;
        if (idx < n - 1) {
          int mm = idx + 1;
          double mx = fabs(A[idx][mm]);
          for (int i = idx + 2; i < n; ++i) {
            const double val = fabs(A[idx][i]);
            if (mx < val) mx = val, mm = i;
          }
          indR[idx] = mm;
        }
        if (idx > 0) {
          int mm = 0;
          double mx = fabs(A[0][idx]);
          for (int i = 1; i < idx; ++i) {
            const double val = fabs(A[i][idx]);
            if (mx < val) mx = val, mm = i;
          }
          indC[idx] = mm;
        }
      }
    }
    for (int k = 0; k < n - 1; ++k) {
      int mm = k;
      for (int i = k + 1; i < n; ++i)
        if (Wv[mm] < Wv[i]) mm = i;
      if (k != mm) {
        double t = Wv[mm];
        Wv[mm] = Wv[k];
        Wv[k] = t;
        for (int i = 0; i < n; ++i) {
          t = V[mm][i];
          V[mm][i] = V[k][i];
          V[k][i] = t;
        }
      }
    }
  }
  double h0[9];
  for (int k = 0; k < 9; ++k) h0[k] = V[8][k];  // the row of the smallest eigenvalue
  // H = invHnorm * H0 * Hnorm2, then / H[2][2]
  const double inv[9] = {1.0 / smx, 0, cmx, 0, 1.0 / smy, cmy, 0, 0, 1};
  const double hn2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
  double t[9], r[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) t[3 * a + b] = inv[3 * a] * h0[b] + inv[3 * a + 1] * h0[3 + b] + inv[3 * a + 2] * h0[6 + b];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) r[3 * a + b] = t[3 * a] * hn2[b] + t[3 * a + 1] * hn2[3 + b] + t[3 * a + 2] * hn2[6 + b];
  const double s = 1.0 / r[8];
  for (int k = 0; k < 9; ++k) H[k] = r[k] * s;
  return 1;
}

struct HJob {
  int off, count, samp;
  float t;  // (float)(threshold * threshold) of the pair
};

__global__ __launch_bounds__(64) void homog_solve(const HJob* __restrict__ jobs, int n_jobs, int chunk,
                                                  const int* __restrict__ samples, const float* __restrict__ p1,
                                                  const float* __restrict__ p2, double* __restrict__ models,
                                                  int* __restrict__ n_models) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_jobs * chunk) return;
  const int j = t / chunk, it = t - j * chunk;
  const HJob jb = jobs[j];
  const int* s = samples + ((size_t)jb.samp + it) * 4;
  float M[4][2], m[4][2];
  bool ok = true;
  for (int k = 0; k < 4; ++k) {
    if (s[k] < 0) {  // (no acceptable sample was found for this iteration: RANSAC has ended there)
      ok = false;
      break;
    }
    const size_t i = (size_t)jb.off + s[k];
    M[k][0] = p1[2 * i]; M[k][1] = p1[2 * i + 1];
    m[k][0] = p2[2 * i]; m[k][1] = p2[2 * i + 1];
  }
  double H[9];
  const int n = ok ? homography_kernel(M, m, H) : 0;
  n_models[t] = n;
  if (n)
    for (int e = 0; e < 9; ++e) models[(size_t)t * 9 + e] = H[e];
}

// HomographyEstimatorCallback::computeError + findInliers, float arithmetic
__device__ __forceinline__ bool homog_inlier(const float* Hf, float Mx, float My, float mx, float my, float t) {
  const float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
  const float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - mx;
  const float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - my;
  return dx * dx + dy * dy <= t;
}

__global__ __launch_bounds__(256) void homog_count(const HJob* __restrict__ jobs, int chunk, const float* __restrict__ p1,
                                                   const float* __restrict__ p2, const double* __restrict__ models,
                                                   const int* __restrict__ n_models, int* __restrict__ counts) {
  __shared__ float sH[8];
  __shared__ int s_cnt;
  const int slot = blockIdx.x;
  const HJob jb = jobs[slot / chunk];
  const int nm = n_models[slot];
  if (threadIdx.x == 0) s_cnt = 0;
  if (threadIdx.x < 8 && nm) sH[threadIdx.x] = (float)models[(size_t)slot * 9 + threadIdx.x];
  __syncthreads();
  if (nm) {
    int cnt = 0;
    for (int i = threadIdx.x; i < jb.count; i += 256) {
      const size_t k = (size_t)jb.off + i;
      cnt += homog_inlier(sH, p1[2 * k], p1[2 * k + 1], p2[2 * k], p2[2 * k + 1], jb.t) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[slot] = nm ? s_cnt : 0;
}

__global__ __launch_bounds__(256) void homog_mask(const int* __restrict__ offsets, const float* __restrict__ p1,
                                                  const float* __restrict__ p2, const double* __restrict__ best_H,
                                                  const unsigned char* __restrict__ has, const float* __restrict__ tt,
                                                  unsigned char* __restrict__ mask) {
  const int pr = blockIdx.x;
  const int o = offsets[pr], n = offsets[pr + 1] - o;
  __shared__ float sH[8];
  if (threadIdx.x < 8) sH[threadIdx.x] = (float)best_H[(size_t)pr * 9 + threadIdx.x];
  __syncthreads();
  const int h = has[pr];
  for (int i = threadIdx.x; i < n; i += 256) {
    const size_t k = (size_t)o + i;
    mask[k] = h == 2 ? 1 : h == 1 ? (homog_inlier(sH, p1[2 * k], p1[2 * k + 1], p2[2 * k], p2[2 * k + 1], tt[pr]) ? 1 : 0) : 0;
  }
}

// ---- host: HomographyEstimatorCallback::checkSubset and the per-pair sample stream
static bool have_collinear(const float (*p)[2], int count) {
  const int i = count - 1;
  for (int j = 0; j < i; ++j) {
    const double dx1 = p[j][0] - p[i][0], dy1 = p[j][1] - p[i][1];
    for (int k = 0; k < j; ++k) {
      const double dx2 = p[k][0] - p[i][0], dy2 = p[k][1] - p[i][1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
    }
  }
  return false;
}
static double det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static bool homography_check_subset(const float (*s1)[2], const float (*s2)[2]) {
  if (have_collinear(s1, 4) || have_collinear(s2, 4)) return false;
  static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
  int negative = 0;
  for (int i = 0; i < 4; ++i) {
    const int* t = tt[i];
    const double A[9] = {s1[t[0]][0], s1[t[0]][1], 1., s1[t[1]][0], s1[t[1]][1], 1., s1[t[2]][0], s1[t[2]][1], 1.};
    const double B[9] = {s2[t[0]][0], s2[t[0]][1], 1., s2[t[1]][0], s2[t[1]][1], 1., s2[t[2]][0], s2[t[2]][1], 1.};
    negative += det3(A) * det3(B) < 0;
  }
  return negative == 0 || negative == 4;
}
struct HSampleStream {  // per pair: the accepted 4-point samples in order (-1: getSubset gave up after 10000 attempts)
  CvRng rng;
  std::vector<int> idx;
  bool dead = false;
  void extend(const float* m1, const float* m2, int count, int n_iters) {
    while ((int)idx.size() < 4 * n_iters) {
      int s[4] = {-1, -1, -1, -1};
      bool found = false;
      for (int attempt = 0; attempt < 10000 && !dead; ++attempt) {
        for (int i = 0; i < 4;) {
          const int v = rng.uniform(0, count);
          int j = 0;
          for (; j < i; ++j)
            if (s[j] == v) break;
          if (j < i) continue;
          s[i++] = v;
        }
        float a[4][2], b[4][2];
        for (int k = 0; k < 4; ++k) {
          a[k][0] = m1[2 * s[k]]; a[k][1] = m1[2 * s[k] + 1];
          b[k][0] = m2[2 * s[k]]; b[k][1] = m2[2 * s[k] + 1];
        }
        if (homography_check_subset(a, b)) {
          found = true;
          break;
        }
      }
      if (!found) {
        dead = true;
        s[0] = s[1] = s[2] = s[3] = -1;
      }
      idx.insert(idx.end(), s, s + 4);
    }
  }
};

// The model of a RANSAC iteration that raised a pair's best count stays on the device: (index of the model in the
// chunk's model array, pair) -> best[pair].  (Downloading every model of every iteration for the host to pick from
// moved 720 bytes per iteration: 0.9 GB for 1225 pairs x 1000 iterations.)
__global__ void score_keep_best(const int2* __restrict__ upd, int n, const double* __restrict__ models, double* __restrict__ best) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * n) return;
  const int2 u = upd[i / 9];
  best[(size_t)u.y * 9 + i % 9] = models[(size_t)u.x * 9 + i % 9];
}

}  // namespace

extern "C" int sfmhip_score_essential(sfmhip_ctx* ctx, int n_pairs, const int32_t* offsets, const double* left_xy,
                                      const double* right_xy, double fx, double fy, double cx, double cy, double prob,
                                      double threshold, int32_t* inliers, uint8_t* mask, int32_t* iterations) {
  if (!ctx || n_pairs < 0 || !offsets || !inliers || !(prob > 0 && prob < 1)) return SFMHIP_ERR_ARG;
  if (n_pairs == 0) return SFMHIP_OK;
  const long long total = offsets[n_pairs];
  if (total < 0 || (total > 0 && (!left_xy || !right_xy))) return SFMHIP_ERR_ARG;
  for (int p = 0; p < n_pairs; ++p)
    if (offsets[p + 1] < offsets[p]) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  constexpr int MAX_ITERS = 1000, MODEL_POINTS = 5;
  const double thr = threshold / ((fx + fy) / 2);
  const float t = (float)(thr * thr);
  const double ax = 1. / fx, bx = -cx * ax, ay = 1. / fy, by = -cy * ay;
  int flags_any = 0;
  ctx->score_flags = 0;

  struct Bufs {
    std::vector<void*> v;
    ~Bufs() {
      for (void* p : v) hipFree(p);
    }
  } bufs;
  auto dalloc = [&](void** p, size_t bytes) -> int {
    if (hipMalloc(p, bytes ? bytes : 8) != hipSuccess) return SFMHIP_ERR_ALLOC;
    bufs.v.push_back(*p);
    return SFMHIP_OK;
  };
  double *d_raw = nullptr, *d_p1 = nullptr, *d_p2 = nullptr;
  SFM_TRY(dalloc((void**)&d_raw, sizeof(double) * 2 * (size_t)std::max<long long>(total, 1)));
  SFM_TRY(dalloc((void**)&d_p1, sizeof(double) * 2 * (size_t)std::max<long long>(total, 1)));
  SFM_TRY(dalloc((void**)&d_p2, sizeof(double) * 2 * (size_t)std::max<long long>(total, 1)));
  if (total > 0) {
    const int nb = (int)((2 * total + 255) / 256);
    SFM_HIP_TRY(hipMemcpyAsync(d_raw, left_xy, sizeof(double) * 2 * total, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(score_normalize, dim3(nb), dim3(256), 0, st, d_raw, d_p1, 2 * total, ax, bx, ay, by);
    SFM_HIP_TRY(hipStreamSynchronize(st));  // (d_raw is reused)
    SFM_HIP_TRY(hipMemcpyAsync(d_raw, right_xy, sizeof(double) * 2 * total, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(score_normalize, dim3(nb), dim3(256), 0, st, d_raw, d_p2, 2 * total, ax, bx, ay, by);
  }
  // per pair: the state of ptsetreg's loop
  struct PairState {
    int count, niters, iter, best, best_it, best_m;
    bool done;
  };
  std::vector<PairState> ps(n_pairs);
  std::vector<unsigned char> has(n_pairs, 0);
  std::map<int, SampleStream> streams;
  double* d_bestE = nullptr;
  int2* d_upd = nullptr;
  std::vector<int2> upd;
  SFM_TRY(dalloc((void**)&d_bestE, sizeof(double) * 9 * n_pairs));
  SFM_TRY(dalloc((void**)&d_upd, sizeof(int2) * n_pairs));
  SFM_HIP_TRY(hipMemsetAsync(d_bestE, 0, sizeof(double) * 9 * n_pairs, st));
  for (int p = 0; p < n_pairs; ++p) {
    PairState& s = ps[p];
    s.count = offsets[p + 1] - offsets[p];
    s.niters = MAX_ITERS;
    s.iter = 0;
    s.best = 0;
    s.best_it = s.best_m = -1;
    s.done = s.count < MODEL_POINTS;  // (run() returns false: no model, empty mask)
  }
  int chunk = 32;
  std::vector<ScoreJob> jobs;
  std::vector<int> job_pair, h_samples, h_nm, h_counts;
  ScoreJob* d_jobs = nullptr;
  int *d_samples = nullptr, *d_nm = nullptr, *d_counts = nullptr;
  double *d_models = nullptr, *d_work = nullptr;
  size_t cap_jobs = 0, cap_slots = 0, cap_samples = 0;
  for (;;) {
    jobs.clear();
    job_pair.clear();
    h_samples.clear();
    for (int p = 0; p < n_pairs; ++p) {
      PairState& s = ps[p];
      if (s.done) continue;
      ScoreJob jb;
      jb.off = offsets[p];
      jb.count = s.count;
      jb.samp = (int)(h_samples.size() / 5);
      if (s.count == MODEL_POINTS) {
        // run(): count == modelPoints -> runKernel on the five, every match an inlier if there is a model
        for (int it = 0; it < chunk; ++it)
          for (int k = 0; k < 5; ++k) h_samples.push_back(k);
      } else {
        SampleStream& ss = streams[s.count];
        ss.extend(s.count, s.iter + chunk);
        h_samples.insert(h_samples.end(), ss.idx.begin() + 5 * (size_t)s.iter, ss.idx.begin() + 5 * (size_t)(s.iter + chunk));
      }
      jobs.push_back(jb);
      job_pair.push_back(p);
    }
    if (jobs.empty()) break;
    const size_t nj = jobs.size(), slots = nj * (size_t)chunk;
    if (nj > cap_jobs) {
      SFM_TRY(dalloc((void**)&d_jobs, sizeof(ScoreJob) * nj));
      cap_jobs = nj;
    }
    if (slots > cap_slots) {
      SFM_TRY(dalloc((void**)&d_nm, sizeof(int) * slots));
      SFM_TRY(dalloc((void**)&d_counts, sizeof(int) * slots * MAX_MODELS));
      SFM_TRY(dalloc((void**)&d_models, sizeof(double) * slots * MAX_MODELS * 9));
      SFM_TRY(dalloc((void**)&d_work, sizeof(double) * ((slots + 63) / 64 * 64) * FP_WORK));
      cap_slots = slots;
    }
    if (h_samples.size() > cap_samples) {
      SFM_TRY(dalloc((void**)&d_samples, sizeof(int) * h_samples.size()));
      cap_samples = h_samples.size();
    }
    SFM_HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(ScoreJob) * nj, hipMemcpyHostToDevice, st));
    SFM_HIP_TRY(hipMemcpyAsync(d_samples, h_samples.data(), sizeof(int) * h_samples.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(score_setup, dim3((unsigned)((slots + 63) / 64)), dim3(64), 0, st, d_jobs, (int)nj, chunk, d_samples, d_p1,
                       d_p2, d_work);
    hipLaunchKernelGGL(score_roots, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, (int)slots, (const double*)d_work, d_models,
                       d_nm);
    hipLaunchKernelGGL(score_count, dim3((unsigned)slots), dim3(256), 0, st, d_jobs, chunk, d_p1, d_p2, d_models, d_nm, t, d_counts);
    SFM_HIP_TRY(hipGetLastError());
    h_nm.resize(slots);
    h_counts.resize(slots * MAX_MODELS);
    SFM_HIP_TRY(hipMemcpyAsync(h_nm.data(), d_nm, sizeof(int) * slots, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(h_counts.data(), d_counts, sizeof(int) * slots * MAX_MODELS, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    upd.clear();
    // ---- ptsetreg.cpp run(): the models of a sample in order, the iteration limit from the best count so far
    for (size_t j = 0; j < nj; ++j) {
      PairState& s = ps[job_pair[j]];
      if (s.count == MODEL_POINTS) {
        const size_t slot = j * chunk;
        flags_any |= h_nm[slot] >> 8;
        if ((h_nm[slot] & 0xff) > 0) {
          s.best = MODEL_POINTS;
          has[job_pair[j]] = 2;
          upd.push_back(int2{(int)(slot * MAX_MODELS), job_pair[j]});
        }
        s.done = true;
        continue;
      }
      long long keep = -1;  // the last model of this chunk that raised the best count
      for (int it = 0; it < chunk && s.iter < s.niters; ++it, ++s.iter) {
        const size_t slot = j * chunk + it;
        flags_any |= h_nm[slot] >> 8;
        for (int m = 0; m < (h_nm[slot] & 0xff); ++m) {
          const int good = h_counts[slot * MAX_MODELS + m];
          if (good > std::max(s.best, MODEL_POINTS - 1)) {
            s.best = good;
            has[job_pair[j]] = 1;
            keep = (long long)(slot * MAX_MODELS + m);
            s.niters = ransac_update_num_iters(prob, (double)(s.count - good) / s.count, MODEL_POINTS, s.niters);
          }
        }
      }
      if (keep >= 0) upd.push_back(int2{(int)keep, job_pair[j]});
      if (s.iter >= s.niters) s.done = true;
    }
    if (!upd.empty()) {  // (before the next chunk's solve overwrites the models; same stream)
      SFM_HIP_TRY(hipMemcpyAsync(d_upd, upd.data(), sizeof(int2) * upd.size(), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(score_keep_best, dim3((unsigned)((9 * upd.size() + 255) / 256)), dim3(256), 0, st, (const int2*)d_upd,
                         (int)upd.size(), (const double*)d_models, d_bestE);
    }
    chunk = std::min(2 * chunk, 256);
  }
  for (int p = 0; p < n_pairs; ++p) {
    inliers[p] = ps[p].best;
    if (iterations) iterations[p] = ps[p].iter;
  }
  ctx->score_flags = flags_any;
  if (mask && total > 0) {
    int* d_off = nullptr;
    unsigned char *d_has = nullptr, *d_mask = nullptr;
    SFM_TRY(dalloc((void**)&d_off, sizeof(int) * (n_pairs + 1)));
    SFM_TRY(dalloc((void**)&d_has, n_pairs));
    SFM_TRY(dalloc((void**)&d_mask, (size_t)total));
    SFM_HIP_TRY(hipMemcpyAsync(d_off, offsets, sizeof(int) * (n_pairs + 1), hipMemcpyHostToDevice, st));
    SFM_HIP_TRY(hipMemcpyAsync(d_has, has.data(), n_pairs, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(score_mask, dim3(n_pairs), dim3(256), 0, st, d_off, d_p1, d_p2, d_bestE, d_has, t, d_mask);
    SFM_HIP_TRY(hipGetLastError());
    SFM_HIP_TRY(hipMemcpyAsync(mask, d_mask, (size_t)total, hipMemcpyDeviceToHost, st));
  }
  SFM_HIP_TRY(hipStreamSynchronize(st));
  return SFMHIP_OK;
}

// EMEstimatorCallback::runKernel for explicit samples (five normalised correspondences each): what score_solve runs per
// (pair, iteration), exposed for sample-level parity checks
__global__ __launch_bounds__(64) void five_point_samples(const double* __restrict__ q1, const double* __restrict__ q2, int n,
                                                         double* __restrict__ models, int* __restrict__ n_models) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double a[5][2], b[5][2];
  for (int k = 0; k < 5; ++k) {
    a[k][0] = q1[10 * (size_t)t + 2 * k];
    a[k][1] = q1[10 * (size_t)t + 2 * k + 1];
    b[k][0] = q2[10 * (size_t)t + 2 * k];
    b[k][1] = q2[10 * (size_t)t + 2 * k + 1];
  }
  double E[MAX_MODELS][9];
  int flags = 0;
  const int nm = five_point(a, b, E, &flags);
  n_models[t] = nm | (flags << 8);
  for (int m = 0; m < nm; ++m)
    for (int e = 0; e < 9; ++e) models[((size_t)t * MAX_MODELS + m) * 9 + e] = E[m][e];
}

extern "C" int sfmhip_score_five_point(sfmhip_ctx* ctx, int n_samples, const double* q1, const double* q2, double* models,
                                       int32_t* n_models) {
  if (!ctx || n_samples < 0 || (n_samples && (!q1 || !q2 || !models || !n_models))) return SFMHIP_ERR_ARG;
  if (n_samples == 0) return SFMHIP_OK;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *d_q1 = nullptr, *d_q2 = nullptr, *d_m = nullptr;
  int* d_n = nullptr;
  int rc = SFMHIP_OK;
  if ((rc = sfm_dev_alloc(&d_q1, 10 * (size_t)n_samples)) || (rc = sfm_dev_alloc(&d_q2, 10 * (size_t)n_samples)) ||
      (rc = sfm_dev_alloc(&d_m, 90 * (size_t)n_samples)) || (rc = sfm_dev_alloc(&d_n, (size_t)n_samples))) {
    hipFree(d_q1), hipFree(d_q2), hipFree(d_m), hipFree(d_n);
    return rc;
  }
  hipError_t e = hipMemcpyAsync(d_q1, q1, sizeof(double) * 10 * n_samples, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_q2, q2, sizeof(double) * 10 * n_samples, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(d_m, 0, sizeof(double) * 90 * n_samples, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(five_point_samples, dim3((n_samples + 63) / 64), dim3(64), 0, st, d_q1, d_q2, n_samples, d_m, d_n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(models, d_m, sizeof(double) * 90 * n_samples, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(n_models, d_n, sizeof(int) * n_samples, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(d_q1), hipFree(d_q2), hipFree(d_m), hipFree(d_n);
  if (e != hipSuccess) {
    g_sfmhip_last_hip_error = (int)e;
    return SFMHIP_ERR_HIP;
  }
  return SFMHIP_OK;
}

// HomographyEstimatorCallback::runKernel for explicit 4-point samples (sample-level parity checks)
__global__ __launch_bounds__(64) void homography_samples(const float* __restrict__ M, const float* __restrict__ m, int n,
                                                         double* __restrict__ H, int* __restrict__ ok) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float a[4][2], b[4][2];
  for (int k = 0; k < 4; ++k) {
    a[k][0] = M[8 * (size_t)t + 2 * k];
    a[k][1] = M[8 * (size_t)t + 2 * k + 1];
    b[k][0] = m[8 * (size_t)t + 2 * k];
    b[k][1] = m[8 * (size_t)t + 2 * k + 1];
  }
  double h[9];
  const int r = homography_kernel(a, b, h);
  ok[t] = r;
  for (int e = 0; e < 9; ++e) H[9 * (size_t)t + e] = r ? h[e] : 0.0;
}

extern "C" int sfmhip_score_homography_kernel(sfmhip_ctx* ctx, int n_samples, const float* M, const float* m, double* H, int32_t* ok) {
  if (!ctx || n_samples < 0 || (n_samples && (!M || !m || !H || !ok))) return SFMHIP_ERR_ARG;
  if (n_samples == 0) return SFMHIP_OK;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  float *d_M = nullptr, *d_m = nullptr;
  double* d_H = nullptr;
  int* d_ok = nullptr;
  int rc = SFMHIP_OK;
  if ((rc = sfm_dev_alloc(&d_M, 8 * (size_t)n_samples)) || (rc = sfm_dev_alloc(&d_m, 8 * (size_t)n_samples)) ||
      (rc = sfm_dev_alloc(&d_H, 9 * (size_t)n_samples)) || (rc = sfm_dev_alloc(&d_ok, (size_t)n_samples))) {
    hipFree(d_M), hipFree(d_m), hipFree(d_H), hipFree(d_ok);
    return rc;
  }
  hipError_t e = hipMemcpyAsync(d_M, M, sizeof(float) * 8 * n_samples, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_m, m, sizeof(float) * 8 * n_samples, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(homography_samples, dim3((n_samples + 63) / 64), dim3(64), 0, st, d_M, d_m, n_samples, d_H, d_ok);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(H, d_H, sizeof(double) * 9 * n_samples, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(ok, d_ok, sizeof(int) * n_samples, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(d_M), hipFree(d_m), hipFree(d_H), hipFree(d_ok);
  if (e != hipSuccess) {
    g_sfmhip_last_hip_error = (int)e;
    return SFMHIP_ERR_HIP;
  }
  return SFMHIP_OK;
}

extern "C" int sfmhip_score_last_flags(sfmhip_ctx* ctx) { return ctx ? ctx->score_flags : 0; }

extern "C" int sfmhip_score_homography(sfmhip_ctx* ctx, int n_pairs, const int32_t* offsets, const double* left_xy,
                                       const double* right_xy, const double* thresholds, double confidence, int max_iters,
                                       int32_t* inliers, uint8_t* mask, int32_t* iterations) {
  if (!ctx || n_pairs < 0 || !offsets || !inliers || !thresholds || !(confidence > 0 && confidence < 1)) return SFMHIP_ERR_ARG;
  if (n_pairs == 0) return SFMHIP_OK;
  const long long total = offsets[n_pairs];
  if (total < 0 || (total > 0 && (!left_xy || !right_xy))) return SFMHIP_ERR_ARG;
  for (int p = 0; p < n_pairs; ++p)
    if (offsets[p + 1] < offsets[p]) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  constexpr int MODEL_POINTS = 4;
  const int MAX_ITERS = std::max(max_iters, 1);
  struct Bufs {
    std::vector<void*> v;
    ~Bufs() {
      for (void* p : v) hipFree(p);
    }
  } bufs;
  auto dalloc = [&](void** p, size_t bytes) -> int {
    if (hipMalloc(p, bytes ? bytes : 8) != hipSuccess) return SFMHIP_ERR_ALLOC;
    bufs.v.push_back(*p);
    return SFMHIP_OK;
  };
  // the points as float (findHomography converts them to CV_32F before anything else)
  std::vector<float> h1(2 * (size_t)std::max<long long>(total, 1)), h2(h1.size());
  for (long long i = 0; i < 2 * total; ++i) {
    h1[i] = (float)left_xy[i];
    h2[i] = (float)right_xy[i];
  }
  float *d_p1 = nullptr, *d_p2 = nullptr;
  SFM_TRY(dalloc((void**)&d_p1, sizeof(float) * h1.size()));
  SFM_TRY(dalloc((void**)&d_p2, sizeof(float) * h2.size()));
  SFM_HIP_TRY(hipMemcpyAsync(d_p1, h1.data(), sizeof(float) * h1.size(), hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemcpyAsync(d_p2, h2.data(), sizeof(float) * h2.size(), hipMemcpyHostToDevice, st));
  struct PairState {
    int count, niters, iter, best;
    float t;
    bool done;
  };
  std::vector<PairState> ps(n_pairs);
  std::vector<HSampleStream> streams(n_pairs);
  std::vector<unsigned char> has(n_pairs, 0);
  std::vector<float> tts(n_pairs);
  double* d_bestH = nullptr;
  int2* d_upd = nullptr;
  std::vector<int2> upd;
  SFM_TRY(dalloc((void**)&d_bestH, sizeof(double) * 9 * n_pairs));
  SFM_TRY(dalloc((void**)&d_upd, sizeof(int2) * n_pairs));
  SFM_HIP_TRY(hipMemsetAsync(d_bestH, 0, sizeof(double) * 9 * n_pairs, st));
  for (int p = 0; p < n_pairs; ++p) {
    PairState& s = ps[p];
    s.count = offsets[p + 1] - offsets[p];
    double thr = thresholds[p];
    if (thr <= 0) thr = 3;  // defaultRANSACReprojThreshold
    s.t = tts[p] = (float)(thr * thr);
    s.niters = MAX_ITERS;
    s.iter = s.best = 0;
    s.done = s.count < MODEL_POINTS;
  }
  int chunk = 32;
  std::vector<HJob> jobs;
  std::vector<int> job_pair, h_samples, h_nm, h_counts;
  HJob* d_jobs = nullptr;
  int *d_samples = nullptr, *d_nm = nullptr, *d_counts = nullptr;
  double* d_models = nullptr;
  size_t cap_jobs = 0, cap_slots = 0, cap_samples = 0;
  for (;;) {
    jobs.clear();
    job_pair.clear();
    h_samples.clear();
    for (int p = 0; p < n_pairs; ++p) {
      PairState& s = ps[p];
      if (s.done) continue;
      HJob jb;
      jb.off = offsets[p];
      jb.count = s.count;
      jb.t = s.t;
      jb.samp = (int)(h_samples.size() / 4);
      if (s.count == MODEL_POINTS) {  // method == 0 || npoints == 4: runKernel on the four, the mask all ones
        for (int it = 0; it < chunk; ++it)
          for (int k = 0; k < 4; ++k) h_samples.push_back(k);
      } else {
        HSampleStream& ss = streams[p];
        ss.extend(h1.data() + 2 * (size_t)offsets[p], h2.data() + 2 * (size_t)offsets[p], s.count, s.iter + chunk);
        h_samples.insert(h_samples.end(), ss.idx.begin() + 4 * (size_t)s.iter, ss.idx.begin() + 4 * (size_t)(s.iter + chunk));
      }
      jobs.push_back(jb);
      job_pair.push_back(p);
    }
    if (jobs.empty()) break;
    const size_t nj = jobs.size(), slots = nj * (size_t)chunk;
    if (nj > cap_jobs) {
      SFM_TRY(dalloc((void**)&d_jobs, sizeof(HJob) * nj));
      cap_jobs = nj;
    }
    if (slots > cap_slots) {
      SFM_TRY(dalloc((void**)&d_nm, sizeof(int) * slots));
      SFM_TRY(dalloc((void**)&d_counts, sizeof(int) * slots));
      SFM_TRY(dalloc((void**)&d_models, sizeof(double) * slots * 9));
      cap_slots = slots;
    }
    if (h_samples.size() > cap_samples) {
      SFM_TRY(dalloc((void**)&d_samples, sizeof(int) * h_samples.size()));
      cap_samples = h_samples.size();
    }
    SFM_HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(HJob) * nj, hipMemcpyHostToDevice, st));
    SFM_HIP_TRY(hipMemcpyAsync(d_samples, h_samples.data(), sizeof(int) * h_samples.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(homog_solve, dim3((unsigned)((slots + 63) / 64)), dim3(64), 0, st, d_jobs, (int)nj, chunk, d_samples, d_p1,
                       d_p2, d_models, d_nm);
    hipLaunchKernelGGL(homog_count, dim3((unsigned)slots), dim3(256), 0, st, d_jobs, chunk, d_p1, d_p2, d_models, d_nm, d_counts);
    SFM_HIP_TRY(hipGetLastError());
    h_nm.resize(slots);
    h_counts.resize(slots);
    SFM_HIP_TRY(hipMemcpyAsync(h_nm.data(), d_nm, sizeof(int) * slots, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(h_counts.data(), d_counts, sizeof(int) * slots, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    upd.clear();
    for (size_t j = 0; j < nj; ++j) {
      const int p = job_pair[j];
      PairState& s = ps[p];
      if (s.count == MODEL_POINTS) {
        if (h_nm[j * chunk] > 0) {
          s.best = MODEL_POINTS;
          has[p] = 2;
        }
        s.done = true;
        continue;
      }
      long long keep = -1;
      for (int it = 0; it < chunk && s.iter < s.niters; ++it, ++s.iter) {
        const size_t slot = j * chunk + it;
        if (streams[p].idx[4 * (size_t)s.iter] < 0) {  // getSubset failed: run() leaves the loop (iter 0: no model at all)
          s.niters = s.iter;
          break;
        }
        if (h_nm[slot] <= 0) continue;
        const int good = h_counts[slot];
        if (good > std::max(s.best, MODEL_POINTS - 1)) {
          s.best = good;
          has[p] = 1;
          keep = (long long)slot;
          s.niters = ransac_update_num_iters(confidence, (double)(s.count - good) / s.count, MODEL_POINTS, s.niters);
        }
      }
      if (keep >= 0) upd.push_back(int2{(int)keep, p});
      if (s.iter >= s.niters) s.done = true;
    }
    if (!upd.empty()) {
      SFM_HIP_TRY(hipMemcpyAsync(d_upd, upd.data(), sizeof(int2) * upd.size(), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(score_keep_best, dim3((unsigned)((9 * upd.size() + 255) / 256)), dim3(256), 0, st, (const int2*)d_upd,
                         (int)upd.size(), (const double*)d_models, d_bestH);
    }
    chunk = std::min(2 * chunk, 256);
  }
  for (int p = 0; p < n_pairs; ++p) {
    inliers[p] = ps[p].best;
    if (iterations) iterations[p] = ps[p].iter;
  }
  if (mask && total > 0) {
    int* d_off = nullptr;
    float* d_tt = nullptr;
    unsigned char *d_has = nullptr, *d_mask = nullptr;
    SFM_TRY(dalloc((void**)&d_off, sizeof(int) * (n_pairs + 1)));
    SFM_TRY(dalloc((void**)&d_tt, sizeof(float) * n_pairs));
    SFM_TRY(dalloc((void**)&d_has, n_pairs));
    SFM_TRY(dalloc((void**)&d_mask, (size_t)total));
    SFM_HIP_TRY(hipMemcpyAsync(d_off, offsets, sizeof(int) * (n_pairs + 1), hipMemcpyHostToDevice, st));
    SFM_HIP_TRY(hipMemcpyAsync(d_tt, tts.data(), sizeof(float) * n_pairs, hipMemcpyHostToDevice, st));
    SFM_HIP_TRY(hipMemcpyAsync(d_has, has.data(), n_pairs, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(homog_mask, dim3(n_pairs), dim3(256), 0, st, d_off, d_p1, d_p2, d_bestH, d_has, d_tt, d_mask);
    SFM_HIP_TRY(hipGetLastError());
    SFM_HIP_TRY(hipMemcpyAsync(mask, d_mask, (size_t)total, hipMemcpyDeviceToHost, st));
  }
  SFM_HIP_TRY(hipStreamSynchronize(st));
  return SFMHIP_OK;
}
