// ba.hip -- Levenberg-Marquardt bundle adjustment with a Schur-complement reduced camera solve
// on gfx950 (MI355X).
//
// Replaces ceres::Solve(DENSE_SCHUR) as configured by BundleAdjustment::adjustBundle (reference
// src/BundleAdjustment.cpp:115-123) for the cost functor SimpleReprojectionError
// (src/BundleAdjustment.cpp:10-35): parameter blocks = camera (angle-axis 3 + translation 3),
// point (3), one shared focal (1); residual 2 per observation; points are the eliminated
// e-blocks, cameras + focal the reduced system of dimension 6*n_cam+1.
//
// Layout in HBM (DESIGN.md "BA"): everything f64.  Points are re-ordered once per problem so
// that points observed by the same ascending camera list ("signature") are contiguous and cut
// into chunks; the reduced matrix S is a dense (6Nc+1)^2 row-major array of which only the
// upper triangle is formed (= a column-major lower triangle for the Cholesky kernels).
//
// Kernels per LM iteration:
//   ba_cam_blocks     F^T F part of the reduced system (camera 6x6 blocks, focal border, F^T b,
//                     cost) from a camera-major copy of the observations.
//   ba_eliminate_mfma<NB>  Schur correction.  A workgroup owns a run of points of one signature;
//                     16 lanes linearise the (<=10) observations of a point (4 points per wave
//                     and iteration), DPP-reduce the 3x3 point block, publish
//                     T_o = (Jc^T Jp) C^-1/2 as 3 rows of a 12 x 64 LDS panel, and the wave adds
//                     the panel's Gram matrix on v_mfma_f64_16x16x4_f64; one f64 atomic scatter
//                     into S per workgroup.
//   ba_finalize       LM diagonal (clamped squared column norms / radius) onto S, gradient max.
//   chol_step/chol_backsolve  dense blocked Cholesky of S with the rhs carried as an extra
//                     row, then the transposed triangular solve.
//   ba_cand_cams      candidate cameras / focal and their rotation tables.
//   ba_backsub        per point: back-substitution, model cost change, candidate point,
//                     candidate cost.
// Multi-GPU: every rank holds all cameras and its own block of points; [S | g | F^T b |
// diag | scalars] is summed by the caller's all-reduce (RCCL over xGMI) once per iteration,
// every rank then solves the same reduced system and back-substitutes its own points.
#include "common.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <thread>
#include <vector>
#include <float.h>

namespace {

constexpr int CAMD = 40;     // doubles per camera table: R[9] t[3] dR/dw[27] pad
constexpr int SC = 16;       // scalar slots at the tail of the all-reduce buffer (+ world)
constexpr int FB_MAXN = 64;  // fallback kernel: max observations per point

struct Chunk {
  int sig_off;  // offset into sig_cams
  int n;        // observations per point in this signature
  int p0;       // first sorted point
  int cnt;      // points in the chunk
};

struct BaDev {
  // problem (sorted order)
  int nc, np, no, dim, ld;
  const int* optr;     // np+1
  const int* ocam;     // no
  const double2* oxy;  // no
  // parameters
  double* cams;    // nc*6
  double* pts;     // np*3 (sorted)
  double* focal;   // 1
  double* camd;    // nc*CAMD
  double* cams_c;  // candidates
  double* pts_c;
  double* focal_c;
  double* camd_c;
  // scaling / LM
  double* scale_c;  // nc*6
  double* scale_p;  // np*3
  double* scale_f;  // 1
  double* diag;     // dim (clamped)
  // reduced system: red = [S ld*ld | g ld | gF ld | dc ld | sc SC+world], ld = dim rounded up
  // to 32 (row stride of S; the padded diagonal is 1, everything else in the padding 0)
  double* red;
  double* z;     // dim solution
  double* dinv;  // ld: 1 / diag(L)
  double* linv;  // ld*32: inverse of every diagonal tile of L, transposed ([k][col][row])
  double* red2;  // 16 scalars of the step evaluation
  int* info;     // cholesky failure flag
};

__device__ __forceinline__ double* red_S(const BaDev& d) { return d.red; }
__device__ __forceinline__ double* red_g(const BaDev& d) { return d.red + (size_t)d.ld * d.ld; }
__device__ __forceinline__ double* red_gF(const BaDev& d) { return d.red + (size_t)d.ld * d.ld + d.ld; }
__device__ __forceinline__ double* red_dc(const BaDev& d) { return d.red + (size_t)d.ld * d.ld + 2 * d.ld; }
__device__ __forceinline__ double* red_sc(const BaDev& d) { return d.red + (size_t)d.ld * d.ld + 3 * d.ld; }

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_max_pos_f64(double* p, double v) {
  // v >= 0: IEEE order == unsigned integer order
  atomicMax((unsigned long long*)p, (unsigned long long)__double_as_longlong(v));
}

// ---------------------------------------------------------------- small f64 helpers
typedef double v4d __attribute__((ext_vector_type(4)));  // accumulator of v_mfma_f64_16x16x4_f64
// v_rcp_f64 / v_rsq_f64 give ~24 bits; two Newton steps bring them to rounding level (the BA
// kernels are tolerance-level f64).
__device__ __forceinline__ double rcp_f64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}
__device__ __forceinline__ double rsqrt_f64(double d) {
  double r = __builtin_amdgcn_rsq(d);
  r = r * (1.5 - 0.5 * d * r * r);
  r = r * (1.5 - 0.5 * d * r * r);
  return r;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over each row of 16 lanes, every lane of the row receives the (bitwise identical) total
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return v;
}

// ---------------------------------------------------------------- camera tables
// R = dp/dX and dR/dw_j of ceres::AngleAxisRotatePoint, same theta^2 > eps branch as the
// expression autodiff differentiates (src/BundleAdjustment.cpp:16).
__device__ void cam_table(const double* cam, double* o, bool want_d) {
  const double a0 = cam[0], a1 = cam[1], a2 = cam[2];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  double R[9], dR[27];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
    const double w[3] = {a0 * ti, a1 * ti, a2 * ti};
    const double oc = 1.0 - c;
    R[0] = c + oc * w[0] * w[0];
    R[1] = -s * w[2] + oc * w[0] * w[1];
    R[2] = s * w[1] + oc * w[0] * w[2];
    R[3] = s * w[2] + oc * w[1] * w[0];
    R[4] = c + oc * w[1] * w[1];
    R[5] = -s * w[0] + oc * w[1] * w[2];
    R[6] = -s * w[1] + oc * w[2] * w[0];
    R[7] = s * w[0] + oc * w[2] * w[1];
    R[8] = c + oc * w[2] * w[2];
    if (want_d) {
      for (int j = 0; j < 3; ++j) {
        double dw[3];
        for (int i = 0; i < 3; ++i) dw[i] = ((i == j ? 1.0 : 0.0) - w[i] * w[j]) * ti;
        const double wj = w[j];
        // d/dw_j [ c I + s [w]x + (1-c) w w^T ]
        double* D = dR + 9 * j;
        const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        const double dK[9] = {0, -dw[2], dw[1], dw[2], 0, -dw[0], -dw[1], dw[0], 0};
        for (int r = 0; r < 3; ++r)
          for (int q = 0; q < 3; ++q) {
            double v = c * wj * K[3 * r + q] + s * dK[3 * r + q] + s * wj * w[r] * w[q] +
                       oc * (dw[r] * w[q] + w[r] * dw[q]);
            if (r == q) v += -s * wj;
            D[3 * r + q] = v;
          }
      }
    }
  } else {
    R[0] = 1; R[1] = -a2; R[2] = a1;
    R[3] = a2; R[4] = 1; R[5] = -a0;
    R[6] = -a1; R[7] = a0; R[8] = 1;
    if (want_d) {
      for (int i = 0; i < 27; ++i) dR[i] = 0;
      // dR/dw_0 = [e0]x, ...
      dR[0 * 9 + 5] = -1; dR[0 * 9 + 7] = 1;
      dR[1 * 9 + 2] = 1;  dR[1 * 9 + 6] = -1;
      dR[2 * 9 + 1] = -1; dR[2 * 9 + 3] = 1;
    }
  }
  for (int i = 0; i < 9; ++i) o[i] = R[i];
  o[9] = cam[3];
  o[10] = cam[4];
  o[11] = cam[5];
  if (want_d)
    for (int i = 0; i < 27; ++i) o[12 + i] = dR[i];
}

__global__ void ba_cam_prep(const double* __restrict__ cams, double* __restrict__ camd, int nc, int want_d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nc) cam_table(cams + 6 * c, camd + (size_t)CAMD * c, want_d != 0);
}

// ---------------------------------------------------------------- per-observation linearisation
struct ObsLin {
  double r0, r1;
  double Jc[12];  // 2x6 row-major
  double Jp[6];   // 2x3
  double Jf[2];
};

template <typename CP>
__device__ __forceinline__ void obs_residual(CP cd, const double X[3], double focal, double ox, double oy,
                                             double& r0, double& r1) {
  const double px = cd[0] * X[0] + cd[1] * X[1] + cd[2] * X[2] + cd[9];
  const double py = cd[3] * X[0] + cd[4] * X[1] + cd[5] * X[2] + cd[10];
  const double pz = cd[6] * X[0] + cd[7] * X[1] + cd[8] * X[2] + cd[11];
  const double iz = rcp_f64(pz);
  const double xp = px * iz, yp = py * iz;
  r0 = focal * xp - ox;
  r1 = focal * yp - oy;
}

// unscaled Jacobian; sc/sp/sf (may be null -> 1) scale the columns
template <typename CP>
__device__ __forceinline__ void obs_linearize(CP cd, const double X[3], double focal, double ox, double oy,
                                              const double* sc, const double* sp, double sf, ObsLin& o) {
  const double px = cd[0] * X[0] + cd[1] * X[1] + cd[2] * X[2] + cd[9];
  const double py = cd[3] * X[0] + cd[4] * X[1] + cd[5] * X[2] + cd[10];
  const double pz = cd[6] * X[0] + cd[7] * X[1] + cd[8] * X[2] + cd[11];
  const double iz = rcp_f64(pz);
  const double xp = px * iz, yp = py * iz;
  o.r0 = focal * xp - ox;
  o.r1 = focal * yp - oy;
  const double d00 = focal * iz, d02 = -focal * xp * iz, d12 = -focal * yp * iz;  // dr/dp rows
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double dx = cd[12 + 9 * j + 0] * X[0] + cd[12 + 9 * j + 1] * X[1] + cd[12 + 9 * j + 2] * X[2];
    const double dy = cd[12 + 9 * j + 3] * X[0] + cd[12 + 9 * j + 4] * X[1] + cd[12 + 9 * j + 5] * X[2];
    const double dz = cd[12 + 9 * j + 6] * X[0] + cd[12 + 9 * j + 7] * X[1] + cd[12 + 9 * j + 8] * X[2];
    const double s = sc ? sc[j] : 1.0;
    o.Jc[j] = (d00 * dx + d02 * dz) * s;
    o.Jc[6 + j] = (d00 * dy + d12 * dz) * s;
  }
  {
    const double s3 = sc ? sc[3] : 1.0, s4 = sc ? sc[4] : 1.0, s5 = sc ? sc[5] : 1.0;
    o.Jc[3] = d00 * s3;
    o.Jc[4] = 0.0;
    o.Jc[5] = d02 * s5;
    o.Jc[9] = 0.0;
    o.Jc[10] = d00 * s4;
    o.Jc[11] = d12 * s5;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double s = sp ? sp[j] : 1.0;
    o.Jp[j] = (d00 * cd[j] + d02 * cd[6 + j]) * s;
    o.Jp[3 + j] = (d00 * cd[3 + j] + d12 * cd[6 + j]) * s;
  }
  o.Jf[0] = xp * sf;
  o.Jf[1] = yp * sf;
}

// 3x3 SPD: inverse of the Cholesky factor (Li lower, C^-1 = Li^T Li); returns false if not PD.
// Division-free: the three pivots go through rsqrt.
__device__ __forceinline__ bool chol3_inv(const double C[6] /*00 10 11 20 21 22*/, double Li[6]) {
  const double c00 = C[0], c10 = C[1], c11 = C[2], c20 = C[3], c21 = C[4], c22 = C[5];
  const double i00 = rsqrt_f64(c00);
  const double l10 = c10 * i00, l20 = c20 * i00;
  const double d11 = c11 - l10 * l10;
  const double i11 = rsqrt_f64(d11);
  const double l21 = (c21 - l20 * l10) * i11;
  const double d22 = c22 - l20 * l20 - l21 * l21;
  const double i22 = rsqrt_f64(d22);
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  Li[0] = i00; Li[1] = i10; Li[2] = i11; Li[3] = i20; Li[4] = i21; Li[5] = i22;
  return (c00 > 0) && (d11 > 0) && (d22 > 0);
}

// ---------------------------------------------------------------- Jacobi scaling (iteration 0)
// one thread per point: unscaled squared column norms of the point's three columns (complete
// locally) and ||x||^2 of the points.  The camera and focal columns come from ba_cam_blocks in
// its norms-only mode (into red_dc, summed across ranks before ba_make_scale).
__global__ __launch_bounds__(256) void ba_point_norms(BaDev d, int jacobi) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double xn2 = 0;
  if (p < d.np) {
    const double X[3] = {d.pts[3 * p], d.pts[3 * p + 1], d.pts[3 * p + 2]};
    double np2[3] = {0, 0, 0};
    if (jacobi) {
      const double focal = *d.focal;
      for (int k = d.optr[p]; k < d.optr[p + 1]; ++k) {
        const int c = d.ocam[k];
        const double2 xy = d.oxy[k];
        ObsLin o;
        obs_linearize(d.camd + (size_t)CAMD * c, X, focal, xy.x, xy.y, nullptr, nullptr, 1.0, o);
#pragma unroll
        for (int j = 0; j < 3; ++j) np2[j] += o.Jp[j] * o.Jp[j] + o.Jp[3 + j] * o.Jp[3 + j];
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      d.scale_p[3 * p + j] = jacobi ? 1.0 / (1.0 + sqrt(np2[j])) : 1.0;
      xn2 += X[j] * X[j];
    }
  }
  __shared__ double sh[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) xn2 += __shfl_down(xn2, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = xn2;
  __syncthreads();
  if (threadIdx.x == 0) atomic_add_f64(red_sc(d) + 1, sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ void ba_make_scale(BaDev d, int jacobi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double* dc = red_dc(d);
  if (i < 6 * d.nc) d.scale_c[i] = jacobi ? 1.0 / (1.0 + sqrt(dc[i])) : 1.0;
  if (i == 6 * d.nc) *d.scale_f = jacobi ? 1.0 / (1.0 + sqrt(dc[i])) : 1.0;
}

// ---------------------------------------------------------------- linearise + Schur eliminate
// The reduced system splits into S = F^T F + D^2 - sum_p (F_p^T E_p) C_p^-1 (E_p^T F_p):
//   * ba_cam_blocks   forms the F^T F part (per-camera 6x6 blocks, the focal border, F^T b,
//                     the cost) from a camera-major copy of the observations, one workgroup
//                     per (camera, slice): every block is owned by few workgroups, 36 atomics
//                     per workgroup;
//   * ba_eliminate_mfma<NB>  forms the Schur correction.  Points with the same ascending camera
//                     list ("signature", n cameras) are contiguous and cut into chunks; with
//                     M_p = [T_0^T .. T_{n-1}^T | t_f | u]  (3 x (6n+2),  T_o = (Jc_o^T Jp_o) C_p^-1/2,
//                     t_f = C_p^-1/2 Jp^T Jf, u = C_p^-1/2 Jp^T r) the chunk's contribution is the
//                     Gram matrix sum_p M_p^T M_p, a (6n+2)^2 <= 64^2 dense block that every
//                     wave accumulates on v_mfma_f64_16x16x4_f64 (4 points = 12 rows = 3 k-steps
//                     per iteration, upper tiles only) and the workgroup scatters into S once.
constexpr int MP = 80;  // LDS row pitch (doubles) of a wave's M panel: the 4 k-rows of one
                        // fragment read sit 160 dwords apart -> disjoint banks

struct CamLds {  // camera table transposed in LDS: element e of camera slot o at [e*16 + o]
  const double* base;
  __device__ __forceinline__ double operator[](int e) const { return base[e * 16]; }
};

template <int NB>
__global__ __launch_bounds__(256, 2) void ba_eliminate_mfma(BaDev d, const Chunk* __restrict__ chunks,
                                                         const int* __restrict__ chunk_ids,
                                                         const int* __restrict__ sig_cams, double inv_radius,
                                                         double lm_lo, double lm_hi, int rank) {
  constexpr int NT = NB * (NB + 1) / 2;
  __shared__ __attribute__((aligned(16))) double s_cam[CAMD * 16];
  extern __shared__ __attribute__((aligned(16))) double s_M[];  // nw x 12 x MP panels; also the cross-wave reduction buffer
  __shared__ int s_gidx[64];  // local Gram index -> row/column of S; -2: the rhs column u; -1: padding
  const int nw = blockDim.x >> 6;  // 4 waves for long runs, 1 for runs of a few points (unstructured visibility)
  const Chunk ch = chunks[chunk_ids[blockIdx.x]];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = ch.n;
  const int* cams = sig_cams + ch.sig_off;
  const int sld = d.ld, fo = 6 * d.nc;
  for (int idx = tid; idx < n * CAMD; idx += (int)blockDim.x) {
    const int o = idx / CAMD, e = idx - o * CAMD;
    s_cam[e * 16 + o] = d.camd[(size_t)CAMD * cams[o] + e];
  }
  for (int idx = tid; idx < nw * 12 * MP; idx += (int)blockDim.x) s_M[idx] = 0.0;
  if (tid < 64) {
    int gi = -1;
    if (tid < 6 * n) gi = 6 * cams[tid / 6] + tid % 6;
    else if (tid == 6 * n) gi = fo;
    else if (tid == 6 * n + 1) gi = -2;
    s_gidx[tid] = gi;
  }
  const int o = lane & 15, q = lane >> 4;
  const bool valid_o = o < n;
  const int oc = valid_o ? o : 0;  // idle lanes shadow observation 0 (finite data), masked out below
  double sc[6];
  {
    const int cam = cams[oc];
#pragma unroll
    for (int j = 0; j < 6; ++j) sc[j] = d.scale_c[6 * cam + j];
  }
  const double sf = *d.scale_f, focal = *d.focal;
  const int kobs0 = d.optr[ch.p0];
  __syncthreads();

  v4d acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  double gmax = 0.0;
  int nfail = 0;
  double* Mw = s_M + wave * (12 * MP);
  const CamLds cd{s_cam + oc};
  const int frow = lane >> 4, fcol = lane & 15;

  for (int quad = wave; 4 * quad < ch.cnt; quad += nw) {
    const int pi = 4 * quad + q;
    const bool pv = pi < ch.cnt;
    const int pl = pv ? pi : ch.cnt - 1;
    const int p = ch.p0 + pl;
    const double X[3] = {d.pts[3 * p], d.pts[3 * p + 1], d.pts[3 * p + 2]};
    const double sp[3] = {d.scale_p[3 * p], d.scale_p[3 * p + 1], d.scale_p[3 * p + 2]};
    ObsLin ol;
    {
      const double2 xy = d.oxy[kobs0 + pl * n + oc];
      obs_linearize(cd, X, focal, xy.x, xy.y, sc, sp, sf, ol);
    }
    const double live = (pv && valid_o) ? 1.0 : 0.0;
    // point block C = sum Jp^T Jp (lower: 00 10 11 20 21 22), gp = Jp^T r, wf = Jp^T Jf
    double red[12];
    red[0] = live * (ol.Jp[0] * ol.Jp[0] + ol.Jp[3] * ol.Jp[3]);
    red[1] = live * (ol.Jp[1] * ol.Jp[0] + ol.Jp[4] * ol.Jp[3]);
    red[2] = live * (ol.Jp[1] * ol.Jp[1] + ol.Jp[4] * ol.Jp[4]);
    red[3] = live * (ol.Jp[2] * ol.Jp[0] + ol.Jp[5] * ol.Jp[3]);
    red[4] = live * (ol.Jp[2] * ol.Jp[1] + ol.Jp[5] * ol.Jp[4]);
    red[5] = live * (ol.Jp[2] * ol.Jp[2] + ol.Jp[5] * ol.Jp[5]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[6 + a] = live * (ol.Jp[a] * ol.r0 + ol.Jp[3 + a] * ol.r1);
      red[9 + a] = live * (ol.Jp[a] * ol.Jf[0] + ol.Jp[3 + a] * ol.Jf[1]);
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) red[e] = row16_sum(red[e]);
    // LM damping of the point block: D^2 = clamp(diag) / radius
    double C[6] = {red[0], red[1], red[2], red[3], red[4], red[5]};
    C[0] += fmin(fmax(C[0], lm_lo), lm_hi) * inv_radius;
    C[2] += fmin(fmax(C[2], lm_lo), lm_hi) * inv_radius;
    C[5] += fmin(fmax(C[5], lm_lo), lm_hi) * inv_radius;
    double Li[6];
    const bool pd = chol3_inv(C, Li);
    if (!pd) {
      if (pv && o == 0) ++nfail;
      Li[0] = Li[1] = Li[2] = Li[3] = Li[4] = Li[5] = 0;
    }
    const double pvf = pv ? 1.0 : 0.0;
    // gradient of the point (unscaled) for the gradient tolerance
    gmax = fmax(gmax, pvf * fmax(fabs(red[6] * rcp_f64(sp[0])), fmax(fabs(red[7] * rcp_f64(sp[1])), fabs(red[8] * rcp_f64(sp[2])))));
    if (valid_o) {
      // W = Jc^T Jp (6x3); T = W Li^T : T[i][k] = sum_{a<=k} W[i][a] Li[k][a]; row k of the panel
      double* row0 = Mw + (3 * q) * MP + 6 * o;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double w0 = live * (ol.Jc[i] * ol.Jp[0] + ol.Jc[6 + i] * ol.Jp[3]);
        const double w1 = live * (ol.Jc[i] * ol.Jp[1] + ol.Jc[6 + i] * ol.Jp[4]);
        const double w2 = live * (ol.Jc[i] * ol.Jp[2] + ol.Jc[6 + i] * ol.Jp[5]);
        row0[i] = w0 * Li[0];
        row0[MP + i] = w0 * Li[1] + w1 * Li[2];
        row0[2 * MP + i] = w0 * Li[3] + w1 * Li[4] + w2 * Li[5];
      }
    }
    if (o == 0) {
      // border columns: t_f = Li wf, u = Li gp
      double* b0 = Mw + (3 * q) * MP + 6 * n;
      b0[0] = pvf * (Li[0] * red[9]);
      b0[MP] = pvf * (Li[1] * red[9] + Li[2] * red[10]);
      b0[2 * MP] = pvf * (Li[3] * red[9] + Li[4] * red[10] + Li[5] * red[11]);
      b0[1] = pvf * (Li[0] * red[6]);
      b0[MP + 1] = pvf * (Li[1] * red[6] + Li[2] * red[7]);
      b0[2 * MP + 1] = pvf * (Li[3] * red[6] + Li[4] * red[7] + Li[5] * red[8]);
    }
    // Gram update: the same fragment serves as A (M^T tile) and B (M tile) operand
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      double fr[NB];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) fr[blk] = Mw[(4 * ks + frow) * MP + 16 * blk + fcol];
      int t = 0;
#pragma unroll
      for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = ti; tj < NB; ++tj, ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[ti], fr[tj], acc[t], 0, 0, 0);
    }
  }

  // ---- cross-wave sum of the Gram tiles (s_M is free now), then one scatter into S per chunk
  __syncthreads();
#pragma unroll 1
  for (int w = 1; w < nw; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) s_M[(t * 4 + g) * 64 + lane] = acc[t][g];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] += s_M[(t * 4 + g) * 64 + lane];
    }
    __syncthreads();
  }
  double* scv = red_sc(d);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    gmax = fmax(gmax, __shfl_down(gmax, off));
    nfail += __shfl_down(nfail, off);
  }
  if (lane == 0) {
    if (nfail) atomic_add_f64(scv + 2, (double)nfail);
    atomic_max_pos_f64(scv + SC + rank, gmax);
  }
  if (wave == 0) {
    double* S = red_S(d);
    double* g = red_g(d);
    int t = 0;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
      for (int tj = ti; tj < NB; ++tj, ++t) {
        const int lc = 16 * tj + fcol;
        const int gc = s_gidx[lc];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int lr = 16 * ti + frow + 4 * gg;
          const int gr = s_gidx[lr];
          if (gr < 0 || lr > lc) continue;
          if (gc >= 0) atomic_add_f64(S + (size_t)gr * sld + gc, -acc[t][gg]);
          else if (gc == -2) atomic_add_f64(g + gr, -acc[t][gg]);
        }
      }
  }
}

// F^T F part of the reduced system from the camera-major observation list: thread per
// observation, register accumulation of the camera's 6x6 block (upper, 21), the focal border
// (6), F^T b (6), and the scalars Jf^2, Jf r, r^2; block-reduced, 36 atomics per workgroup.
__global__ __launch_bounds__(256) void ba_cam_blocks(BaDev d, const int* __restrict__ cptr,
                                                     const int* __restrict__ cpt,
                                                     const double2* __restrict__ cxy, int nsplit,
                                                     int norms_only /* unscaled diagonal into dc only */) {
  __shared__ double sh[4][36];
  const int c = blockIdx.x / nsplit, part = blockIdx.x - c * nsplit;
  const int k0 = cptr[c], k1 = cptr[c + 1];
  const int len = k1 - k0;
  const int per = (len + nsplit - 1) / nsplit;
  const int kb = k0 + part * per, ke = min(k1, kb + per);
  double a[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) a[e] = 0.0;
  const double* cd = d.camd + (size_t)CAMD * c;
  const double* scp = norms_only ? nullptr : d.scale_c + 6 * c;
  const double sf = norms_only ? 1.0 : *d.scale_f, focal = *d.focal;
  for (int k = kb + threadIdx.x; k < ke; k += 256) {
    const int p = cpt[k];
    const double2 xy = cxy[k];
    const double X[3] = {d.pts[3 * p], d.pts[3 * p + 1], d.pts[3 * p + 2]};
    double sp[3] = {1.0, 1.0, 1.0};
    if (!norms_only) {
      sp[0] = d.scale_p[3 * p];
      sp[1] = d.scale_p[3 * p + 1];
      sp[2] = d.scale_p[3 * p + 2];
    }
    ObsLin o;
    obs_linearize(cd, X, focal, xy.x, xy.y, scp, sp, sf, o);
    int e = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = i; j < 6; ++j) a[e++] += o.Jc[i] * o.Jc[j] + o.Jc[6 + i] * o.Jc[6 + j];
      a[21 + i] += o.Jc[i] * o.Jf[0] + o.Jc[6 + i] * o.Jf[1];
      a[27 + i] += o.Jc[i] * o.r0 + o.Jc[6 + i] * o.r1;
    }
    a[33] += o.Jf[0] * o.Jf[0] + o.Jf[1] * o.Jf[1];
    a[34] += o.Jf[0] * o.r0 + o.Jf[1] * o.r1;
    a[35] += o.r0 * o.r0 + o.r1 * o.r1;
  }
#pragma unroll
  for (int e = 0; e < 36; ++e) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[e] += __shfl_down(a[e], off);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < 36; ++e) sh[wave][e] = a[e];
  }
  __syncthreads();
  if (threadIdx.x < 36 && len > 0) {
    const int e = threadIdx.x;
    const double v = sh[0][e] + sh[1][e] + sh[2][e] + sh[3][e];
    double* S = red_S(d);
    double* g = red_g(d);
    double* gF = red_gF(d);
    double* dc = red_dc(d);
    const int sld = d.ld, fo = 6 * d.nc, r0 = 6 * c;
    if (norms_only) {
      int i = 0, rem = e;
      while (e < 21 && rem >= 6 - i) {
        rem -= 6 - i;
        ++i;
      }
      if (e < 21 && rem == 0) atomic_add_f64(dc + r0 + i, v);
      if (e == 33) atomic_add_f64(dc + fo, v);
    } else if (e < 21) {
      int i = 0, rem = e;
      while (rem >= 6 - i) {
        rem -= 6 - i;
        ++i;
      }
      const int j = i + rem;
      atomic_add_f64(S + (size_t)(r0 + i) * sld + r0 + j, v);
      if (i == j) atomic_add_f64(dc + r0 + i, v);
    } else if (e < 27) {
      atomic_add_f64(S + (size_t)(r0 + e - 21) * sld + fo, v);
    } else if (e < 33) {
      atomic_add_f64(g + r0 + e - 27, v);
      atomic_add_f64(gF + r0 + e - 27, v);
    } else if (e == 33) {
      atomic_add_f64(S + (size_t)fo * sld + fo, v);
      atomic_add_f64(dc + fo, v);
    } else if (e == 34) {
      atomic_add_f64(g + fo, v);
      atomic_add_f64(gF + fo, v);
    } else {
      atomic_add_f64(red_sc(d) + 0, v);
    }
  }
}

// Generic path of the Schur correction: one wave per point; any observation count up to FB_MAXN,
// cameras in any order, repeated cameras allowed.  Per-point atomics (no accumulation across
// points).  The F^T F part of these points comes from ba_cam_blocks like everyone else's.
__global__ __launch_bounds__(64) void ba_eliminate_generic(BaDev d, const int* __restrict__ plist, double radius,
                                                           double lm_lo, double lm_hi, int rank) {
  __shared__ __attribute__((aligned(16))) double s_T[(FB_MAXN + 1) * 18];
  __shared__ int s_cam[FB_MAXN];
  const int p = plist[blockIdx.x];
  const int lane = threadIdx.x;
  const int k0 = d.optr[p], n = d.optr[p + 1] - k0;
  const int dim = d.ld /* row stride of S */, fo = 6 * d.nc;
  const bool is_obs = lane < n;
  const int k = k0 + (is_obs ? lane : 0);
  const int mycam = d.ocam[k];
  if (is_obs) s_cam[lane] = mycam;
  const double X[3] = {d.pts[3 * p], d.pts[3 * p + 1], d.pts[3 * p + 2]};
  const double sp[3] = {d.scale_p[3 * p], d.scale_p[3 * p + 1], d.scale_p[3 * p + 2]};
  const double sf = *d.scale_f, focal = *d.focal;
  ObsLin o;
  {
    const double2 xy = d.oxy[k];
    obs_linearize(d.camd + (size_t)CAMD * mycam, X, focal, xy.x, xy.y, d.scale_c + 6 * mycam, sp, sf, o);
  }
  const double live = is_obs ? 1.0 : 0.0;
  double red[12];
  red[0] = live * (o.Jp[0] * o.Jp[0] + o.Jp[3] * o.Jp[3]);
  red[1] = live * (o.Jp[1] * o.Jp[0] + o.Jp[4] * o.Jp[3]);
  red[2] = live * (o.Jp[1] * o.Jp[1] + o.Jp[4] * o.Jp[4]);
  red[3] = live * (o.Jp[2] * o.Jp[0] + o.Jp[5] * o.Jp[3]);
  red[4] = live * (o.Jp[2] * o.Jp[1] + o.Jp[5] * o.Jp[4]);
  red[5] = live * (o.Jp[2] * o.Jp[2] + o.Jp[5] * o.Jp[5]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    red[6 + a] = live * (o.Jp[a] * o.r0 + o.Jp[3 + a] * o.r1);
    red[9 + a] = live * (o.Jp[a] * o.Jf[0] + o.Jp[3 + a] * o.Jf[1]);
  }
#pragma unroll
  for (int e = 0; e < 12; ++e) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) red[e] += __shfl_xor(red[e], off);
  }
  double C[6] = {red[0], red[1], red[2], red[3], red[4], red[5]};
  C[0] += fmin(fmax(C[0], lm_lo), lm_hi) / radius;
  C[2] += fmin(fmax(C[2], lm_lo), lm_hi) / radius;
  C[5] += fmin(fmax(C[5], lm_lo), lm_hi) / radius;
  double Li[6];
  const bool pd = chol3_inv(C, Li);
  if (!pd) Li[0] = Li[1] = Li[2] = Li[3] = Li[4] = Li[5] = 0;
  double* S = red_S(d);
  double* g = red_g(d);
  double* scv = red_sc(d);
  if (is_obs) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double w0 = o.Jc[i] * o.Jp[0] + o.Jc[6 + i] * o.Jp[3];
      const double w1 = o.Jc[i] * o.Jp[1] + o.Jc[6 + i] * o.Jp[4];
      const double w2 = o.Jc[i] * o.Jp[2] + o.Jc[6 + i] * o.Jp[5];
      s_T[lane * 18 + 3 * i + 0] = w0 * Li[0];
      s_T[lane * 18 + 3 * i + 1] = w0 * Li[1] + w1 * Li[2];
      s_T[lane * 18 + 3 * i + 2] = w0 * Li[3] + w1 * Li[4] + w2 * Li[5];
    }
  }
  double tf[3], u[3];
  tf[0] = Li[0] * red[9];
  tf[1] = Li[1] * red[9] + Li[2] * red[10];
  tf[2] = Li[3] * red[9] + Li[4] * red[10] + Li[5] * red[11];
  u[0] = Li[0] * red[6];
  u[1] = Li[1] * red[6] + Li[2] * red[7];
  u[2] = Li[3] * red[6] + Li[4] * red[7] + Li[5] * red[8];
  if (lane == 0) {
    atomic_add_f64(S + (size_t)fo * dim + fo, -(tf[0] * tf[0] + tf[1] * tf[1] + tf[2] * tf[2]));
    atomic_add_f64(g + fo, -(tf[0] * u[0] + tf[1] * u[1] + tf[2] * u[2]));
    if (!pd) atomic_add_f64(scv + 2, 1.0);
    const double gm = fmax(fabs(red[6] / sp[0]), fmax(fabs(red[7] / sp[1]), fabs(red[8] / sp[2])));
    atomic_max_pos_f64(scv + SC + rank, gm);
  }
  __syncthreads();
  if (is_obs) {  // border: S[cam][focal] -= T t_f ; g[cam] -= T u
    const int r0 = 6 * mycam;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double* T = s_T + lane * 18 + 3 * i;
      atomic_add_f64(S + (size_t)(r0 + i) * dim + fo, -(T[0] * tf[0] + T[1] * tf[1] + T[2] * tf[2]));
      atomic_add_f64(g + r0 + i, -(T[0] * u[0] + T[1] * u[1] + T[2] * u[2]));
    }
  }
  const int npairs = n * (n + 1) / 2;
  for (int q = lane; q < npairs; q += 64) {
    // q -> (a <= b)
    int a = 0, rem = q;
    while (rem >= n - a) {
      rem -= n - a;
      ++a;
    }
    const int b = a + rem;
    const int ca = s_cam[a], cb = s_cam[b];
    const double* Ta = s_T + a * 18;
    const double* Tb = s_T + b * 18;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        const double v = Ta[3 * i] * Tb[3 * j] + Ta[3 * i + 1] * Tb[3 * j + 1] + Ta[3 * i + 2] * Tb[3 * j + 2];
        if (ca < cb) {
          atomic_add_f64(S + (size_t)(6 * ca + i) * dim + 6 * cb + j, -v);
        } else if (ca > cb) {
          atomic_add_f64(S + (size_t)(6 * cb + j) * dim + 6 * ca + i, -v);
        } else if (a == b) {
          if (i <= j) atomic_add_f64(S + (size_t)(6 * ca + i) * dim + 6 * ca + j, -v);
        } else {  // same camera twice in one point: M + M^T, upper part
          if (i <= j) atomic_add_f64(S + (size_t)(6 * ca + i) * dim + 6 * ca + j, -v);
          if (j <= i) atomic_add_f64(S + (size_t)(6 * ca + j) * dim + 6 * ca + i, -v);
        }
      }
  }
}

// LM diagonal of the camera/focal columns onto S; gradient max over those columns
__global__ __launch_bounds__(1024) void ba_finalize(BaDev d, double radius, double lm_lo, double lm_hi, int world,
                                                    int add_diag) {
  __shared__ double sh[16];
  double* S = red_S(d);
  const double* gF = red_gF(d);
  const double* dc = red_dc(d);
  double* scv = red_sc(d);
  double gm = 0;
  for (int i = threadIdx.x; i < d.dim; i += blockDim.x) {
    const double v = fmin(fmax(dc[i], lm_lo), lm_hi);
    d.diag[i] = v;
    if (add_diag) S[(size_t)i * d.ld + i] += v / radius;
    const double s = i < 6 * d.nc ? d.scale_c[i] : *d.scale_f;
    gm = fmax(gm, fabs(gF[i] / s));
  }
  for (int i = d.dim + threadIdx.x; i < d.ld; i += blockDim.x) S[(size_t)i * d.ld + i] = 1.0;  // padding
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_down(gm, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) gm = fmax(gm, sh[w]);
    for (int r = 0; r < world; ++r) gm = fmax(gm, scv[SC + r]);
    scv[3] = gm;  // gradient max norm (unscaled), all parameter blocks, all ranks
  }
}

// ---------------------------------------------------------------- dense Cholesky (f64)
// A is the row-major upper triangle of S == column-major lower triangle: L(r,c) = A[c*ld + r],
// r >= c, so a column of L is contiguous.  ld = dim rounded up to 32 (identity on the padded
// diagonal), so no tile needs a bounds check.  The rhs g is carried as one extra tile row
// (row 0 of tile row nt, kept in y) so that y = L^-1 g falls out of the factorisation;
// chol_backsolve then solves L^T z = y.
//
// One launch per 32-column panel (right-looking, the update of the previous panel fused in).
// Launch k, one 2-wave workgroup per remaining tile (ti >= tj >= k):
//   * tiles right of block column k only take the pending rank-32 update of panel k-1:
//     T -= L(ti,k-1) L(tj,k-1)^T on v_mfma_f64_16x16x4_f64, operands straight from global
//     memory (16 contiguous doubles per k index), each wave one 16-column half of the tile;
//   * the tiles of block column k carry the factorisation, with no inter-workgroup dependency
//     inside the launch: wave 0 updates + factors the diagonal tile (every such workgroup
//     redundantly: POTRF in registers, row per lane, pivot and next-column terms by
//     v_readlane, the rest of the column broadcast through LDS), wave 1 updates the
//     workgroup's own tile and solves it against L_kk (row per lane), trailing the
//     factorisation by one 4-column block (progress flag in LDS; the factorising wave never waits).
#ifdef SFM_CHOL_STAMPS
// diagnostic build only (scripts/chol_stamps.py): s_memtime at the phase boundaries of the panel
// workgroup ti_rel == 1 of launch k == 4, written to a buffer nothing else reads
__device__ unsigned long long g_chol_stamps[16];
#define CHOL_STAMP(slot)                                                                        \
  do {                                                                                          \
    if (k == 4 && ti_rel == 1 && tj_rel == 0 && lane == 0) {                                    \
      unsigned long long t_;                                                                    \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
      g_chol_stamps[slot] = t_;                                                                 \
    }                                                                                           \
  } while (0)
#else
#define CHOL_STAMP(slot)
#endif
constexpr int CB = 32;
constexpr int CBP = 34;  // LDS row pitch in doubles: 16-byte aligned rows, conflict-free tile writes

// P[c][r] -= sum_kk Lc[c][kk] * Lr[r][kk] for the 16x16 sub-tiles (ci, ri) of a 32x32 tile.
// MFMA roles: A operand = Lc (lane: row c = lane&15, k = lane>>4), B operand = Lr (col r =
// lane&15), so that the result's lane index is the memory-contiguous tile row r and its 4
// registers are tile columns c = (lane>>4) + 4g.
#define CHOL_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0)

template <bool RHS>
__device__ __forceinline__ void chol_body(double* __restrict__ A, double* __restrict__ y, double* __restrict__ dinv,
                                          double* __restrict__ linv, int ld, int nt, int k, int* __restrict__ info,
                                          int ti_rel, int tj_rel, double* sD, double* sT, double* sLr, double* sdi,
                                          int* s_prog_p) {
  const int m = nt - k;
  constexpr bool is_rhs = RHS;  // the rhs tile row is its own instantiation: plain loads for everybody else
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j16 = lane & 15, q = lane >> 4;
  const int r0 = (k + ti_rel) * CB, c0 = (k + tj_rel) * CB, p0 = (k - 1) * CB;
  const bool rlane = j16 == 0;  // the rhs tile row has one real row: tile row 0
  if (threadIdx.x == 0) (*s_prog_p) = 0;
  __syncthreads();
  CHOL_STAMP(wave == 0 ? 0 : 8);

  // element (tile row r, column index col) of the workgroup's tile row: L(r0+r, col) or, for the
  // rhs row, y[col] on tile row 0 and zero elsewhere
  auto ld_row = [&](int r, int col) -> double {
    if (is_rhs) return r == 0 ? y[col] : 0.0;
    return A[(size_t)col * ld + r0 + r];
  };

  if (tj_rel != 0) {
    // ---------------- trailing tile: pending update of panel k-1 only (k >= 1 here)
    const int ci = wave;
    double a[8], b[2][8];
    v4d acc[2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int kk = p0 + 4 * ks + q;
      a[ks] = -A[(size_t)kk * ld + c0 + 16 * ci + j16];
      b[0][ks] = ld_row(j16, kk);
      b[1][ks] = ld_row(16 + j16, kk);
    }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[ri][g] = ld_row(16 * ri + j16, c0 + 16 * ci + q + 4 * g);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      CHOL_MFMA(acc[0], a[ks], b[0][ks]);
      CHOL_MFMA(acc[1], a[ks], b[1][ks]);
    }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = c0 + 16 * ci + q + 4 * g;
        if (is_rhs) {
          if (ri == 0 && rlane) y[col] = acc[ri][g];
        } else {
          A[(size_t)col * ld + r0 + 16 * ri + j16] = acc[ri][g];
        }
      }
    return;
  }

  // ---------------- block column k: update, factor the diagonal tile, solve the own tile
  // the factorising wave is the launch's critical path: it outranks the solving wave that trails
  // it (and polls its progress through LDS), and both outrank the trailing-tile filler
  if (wave == 0) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(1);
  const bool owner = ti_rel == 0;
  const int i = lane & 31;  // row of the tile handled by this lane in the row-per-lane phases
  if (wave == 0 || !owner) {
    // wave 0: diagonal tile (rows c0..); wave 1: own tile (rows r0.. / the rhs row)
    const bool diag = wave == 0;
    double a[2][8], b[2][8];
    v4d acc[2][2];  // [ci][ri]
    if (k > 0) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = p0 + 4 * ks + q;
        a[0][ks] = -A[(size_t)kk * ld + c0 + j16];
        a[1][ks] = -A[(size_t)kk * ld + c0 + 16 + j16];
        if (!diag) {
          b[0][ks] = ld_row(j16, kk);
          b[1][ks] = ld_row(16 + j16, kk);
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = c0 + 16 * ci + q + 4 * g;
          acc[ci][ri][g] = diag ? A[(size_t)col * ld + c0 + 16 * ri + j16] : ld_row(16 * ri + j16, col);
        }
#ifdef SFM_CHOL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CHOL_STAMP(wave == 0 ? 1 : 9);
    if (k > 0) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int ri = 0; ri < 2; ++ri) CHOL_MFMA(acc[ci][ri], a[ci][ks], diag ? -a[ri][ks] : b[ri][ks]);
    }
    double* dst = diag ? sD : sT;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[(16 * ri + j16) * CBP + 16 * ci + q + 4 * g] = acc[ci][ri][g];
  }
  // (each wave reads back only what it wrote itself: LDS operations of one wave stay in order)
  CHOL_STAMP(wave == 0 ? 2 : 10);

  if (wave == 0) {
    // ---- POTRF of the diagonal tile: lane i = row i (both half-waves alike)
    double d[CB];
#pragma unroll
    for (int c = 0; c < CB; c += 2) {
      const double2 v = *(const double2*)(sD + i * CBP + c);
      d[c] = v.x;
      d[c + 1] = v.y;
    }
    // left-looking inside the tile: column j first takes the products with the finished columns
    // (row j of L broadcast from LDS, the newest column through v_readlane so that the pivot
    // chain does not wait for an LDS round trip), then pivot -> rsq -> scale.  Lanes 32..63
    // mirror lanes 0..31 (same values to the same LDS addresses).
    // Software-pipelined: the products of column j+1 with the columns before j are formed in
    // the shadow of column j's pivot chain (readlane -> rsq -> two Newton steps -> scale), so the
    // next chain starts with one FMA.  (Measured, scripts/ubench/op_rate64: a lone wave issues
    // an f64 op every ~5.4 cycles, 8.4 when dependent, v_rsq_f64 16: the ~60 instructions of a
    // column, not the chain, set its ~450 cycles.)
    bool bad = false;
    double pre = d[0];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      double v = pre;
      if (j >= 1) v -= d[j - 1] * readlane_f64(d[j - 1], j);
      const double djj = readlane_f64(v, j);
      bad |= !(djj > 0.0);
      const double r = rsqrt_f64(djj);
      const double l = i >= j ? v * r : 0.0;  // lane j: djj * r = sqrt(djj)
      if (j + 1 < CB) {
        double pa[4] = {d[j + 1], 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c + 1 < j; c += 2) {
          const double2 x = *(const double2*)(sLr + (j + 1) * CBP + c);
          pa[c & 3] -= d[c] * x.x;
          pa[(c + 1) & 3] -= d[c + 1] * x.y;
        }
        if (j & 1) pa[(j - 1) & 3] -= d[j - 1] * sLr[(j + 1) * CBP + j - 1];
        pre = (pa[0] + pa[1]) + (pa[2] + pa[3]);
      }
      d[j] = l;
      sLr[i * CBP + j] = l;
      sdi[j] = r;
      __builtin_amdgcn_sched_barrier(0);  // keep the columns in order
      // progress flag for the solving wave (LDS operations of one wave stay in order, so the
      // columns are in LDS before the flag); the factorising wave never waits for anybody
      if ((j & 3) == 3) {
        asm volatile("" ::: "memory");
        *(volatile int*)&(*s_prog_p) = j + 1;
      }
    }
    CHOL_STAMP(3);
    if (owner) {
      // the diagonal tile's owner publishes L_kk and 1/diag
      if (bad && lane == 0) atomicExch(info, k * CB + 1);  // the host discards the step
      if (lane < CB) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
          if (c <= i) A[(size_t)(c0 + c) * ld + c0 + i] = d[c];
        dinv[c0 + i] = sdi[i];
      }
    }
  } else {
    // ---- own tile (or the rhs row): X = T L_kk^-T, lane i = row i, one 8-column block behind
    // (the diagonal tile's owner solves the identity instead: X = L_kk^-T, kept for the
    // backward substitution, whose diagonal solves then are plain 32x32 products)
    const unsigned prog_addr = (unsigned)(size_t)(__attribute__((address_space(3))) int*)&(*s_prog_p);
    double t[CB];
    if (!owner) {
#pragma unroll
      for (int c = 0; c < CB; c += 2) {
        const double2 v = *(const double2*)(sT + i * CBP + c);
        t[c] = v.x;
        t[c + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int c = 0; c < CB; ++c) t[c] = c == i ? 1.0 : 0.0;
    }
#pragma unroll
    for (int jb = 0; jb < CB / 4; ++jb) {
      {
        // wait until the factorising wave has published 4*jb+4 columns: one opaque asm block (a
        // C loop here splits the unrolled solve into basic blocks and the allocator spills)
        int seen_;
        asm volatile(
            "1:\n\t"
            "ds_read_b32 %0, %1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_lt_i32 vcc, %0, %2\n\t"
            "s_cbranch_vccz 2f\n\t"
            "s_sleep 4\n\t"
            "s_branch 1b\n\t"
            "2:\n\t"
            : "=&v"(seen_)
            : "v"(prog_addr), "v"(4 * jb + 4)
            : "vcc", "memory");
      }
      {
#pragma unroll
        for (int j = 4 * jb; j < 4 * jb + 4; ++j) {
          double ta[4] = {t[j], 0.0, 0.0, 0.0};
#pragma unroll
          for (int c = 0; c + 1 < j; c += 2) {
            const double2 x = *(const double2*)(sLr + j * CBP + c);
            ta[c & 3] -= t[c] * x.x;
            ta[(c + 1) & 3] -= t[c + 1] * x.y;
          }
          if (j & 1) ta[(j - 1) & 3] -= t[j - 1] * sLr[j * CBP + j - 1];
          t[j] = ((ta[0] + ta[1]) + (ta[2] + ta[3])) * sdi[j];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    CHOL_STAMP(11);
    if (owner) {
      if (lane < CB) {
#pragma unroll
        for (int c = 0; c < CB; ++c) linv[(size_t)(c0 + c) * CB + i] = t[c];  // (L_kk^-T)[i][c], column-major
      }
    } else if (is_rhs) {
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < CB; ++c) y[c0 + c] = t[c];
      }
    } else if (lane < CB) {
#pragma unroll
      for (int c = 0; c < CB; ++c) A[(size_t)(c0 + c) * ld + r0 + i] = t[c];
    }
    CHOL_STAMP(12);
  }
}

__global__ __launch_bounds__(128) void chol_step(double* __restrict__ A, double* __restrict__ y,
                                                 double* __restrict__ dinv, double* __restrict__ linv, int ld,
                                                 int nt, int k, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double sD[CB * CBP];   // updated diagonal tile [row][col]
  __shared__ __attribute__((aligned(16))) double sT[CB * CBP];   // updated own tile      [row][col]
  __shared__ __attribute__((aligned(16))) double sLr[CB * CBP];  // L_kk [row][col]
  __shared__ double sdi[CB];                                     // 1 / diag(L_kk)
  __shared__ int s_prog;                                         // columns of L_kk finished so far
  const int m = nt - k;  // remaining tile rows (the rhs row comes on top)
  // block -> tile: the m+1 tiles of block column k first (they carry the factorisation)
  int ti_rel, tj_rel;
  if ((int)blockIdx.x <= m) {
    ti_rel = blockIdx.x;
    tj_rel = 0;
  } else {
    int t = blockIdx.x - (m + 1);
    ti_rel = 1;
    while (true) {
      const int w = ti_rel < m ? ti_rel : m - 1;  // tj_rel in 1..min(ti_rel, m-1)
      if (t < w) break;
      t -= w;
      ++ti_rel;
    }
    tj_rel = 1 + t;
  }
  if (ti_rel == m) chol_body<true>(A, y, dinv, linv, ld, nt, k, info, ti_rel, tj_rel, sD, sT, sLr, sdi, &s_prog);
  else chol_body<false>(A, y, dinv, linv, ld, nt, k, info, ti_rel, tj_rel, sD, sT, sLr, sdi, &s_prog);
}
#undef CHOL_MFMA

// L^T z = y.  U = L^T is upper triangular and row-major in this storage (U[i][j] = A[i*ld + j]),
// so row i of U is contiguous.  Block rows are taken in groups of GB (256 rows) from the bottom:
//   chol_backsolve_group  one workgroup of 8 waves per group: a chain wave (32x32 products with
//                         the diagonal tiles' inverses) and seven owner waves that fold finished
//                         blocks into the ones further up (see the kernel);
//   chol_backsolve_gemv   folds the group's solution into y of every row above the group, one
//                         wave per row (coalesced 2 KB row segments), all CUs.
// A single workgroup pulling the whole triangle (5.8 MB at cfg4) is bound by one CU's load
// bandwidth; the grouping leaves it 1/GB of the triangle.
constexpr int GB = 8;

// Bounded LDS spin (one opaque asm block, see chol_step): until *flag >= need.
__device__ __forceinline__ void lds_wait_ge(const int* flag, int need) {
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) int*)flag;
  int seen_, budget_ = 1 << 20;
  asm volatile(
      "1:\n\t"
      "ds_read_b32 %0, %2\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_lt_i32 vcc, %0, %3\n\t"
      "s_cbranch_vccz 2f\n\t"
      "s_sub_u32 %1, %1, 1\n\t"
      "s_cmp_eq_u32 %1, 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_sleep 1\n\t"
      "s_branch 1b\n\t"
      "2:\n\t"
      : "=&v"(seen_), "+s"(budget_)
      : "v"(addr), "v"(need)
      : "vcc", "scc", "memory");
}

// One workgroup of 8 waves per group of GB block rows.  All waves first stage what the chain
// will need -- the inverses of the group's diagonal tiles and the tiles right above the diagonal,
// 16 KB per block -- into LDS.  Wave 0 then is the dependency chain: for each block from the
// bottom, z_b = L_bb^-T y_b (a 32x32 product) and the fold of z_b into the block right above.
// Waves 1..7 own the other targets: the owner of block b folds the solutions of the blocks two
// or more below into y_b as the chain publishes them, and hands y_b to the chain just before
// it is needed.  Lane = row of the target block; hand-offs go through LDS flags.
__global__ __launch_bounds__(512) void chol_backsolve_group(const double* __restrict__ A, const double* __restrict__ y,
                                                             const double* __restrict__ linv, double* __restrict__ z,
                                                             int ld, int kb_lo, int kb_hi) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];  // [GB][CB*CB] inverses | [GB][CB*CB] adjacent tiles
  __shared__ __attribute__((aligned(16))) double s_z[GB][CB];    // published by the chain
  __shared__ __attribute__((aligned(16))) double s_y[GB][CB];    // published by the owners
  __shared__ __attribute__((aligned(16))) double s_tmp[CB];
  __shared__ int s_zready[GB], s_ydone[GB];
  double* s_inv = s_dyn;                  // [b][j*CB + i] = (L_bb^-T)[i][j]
  double* s_adj = s_dyn + GB * CB * CB;   // [b][r*CB + i] = L(row r of block b, column i of block b-1)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31;
  const int nb = kb_hi - kb_lo;
  if (tid < GB) {
    s_zready[tid] = 0;
    s_ydone[tid] = 0;
  }
  for (int e = tid; e < nb * CB * CB / 2; e += 512) {  // 16-byte pieces
    const int b = e / (CB * CB / 2), w = e % (CB * CB / 2);
    const int k0 = (kb_lo + b) * CB;
    *(double2*)(s_inv + b * CB * CB + 2 * w) = *(const double2*)(linv + (size_t)k0 * CB + 2 * w);
    if (b > 0) {
      const int c = w % CB, r = 2 * (w / CB);  // column c of block b-1, rows r, r+1 of block b (conflict-free LDS writes)
      const double2 v = *(const double2*)(A + (size_t)(k0 - CB + c) * ld + k0 + r);
      s_adj[b * CB * CB + r * CB + c] = v.x;
      s_adj[b * CB * CB + (r + 1) * CB + c] = v.y;
    }
  }
  __syncthreads();
  if (wave == 0) {
    // ---- the chain
    double adj = 0.0;
    for (int b = nb - 1; b >= 0; --b) {
      const double* inv = s_inv + b * CB * CB + i;
      lds_wait_ge(&s_ydone[b], 1);
      const double yv = s_y[b][i] - adj;
      if (lane < CB) s_tmp[i] = yv;
      double z0 = 0.0, z1 = 0.0;
#pragma unroll
      for (int j = 0; j < CB; j += 2) {
        const double2 t = *(const double2*)(s_tmp + j);
        z0 += inv[j * CB] * t.x;
        z1 += inv[(j + 1) * CB] * t.y;
      }
      const double zi = z0 + z1;
      if (lane < CB) {
        s_z[b][i] = zi;
        z[(size_t)(kb_lo + b) * CB + i] = zi;
      }
      asm volatile("" ::: "memory");
      *(volatile int*)&s_zready[b] = 1;
      if (b > 0) {
        const double* ad = s_adj + b * CB * CB + i;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int r = 0; r < CB; r += 2) {
          const double2 t = *(const double2*)(&s_z[b][r]);
          a0 += ad[r * CB] * t.x;
          a1 += ad[(r + 1) * CB] * t.y;
        }
        adj = a0 + a1;
      }
    }
    return;
  }
  // ---- owners: wave w owns the targets b with 1 + b % 7 == w
  for (int b = nb - 1; b >= 0; --b) {
    if (1 + b % 7 != wave) continue;
    const int c0 = (kb_lo + b) * CB;
    double yacc = y[c0 + i];
    for (int sblk = nb - 1; sblk >= b + 2; --sblk) {
      const int r0 = (kb_lo + sblk) * CB;
      double col[CB];  // column i of tile (sblk, b): fetched before waiting for z
#pragma unroll
      for (int r = 0; r < CB; r += 2) {
        const double2 v = *(const double2*)(A + (size_t)(c0 + i) * ld + r0 + r);
        col[r] = v.x;
        col[r + 1] = v.y;
      }
      lds_wait_ge(&s_zready[sblk], 1);
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int r = 0; r < CB; r += 2) {
        const double2 t = *(const double2*)(&s_z[sblk][r]);
        a0 += col[r] * t.x;
        a1 += col[r + 1] * t.y;
      }
      yacc -= a0 + a1;
    }
    if (lane < CB) s_y[b][i] = yacc;
    asm volatile("" ::: "memory");
    *(volatile int*)&s_ydone[b] = 1;
  }
}

// y[i] -= sum_{j in [c_lo, c_hi)} U[i][j] z[j] for the rows above the group, one wave per row
__global__ __launch_bounds__(256) void chol_backsolve_gemv(const double* __restrict__ A, double* __restrict__ y,
                                                           const double* __restrict__ z, int ld, int c_lo, int c_hi) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= c_lo) return;
  const double* ui = A + (size_t)row * ld;
  double acc = 0.0;
  for (int j = c_lo + 2 * lane; j < c_hi; j += 128) {
    const double2 uu = *(const double2*)(ui + j);
    const double2 zz = *(const double2*)(z + j);
    acc += uu.x * zz.x + uu.y * zz.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) y[row] -= acc;
}

// ---------------------------------------------------------------- step application
// candidate cameras / focal: x + (-z)*scale, their tables, and the camera part of the norms
__global__ void ba_cand_cams(BaDev d, const unsigned char* __restrict__ cam_used, int rank) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  double sn2 = 0, cn2 = 0;
  if (c < d.nc) {
    double cam[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double dl = -d.z[6 * c + j] * d.scale_c[6 * c + j];
      cam[j] = d.cams[6 * c + j] + dl;
      d.cams_c[6 * c + j] = cam[j];
      if (cam_used[c]) {
        sn2 += dl * dl;
        cn2 += cam[j] * cam[j];
      }
    }
    cam_table(cam, d.camd_c + (size_t)CAMD * c, true);
  }
  if (c == d.nc) {
    const double dl = -d.z[6 * d.nc] * (*d.scale_f);
    const double f = *d.focal + dl;
    *d.focal_c = f;
    sn2 += dl * dl;
    cn2 += f * f;
  }
  if (rank == 0 && (sn2 != 0 || cn2 != 0)) {
    atomic_add_f64(d.red2 + 2, sn2);
    atomic_add_f64(d.red2 + 3, cn2);
  }
}

// per point: back-substitute, model cost change, candidate point + candidate cost
__global__ __launch_bounds__(256) void ba_backsub(BaDev d, double radius, double lm_lo, double lm_hi) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double mcc = 0, cost_c = 0, sn2 = 0, cn2 = 0;
  if (p < d.np) {
    const double X[3] = {d.pts[3 * p], d.pts[3 * p + 1], d.pts[3 * p + 2]};
    const double sp[3] = {d.scale_p[3 * p], d.scale_p[3 * p + 1], d.scale_p[3 * p + 2]};
    const double sf = *d.scale_f, focal = *d.focal, focal_c = *d.focal_c;
    const double zf = d.z[6 * d.nc];
    const int k0 = d.optr[p], k1 = d.optr[p + 1];
    double C[6] = {0, 0, 0, 0, 0, 0}, e[3] = {0, 0, 0};
    for (int k = k0; k < k1; ++k) {
      const int c = d.ocam[k];
      const double2 xy = d.oxy[k];
      ObsLin o;
      obs_linearize(d.camd + (size_t)CAMD * c, X, focal, xy.x, xy.y, d.scale_c + 6 * c, sp, sf, o);
      double s0 = o.r0 - o.Jf[0] * zf, s1 = o.r1 - o.Jf[1] * zf;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double zj = d.z[6 * c + j];
        s0 -= o.Jc[j] * zj;
        s1 -= o.Jc[6 + j] * zj;
      }
      C[0] += o.Jp[0] * o.Jp[0] + o.Jp[3] * o.Jp[3];
      C[1] += o.Jp[1] * o.Jp[0] + o.Jp[4] * o.Jp[3];
      C[2] += o.Jp[1] * o.Jp[1] + o.Jp[4] * o.Jp[4];
      C[3] += o.Jp[2] * o.Jp[0] + o.Jp[5] * o.Jp[3];
      C[4] += o.Jp[2] * o.Jp[1] + o.Jp[5] * o.Jp[4];
      C[5] += o.Jp[2] * o.Jp[2] + o.Jp[5] * o.Jp[5];
#pragma unroll
      for (int a = 0; a < 3; ++a) e[a] += o.Jp[a] * s0 + o.Jp[3 + a] * s1;
    }
    C[0] += fmin(fmax(C[0], lm_lo), lm_hi) / radius;
    C[2] += fmin(fmax(C[2], lm_lo), lm_hi) / radius;
    C[5] += fmin(fmax(C[5], lm_lo), lm_hi) / radius;
    double Li[6];
    if (!chol3_inv(C, Li)) Li[0] = Li[1] = Li[2] = Li[3] = Li[4] = Li[5] = 0;
    // y = C^-1 e = Li^T (Li e); step = -y
    const double t0 = Li[0] * e[0], t1 = Li[1] * e[0] + Li[2] * e[1], t2 = Li[3] * e[0] + Li[4] * e[1] + Li[5] * e[2];
    const double stp[3] = {-(Li[0] * t0 + Li[1] * t1 + Li[3] * t2), -(Li[2] * t1 + Li[4] * t2), -(Li[5] * t2)};
    double Xc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double dl = stp[j] * sp[j];
      Xc[j] = X[j] + dl;
      d.pts_c[3 * p + j] = Xc[j];
      sn2 += dl * dl;
      cn2 += Xc[j] * Xc[j];
    }
    for (int k = k0; k < k1; ++k) {
      const int c = d.ocam[k];
      const double2 xy = d.oxy[k];
      ObsLin o;
      obs_linearize(d.camd + (size_t)CAMD * c, X, focal, xy.x, xy.y, d.scale_c + 6 * c, sp, sf, o);
      double m0 = -o.Jf[0] * zf, m1 = -o.Jf[1] * zf;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double zj = d.z[6 * c + j];
        m0 -= o.Jc[j] * zj;
        m1 -= o.Jc[6 + j] * zj;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        m0 += o.Jp[j] * stp[j];
        m1 += o.Jp[3 + j] * stp[j];
      }
      mcc += m0 * (o.r0 + m0 / 2.0) + m1 * (o.r1 + m1 / 2.0);
      double r0, r1;
      obs_residual(d.camd_c + (size_t)CAMD * c, Xc, focal_c, xy.x, xy.y, r0, r1);
      cost_c += r0 * r0 + r1 * r1;
    }
  }
  __shared__ double sh[4][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mcc += __shfl_down(mcc, o);
    cost_c += __shfl_down(cost_c, o);
    sn2 += __shfl_down(sn2, o);
    cn2 += __shfl_down(cn2, o);
  }
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    sh[0][w] = mcc;
    sh[1][w] = cost_c;
    sh[2][w] = sn2;
    sh[3][w] = cn2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_add_f64(d.red2 + 0, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    atomic_add_f64(d.red2 + 1, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
    atomic_add_f64(d.red2 + 2, sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3]);
    atomic_add_f64(d.red2 + 3, sh[3][0] + sh[3][1] + sh[3][2] + sh[3][3]);
  }
}

__global__ void ba_cam_norm(BaDev d, const unsigned char* __restrict__ cam_used, double* out) {
  // ||x||^2 over used cameras + focal (single block)
  double s = 0;
  for (int i = threadIdx.x; i < 6 * d.nc; i += blockDim.x)
    if (cam_used[i / 6]) s += d.cams[i] * d.cams[i];
  if (threadIdx.x == 0) s += (*d.focal) * (*d.focal);
  __shared__ double sh[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    *out = s;
  }
}

}  // namespace

// ================================================================= host side
struct LmState {
  bool started = false, have_lin = false;
  bool lin_unread = false;  // cost / gradient of the enqueued linearisation not read back yet
  double radius = 1e4, decrease_factor = 2.0;
  int invalid = 0, iter = 0, nsucc = 0;
  double x_norm = 0, cost = 0, initial_cost = 0, gmax = 0;
};

struct sfmhip_ba {
  sfmhip_ctx* ctx = nullptr;
  LmState lm;
  int nc = 0, np_in = 0, no_in = 0;  // as given
  int np = 0, no = 0;                // with >= 1 observation, sorted order
  int dim = 0, ld = 0;
  size_t ssz = 0;  // ld*ld: doubles of the S part of `red`
  BaDev d{};
  std::vector<int> perm;  // sorted point -> input point
  std::vector<unsigned char> h_cam_used;
  unsigned char* d_cam_used = nullptr;
  bool cam_used_known = false;
  // plan
  Chunk* d_chunks = nullptr;
  int* d_chunk_ids[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int n_chunk_ids[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [NB-1]: 4-wave workgroups (long runs), [4 + NB-1]: 1-wave (short runs)
  int* d_sig_cams = nullptr;
  int* d_cptr = nullptr;  // camera-major observation list: cptr[nc+1], cpt[no], cxy[no]
  int* d_cpt = nullptr;
  double2* d_cxy = nullptr;
  int cam_split = 1;
  int* d_fb_points = nullptr;
  int n_fb = 0;
  // device storage owned
  std::vector<void*> allocs;
  size_t red_count = 0;
  double* h_sc = nullptr;  // pinned: scalars read back per iteration
  // comm
  sfmhip_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int rank = 0, world = 1;
  // state
  bool scale_ready = false;
  std::vector<double> h_pts_unused;  // input points without observations keep their values
  std::vector<double> h_pts_in;
  double x_norm = 0;
  // timing
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double t_acc[4] = {0, 0, 0, 0};
  int launches = 0;
};

// fn(lo, hi) over [0, n) on up to 8 host threads (problem set-up only)
template <typename F>
static void host_parallel_for(int n, F fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int nth = (int)std::max(1u, std::min(8u, hw ? hw : 1u));
  if (n < 20000 || nth == 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t) {
    const int lo = (int)((long long)n * t / nth), hi = (int)((long long)n * (t + 1) / nth);
    th.emplace_back([=, &fn]() { fn(lo, hi); });
  }
  for (auto& x : th) x.join();
}

template <typename T>
static int ba_alloc(sfmhip_ba* b, T** p, size_t n) {
  SFM_TRY(sfm_dev_alloc(p, n));
  b->allocs.push_back((void*)*p);
  return SFMHIP_OK;
}

#ifdef SFM_CHOL_STAMPS
extern "C" int sfmhip_debug_chol_stamps(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_chol_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
#endif

extern "C" void sfmhip_ba_default_opts(sfmhip_ba_opts* o) {
  if (!o) return;
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

extern "C" int sfmhip_ba_create(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, const int32_t* obs_cam,
                                const int32_t* obs_pt, const double* obs_xy, sfmhip_ba** out) {
  if (!ctx || !out || n_cam <= 0 || n_pt < 0 || n_obs < 0) return SFMHIP_ERR_ARG;
  if (n_obs && (!obs_cam || !obs_pt || !obs_xy)) return SFMHIP_ERR_ARG;
  for (int o = 0; o < n_obs; ++o)
    if (obs_cam[o] < 0 || obs_cam[o] >= n_cam || obs_pt[o] < 0 || obs_pt[o] >= n_pt) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  const bool prof_ = getenv("SFMHIP_PROFILE_CREATE") != nullptr;
  auto tp_ = std::chrono::steady_clock::now();
  auto lap_ = [&](const char* what) {
    if (!prof_) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[sfmhip_ba_create] %-22s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp_).count());
    tp_ = now;
  };
  sfmhip_ba* b = new sfmhip_ba();
  b->ctx = ctx;
  b->nc = n_cam;
  b->np_in = n_pt;
  b->no_in = n_obs;
  b->dim = 6 * n_cam + 1;
  b->ld = (b->dim + CB - 1) / CB * CB;
  b->ssz = (size_t)b->ld * b->ld;
  // ---- group observations by point, ascending camera inside a point (std::map order of
  //      Point3D::idxImage, reference src/BundleAdjustment.cpp:87)
  std::vector<int> cnt(n_pt + 1, 0);
  for (int o = 0; o < n_obs; ++o) cnt[obs_pt[o] + 1]++;
  for (int p = 0; p < n_pt; ++p) cnt[p + 1] += cnt[p];
  std::vector<int> slot(n_obs), fill(n_pt, 0);
  for (int o = 0; o < n_obs; ++o) slot[cnt[obs_pt[o]] + fill[obs_pt[o]]++] = o;
  for (int p = 0; p < n_pt; ++p) {  // stable insertion sort: a point has a handful of observations
    for (int i = cnt[p] + 1; i < cnt[p + 1]; ++i) {
      const int v = slot[i], cv = obs_cam[v];
      int j = i - 1;
      for (; j >= cnt[p] && obs_cam[slot[j]] > cv; --j) slot[j + 1] = slot[j];
      slot[j + 1] = v;
    }
  }
  // the point's ascending camera list, flat, and a hash of it: the signature sort below compares
  // (length, hash) first and walks the lists only on equal hashes
  std::vector<int> scam(n_obs);
  std::vector<uint64_t> sig_hash(n_pt, 0);
  for (int p = 0; p < n_pt; ++p) {
    uint64_t h = 1469598103934665603ull;
    for (int k = cnt[p]; k < cnt[p + 1]; ++k) {
      scam[k] = obs_cam[slot[k]];
      h = (h ^ (uint64_t)(uint32_t)scam[k]) * 1099511628211ull;
    }
    sig_hash[p] = h;
  }
  lap_("group by point");
  // ---- group the points that have observations by signature (their ascending camera list): a
  //      hash table assigns run ids in order of first appearance, a counting sort makes the runs
  //      contiguous (stable: ascending point index inside a run)
  auto sig_equal = [&](int x, int y) {  // x, y: input point indices
    const int nx = cnt[x + 1] - cnt[x];
    if (nx != cnt[y + 1] - cnt[y] || sig_hash[x] != sig_hash[y]) return false;
    for (int k = 0; k < nx; ++k)
      if (scam[cnt[x] + k] != scam[cnt[y] + k]) return false;
    return true;
  };
  std::vector<int> run_of(n_pt, -1), run_rep, run_cnt;
  {
    size_t cap = 64;
    while (cap < 2 * (size_t)n_pt + 16) cap <<= 1;
    std::vector<int> table(cap, -1);  // open addressing: run id, keyed by the signature hash
    for (int p = 0; p < n_pt; ++p) {
      const int n = cnt[p + 1] - cnt[p];
      if (n > FB_MAXN) {
        delete b;
        return SFMHIP_ERR_UNSUPPORTED;  // more than FB_MAXN observations of one point
      }
      if (n == 0) continue;
      size_t slot_i = (size_t)(sig_hash[p] ^ (sig_hash[p] >> 29)) & (cap - 1);
      for (;; slot_i = (slot_i + 1) & (cap - 1)) {
        const int r = table[slot_i];
        if (r < 0) {
          table[slot_i] = (int)run_rep.size();
          run_of[p] = (int)run_rep.size();
          run_rep.push_back(p);
          run_cnt.push_back(1);
          break;
        }
        if (sig_equal(run_rep[r], p)) {
          run_of[p] = r;
          ++run_cnt[r];
          break;
        }
      }
    }
  }
  std::vector<int> run_start(run_rep.size() + 1, 0);
  for (size_t r = 0; r < run_rep.size(); ++r) run_start[r + 1] = run_start[r] + run_cnt[r];
  std::vector<int> order(run_start.back());
  {
    std::vector<int> fillr(run_start.begin(), run_start.end() - 1);
    for (int p = 0; p < n_pt; ++p)
      if (run_of[p] >= 0) order[fillr[run_of[p]]++] = p;
  }
  b->np = (int)order.size();
  b->perm = order;
  std::vector<int> optr(b->np + 1, 0);
  for (int sp = 0; sp < b->np; ++sp) optr[sp + 1] = optr[sp] + (cnt[order[sp] + 1] - cnt[order[sp]]);
  b->no = optr[b->np];
  std::vector<int> ocam(b->no);
  std::vector<double> oxy(2 * (size_t)b->no);
  b->h_cam_used.assign(n_cam, 0);
  // the gather of a million observations is a cache-miss chain on one core: split it over a few
  host_parallel_for(b->np, [&](int lo, int hi) {
    for (int sp = lo; sp < hi; ++sp) {
      const int p = order[sp];
      int w = optr[sp];
      for (int k = cnt[p]; k < cnt[p + 1]; ++k, ++w) {
        const int o = slot[k];
        ocam[w] = obs_cam[o];
        oxy[2 * (size_t)w] = obs_xy[2 * (size_t)o];
        oxy[2 * (size_t)w + 1] = obs_xy[2 * (size_t)o + 1];
      }
    }
  });
  for (int k = 0; k < b->no; ++k) b->h_cam_used[ocam[k]] = 1;
  lap_("signature sort + csr");
  // ---- chunks: runs of equal signature with strictly ascending cameras, n <= 10 -> MFMA path,
  //      classed by the width of the local Gram matrix: NB = ceil((6n+2)/16) column blocks
  std::vector<Chunk> chunks;
  std::vector<int> ids[8], sig_cams, fb;
  constexpr int SHORT_RUN = 12;  // runs of at most this many points go to one-wave workgroups
  // points per workgroup: 2 workgroups of 4 waves are resident per CU (register-bound), so the
  // launch runs in rounds of 512 workgroups; pick the run length that minimises
  // rounds x (run length + fixed per-workgroup cost, ~40 points' worth of prologue + scatter)
  int target = 64;
  const std::vector<int>& gstart = run_start;  // first sorted point of every run, + np
  {
    std::vector<int> gsz;
    for (size_t gi = 0; gi + 1 < gstart.size(); ++gi)
      if (gstart[gi + 1] - gstart[gi] > SHORT_RUN) gsz.push_back(gstart[gi + 1] - gstart[gi]);
    double best = 1e300;
    for (int t = 32; t <= 512; t += 4) {
      long long w = 0;
      for (int g : gsz) w += (g + t - 1) / t;
      const double cost = (double)((w + 511) / 512) * (t + 40);
      if (cost < best) {
        best = cost;
        target = t;
      }
    }
  }
  for (size_t gi = 0; gi + 1 < gstart.size(); ++gi) {
    const int sp = gstart[gi], e = gstart[gi + 1];
    const int n = optr[sp + 1] - optr[sp];
    bool strict = true;
    for (int k = 1; k < n; ++k) strict = strict && ocam[optr[sp] + k - 1] < ocam[optr[sp] + k];
    if (n <= 10 && strict) {
      const int so = (int)sig_cams.size();
      for (int k = 0; k < n; ++k) sig_cams.push_back(ocam[optr[sp] + k]);
      const int nb = (6 * n + 2 + 15) / 16;
      if (e - sp <= SHORT_RUN) {
        ids[4 + nb - 1].push_back((int)chunks.size());
        chunks.push_back(Chunk{so, n, sp, e - sp});
      } else {
        const int parts = (e - sp + target - 1) / target;
        for (int q = 0; q < parts; ++q) {
          const int lo = sp + (int)((long long)(e - sp) * q / parts), hi = sp + (int)((long long)(e - sp) * (q + 1) / parts);
          ids[nb - 1].push_back((int)chunks.size());
          chunks.push_back(Chunk{so, n, lo, hi - lo});
        }
      }
    } else {
      for (int q = sp; q < e; ++q) fb.push_back(q);
    }
  }
  lap_("chunks");
  // ---- camera-major copy of the observations (sorted point index, xy) for ba_cam_blocks
  std::vector<int> cptr(n_cam + 1, 0), cpt(b->no);
  std::vector<double> cxy(2 * (size_t)b->no);
  {
    for (int k = 0; k < b->no; ++k) cptr[ocam[k] + 1]++;
    for (int c = 0; c < n_cam; ++c) cptr[c + 1] += cptr[c];
    std::vector<int> fillc(cptr.begin(), cptr.end() - 1);
    for (int sp = 0; sp < b->np; ++sp)
      for (int k = optr[sp]; k < optr[sp + 1]; ++k) {
        const int dst = fillc[ocam[k]]++;
        cpt[dst] = sp;
        cxy[2 * (size_t)dst] = oxy[2 * (size_t)k];
        cxy[2 * (size_t)dst + 1] = oxy[2 * (size_t)k + 1];
      }
    b->cam_split = std::max(1, std::min(64, 1024 / std::max(n_cam, 1)));
  }
  lap_("camera-major copy");
  // ---- device storage
  BaDev& d = b->d;
  d.nc = n_cam;
  d.np = b->np;
  d.no = b->no;
  d.dim = b->dim;
  d.ld = b->ld;
  int rc = SFMHIP_OK;
  int *d_optr = nullptr, *d_ocam = nullptr;
  double2* d_oxy = nullptr;
  b->red_count = b->ssz + 3 * (size_t)b->ld + SC + 64;
#define BA_A(ptr, n)                         \
  if (rc == SFMHIP_OK) rc = ba_alloc(b, &(ptr), (size_t)(n))
  BA_A(d_optr, b->np + 1);
  BA_A(d_ocam, b->no);
  BA_A(d_oxy, b->no);
  BA_A(d.cams, 6 * n_cam);
  BA_A(d.pts, 3 * (size_t)b->np);
  BA_A(d.focal, 1);
  BA_A(d.camd, (size_t)CAMD * n_cam);
  BA_A(d.cams_c, 6 * n_cam);
  BA_A(d.pts_c, 3 * (size_t)b->np);
  BA_A(d.focal_c, 1);
  BA_A(d.camd_c, (size_t)CAMD * n_cam);
  BA_A(d.scale_c, 6 * n_cam);
  BA_A(d.scale_p, 3 * (size_t)b->np);
  BA_A(d.scale_f, 1);
  BA_A(d.diag, b->ld);
  BA_A(d.red, b->red_count);
  BA_A(d.z, b->ld);
  BA_A(d.dinv, b->ld);
  BA_A(d.linv, (size_t)b->ld * CB);
  BA_A(d.red2, 16);
  BA_A(d.info, 1);
  BA_A(b->d_cam_used, n_cam);
  BA_A(b->d_chunks, chunks.size());
  for (int c = 0; c < 8; ++c) BA_A(b->d_chunk_ids[c], ids[c].size());
  BA_A(b->d_sig_cams, sig_cams.size());
  BA_A(b->d_cptr, cptr.size());
  BA_A(b->d_cpt, cpt.size());
  BA_A(b->d_cxy, cpt.size());
  BA_A(b->d_fb_points, fb.size());
#undef BA_A
  if (rc != SFMHIP_OK) {
    sfmhip_ba_destroy(b);
    return rc;
  }
  d.optr = d_optr;
  d.ocam = d_ocam;
  d.oxy = d_oxy;
  for (int c = 0; c < 8; ++c) b->n_chunk_ids[c] = (int)ids[c].size();
  b->n_fb = (int)fb.size();
  auto up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  lap_("hipMalloc x27");
  SFM_HIP_TRY(up(d_optr, optr.data(), optr.size() * 4));
  SFM_HIP_TRY(up(d_ocam, ocam.data(), ocam.size() * 4));
  SFM_HIP_TRY(up(d_oxy, oxy.data(), oxy.size() * 8));
  SFM_HIP_TRY(up(b->d_cam_used, b->h_cam_used.data(), n_cam));
  SFM_HIP_TRY(up(b->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk)));
  for (int c = 0; c < 8; ++c) SFM_HIP_TRY(up(b->d_chunk_ids[c], ids[c].data(), ids[c].size() * 4));
  SFM_HIP_TRY(up(b->d_sig_cams, sig_cams.data(), sig_cams.size() * 4));
  SFM_HIP_TRY(up(b->d_cptr, cptr.data(), cptr.size() * 4));
  SFM_HIP_TRY(up(b->d_cpt, cpt.data(), cpt.size() * 4));
  SFM_HIP_TRY(up(b->d_cxy, cxy.data(), cxy.size() * 8));
  SFM_HIP_TRY(up(b->d_fb_points, fb.data(), fb.size() * 4));
  lap_("uploads");
  SFM_HIP_TRY(hipHostMalloc((void**)&b->h_sc, sizeof(double) * (SC + 64 + 16), hipHostMallocDefault));
  for (auto& e : b->ev) SFM_HIP_TRY(hipEventCreate(&e));
  b->h_pts_in.assign(3 * (size_t)n_pt, 0.0);
  lap_("pinned + events");
  *out = b;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_set_allreduce(sfmhip_ba* b, sfmhip_allreduce_fn fn, void* user, int rank, int world) {
  if (!b || world < 1 || rank < 0 || rank >= world || world > 64) return SFMHIP_ERR_ARG;
  if (world > 1 && !fn) return SFMHIP_ERR_ARG;
  b->allreduce = fn;
  b->allreduce_user = user;
  b->rank = rank;
  b->world = world;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_set_params(sfmhip_ba* b, const double* cams6, const double* pts3, double focal) {
  if (!b || !cams6 || (!pts3 && b->np_in)) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  hipStream_t st = b->ctx->stream;
  if (b->np_in) memcpy(b->h_pts_in.data(), pts3, sizeof(double) * 3 * (size_t)b->np_in);
  std::vector<double> sorted(3 * (size_t)std::max(b->np, 1));
  for (int sp = 0; sp < b->np; ++sp)
    for (int j = 0; j < 3; ++j) sorted[3 * (size_t)sp + j] = pts3[3 * (size_t)b->perm[sp] + j];
  SFM_HIP_TRY(hipMemcpyAsync(b->d.cams, cams6, sizeof(double) * 6 * b->nc, hipMemcpyHostToDevice, st));
  if (b->np) SFM_HIP_TRY(hipMemcpyAsync(b->d.pts, sorted.data(), sizeof(double) * 3 * (size_t)b->np, hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemcpyAsync(b->d.focal, &focal, sizeof(double), hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  b->scale_ready = false;
  b->lm.started = false;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_get_params(sfmhip_ba* b, double* cams6, double* pts3, double* focal) {
  if (!b) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  hipStream_t st = b->ctx->stream;
  std::vector<double> sorted(3 * (size_t)std::max(b->np, 1));
  if (cams6) SFM_HIP_TRY(hipMemcpyAsync(cams6, b->d.cams, sizeof(double) * 6 * b->nc, hipMemcpyDeviceToHost, st));
  if (pts3 && b->np) SFM_HIP_TRY(hipMemcpyAsync(sorted.data(), b->d.pts, sizeof(double) * 3 * (size_t)b->np, hipMemcpyDeviceToHost, st));
  if (focal) SFM_HIP_TRY(hipMemcpyAsync(focal, b->d.focal, sizeof(double), hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (pts3) {
    memcpy(pts3, b->h_pts_in.data(), sizeof(double) * 3 * (size_t)b->np_in);  // points without observations
    for (int sp = 0; sp < b->np; ++sp)
      for (int j = 0; j < 3; ++j) pts3[3 * (size_t)b->perm[sp] + j] = sorted[3 * (size_t)sp + j];
  }
  return SFMHIP_OK;
}

// -------- building blocks of one LM iteration (all asynchronous on the context stream)
static int ba_allreduce(sfmhip_ba* b, double* buf, size_t count) {
  if (b->world <= 1) return SFMHIP_OK;
  const int rc = b->allreduce(buf, count, b->allreduce_user);
  return rc == 0 ? SFMHIP_OK : SFMHIP_ERR_COMM;
}

static int ba_prepare_scale(sfmhip_ba* b, int jacobi) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  const size_t tail = b->ssz + 2 * (size_t)b->ld;  // dc | sc
  SFM_HIP_TRY(hipMemsetAsync(d.red + tail, 0, sizeof(double) * ((size_t)b->ld + SC + 64), st));
  hipLaunchKernelGGL(ba_cam_prep, dim3((b->nc + 63) / 64), dim3(64), 0, st, d.cams, d.camd, b->nc, 1);
  if (b->np) hipLaunchKernelGGL(ba_point_norms, dim3((b->np + 255) / 256), dim3(256), 0, st, d, jacobi);
  if (b->no && jacobi)
    hipLaunchKernelGGL(ba_cam_blocks, dim3(b->nc * b->cam_split), dim3(256), 0, st, d, b->d_cptr, b->d_cpt, b->d_cxy,
                       b->cam_split, 1);
  SFM_HIP_TRY(hipGetLastError());
  SFM_TRY(ba_allreduce(b, d.red + tail, (size_t)b->ld + SC));
  hipLaunchKernelGGL(ba_make_scale, dim3((b->dim + 255) / 256), dim3(256), 0, st, d, jacobi);
  SFM_HIP_TRY(hipGetLastError());
  // cameras observed by any rank: nonzero translation-column norm
  std::vector<double> dc(b->ld + SC);
  SFM_HIP_TRY(hipMemcpyAsync(dc.data(), d.red + tail, sizeof(double) * (b->ld + SC), hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (jacobi || b->world > 1) {
    if (jacobi)
      for (int c = 0; c < b->nc; ++c) b->h_cam_used[c] = dc[6 * c + 3] > 0 ? 1 : 0;
    SFM_HIP_TRY(hipMemcpyAsync(b->d_cam_used, b->h_cam_used.data(), b->nc, hipMemcpyHostToDevice, st));
  }
  const double pts_n2 = dc[b->ld + 1];
  double* tmp = d.red2 + 8;
  hipLaunchKernelGGL(ba_cam_norm, dim3(1), dim3(256), 0, st, d, b->d_cam_used, tmp);
  double cam_n2 = 0;
  SFM_HIP_TRY(hipMemcpyAsync(&cam_n2, tmp, sizeof(double), hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  b->x_norm = std::sqrt(pts_n2 + cam_n2);
  b->scale_ready = true;
  return SFMHIP_OK;
}

// linearise at the current x + eliminate with `radius`; leaves [S|g|gF|dc|sc] summed over ranks
static int ba_linearize_eliminate(sfmhip_ba* b, double radius, const sfmhip_ba_opts* o, bool add_diag) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  SFM_HIP_TRY(hipEventRecord(b->ev[0], st));
  SFM_HIP_TRY(hipMemsetAsync(d.red, 0, sizeof(double) * b->red_count, st));
  hipLaunchKernelGGL(ba_cam_prep, dim3((b->nc + 63) / 64), dim3(64), 0, st, d.cams, d.camd, b->nc, 1);
  const double inv_radius = 1.0 / radius;
  int nl = 0;
  if (b->no) {
    hipLaunchKernelGGL(ba_cam_blocks, dim3(b->nc * b->cam_split), dim3(256), 0, st, d, b->d_cptr, b->d_cpt, b->d_cxy,
                       b->cam_split, 0);
    ++nl;
  }
#define BA_ELIM(NB)                                                                                                   \
  for (int cls = 0; cls < 2; ++cls) {                                                                                 \
    const int li = 4 * cls + NB - 1, nthreads = cls ? 64 : 256;                                                       \
    if (!b->n_chunk_ids[li]) continue;                                                                                \
    hipLaunchKernelGGL((ba_eliminate_mfma<NB>), dim3(b->n_chunk_ids[li]), dim3(nthreads),                             \
                       sizeof(double) * (nthreads / 64) * 12 * MP, st, d, b->d_chunks, b->d_chunk_ids[li],            \
                       b->d_sig_cams, inv_radius, o->min_lm_diagonal, o->max_lm_diagonal, b->rank);                   \
    ++nl;                                                                                                             \
  }
  BA_ELIM(1)
  BA_ELIM(2)
  BA_ELIM(3)
  BA_ELIM(4)
#undef BA_ELIM
  if (b->n_fb)
    hipLaunchKernelGGL(ba_eliminate_generic, dim3(b->n_fb), dim3(64), 0, st, d, b->d_fb_points, radius,
                       o->min_lm_diagonal, o->max_lm_diagonal, b->rank);
  SFM_HIP_TRY(hipGetLastError());
  b->launches += 2 + nl + (b->n_fb > 0);
  SFM_HIP_TRY(hipEventRecord(b->ev[1], st));
  SFM_TRY(ba_allreduce(b, d.red, b->ssz + 3 * (size_t)b->ld + SC + b->world));
  SFM_HIP_TRY(hipEventRecord(b->ev[2], st));
  hipLaunchKernelGGL(ba_finalize, dim3(1), dim3(1024), 0, st, d, radius, o->min_lm_diagonal, o->max_lm_diagonal,
                     b->world, add_diag ? 1 : 0);
  SFM_HIP_TRY(hipGetLastError());
  b->launches += 1;
  return SFMHIP_OK;
}

static int ba_reduced_solve(sfmhip_ba* b) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  double* A = d.red;
  double* y = d.red + b->ssz;  // g becomes y = L^-1 g
  SFM_HIP_TRY(hipMemsetAsync(d.info, 0, sizeof(int), st));
  const int nt = b->ld / CB;
  for (int k = 0; k < nt; ++k) {
    const int m = nt - k;
    const int nblk = k == 0 ? m + 1 : m * (m + 1) / 2 + m;  // launch 0 has no pending update
    hipLaunchKernelGGL(chol_step, dim3(nblk), dim3(128), 0, st, A, y, d.dinv, d.linv, d.ld, nt, k, d.info);
  }
  int nbs = 0;
  constexpr int kBsLds = 2 * GB * CB * CB * (int)sizeof(double);  // 128 KiB of dynamic LDS
  static bool bs_attr = false;
  if (!bs_attr) {
    SFM_HIP_TRY(hipFuncSetAttribute((const void*)chol_backsolve_group, hipFuncAttributeMaxDynamicSharedMemorySize, kBsLds));
    bs_attr = true;
  }
  for (int hi = nt; hi > 0; hi -= GB) {
    const int lo = std::max(0, hi - GB);
    hipLaunchKernelGGL(chol_backsolve_group, dim3(1), dim3(512), kBsLds, st, A, y, d.linv, d.z, d.ld, lo, hi);
    ++nbs;
    if (lo > 0) {
      hipLaunchKernelGGL(chol_backsolve_gemv, dim3((lo * CB + 3) / 4), dim3(256), 0, st, A, y, d.z, d.ld, lo * CB, hi * CB);
      ++nbs;
    }
  }
  SFM_HIP_TRY(hipGetLastError());
  b->launches += nt + nbs;
  return SFMHIP_OK;
}

static int ba_step_eval(sfmhip_ba* b, double radius, const sfmhip_ba_opts* o) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  SFM_HIP_TRY(hipMemsetAsync(d.red2, 0, sizeof(double) * 8, st));
  hipLaunchKernelGGL(ba_cand_cams, dim3((b->nc + 1 + 63) / 64), dim3(64), 0, st, d, b->d_cam_used, b->rank);
  if (b->np)
    hipLaunchKernelGGL(ba_backsub, dim3((b->np + 255) / 256), dim3(256), 0, st, d, radius, o->min_lm_diagonal,
                       o->max_lm_diagonal);
  SFM_HIP_TRY(hipGetLastError());
  b->launches += 2;
  SFM_TRY(ba_allreduce(b, d.red2, 8));
  return SFMHIP_OK;
}

static void ba_swap_candidate(sfmhip_ba* b) {
  std::swap(b->d.cams, b->d.cams_c);
  std::swap(b->d.pts, b->d.pts_c);
  std::swap(b->d.focal, b->d.focal_c);
  std::swap(b->d.camd, b->d.camd_c);
}

struct IterScalars {
  double cost, nfail, gmax;           // from the eliminate pass (at x)
  double cost_c, mcc, step_n2, cand_n2;  // from the step evaluation
  int info;
};

static int ba_read_scalars(sfmhip_ba* b, IterScalars* s, bool with_step) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  const size_t sc_off = b->ssz + 3 * (size_t)b->ld;
  SFM_HIP_TRY(hipMemcpyAsync(b->h_sc, d.red + sc_off, sizeof(double) * SC, hipMemcpyDeviceToHost, st));
  if (with_step) {
    SFM_HIP_TRY(hipMemcpyAsync(b->h_sc + SC, d.red2, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(b->h_sc + SC + 8, d.info, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  SFM_HIP_TRY(hipStreamSynchronize(st));
  s->cost = 0.5 * b->h_sc[0];
  s->nfail = b->h_sc[2];
  s->gmax = b->h_sc[3];
  if (with_step) {
    s->cost_c = 0.5 * b->h_sc[SC + 0];
    s->mcc = -b->h_sc[SC + 1];
    s->step_n2 = b->h_sc[SC + 2];
    s->cand_n2 = b->h_sc[SC + 3];
    memcpy(&s->info, b->h_sc + SC + 8, sizeof(int));
  }
  return SFMHIP_OK;
}

static void ba_acc_timing(sfmhip_ba* b) {
  float ms;
  if (hipEventElapsedTime(&ms, b->ev[0], b->ev[1]) == hipSuccess) b->t_acc[0] += ms * 1e-3;
  if (hipEventElapsedTime(&ms, b->ev[1], b->ev[2]) == hipSuccess) b->t_acc[1] += ms * 1e-3;
  if (hipEventElapsedTime(&ms, b->ev[2], b->ev[3]) == hipSuccess) b->t_acc[2] += ms * 1e-3;
  if (hipEventElapsedTime(&ms, b->ev[3], b->ev[4]) == hipSuccess) b->t_acc[3] += ms * 1e-3;
}

// TrustRegionMinimizer::Minimize (Ceres 1.13) + LevenbergMarquardtStrategy, host control loop.
// The loop state lives in the sfmhip_ba object so that sfmhip_ba_iterate can be called one
// iteration at a time (bench.py interleaves it with matching sweeps).
static int ba_begin(sfmhip_ba* b, const sfmhip_ba_opts* o) {
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  for (double& t : b->t_acc) t = 0;
  b->launches = 0;
  SFM_TRY(ba_prepare_scale(b, o->jacobi_scaling));
  LmState& s = b->lm;
  s = LmState();
  s.radius = o->initial_radius;
  s.x_norm = b->x_norm;
  // iteration 0: cost + gradient at x0 (the same pass also forms the first reduced system)
  IterScalars sc{};
  SFM_TRY(ba_linearize_eliminate(b, s.radius, o, true));
  SFM_TRY(ba_read_scalars(b, &sc, false));
  s.cost = s.initial_cost = sc.cost;
  s.gmax = sc.gmax;
  s.have_lin = true;
  s.started = true;
  return SFMHIP_OK;
}

// one LM iteration; *stop receives the termination type once a stopping rule fires (-1 else)
static int ba_one_iteration(sfmhip_ba* b, const sfmhip_ba_opts* o, bool timing_only, int* stop) {
  hipStream_t st = b->ctx->stream;
  LmState& s = b->lm;
  IterScalars sc{};
  *stop = -1;
  ++s.iter;
  if (!s.have_lin) SFM_TRY(ba_linearize_eliminate(b, s.radius, o, true));
  s.have_lin = false;
  SFM_TRY(ba_reduced_solve(b));
  SFM_HIP_TRY(hipEventRecord(b->ev[3], st));
  SFM_TRY(ba_step_eval(b, s.radius, o));
  SFM_HIP_TRY(hipEventRecord(b->ev[4], st));
  SFM_TRY(ba_read_scalars(b, &sc, true));
  ba_acc_timing(b);
  if (s.lin_unread) {
    // the linearisation enqueued after the last accepted step is read together with this step's
    // scalars (one host synchronisation per iteration); Ceres tests the gradient tolerance right
    // after accepting a step, so a converged gradient discards the step evaluated above
    s.lin_unread = false;
    s.cost = sc.cost;
    s.gmax = sc.gmax;
    if (!timing_only && s.gmax <= o->gradient_tolerance) {
      --s.iter;
      *stop = SFMHIP_BA_CONVERGENCE;
      return SFMHIP_OK;
    }
  }
  const bool finite = std::isfinite(sc.step_n2) && std::isfinite(sc.mcc) && std::isfinite(sc.cost_c);
  const bool bad = sc.info != 0 || sc.nfail > 0 || !finite;
  if (bad || !(sc.mcc > 0.0)) {  // HandleInvalidStep
    if (++s.invalid >= o->max_consecutive_invalid && !timing_only) {
      *stop = SFMHIP_BA_FAILURE;
      return SFMHIP_OK;
    }
    s.radius /= s.decrease_factor;
    s.decrease_factor *= 2.0;
    if (o->verbose) fprintf(stderr, "[sfmhip-ba] it %d invalid step (info %d), radius %.3e\n", s.iter, sc.info, s.radius);
    return SFMHIP_OK;
  }
  s.invalid = 0;
  const double step_norm = std::sqrt(sc.step_n2);
  if (!timing_only) {
    if (step_norm <= o->parameter_tolerance * (s.x_norm + o->parameter_tolerance)) {
      *stop = SFMHIP_BA_CONVERGENCE;  // ParameterToleranceReached: candidate not taken
      return SFMHIP_OK;
    }
    if (std::fabs(s.cost - sc.cost_c) <= o->function_tolerance * s.cost) {
      *stop = SFMHIP_BA_CONVERGENCE;  // FunctionToleranceReached: candidate not taken
      return SFMHIP_OK;
    }
  }
  const double rho = (s.cost - sc.cost_c) / sc.mcc;
  if (o->verbose)
    fprintf(stderr, "[sfmhip-ba] it %d cost %.9e -> %.9e rho %.3e radius %.3e |step| %.3e\n", s.iter, s.cost, sc.cost_c,
            rho, s.radius, step_norm);
  if (rho > o->min_relative_decrease) {  // HandleSuccessfulStep
    ba_swap_candidate(b);
    s.x_norm = std::sqrt(sc.cand_n2);
    ++s.nsucc;
    const double q = 2.0 * rho - 1.0;
    s.radius = s.radius / std::fmax(1.0 / 3.0, 1.0 - q * q * q);
    s.radius = std::fmin(o->max_radius, s.radius);
    s.decrease_factor = 2.0;
    // re-linearise at the new x; with the new radius this is also the next reduced system.
    // Enqueued only: its cost / gradient come back with the next iteration's scalars.
    SFM_TRY(ba_linearize_eliminate(b, s.radius, o, true));
    s.cost = sc.cost_c;  // provisional (same residuals, other summation order)
    s.have_lin = true;
    s.lin_unread = true;
  } else {  // HandleUnsuccessfulStep
    s.radius /= s.decrease_factor;
    s.decrease_factor *= 2.0;
  }
  return SFMHIP_OK;
}

// cost / gradient of a linearisation that is still only enqueued
static int ba_flush_lin(sfmhip_ba* b) {
  LmState& s = b->lm;
  if (!s.lin_unread) return SFMHIP_OK;
  IterScalars sc{};
  SFM_TRY(ba_read_scalars(b, &sc, false));
  s.cost = sc.cost;
  s.gmax = sc.gmax;
  s.lin_unread = false;
  return SFMHIP_OK;
}

static void ba_fill_summary(sfmhip_ba* b, int term, double time_s, sfmhip_ba_summary* sum) {
  const LmState& s = b->lm;
  b->x_norm = s.x_norm;
  sum->termination = term;
  sum->iterations = s.iter;
  sum->successful_steps = s.nsucc;
  sum->initial_cost = s.initial_cost;
  sum->final_cost = s.cost;
  sum->final_radius = s.radius;
  sum->gradient_max_norm = s.gmax;
  sum->time_s = time_s;
}

extern "C" int sfmhip_ba_run(sfmhip_ba* b, const sfmhip_ba_opts* opts, sfmhip_ba_summary* summary) {
  if (!b) return SFMHIP_ERR_ARG;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(clk::now() - t0).count(); };
  sfmhip_ba_opts od;
  if (!opts) {
    sfmhip_ba_default_opts(&od);
    opts = &od;
  }
  sfmhip_ba_summary sm;
  memset(&sm, 0, sizeof sm);
  SFM_TRY(ba_begin(b, opts));
  LmState& s = b->lm;
  int term = SFMHIP_BA_NO_CONVERGENCE;
  if (s.gmax <= opts->gradient_tolerance) {
    term = SFMHIP_BA_CONVERGENCE;
  } else {
    for (;;) {
      if (s.iter >= opts->max_iterations) break;                            // NO_CONVERGENCE
      if (opts->max_time_s > 0 && elapsed() >= opts->max_time_s) break;     // NO_CONVERGENCE
      if (s.radius < opts->min_radius) {
        term = SFMHIP_BA_CONVERGENCE;
        break;
      }
      int stop = -1;
      SFM_TRY(ba_one_iteration(b, opts, false, &stop));
      if (stop >= 0) {
        term = stop;
        break;
      }
    }
  }
  SFM_TRY(ba_flush_lin(b));
  if (term == SFMHIP_BA_NO_CONVERGENCE && s.gmax <= opts->gradient_tolerance) term = SFMHIP_BA_CONVERGENCE;
  s.started = false;  // a finished solve is not resumable
  ba_fill_summary(b, term, elapsed(), &sm);
  if (summary) *summary = sm;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_iterate(sfmhip_ba* b, int iters, sfmhip_ba_summary* summary) {
  if (!b || iters < 0) return SFMHIP_ERR_ARG;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  sfmhip_ba_opts o;
  sfmhip_ba_default_opts(&o);
  sfmhip_ba_summary sm;
  memset(&sm, 0, sizeof sm);
  if (!b->lm.started) SFM_TRY(ba_begin(b, &o));
  for (int i = 0; i < iters; ++i) {
    int stop = -1;
    SFM_TRY(ba_one_iteration(b, &o, true, &stop));
  }
  SFM_TRY(ba_flush_lin(b));
  ba_fill_summary(b, SFMHIP_BA_NO_CONVERGENCE, std::chrono::duration<double>(clk::now() - t0).count(), &sm);
  if (summary) *summary = sm;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_reduced_system(sfmhip_ba* b, double radius, double* S, double* g, double* cost) {
  if (!b || !(radius > 0)) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  sfmhip_ba_opts o;
  sfmhip_ba_default_opts(&o);
  if (!b->scale_ready) SFM_TRY(ba_prepare_scale(b, o.jacobi_scaling));
  // this rank's points only: no all-reduce, and the camera/focal LM diagonal (a global
  // quantity) is added only when there is a single rank
  const int world = b->world;
  b->world = 1;
  int rc = ba_linearize_eliminate(b, radius, &o, world == 1);
  b->world = world;
  SFM_TRY(rc);
  hipStream_t st = b->ctx->stream;
  const int n = b->dim;
  if (S) {
    SFM_HIP_TRY(hipMemcpy2DAsync(S, sizeof(double) * n, b->d.red, sizeof(double) * b->ld, sizeof(double) * n, n,
                                 hipMemcpyDeviceToHost, st));
  }
  if (g) SFM_HIP_TRY(hipMemcpyAsync(g, b->d.red + b->ssz, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  double sc0 = 0;
  SFM_HIP_TRY(hipMemcpyAsync(&sc0, b->d.red + b->ssz + 3 * (size_t)b->ld, sizeof(double), hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (S)
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) S[(size_t)j * n + i] = S[(size_t)i * n + j];
  if (cost) *cost = 0.5 * sc0;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_last_timing(sfmhip_ba* b, double seconds[4], int* launches) {
  if (!b || !seconds) return SFMHIP_ERR_ARG;
  for (int i = 0; i < 4; ++i) seconds[i] = b->t_acc[i];
  if (launches) *launches = b->launches;
  return SFMHIP_OK;
}

extern "C" void sfmhip_ba_destroy(sfmhip_ba* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  for (void* p : b->allocs) hipFree(p);
  if (b->h_sc) hipHostFree(b->h_sc);
  for (auto& e : b->ev)
    if (e) hipEventDestroy(e);
  delete b;
}

extern "C" int sfmhip_ba_solve(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, double* cams6, double* pts3,
                               double* focal, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                               const sfmhip_ba_opts* opts, sfmhip_ba_summary* summary) {
  if (!ctx || !cams6 || !focal) return SFMHIP_ERR_ARG;
  sfmhip_ba* b = nullptr;
  int rc = sfmhip_ba_create(ctx, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_xy, &b);
  if (rc == SFMHIP_OK) rc = sfmhip_ba_set_params(b, cams6, pts3, *focal);
  if (rc == SFMHIP_OK) rc = sfmhip_ba_run(b, opts, summary);
  if (rc == SFMHIP_OK) rc = sfmhip_ba_get_params(b, cams6, pts3, focal);
  sfmhip_ba_destroy(b);
  return rc;
}
