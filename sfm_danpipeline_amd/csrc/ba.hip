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
//   ba_eliminate_mfma<NB>  ONE linearisation forms all of S for the points in long runs.  A workgroup owns a run of
//                     points of one signature; 16 lanes linearise the (<=10) observations of a point (4 points per
//                     wave and iteration), DPP-reduce the 3x3 point block, publish T_o = (Jc^T Jp) C^-1/2 as 3 rows
//                     of a 12 x 64 LDS panel, and the wave adds the panel's Gram matrix on v_mfma_f64_16x16x4_f64;
//                     F^T F accumulates in LDS next to it; one f64 atomic per entry of the block and workgroup.
//   ba_pp_points / ba_pp_pairs / ba_cam_blocks   the pair path for every other point (short runs, unsorted or
//                     repeated cameras, long tracks): T rows stored per observation, S blocks summed per camera
//                     pair and per camera.
//   ba_finalize       LM diagonal (clamped squared column norms / radius) onto S, gradient max.
//   chol_step2 / chol_step2_chains   blocked Cholesky of S (two 32-column panels per launch) with the rhs carried
//                     as an extra row (y = L^-1 g) and the identity as extra tile rows (X = L^-T): dense, or on
//                     the independent chains + separator of a dissected camera graph (nd_gather, nd_combine,
//                     nd_xy, nd_w).
//   ba_cand_cams      candidate cameras / focal and their rotation tables.
//   ba_backsub        per point: back-substitution, model cost change, candidate point,
//                     candidate cost.
// Multi-GPU: every rank holds all cameras and its own block of points; [S | g | F^T b |
// diag | scalars] is summed by the caller's all-reduce (RCCL over xGMI) once per iteration,
// every rank then solves the same reduced system and back-substitutes its own points.
#include "common.h"
#include "ba_front_plan.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <functional>
#include <map>
#include <thread>
#include <vector>
#include <float.h>
#include <unistd.h>

namespace {

constexpr int CAMD = 40;     // doubles per camera table: R[9] t[3] dR/dw[27] pad
constexpr int SC = 16;       // scalar slots at the tail of the all-reduce buffer (+ world)
// red2, the step evaluation's scalars at the tail of the reduced-system buffer: [0..4) the four totals (candidate cost, model cost
// change, |step|^2, |candidate|^2) that step_finish leaves -- the sums of the workgroups' slots in BaDev::step_part, added in a
// fixed order; [0, 8) is what several ranks all-reduce -- | (RED2_SLOTS x 4: the slots round 4's atomics landed in, unused since
// round 5, kept for the layout) | RED2_TMP | RED2_INFO (an int: the reduced solve's status) | pad.
constexpr int RED2_SLOTS = 32, RED2_SUM_N = 8 + 4 * RED2_SLOTS, RED2_TMP = RED2_SUM_N, RED2_INFO = RED2_SUM_N + 1,
              RED2_N = RED2_SUM_N + 8;
// red2[RED2_TIMEOUT]: 1 when a bounded spin of this rank's reduced solve or step evaluation ran out; summed over the ranks by the
// step evaluation's all-reduce, so that EVERY rank stops and repeats the solve (a rank-local stop would leave its peers waiting
// in an all-reduce the stopped rank never issues)
constexpr int RED2_TIMEOUT = 4;
// the reduced solve's status word (RED2_INFO): > 0 a pivot was not positive; -1 a hand-off of the reduced solve never arrived
// (the front tree then runs level by level); -2 a slot of the step evaluation's sums never arrived (no fallback: SFMHIP_ERR_TIMEOUT)
constexpr int INFO_FINISHER_TIMEOUT = -2;
constexpr int FB_MAXN = 4096;  // sanity cap on the observations of one point (the pair path has no structural limit)

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
  double* iscale_p; // np*3: 1 / scale_p (the elimination's gradient maximum multiplies by it: three reciprocals per point and iteration less)
  double* scale_f;  // 1
  double* diag;     // dim (clamped)
  // reduced system: red = [S ld*ld | g ld | gF ld | dc ld | sc SC+world], ld = dim rounded up
  // to 32 (row stride of S; the padded diagonal is 1, everything else in the padding 0)
  double* red;
  double* z;     // dim solution
  double* xinv;  // ld x ld, column-major like S's triangle: the identity on entry to the factorisation, L^-T after it
  double* red2;  // 16 scalars of the step evaluation (the slots behind them are no longer used: step_part)
  int* info;     // cholesky failure flag
  // trust-region loop on the device (round 5)
  struct LmDev* lm;   // null: the host decides (radius by kernel argument, the host swaps the parameter sets)
  double* step_part;  // the step evaluation's sums per workgroup, 4 each (candidate cost, model cost change, |step|^2, |candidate|^2),
                      // then the camera parts (cam_parts x 2: |step|^2, |candidate|^2), added up in a FIXED order by the finisher -- the
                      // last workgroup of the step evaluation's last kernel (step_finish): the sums -- hence rho, the radius, the
                      // trajectory -- are the same bits every run
  int step_total;     // workgroups of the step evaluation's kernels
  int cam_parts;
  int decide_here;    // 1: that last workgroup also takes the LM decision (single rank); 0: ba_decide does, after the all-reduce
  int rank;           // of this process (the camera parts of the norms are counted by rank 0 only)
  int step_last;      // this launch is the step evaluation's last kernel (behind a stop its first thread still counts the decision)
  double* lm_host;    // pinned, as the device sees it: the host's copies of the record, a ring of LM_RING
};

// ---------------------------------------------------------------- the trust-region decision (host and device alike)
// TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy of Ceres 1.13 with the options of reference
// src/BundleAdjustment.cpp:115-121: one function, compiled for the host (the loop behind SFMHIP_BA_HOST_LOOP=1, and stage
// timing) and for the device (the default: the last workgroup of the step evaluation calls it, the next linearisation reads
// the radius and which parameter set is x from the record -- the host enqueues iterations ahead and reads records behind
// the GPU).  Contraction is off in it: +, *, /, sqrt of gfx950 are bit-equal to the host's (scripts/ubench/f64_rounding.hip),
// so both loops walk the same trajectory bit for bit (tests/test_gpu_geometry.py::test_device_loop_equals_host_loop).
enum { LM_RUNNING = -1, LM_STOP_TIMEOUT = 100 };
enum { LM_KIND_NONE = 0, LM_KIND_INVALID = 1, LM_KIND_ACCEPTED = 2, LM_KIND_REJECTED = 3, LM_KIND_STOP = 4 };
constexpr int LM_RING = 64;
struct LmDev {
  // options of the run (written by the host when a run starts)
  double gtol, ptol, ftol, min_rel_dec, max_radius, min_radius;
  int max_invalid, max_iter, timing_only, pad0;
  // state
  double radius, dec_factor, cost, gmax, x_norm, initial_cost;
  int iter, nsucc, invalid, lin_unread;
  int parity;    // 1: the "candidate" buffers hold x (kernels swap their view, lm_view); the host puts it back to 0 when it takes over
  int stop;      // LM_RUNNING, SFMHIP_BA_* once a stopping rule has fired (later decisions change nothing), LM_STOP_TIMEOUT
  int timeouts;  // reduced solves that reported a spin that ran out (info < 0): the step is neither taken nor counted
  unsigned seq;  // decisions made on this problem (the host waits for a value of it)
  unsigned stop_seq;  // the decision that set `stop`
  int pad1;
  // the last decision, for the log
  double log_cost0, log_cost_c, log_rho, log_step_norm, log_mcc, log_nfail;
  int log_kind, log_info;
};
struct LmIn {
  double lin_cost, lin_nfail, lin_gmax;      // of the linearisation at x (its cost is 0.5 * sum r^2)
  double cost_c, mcc, step_n2, cand_n2;      // of the step evaluation
  int info;                                  // of the reduced solve: > 0 a pivot was not positive, < 0 a bounded spin ran out
};

__host__ __device__ inline void lm_decide_step(LmDev& s, const LmIn& in) {
#pragma clang fp contract(off)
  s.seq += 1u;
  s.log_kind = LM_KIND_NONE;
  if (s.stop != LM_RUNNING) return;  // (iterations enqueued past a stop: nothing is accepted, nothing changes)
  s.log_info = in.info;
  if (in.info < 0) {  // a scheduling artefact, not a property of the matrix: the host repeats the solve level by level
    s.timeouts += 1;
    s.stop = LM_STOP_TIMEOUT;
    return;
  }
  s.iter += 1;
  const bool t_only = s.timing_only != 0;
  s.log_kind = LM_KIND_STOP;
  if (s.lin_unread) {
    // cost / gradient of the linearisation enqueued behind the last accepted step; Ceres tests the gradient tolerance
    // right after accepting a step, so a converged gradient discards the step evaluated since
    s.lin_unread = 0;
    s.cost = in.lin_cost;
    s.gmax = in.lin_gmax;
    if (!t_only && s.gmax <= s.gtol) {
      s.iter -= 1;
      s.stop = SFMHIP_BA_CONVERGENCE;
      return;
    }
  }
  s.log_cost0 = s.cost, s.log_cost_c = in.cost_c, s.log_mcc = in.mcc, s.log_nfail = in.lin_nfail, s.log_rho = 0.0, s.log_step_norm = 0.0;
  const bool finite = __builtin_isfinite(in.step_n2) && __builtin_isfinite(in.mcc) && __builtin_isfinite(in.cost_c);
  const bool bad = in.info != 0 || in.lin_nfail > 0 || !finite;
  if (bad || !(in.mcc > 0.0)) {  // HandleInvalidStep
    s.invalid += 1;
    if (s.invalid >= s.max_invalid && !t_only) {
      s.stop = SFMHIP_BA_FAILURE;
      return;
    }
    s.radius = s.radius / s.dec_factor;
    s.dec_factor = s.dec_factor * 2.0;
    s.log_kind = LM_KIND_INVALID;
    s.log_step_norm = in.step_n2;
  } else {
    s.invalid = 0;
    const double step_norm = sqrt(in.step_n2);
    s.log_step_norm = step_norm;
    if (!t_only) {
      if (step_norm <= s.ptol * (s.x_norm + s.ptol)) {  // ParameterToleranceReached: candidate not taken
        s.stop = SFMHIP_BA_CONVERGENCE;
        return;
      }
      if (fabs(s.cost - in.cost_c) <= s.ftol * s.cost) {  // FunctionToleranceReached: candidate not taken
        s.stop = SFMHIP_BA_CONVERGENCE;
        return;
      }
    }
    const double rho = (s.cost - in.cost_c) / in.mcc;
    s.log_rho = rho;
    if (rho > s.min_rel_dec) {  // HandleSuccessfulStep
      s.parity ^= 1;
      s.x_norm = sqrt(in.cand_n2);
      s.nsucc += 1;
      const double q = 2.0 * rho - 1.0;
      s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
      s.radius = fmin(s.max_radius, s.radius);
      s.dec_factor = 2.0;
      s.cost = in.cost_c;  // provisional (same residuals, other summation order) until the next linearisation's scalars
      s.lin_unread = 1;
      s.log_kind = LM_KIND_ACCEPTED;
    } else {  // HandleUnsuccessfulStep
      s.radius = s.radius / s.dec_factor;
      s.dec_factor = s.dec_factor * 2.0;
      s.log_kind = LM_KIND_REJECTED;
    }
  }
  // what TrustRegionMinimizer tests before it starts the next iteration (the wall clock is the host's)
  if (!t_only) {
    if (s.iter >= s.max_iter) s.stop = SFMHIP_BA_NO_CONVERGENCE;
    else if (s.radius < s.min_radius) s.stop = SFMHIP_BA_CONVERGENCE;
  }
}

__host__ __device__ inline void lm_decide(LmDev& s, const LmIn& in) {
  const int was = s.stop;
  lm_decide_step(s, in);
  if (was == LM_RUNNING && s.stop != LM_RUNNING) s.stop_seq = s.seq;
}

// iterations enqueued behind a stop: their kernels return at once
__device__ __forceinline__ bool lm_stopped(const BaDev& d) { return d.lm && d.lm->stop != LM_RUNNING; }

// a kernel's view of the parameter sets and the radius: the record's, when the loop runs on the device
__device__ __forceinline__ void lm_view(BaDev& d, double& radius) {
  if (d.lm) {
    const int par = d.lm->parity;
    radius = d.lm->radius;
    if (par) {
      double* t;
      t = d.cams, d.cams = d.cams_c, d.cams_c = t;
      t = d.pts, d.pts = d.pts_c, d.pts_c = t;
      t = d.focal, d.focal = d.focal_c, d.focal_c = t;
      t = d.camd, d.camd = d.camd_c, d.camd_c = t;
    }
  }
}

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
// v_rsq_f64 (about 26 good bits) + one third-order step: e = 1 - d r^2, r' = r (1 + e/2 + 3 e^2/8);
// error ~ e^3: five operations instead of the eight of two Newton steps, on the factorisation chain
__device__ __forceinline__ double rsqrt_f64_h(double d) {
  const double r = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-d * r, r, 1.0);
  const double p = __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(r * e, p, r);
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


// Reduce-scatter of P per-lane values over the wave: at every step (partner = lane ^ OFF, OFF = 32 .. 1) a lane keeps
// one half of its list and adds the partner's copy of that half (zero-padded when the length is odd), so that after
// six steps the lane holds the wave's sums of the entries [base, base + n) of the original list (n <= ceil(P / 64)):
// about P exchanges in all instead of 6 P for a butterfly that leaves every sum on lane 0, and whatever follows
// (atomics, LDS stores) is one instruction of many lanes instead of many instructions of one lane.
template <int P, int OFF, int NOUT>
__device__ __forceinline__ void wave_reduce_scatter(const double (&cur)[P], int lane, int& base, int& n, double (&out)[NOUT]) {
  if constexpr (OFF == 0) {
    static_assert(P <= NOUT, "out holds ceil(P0 / 64) entries");
#pragma unroll
    for (int k = 0; k < P; ++k) out[k] = cur[k];
  } else {
    constexpr int H = (P + 1) / 2;
    const bool bit = (lane & OFF) != 0;
    double nxt[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double lo = cur[k], hi = H + k < P ? cur[H + k] : 0.0;
      nxt[k] = (bit ? hi : lo) + __shfl_xor(bit ? lo : hi, OFF);
    }
    if (bit) {
      base += H;
      n = n > H ? n - H : 0;
    } else {
      n = n < H ? n : H;
    }
    wave_reduce_scatter<H, OFF / 2, NOUT>(nxt, lane, base, n, out);
  }
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
// ROT_SCALED: the three derivative matrices the table holds are scaled by the rotation columns' scale already
template <typename CP, typename SP, bool ROT_SCALED = false>
__device__ __forceinline__ void obs_linearize_g(CP cd, const double X[3], double focal, double ox, double oy,
                                                SP sc, const double* sp, double sf, ObsLin& o) {
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
    if (ROT_SCALED) {
      o.Jc[j] = d00 * dx + d02 * dz;
      o.Jc[6 + j] = d00 * dy + d12 * dz;
    } else {
      const double s = sc ? sc[j] : 1.0;
      o.Jc[j] = (d00 * dx + d02 * dz) * s;
      o.Jc[6 + j] = (d00 * dy + d12 * dz) * s;
    }
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

template <typename CP>
__device__ __forceinline__ void obs_linearize(CP cd, const double X[3], double focal, double ox, double oy,
                                              const double* sc, const double* sp, double sf, ObsLin& o) {
  obs_linearize_g<CP, const double*>(cd, X, focal, ox, oy, sc, sp, sf, o);
}
// A pointer the whole wave shares, in the constant address space: loads through it are scalar loads (one request per
// wave into SGPRs) instead of 64 lanes fetching the same address through the vector memory pipe.
typedef __attribute__((address_space(4))) const double cst_double;
__device__ __forceinline__ cst_double* wave_uniform_ptr(const double* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (cst_double*)(((unsigned long long)hi << 32) | lo);
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
// locally) and ||x||^2 of the points.  The camera and focal columns come from the linearisation
// kernels in their norms-only mode (into red_dc, summed across ranks before ba_make_scale).
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
      d.iscale_p[3 * p + j] = jacobi ? 1.0 + sqrt(np2[j]) : 1.0;
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
#ifdef SFM_ELIM_STAMPS
// diagnostic build only (scripts/elim_stamps.py): s_memtime at the phase boundaries of workgroup 0 / wave 0
__device__ unsigned long long g_elim_stamps[32];
__device__ unsigned long long g_elim_dump[32];
// per workgroup of the last launch: {s_memtime at its start, at its end, s_memrealtime (100 MHz) at its start, at its end, XCC/CU id}
__device__ unsigned long long g_elim_wg[2048 * 5];
#define SFM_ELIM_WG_RECORDS 1
#define EL_WG(slot)                                                                                  \
  do {                                                                                               \
    if (tid == 0 && blockIdx.x < 2048) {                                                             \
      unsigned long long t_, r_;                                                                     \
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory"); \
      g_elim_wg[5 * blockIdx.x + slot] = t_;                                                         \
      g_elim_wg[5 * blockIdx.x + 2 + slot] = r_;                                                     \
      if (slot == 0) {                                                                               \
        unsigned id_, xcc_;                                                                          \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(id_), "=s"(xcc_)); \
        g_elim_wg[5 * blockIdx.x + 4] = ((unsigned long long)xcc_ << 32) | id_;                      \
      }                                                                                              \
    }                                                                                                \
  } while (0)
#define EL_STAMP(slot, cond)                                                         \
  do {                                                                               \
    unsigned long long t_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    ((blockIdx.x == 0 && wave == 0 && (cond)) ? g_elim_stamps : g_elim_dump)[slot] = t_; \
  } while (0)
#define EL_STAMPW(slot)                                                        \
  do {                                                                         \
    unsigned long long t_;                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    ((blockIdx.x == 0) ? g_elim_stamps : g_elim_dump)[slot] = t_;              \
  } while (0)
#else
#define EL_STAMP(slot, cond)
#define EL_STAMPW(slot)
#define EL_WG(slot)
#endif
// a chunk's slab: [Gram block, MFMA layout, NT x 256 <= 2560 | F^T F sums 36 x FP | Jf^2, Jf r, r^2 | gmax | nfail]
constexpr int ELIM_SLAB_FF = 2560, ELIM_SLAB = 2944;
constexpr int FP = 10;  // slots per row of the F^T F accumulators in LDS (the MFMA path takes signatures of n <= 10 cameras)
constexpr int MP = 80;  // LDS row pitch (doubles) of a wave's M panel: the 4 k-rows of one
                        // fragment read sit 160 dwords apart -> disjoint banks

typedef __attribute__((address_space(3))) double lds_double;
struct CamLds {  // camera table transposed in LDS: element e of camera slot o at [e*16 + o]
  const lds_double* base;   // R, t (12 entries): loop-invariant per lane, the compiler keeps them in registers
  const lds_double* dbase;  // dR/dw (27 entries): re-read every iteration through a pinned pointer (54 VGPRs
                            // that the F^T F accumulators need more)
  __device__ __forceinline__ double operator[](int e) const { return e < 12 ? base[e * 16] : dbase[e * 16]; }
};

#ifndef SFM_ELIM_LB
#define SFM_ELIM_LB __launch_bounds__(512)
#endif
static bool elim_lp10() {
  static const bool v = !(getenv("SFMHIP_BA_ELIM_LP") && atoi(getenv("SFMHIP_BA_ELIM_LP")) == 16);
  return v;
}
// LP = lanes per point.  16 (rounds 2-5): a point's observations on one 16-lane DPP row, four points per wave and iteration; a
// ten-camera point leaves six lanes of sixteen with the constant camera slot.  10 (round 6): six points per wave and iteration
// (lanes 60-63 idle), the same instructions for 1.5 x the points; the twelve sums over a point's lanes, which no longer sit in one
// DPP row, meet through the wave's panel in LDS (group_sum12), and the Gram panel is 18 rows -- five k-steps -- instead of 12.
template <int LP>
struct ElimShape {
  static constexpr int PPW = 64 / LP;               // points per wave and iteration
  static constexpr int KST = (3 * PPW + 3) / 4;     // k-steps of the Gram update (4 panel rows each)
  static constexpr int PROWS = 4 * KST;             // panel rows per wave (rows beyond 3 * PPW stay zero)
};
// the twelve sums over the LP = 10 lanes of a point, every lane of the point ending up with all twelve: partials [12][66] into
// the wave's panel (it is rewritten afterwards), lane o of a point adds value o over the point's ten lanes in lane order (lanes
// 0 and 1 also values 10 and 11), the totals [6][12] behind the partials, read back by every lane of the point.  A wave's LDS
// operations execute in order; the fences keep the compiler from moving them.
__device__ __forceinline__ void group10_sum12(double (&red)[12], double* R, int lane, int q, int o) {
  constexpr int PR = 66;
  typedef __attribute__((address_space(3))) double ldsd;
  ldsd* Rl = (ldsd*)R;
#pragma unroll
  for (int v = 0; v < 12; ++v) Rl[v * PR + lane] = red[v];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int qq = q < 6 ? q : 5;  // (the idle lanes 60-63 read point 5's: finite numbers, masked out by the caller)
  const ldsd* a = Rl + o * PR + qq * 10;
  const ldsd* b = Rl + (10 + (o & 1)) * PR + qq * 10;
  double s0 = a[0], s1 = b[0];
#pragma unroll
  for (int j = 1; j < 10; ++j) {
    s0 += a[j];
    s1 += b[j];
  }
  ldsd* Tt = Rl + 12 * PR;
  if (q < 6) {
    Tt[q * 12 + o] = s0;
    if (o < 2) Tt[q * 12 + 10 + o] = s1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int v = 0; v < 12; ++v) red[v] = Tt[qq * 12 + v];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NB, int LP>
__device__ __forceinline__ void elim_chunk(const BaDev& d, const Chunk* __restrict__ chunks, const int chunk_index,
                                                         const int* __restrict__ sig_cams, double inv_radius,
                                                         double lm_lo, double lm_hi, int rank,
                                                         int norms /* 1: unscaled squared column norms of the cameras and the focal into dc, nothing else */,
                                                         double* __restrict__ slab /* non-null: the workgroup's sums go to its slab (ELIM_SLAB doubles per chunk) instead of atomics on S */) {
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int PPW = ElimShape<LP>::PPW, KST = ElimShape<LP>::KST, PROWS = ElimShape<LP>::PROWS;
  __shared__ __attribute__((aligned(16))) double s_cam[CAMD * 16];
  extern __shared__ __attribute__((aligned(16))) double s_M[];  // nw x PROWS x MP panels; also the cross-wave reduction buffer
  __shared__ double s_tsc[3 * 16];  // LP = 10: the cameras' translation scales by slot (read per iteration: six registers for F^T F sums)
  __shared__ int s_gidx[64];  // local Gram index -> row/column of S; -2: the rhs column u; -1: padding
  __shared__ double s_wv[16];  // per wave: gradient maximum, failed point blocks (slab epilogue)
  const int nw = blockDim.x >> 6;  // 4 waves for long runs, 1 for runs of a few points (unstructured visibility)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  EL_STAMP(0, true);
  EL_WG(0);
  const Chunk ch = chunks[chunk_index];
  const int n = ch.n;
  const int* cams = sig_cams + ch.sig_off;
  const int sld = d.ld, fo = 6 * d.nc;
  // (the three derivative matrices of a camera come out of the staging scaled by their rotation column's scale: 27
  // multiplications per camera and workgroup instead of six per observation and iteration)
  for (int idx = tid; idx < n * CAMD; idx += (int)blockDim.x) {
    const int o = idx / CAMD, e = idx - o * CAMD;
    double v = d.camd[(size_t)CAMD * cams[o] + e];
    if (!norms && e >= 12 && e < 39) v *= d.scale_c[6 * cams[o] + (e - 12) / 9];
    s_cam[e * 16 + o] = v;
  }
  // slot 15 (a signature has at most ten cameras): R = 0, t = (0, 0, 1), no derivatives -- what the lanes WITHOUT an observation
  // linearise instead of shadowing observation 0: constants in, constants out, no switching.  The launch runs at the clock the
  // power limit leaves it (2.1 GHz on a moving solve, 2.34 on zeros: scripts/elim_stamps.py): 69.1 -> 68.4 us (round 5).
  if (tid < CAMD) s_cam[tid * 16 + 15] = tid == 11 ? 1.0 : 0.0;
  if (tid < 48) {
    const int j = tid >> 4, so = tid & 15;
    s_tsc[tid] = (norms || so >= n) ? 1.0 : d.scale_c[6 * cams[so] + 3 + j];
  }
  // dynamic LDS: the F^T F accumulators [wave][e][slot] (nw x 36 x 16) + [3][16] | the waves' panels, later the
  // cross-wave reduction and the staged Gram block
  const int ff_sz = nw * 36 * FP + 3 * FP;
  double* s_P = s_M + ff_sz;
  for (int idx = tid; idx < ff_sz + nw * PROWS * MP; idx += (int)blockDim.x) s_M[idx] = 0.0;
  if (tid < 64) {
    int gi = -1;
    if (tid < 6 * n) gi = 6 * cams[tid / 6] + tid % 6;
    else if (tid == 6 * n) gi = fo;
    else if (tid == 6 * n + 1) gi = -2;
    s_gidx[tid] = gi;
  }
  const int o = LP == 16 ? (lane & 15) : lane % LP, q = LP == 16 ? (lane >> 4) : lane / LP;
  const bool lane_ok = q < PPW;  // (LP = 10: lanes 60-63 belong to no point)
  const bool valid_o = lane_ok && o < n;
  const int oc = valid_o ? o : 0;  // idle lanes take observation 0's addresses (finite data) and slot 15's camera table (above); masked out below
  double sc[6];
  {
    const int cam = cams[oc];
#pragma unroll
    for (int j = 0; j < 6; ++j) sc[j] = norms ? 1.0 : d.scale_c[6 * cam + j];
  }
  const double sf = norms ? 1.0 : *d.scale_f, focal = *d.focal;
  const int kobs0 = d.optr[ch.p0];
  __syncthreads();
  EL_STAMP(1, true);

  v4d acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  double gmax = 0.0;
  int nfail = 0;
  double* Mw = s_P + wave * (PROWS * MP);
  const lds_double* cam_lds = (const lds_double*)s_cam + (valid_o ? o : 15);
  const int frow = lane >> 4, fcol = lane & 15;
  // F^T F part of a camera slot (the 6x6 block (upper, 21), the focal border (6), F^T b (6), Jf^2, Jf r, r^2 --
  // what ba_cam_blocks formed from a second linearisation) meets in LDS, ds_add_f64 by the slot's four point lanes -- per
  // iteration for the sums that have no register (below), once behind the loop for those that have
  lds_double* ffw = (lds_double*)s_M + wave * (36 * FP) + oc;
  // point data of the next iteration is loaded one iteration ahead (a lone wave per SIMD otherwise waits a global
  // round trip per iteration)
  double nX[3];
  double2 nxy;
  auto fetch = [&](int quad_) {
    const int pi_ = PPW * quad_ + q;
    const int pl_ = pi_ < ch.cnt ? pi_ : ch.cnt - 1;
    const int p_ = ch.p0 + pl_;
    nX[0] = d.pts[3 * p_];
    nX[1] = d.pts[3 * p_ + 1];
    nX[2] = d.pts[3 * p_ + 2];
    nxy = d.oxy[kobs0 + pl_ * n + oc];
  };
  // The first FFREG of the 36 F^T F sums of the lane's camera slot are registers of the lane, added to the LDS cells once,
  // behind the loop (round 5).  The kernel lasts as long as its longest piece takes ALONE -- the older workgroup of a compute
  // unit runs at the pace of its own dependent chain, the younger one fills the gaps (scripts/elim_stamps.py) -- and 35
  // ds_add_f64 per iteration, four lanes to a cell, were 1.5 k of an iteration's 8.9 k cycles of that chain: every sum moved
  // to a register took 0.2 us off the launch (73.3 us with none, 71.9 / 69.7 / 68.7 with 8 / 14 / 21, 66.5 with 29 -- averages
  // that include iterations past the solve's convergence; 73.9 -> 68.4 us on a solve that still moves).  As many
  // as fit beside the Gram accumulators without spilling: 29 with ten tiles (256 registers, two waves per SIMD either way),
  // all 36 with six tiles or fewer.
#ifdef SFM_ELIM_FFREG
  constexpr int FFREG = SFM_ELIM_FFREG;  // (measurement builds)
#else
  // (LP = 10: the translation scales come from LDS every iteration, which leaves registers for three more sums; the four that
  // still have none go to cells of the LANE's own -- [e - FFREG][lane] in the wave's part of the accumulator array, whose cells
  // for the register sums are idle until the loop is over --, so that their ds_add_f64 hit no cell twice: with six lanes to a
  // cell they took 2 k cycles that the point sums' LDS round trip, next in the wave's LDS queue, had to wait for)
  constexpr int FFREG = NB == 4 ? (LP == 10 ? 32 : 29) : 36;
#endif
  constexpr bool FF_PRIVATE = LP == 10 && FFREG < 36;
  static_assert(!FF_PRIVATE || (36 - FFREG) * 64 <= FFREG * FP, "the lanes' own cells fit the idle part of the accumulator array");
  lds_double* ffp = (lds_double*)s_M + wave * (36 * FP) + lane;
  double ffr[FFREG > 0 ? FFREG : 1];
#pragma unroll
  for (int e = 0; e < FFREG; ++e) ffr[e] = 0.0;
  fetch(wave);
  EL_STAMP(2, true);
  EL_STAMPW(24 + (wave & 3));

  for (int quad = wave; PPW * quad < ch.cnt; quad += nw) {
    EL_STAMP(8, quad == wave + 2 * nw);
    const lds_double* cam_d = cam_lds;
    asm volatile("" : "+v"(cam_d));
    const CamLds cd{cam_lds, cam_d};
    const int pi = PPW * quad + q;
    const bool pv = lane_ok && pi < ch.cnt;
    const double X[3] = {nX[0], nX[1], nX[2]};
    // (the point's column scale and its inverse are not needed before the F^T F sums are through -- the scale goes onto Jp with
    // the mask below --, so they are loaded here and not an iteration ahead with the point: twelve registers for F^T F sums)
    double sp[3] = {1.0, 1.0, 1.0}, isp[3] = {1.0, 1.0, 1.0};
    if (!norms) {
      const int p_ = ch.p0 + (pv ? pi : ch.cnt - 1);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        sp[a] = d.scale_p[3 * p_ + a];
        isp[a] = d.iscale_p[3 * p_ + a];
      }
    }
    const double2 xy = nxy;
    if (PPW * (quad + nw) < ch.cnt) fetch(quad + nw);
    ObsLin ol;
    if (LP == 10) {
      const lds_double* tsc = (const lds_double*)s_tsc + (valid_o ? o : 15);
      asm volatile("" : "+v"(tsc));  // (re-read every iteration, not hoisted into six registers)
      struct {
        const lds_double* p;
        __device__ __forceinline__ explicit operator bool() const { return true; }
        __device__ __forceinline__ double operator[](int j) const { return j >= 3 ? p[(j - 3) * 16] : 1.0; }
      } scl{tsc};
      obs_linearize_g<CamLds, decltype(scl), true>(cd, X, focal, xy.x, xy.y, scl, (const double*)nullptr, sf, ol);
    } else {
      obs_linearize_g<CamLds, const double*, true>(cd, X, focal, xy.x, xy.y, sc, (const double*)nullptr, sf, ol);
    }
    EL_STAMP(9, quad == wave + 2 * nw);
    const double live = (pv && valid_o) ? 1.0 : 0.0;
    if (pv && valid_o) {
      // (Jc[4] = Jc[9] = 0: row 0 has no t_y column, row 1 no t_x column; entry (3,4) is identically zero)
      auto ff_add = [&](int e, double v) {
        if (FF_PRIVATE) __hip_atomic_fetch_add(ffp + (e - FFREG) * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(ffw + e * FP, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // one product / the sum of two products into sum e: a register of the lane while they last, else the LDS cell
      auto ff1 = [&](int e, double a, double c) {
        if (e < FFREG) ffr[e < FFREG ? e : 0] = fma(a, c, ffr[e < FFREG ? e : 0]);
        else ff_add(e, a * c);
      };
      auto ff2 = [&](int e, double a, double c, double a2, double c2) {
        if (e < FFREG) ffr[e < FFREG ? e : 0] = fma(a, c, fma(a2, c2, ffr[e < FFREG ? e : 0]));
        else ff_add(e, fma(a, c, a2 * c2));
      };
      int e = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j, ++e) {
          const bool a0 = i != 4 && j != 4, a1 = i != 3 && j != 3;
          if (a0 && a1) ff2(e, ol.Jc[i], ol.Jc[j], ol.Jc[6 + i], ol.Jc[6 + j]);
          else if (a0) ff1(e, ol.Jc[i], ol.Jc[j]);
          else if (a1) ff1(e, ol.Jc[6 + i], ol.Jc[6 + j]);
        }
        if (i == 4) {
          ff1(21 + i, ol.Jc[6 + i], ol.Jf[1]);
          ff1(27 + i, ol.Jc[6 + i], ol.r1);
        } else if (i == 3) {
          ff1(21 + i, ol.Jc[i], ol.Jf[0]);
          ff1(27 + i, ol.Jc[i], ol.r0);
        } else {
          ff2(21 + i, ol.Jc[i], ol.Jf[0], ol.Jc[6 + i], ol.Jf[1]);
          ff2(27 + i, ol.Jc[i], ol.r0, ol.Jc[6 + i], ol.r1);
        }
      }
      ff2(33, ol.Jf[0], ol.Jf[0], ol.Jf[1], ol.Jf[1]);
      ff2(34, ol.Jf[0], ol.r0, ol.Jf[1], ol.r1);
      ff2(35, ol.r0, ol.r0, ol.r1, ol.r1);
    }
    EL_STAMP(10, quad == wave + 2 * nw);
    // point block C = sum Jp^T Jp (lower: 00 10 11 20 21 22), gp = Jp^T r, wf = Jp^T Jf.  Idle lanes (no such
    // observation, or no such point in the last quad) carry Jp = 0: every product below -- the sums, W = Jc^T Jp, the
    // panel rows -- has Jp as a factor, so six multiplications mask them all (and carry the point's column scale)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double ls = live * sp[a];
      ol.Jp[a] *= ls;
      ol.Jp[3 + a] *= ls;
    }
    double red[12];
    red[0] = ol.Jp[0] * ol.Jp[0] + ol.Jp[3] * ol.Jp[3];
    red[1] = ol.Jp[1] * ol.Jp[0] + ol.Jp[4] * ol.Jp[3];
    red[2] = ol.Jp[1] * ol.Jp[1] + ol.Jp[4] * ol.Jp[4];
    red[3] = ol.Jp[2] * ol.Jp[0] + ol.Jp[5] * ol.Jp[3];
    red[4] = ol.Jp[2] * ol.Jp[1] + ol.Jp[5] * ol.Jp[4];
    red[5] = ol.Jp[2] * ol.Jp[2] + ol.Jp[5] * ol.Jp[5];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[6 + a] = ol.Jp[a] * ol.r0 + ol.Jp[3 + a] * ol.r1;
      red[9 + a] = ol.Jp[a] * ol.Jf[0] + ol.Jp[3 + a] * ol.Jf[1];
    }
    if (LP == 16) {
#pragma unroll
      for (int e = 0; e < 12; ++e) red[e] = row16_sum(red[e]);
    } else {
      group10_sum12(red, Mw, lane, q, o);
    }
    // LM damping of the point block: D^2 = clamp(diag) / radius
    double C[6] = {red[0], red[1], red[2], red[3], red[4], red[5]};
    C[0] += fmin(fmax(C[0], lm_lo), lm_hi) * inv_radius;
    C[2] += fmin(fmax(C[2], lm_lo), lm_hi) * inv_radius;
    C[5] += fmin(fmax(C[5], lm_lo), lm_hi) * inv_radius;
    EL_STAMP(11, quad == wave + 2 * nw);
    double Li[6];
    const bool pd = chol3_inv(C, Li);
    if (!pd) {
      if (pv && o == 0) ++nfail;
      Li[0] = Li[1] = Li[2] = Li[3] = Li[4] = Li[5] = 0;
    }
    const double pvf = pv ? 1.0 : 0.0;
    // gradient of the point (unscaled) for the gradient tolerance
    gmax = fmax(gmax, pvf * fmax(fabs(red[6] * isp[0]), fmax(fabs(red[7] * isp[1]), fabs(red[8] * isp[2]))));
    EL_STAMP(12, quad == wave + 2 * nw);
    // (LP = 10: the sums above went through the panel's rows, so EVERY column of a point's rows is written -- a lane without an
    // observation writes the zeros that Jp = 0 makes of its products, where LP = 16 leaves the zeros of the set-up alone)
    if (LP == 16 ? valid_o : lane_ok) {
      // T = (Jc^T Jp) Li^T = Jc^T (Jp Li^T): Q = Jp Li^T first (2 x 3, twelve operations), then T[i][k] = Jc[0][i] Q[0][k] +
      // Jc[1][i] Q[1][k] -- 48 operations where forming W = Jc^T Jp and then W Li^T took 72; row k of the panel
      double* row0 = Mw + (3 * q) * MP + 6 * o;
      const double q00 = ol.Jp[0] * Li[0], q10 = ol.Jp[3] * Li[0];
      const double q01 = ol.Jp[0] * Li[1] + ol.Jp[1] * Li[2], q11 = ol.Jp[3] * Li[1] + ol.Jp[4] * Li[2];
      const double q02 = ol.Jp[0] * Li[3] + ol.Jp[1] * Li[4] + ol.Jp[2] * Li[5], q12 = ol.Jp[3] * Li[3] + ol.Jp[4] * Li[4] + ol.Jp[5] * Li[5];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        row0[i] = ol.Jc[i] * q00 + ol.Jc[6 + i] * q10;
        row0[MP + i] = ol.Jc[i] * q01 + ol.Jc[6 + i] * q11;
        row0[2 * MP + i] = ol.Jc[i] * q02 + ol.Jc[6 + i] * q12;
      }
    }
    if (o == 0 && lane_ok) {
      // border columns: t_f = Li wf, u = Li gp
      double* b0 = Mw + (3 * q) * MP + 6 * n;
      b0[0] = pvf * (Li[0] * red[9]);
      b0[MP] = pvf * (Li[1] * red[9] + Li[2] * red[10]);
      b0[2 * MP] = pvf * (Li[3] * red[9] + Li[4] * red[10] + Li[5] * red[11]);
      b0[1] = pvf * (Li[0] * red[6]);
      b0[MP + 1] = pvf * (Li[1] * red[6] + Li[2] * red[7]);
      b0[2 * MP + 1] = pvf * (Li[3] * red[6] + Li[4] * red[7] + Li[5] * red[8]);
    }
    EL_STAMP(13, quad == wave + 2 * nw);
    // Gram update: the same fragment serves as A (M^T tile) and B (M tile) operand
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
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
    EL_STAMP(14, quad == wave + 2 * nw);
  }
  EL_STAMP(3, true);
  EL_STAMPW(20 + (wave & 7));
  double ffx[FF_PRIVATE ? 36 - FFREG : 1];
  if (FF_PRIVATE) {
    // the lanes' own cells: read, put back to zero (they are the register sums' cells again), then added like the register sums
#pragma unroll
    for (int e = 0; e < 36 - FFREG; ++e) ffx[e] = ffp[e * 64];
#pragma unroll
    for (int e = 0; e < 36 - FFREG; ++e) ffp[e * 64] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (valid_o) {
#pragma unroll
    for (int e = 0; e < FFREG; ++e)
      __hip_atomic_fetch_add(ffw + e * FP, ffr[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (FF_PRIVATE) {
#pragma unroll
      for (int e = 0; e < 36 - FFREG; ++e)
        __hip_atomic_fetch_add(ffw + (FFREG + e) * FP, ffx[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }

  // ---- epilogue.  Only LDS traffic between the barriers (a barrier waits for the wave's outstanding global
  // atomics: ~2 us each), every global atomic in the last phase:
  //   A  all loops done; wave 0 stores its Gram tiles to s_G (MFMA layout [tile][register][lane], over the panels)
  //   B  the other waves add theirs (ds_add_f64); the F^T F sums over the waves are folded in (S += F^T F - Gram)
  //   C  the three scalar sums over the slots
  //   D  one atomic per entry of the block and workgroup, issued by all waves, every workgroup starting at a
  //      different entry: the workgroups of a round end together and neighbouring signatures share most of their
  //      destinations -- same-address atomics serialise in L2
  lds_double* s_G = (lds_double*)s_P;  // NT x 256
  lds_double* s_F = (lds_double*)s_M;  // [wave][e][slot] nw x 36 x 16 (the accumulators) | [3][16] the scalars per slot
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    gmax = fmax(gmax, __shfl_down(gmax, off));
    nfail += __shfl_down(nfail, off);
  }
  if (lane == 0) {
    s_wv[2 * wave] = gmax;
    s_wv[2 * wave + 1] = (double)nfail;
  }
  __syncthreads();
  EL_STAMP(15, true);
  if (nw == 4) {
    // Four waves: every wave stores its tiles, five at a time, in a buffer of its own, and every thread adds the four copies of
    // its entries -- ((w0 + w1) + w2) + w3, the bits the waves' additions one after the other gave -- into registers; two passes
    // for ten tiles, then the sums go to s_G.  (Round 5: the sequential form, wave 0 stores | wave 1 adds | wave 2 adds | wave 3
    // adds, was 6.3 k cycles at the END of the longest piece, which is what the launch waits for.)
    constexpr int TP = NT < 5 ? NT : 5;
    lds_double* buf = (lds_double*)s_P;
    double r[NT];
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TP) {
#pragma unroll
      for (int t = t0; t < NT && t < t0 + TP; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) buf[wave * (TP * 256) + ((t - t0) * 4 + g) * 64 + lane] = acc[t][g];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TP && t0 + j < NT; ++j) {
        const int k = j * 256 + tid;
        r[t0 + j] = ((buf[k] + buf[TP * 256 + k]) + buf[2 * TP * 256 + k]) + buf[3 * TP * 256 + k];
      }
      __syncthreads();
    }
    if (!norms) {
#pragma unroll
      for (int j = 0; j < NT; ++j) s_G[j * 256 + tid] = r[j];
    }
    __syncthreads();
    EL_STAMP(16, true);
  } else {
  if (wave == 0 && !norms) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) s_G[(t * 4 + g) * 64 + lane] = acc[t][g];
  }
  __syncthreads();
  EL_STAMP(16, true);
  }
  if (nw == 4) {
  } else if (slab && !norms) {
    // (deterministic mode: the waves add their tiles one after the other -- the order they arrive in is not fixed)
    for (int w = 1; w < nw; ++w) {
      if (wave == w) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) s_G[(t * 4 + g) * 64 + lane] += acc[t][g];
      }
      __syncthreads();
    }
  } else if (wave != 0 && !norms) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        __hip_atomic_fetch_add(s_G + (t * 4 + g) * 64 + lane, acc[t][g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // local (row, column) of the Gram block -> its slot in s_G (row <= column)
  auto g_slot = [&](int lr, int lc) {
    const int ti = lr >> 4, tj = lc >> 4;
    const int t = ti * NB - ti * (ti - 1) / 2 + (tj - ti);
    return (t * 4 + ((lr & 15) >> 2)) * 64 + (lr & 3) * 16 + (lc & 15);
  };
  auto g_sub = [&](int lr, int lc, double v) {
    __hip_atomic_fetch_add(s_G + g_slot(lr, lc), -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  for (int idx = tid; idx < 36 * FP; idx += (int)blockDim.x) {
    const int e = idx / FP, slot = idx - e * FP;
    if (slot >= n) continue;
    double v = 0.0;
    for (int w = 0; w < nw; ++w) v += s_F[(w * 36 + e) * FP + slot];
    s_F[e * FP + slot] = v;  // (the sum over the waves, for phases C and D)
    if (norms || e >= 33) continue;
    if (e < 21) {
      int i = 0, rem = e;
      while (rem >= 6 - i) {
        rem -= 6 - i;
        ++i;
      }
      g_sub(6 * slot + i, 6 * slot + i + rem, v);
    } else if (e < 27) {
      g_sub(6 * slot + e - 21, 6 * n, v);      // the focal column
    } else {
      g_sub(6 * slot + e - 27, 6 * n + 1, v);  // the rhs column
    }
  }
  __syncthreads();
  EL_STAMP(4, true);
  if (tid < 3) {
    double v = 0.0;
    for (int slot = 0; slot < n; ++slot) v += s_F[(33 + tid) * FP + slot];
    s_F[(nw * 36 + tid) * FP] = v;
    if (!norms && tid < 2) g_sub(6 * n, 6 * n + tid, v);
  }
  __syncthreads();
  EL_STAMP(5, true);
  if (slab) {
    // ---- slab epilogue: every sum of this workgroup as plain stores into its own slab; ba_gather_slabs adds the
    // slabs into S, g, F^T b, the diagonal and the scalars in a fixed order.  The atomic scatter this replaces took
    // 36 k of a workgroup's 186 k cycles at cfg4 (all workgroups end together and ~25 of them add to the same entries:
    // device-scope f64 atomics execute at the memory side, same-address ones one after the other) and made S depend
    // on the order they landed in.
    double* my = slab + (size_t)chunk_index * ELIM_SLAB;
    if (!norms)
      for (int k = tid; k < NT * 256; k += (int)blockDim.x) my[k] = s_G[k];
    for (int idx = tid; idx < 36 * FP; idx += (int)blockDim.x) my[ELIM_SLAB_FF + idx] = s_F[idx];
    if (tid < 3) my[ELIM_SLAB_FF + 36 * FP + tid] = s_F[(nw * 36 + tid) * FP];
    if (tid == 0) {
      double gm = 0.0, nf = 0.0;
      for (int w = 0; w < nw; ++w) {
        gm = fmax(gm, s_wv[2 * w]);
        nf += s_wv[2 * w + 1];
      }
      my[ELIM_SLAB_FF + 36 * FP + 3] = gm;
      my[ELIM_SLAB_FF + 36 * FP + 4] = nf;
      // (the focal parameter's diagonal entry and gradient once more, where ba_gather_rows finds them without knowing n)
      my[ELIM_SLAB_FF + 36 * FP + 5] = norms ? 0.0 : s_G[g_slot(6 * n, 6 * n)];
      my[ELIM_SLAB_FF + 36 * FP + 6] = norms ? 0.0 : s_G[g_slot(6 * n, 6 * n + 1)];
    }
    EL_STAMP(6, true);
    EL_WG(1);
#ifdef SFM_ELIM_WG_RECORDS
    if (tid == 0 && blockIdx.x < 2048) g_elim_wg[5 * blockIdx.x + 4] |= (unsigned long long)ch.cnt << 48;
#endif
    return;
  }
  double* scv = red_sc(d);
  if (lane == 0 && !norms) {
    if (nfail) atomic_add_f64(scv + 2, (double)nfail);
    atomic_max_pos_f64(scv + SC + rank, gmax);
  }
  for (int idx = tid; idx < 33 * FP; idx += (int)blockDim.x) {  // F^T F diagonal -> dc, F^T b -> gF
    const int e = idx / FP, slot = idx - e * FP;
    if (slot >= n) continue;
    const int r0 = 6 * cams[slot];
    if (e < 21) {
      int i = 0, rem = e;
      while (rem >= 6 - i) {
        rem -= 6 - i;
        ++i;
      }
      if (rem == 0) atomic_add_f64(red_dc(d) + r0 + i, s_F[e * FP + slot]);
    } else if (e >= 27 && !norms) {
      atomic_add_f64(red_gF(d) + r0 + e - 27, s_F[e * FP + slot]);
    }
  }
  if (tid < 3) {
    const double v = s_F[(nw * 36 + tid) * FP];
    if (tid == 0) atomic_add_f64(red_dc(d) + fo, v);
    else if (norms) {
    } else if (tid == 1) atomic_add_f64(red_gF(d) + fo, v);
    else atomic_add_f64(scv + 0, v);
  }
  if (norms) return;
  {
    double* S = red_S(d);
    double* g = red_g(d);
    constexpr int TOT = NT * 256;
    const int rot = (int)((blockIdx.x * 7u) % (unsigned)NT) * 256 + (int)((blockIdx.x * 3u) & 3u) * 64;
    for (int k = tid; k < TOT; k += (int)blockDim.x) {
      int idx = k + rot;
      if (idx >= TOT) idx -= TOT;
      int t = idx >> 8, ti = 0;
      while (t >= NB - ti) {
        t -= NB - ti;
        ++ti;
      }
      const int tj = ti + t, gg = (idx >> 6) & 3, ln = idx & 63;
      const int lr = 16 * ti + (ln >> 4) + 4 * gg, lc = 16 * tj + (ln & 15);
      const int gr = s_gidx[lr], gc = s_gidx[lc];
      if (gr < 0 || lr > lc || gc == -1) continue;
      const double v = -s_G[idx];
      if (gc >= 0) atomic_add_f64(S + (size_t)gr * sld + gc, v);
      else atomic_add_f64(g + gr, v);
    }
  }
  EL_STAMP(6, true);
}

template <int NB, int LP>
__global__ SFM_ELIM_LB void ba_eliminate_mfma(BaDev d, const Chunk* __restrict__ chunks, const int* __restrict__ chunk_ids,
                                              const int* __restrict__ sig_cams, double inv_radius, double lm_lo, double lm_hi,
                                              int rank, int norms, double* __restrict__ slab) {
  // (one chunk per workgroup.  Round 3 tried several: with the slab epilogue a chunk's fixed cost is small, so the runs were cut
  // into unequal pieces and the short ones packed two to a workgroup by a cost model -- 97-104 us per linearisation stage at
  // cfg4 against 92 for this cut (scripts/gpu_ba_elim_pack.py at commit time): the loop around the body costs 35 VGPRs and SGPR
  // spills, and a workgroup's time is not the sum the model assumed)
  if (lm_stopped(d)) return;
  if (d.lm) {  // (the loop on the device: the radius and which parameter set is x come from the record of the last decision)
    double radius;
    lm_view(d, radius);
    inv_radius = 1.0 / radius;
  }
  elim_chunk<NB, LP>(d, chunks, chunk_ids[blockIdx.x], sig_cams, inv_radius, lm_lo, lm_hi, rank, norms, slab);
}

// Sums the slabs of ba_eliminate_mfma into the reduced-system buffer: one thread per destination (an entry of S's
// upper triangle, of g, F^T b, the diagonal, a scalar), its sources listed by the host in chunk order (bit 31 of a
// source = subtract).  dest < 0: the gradient maximum of this rank (a max, not a sum).  Nothing else writes `red` while
// this runs (same stream), so the read-modify-write is plain -- and S is the same bit pattern run after run.
__global__ __launch_bounds__(256) void ba_gather_slabs(const double* __restrict__ slab, const int* __restrict__ ptr,
                                                       const unsigned* __restrict__ src, const int* __restrict__ dest, int nd,
                                                       double* __restrict__ red, long long gmax_off, const LmDev* __restrict__ lm) {
  if (lm && lm->stop != LM_RUNNING) return;
  // sixteen lanes per destination (a destination has 10-30 sources, each in another chunk's slab: one load deep instead
  // of a chain of dependent loads per thread); lane j takes sources j, j + 16, ... in order, the lanes meet by the
  // fixed tree of row16_sum -- the same order every run
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, j = threadIdx.x & 15;
  const bool live = i < nd;
  const int d = live ? dest[i] : 0;
  const int k0 = live ? ptr[i] : 0, k1 = live ? ptr[i + 1] : 0;
  if (live && d < 0) {  // (one destination: this rank's gradient maximum)
    double v = 0.0;
    for (int k = k0 + j; k < k1; k += 16) v = fmax(v, slab[src[k]]);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 16));
    if (j == 0) red[gmax_off] = fmax(red[gmax_off], v);
    return;
  }
  double v = 0.0;
  for (int k = k0 + j; k < k1; k += 16) {
    const unsigned sk = src[k];
    const double x = slab[sk & 0x7fffffffu];
    v += (sk >> 31) ? -x : x;
  }
  v = row16_sum(v);
  if (live && j == 0) red[d] += v;
}

// The same sums, row by row (round 3: the slabs' Gram blocks are 97 % of the gather's sources, and a thread per destination
// reads them 8 bytes at a time from a different cache line each).  Three roles by workgroup index:
//  * [0, row_wgs): a wave per row of S that the MFMA path writes.  Its sources are (chunk, local row) records in chunk order
//    (header + first 32 records at addresses that follow from the row number: one round trip; the rest in an overflow list).
//    The wave reads a source's row of the Gram block with one lane per local column (four contiguous 128-byte segments in the
//    MFMA layout; lanes 62 / 63 take the row's F^T F diagonal and F^T b sums instead), adds it (ds_add_f64) into a row-long
//    accumulator in LDS at the column the signature's column map gives, and at the end writes S's row, g's, the diagonal's and
//    F^T b's entries.  One wave adds a row's sources one after the other: the same order, the same bits, every run.
//  * the workgroups behind them: ba_gather_slabs' code for whatever else the host lists (nothing at present).
//  * the last workgroup: the seven sums that every chunk adds to (the focal parameter's entries, the cost, the gradient
//    maximum, the failed point blocks), thread t over chunks t, t + T, ..., a fixed tree across the threads.
__global__ __launch_bounds__(256) void ba_gather_rows(const double* __restrict__ slab,
                                                      const int* __restrict__ colmap, const int4* __restrict__ row_hdr,
                                                      const int4* __restrict__ row_head, const int4* __restrict__ row_src, int nrows,
                                                      int row_wgs, int ld, int accw, int fo, int nchunks, const int* __restrict__ ptr,
                                                      const unsigned* __restrict__ src, const int* __restrict__ dest, int nd,
                                                      double* __restrict__ red, long long gmax_off, int rmw, const LmDev* __restrict__ lm) {
  extern __shared__ double s_acc[];  // per wave: accw >= 6 |cameras of the row's camera's list| + 4 (those columns of S | the focal column | g's entry | the F^T F diagonal's | F^T b's)
  if (lm && lm->stop != LM_RUNNING) return;
  if ((int)blockIdx.x == (int)gridDim.x - 1) {
    // the last workgroup: the seven destinations that every chunk adds to (the focal parameter's diagonal entry and gradient
    // from the Gram block's border, its F^T F diagonal and F^T b, the cost, the gradient maximum, the failed point blocks):
    // thread t sums chunks t, t + T, ... in order, the threads meet by a fixed tree
    const int T = blockDim.x, t = threadIdx.x;
    double p[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int c0 = t; c0 < nchunks; c0 += 4 * T) {  // (four chunks' loads in flight; summed in chunk order)
      double x[4][7];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* tail = slab + (size_t)(c0 + u * T < nchunks ? c0 + u * T : c0) * ELIM_SLAB + ELIM_SLAB_FF + 36 * FP;
        x[u][0] = tail[5];
        x[u][1] = tail[6];
#pragma unroll
        for (int q = 0; q < 5; ++q) x[u][2 + q] = tail[q];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + u * T < nchunks) {
#pragma unroll
          for (int q = 0; q < 7; ++q) p[q] = q == 5 ? fmax(p[q], x[u][q]) : p[q] + x[u][q];
        }
    }
    for (int q = 0; q < 7; ++q) s_acc[q * T + t] = p[q];
    __syncthreads();
    for (int h = T >> 1; h > 0; h >>= 1) {
      if (t < h)
        for (int q = 0; q < 7; ++q) s_acc[q * T + t] = q == 5 ? fmax(s_acc[q * T + t], s_acc[q * T + t + h]) : s_acc[q * T + t] + s_acc[q * T + t + h];
      __syncthreads();
    }
    if (t < 7) {  // (seven different addresses: one read-modify-write each, side by side)
      const size_t ssz = (size_t)ld * ld;
      const size_t at[7] = {(size_t)fo * ld + fo, ssz + fo, ssz + 2 * (size_t)ld + fo, ssz + (size_t)ld + fo, ssz + 3 * (size_t)ld,
                            (size_t)gmax_off, ssz + 3 * (size_t)ld + 2};
      const double v = s_acc[t * T], o = red[at[t]];
      red[at[t]] = t < 2 ? o - v : t == 5 ? fmax(o, v) : o + v;  // S -= Gram (F^T F folded in), g likewise; dc, F^T b, cost, nfail +=
    }
    return;
  }
  if ((int)blockIdx.x >= row_wgs) {
    const int i = (((int)blockIdx.x - row_wgs) * blockDim.x + threadIdx.x) >> 4, j = threadIdx.x & 15;
    const bool live = i < nd;
    const int d = live ? dest[i] : 0;
    const int k0 = live ? ptr[i] : 0, k1 = live ? ptr[i + 1] : 0;
    if (live && d < 0) {  // (one destination: this rank's gradient maximum)
      double v = 0.0;
      for (int k = k0 + j; k < k1; k += 16) v = fmax(v, slab[src[k]]);
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 16));
      if (j == 0) red[gmax_off] = fmax(red[gmax_off], v);
      return;
    }
    double v = 0.0;
    for (int k = k0 + j; k < k1; k += 64) {  // (four sources of the lane in flight; added in list order)
      unsigned sk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sk[u] = k + 16 * u < k1 ? src[k + 16 * u] : 0u;
      double x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = slab[sk[u] & 0x7fffffffu];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + 16 * u < k1) v += (sk[u] >> 31) ? -x[u] : x[u];
    }
    v = row16_sum(v);
    if (live && j == 0) red[d] += v;
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int w = blockIdx.x * nw + wave;
  if (w >= nrows) return;
  double* acc = s_acc + (size_t)wave * accw;
  // (the row's header and its first 32 source records sit at addresses that follow from w alone: one round trip for both)
  const int4 hdr = row_hdr[w];                      // {row of S, sources, first source record beyond the head, the camera's list (in colmap, behind its length)}
  int4 my = lane < 32 ? row_head[(size_t)w * 32 + lane] : make_int4(0, 0, 0, 0);
  const int* const clist = colmap + hdr.w;
  const int nT = 6 * clist[-1], nacc = nT + 4;  // (needed for the last loop only: in flight beside the sources)
  for (int c = lane; c < accw; c += 64) acc[c] = 0.0;
  const int gr = hdr.x, nsrc = hdr.y;
  // per-lane constants: where local column `lane` sits inside a row of the Gram block's MFMA layout
  const int lane_off = (lane >> 4) * 256 + (lane & 15);
  for (int kb = 0; kb < nsrc; kb += 32) {
    // source records: {slab offset of the chunk, local row | n << 8 | offset of the row inside the Gram block << 16, offset of
    // the signature's column map, offsets of the row's F^T F diagonal / F^T b sums}; up to GB sources in flight, each one load
    // of the Gram row and one of the column map (no load depends on another: a batch is one round trip to memory, and a cfg4
    // row has 25 sources)
    constexpr int GB = 32;
    const int m = nsrc - kb < 32 ? nsrc - kb : 32;
    if (kb > 0) my = lane < m ? row_src[hdr.z + (kb - 32) + lane] : make_int4(0, 0, 0, 0);  // (beyond the head: the overflow list)
    for (int u0 = 0; u0 < m; u0 += GB) {
      double v[GB];
      int col[GB];
      bool ok[GB];
      // phase 1: every load of the batch is issued (both unconditional, at clamped addresses) ...
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int uu = u0 + u < m ? u0 + u : m - 1;  // (wave-uniform)
        const int base = __builtin_amdgcn_readlane(my.x, uu), rec = __builtin_amdgcn_readlane(my.y, uu);
        const int cm = __builtin_amdgcn_readlane(my.z, uu), ff = __builtin_amdgcn_readlane(my.w, uu);
        const int lr = rec & 255, n = (rec >> 8) & 255, roff = (int)((unsigned)rec >> 16);
        ok[u] = u0 + u < m && ((lane >= lr && lane < 6 * n + 2) || lane >= 62);
        // lanes 62, 63 (beyond any Gram block of n <= 10 cameras): the row's entry of the F^T F diagonal and of F^T b
        const int idx = lane >= 62 ? ELIM_SLAB_FF + (lane == 62 ? (ff & 0xFFFF) : (int)((unsigned)ff >> 16)) : roff + lane_off;
        v[u] = slab[(size_t)(unsigned)base + (ok[u] ? idx : 0)];
        col[u] = colmap[cm + lane];
      }
      // ... before the first of them is waited for (a use next to its load makes the row's sources go to memory one after
      // the other)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < GB; ++u)  // (ds_add_f64: a wave's LDS operations execute in order -- the sources add one after the other)
        if (ok[u]) __hip_atomic_fetch_add((lds_double*)acc + col[u], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  double* Srow = red + (size_t)gr * ld;
  const size_t ssz = (size_t)ld * ld;
  for (int c = lane; c < nacc; c += 64) {
    const double a = acc[c];
    if (a != 0.0) {
      double* at = c < nT       ? Srow + 6 * clist[c / 6] + c % 6
                   : c == nT     ? Srow + fo
                   : c == nT + 1 ? red + ssz + gr
                   : c == nT + 2 ? red + ssz + 2 * (size_t)ld + gr
                                 : red + ssz + (size_t)ld + gr;
      // (rmw == 0: the MFMA path is the only writer of these entries between the memset and here -- no pair-path points)
      const double o = rmw ? *at : 0.0;
      *at = c <= nT + 1 ? o - a : o + a;  // S -= Gram (F^T F folded in), g likewise; the diagonal and F^T b +=
    }
  }
}

// The camera's own blocks for the pair path's points, from the camera-major list of their observations: thread per
// observation, register accumulation of the 6x6 block (upper, 21), the focal border (6) and the rhs (6) -- F^T F
// from a linearisation, minus the Schur terms T T^T, T t_f, T u from what ba_pp_points stored --; block-reduced,
// 33 (+ 12) atomics per camera: one addend per entry, the slices of a long camera list summed in slice order first, so
// that -- ba_pp_pairs adding one addend per entry too, and the kernels following each other in stream order -- the pair
// path's S, g and norms are the same bit patterns run after run, like the runs' (slab epilogue).
__global__ __launch_bounds__(256) void ba_cam_blocks(BaDev d, const int* __restrict__ cptr,
                                                     const int* __restrict__ cpt,
                                                     const double2* __restrict__ cxy, int nsplit,
                                                     int norms_only /* unscaled diagonal into dc only */,
                                                     const int2* __restrict__ cslot /* (T row, point slot) per entry */,
                                                     const double* __restrict__ T, const double* __restrict__ tfu,
                                                     const double* __restrict__ pp_part, int n_part, int rank,
                                                     double* __restrict__ cb_part /* nsplit > 1: 66 sums per (camera, slice) */,
                                                     int* __restrict__ cb_cnt /* nsplit > 1: slices of the camera that have stored theirs */) {
  __shared__ double sh[4][36];
  __shared__ double sh2[4][33];
  __shared__ int s_last;
  {
    double radius_unused;
    lm_view(d, radius_unused);
  }
  double q[33];  // the Schur part of the same 33 slots: T T^T (upper 21), T t_f (6), T u (6)
#pragma unroll
  for (int e = 0; e < 33; ++e) q[e] = 0.0;
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
    if (!norms_only) {
      const int2 cs = cslot[k];
      const double* Tt = T + 18 * (size_t)cs.x;
      const double* tu = tfu + 6 * (size_t)cs.y;
      double t[18];
#pragma unroll
      for (int m = 0; m < 18; ++m) t[m] = Tt[m];
      int e2 = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) q[e2++] += t[3 * i] * t[3 * j] + t[3 * i + 1] * t[3 * j + 1] + t[3 * i + 2] * t[3 * j + 2];
        q[21 + i] += t[3 * i] * tu[0] + t[3 * i + 1] * tu[1] + t[3 * i + 2] * tu[2];
        q[27 + i] += t[3 * i] * tu[3] + t[3 * i + 1] * tu[4] + t[3 * i + 2] * tu[5];
      }
    }
  }
  // (the focal's own terms Jf^2, Jf r and the cost r^2 come from ba_pp_points: a[33..35] stay zero here)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    double o1[1];
    int base = 0, n = 36;
    wave_reduce_scatter<36, 32, 1>(a, lane, base, n, o1);
    if (n > 0) sh[wave][base] = o1[0];
    base = 0, n = 33;
    wave_reduce_scatter<33, 32, 1>(q, lane, base, n, o1);
    if (n > 0) sh2[wave][base] = o1[0];
  }
  __syncthreads();
  // A camera with many observations is cut into nsplit slices, a workgroup each.  Their sums meet in a fixed order: every
  // slice stores its 66 sums (write-through), the last one to arrive -- whichever it is -- adds them up slice by slice and
  // issues the camera's atomics, so S, g and the norms get ONE addend per entry from this kernel and the result does not
  // depend on which slice finished first (the elimination's slabs do the same for the runs: bitwise reproducible sums).
  bool mine = len > 0;
  if (nsplit > 1 && len > 0) {
    if (threadIdx.x < 33) {
      const int e = threadIdx.x;
      double* my = cb_part + ((size_t)c * nsplit + part) * 66;
      __hip_atomic_store(my + e, sh[0][e] + sh[1][e] + sh[2][e] + sh[3][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(my + 33 + e, sh2[0][e] + sh2[1][e] + sh2[2][e] + sh2[3][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the stores are through before the slice counts itself in; no fence: a __threadfence() per workgroup is an L2
      // write-back each -- measured elsewhere at 29 us per iteration for 200 workgroups -- and the sums are read back with
      // device-scope loads)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int prev = atomicAdd(cb_cnt + c, 1);
      s_last = prev == nsplit - 1;
      if (s_last) cb_cnt[c] = 0;  // (nobody else touches it before the next launch)
    }
    __syncthreads();
    mine = s_last != 0;
  }
  if (threadIdx.x < 33 && mine) {
    const int e = threadIdx.x;
    double v = sh[0][e] + sh[1][e] + sh[2][e] + sh[3][e];
    double sq = sh2[0][e] + sh2[1][e] + sh2[2][e] + sh2[3][e];
    if (nsplit > 1) {
      v = 0.0, sq = 0.0;
      for (int sl = 0; sl < nsplit; ++sl) {
        const double* pt = cb_part + ((size_t)c * nsplit + sl) * 66;
        v += __hip_atomic_load(pt + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sq += __hip_atomic_load(pt + 33 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
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
    } else if (e < 21) {
      int i = 0, rem = e;
      while (rem >= 6 - i) {
        rem -= 6 - i;
        ++i;
      }
      const int j = i + rem;
      atomic_add_f64(S + (size_t)(r0 + i) * sld + r0 + j, v - sq);
      if (i == j) atomic_add_f64(dc + r0 + i, v);
    } else if (e < 27) {
      atomic_add_f64(S + (size_t)(r0 + e - 21) * sld + fo, v - sq);
    } else {
      atomic_add_f64(g + r0 + e - 27, v - sq);
      atomic_add_f64(gF + r0 + e - 27, v);
    }
  }
  // ---- workgroup 0: the scalars of ba_pp_points (its workgroups' partial sums: the focal's diagonal / rhs / column
  // norm, the cost, failures, the gradient maximum)
  if (blockIdx.x != 0 || n_part <= 0) return;
  __syncthreads();
  double v[7];
#pragma unroll
  for (int e = 0; e < 7; ++e) v[e] = 0.0;
  for (int w = threadIdx.x; w < n_part; w += 256)
#pragma unroll
    for (int e = 0; e < 7; ++e) v[e] = e == 6 ? fmax(v[e], pp_part[8 * (size_t)w + e]) : v[e] + pp_part[8 * (size_t)w + e];
#pragma unroll
  for (int e = 0; e < 7; ++e)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[e] = e == 6 ? fmax(v[e], __shfl_down(v[e], off)) : v[e] + __shfl_down(v[e], off);
  if (lane == 0)
    for (int e = 0; e < 7; ++e) sh[wave][e] = v[e];
  __syncthreads();
  if (threadIdx.x < 7) {
    const int e = threadIdx.x;
    const double t = e == 6 ? fmax(fmax(sh[0][e], sh[1][e]), fmax(sh[2][e], sh[3][e])) : sh[0][e] + sh[1][e] + sh[2][e] + sh[3][e];
    const int fo = 6 * d.nc;
    double* scv = red_sc(d);
    if (norms_only) {
      if (e == 2) atomic_add_f64(red_dc(d) + fo, t);
    } else if (e == 0) atomic_add_f64(red_S(d) + (size_t)fo * d.ld + fo, t);
    else if (e == 1) atomic_add_f64(red_g(d) + fo, t);
    else if (e == 2) atomic_add_f64(red_dc(d) + fo, t);
    else if (e == 3) atomic_add_f64(red_gF(d) + fo, t);
    else if (e == 4) atomic_add_f64(scv + 0, t);
    else if (e == 5) {
      if (t != 0.0) atomic_add_f64(scv + 2, t);
    } else atomic_max_pos_f64(scv + SC + rank, t);
  }
}

// Generic path of the Schur correction ("pair path"): points whose camera list is shared by few others (short runs),
// is unsorted, has a camera twice, or is longer than 10.  Per-point atomic scatters cost ~2000 atomics per point (random
// visibility at 200 cameras / 20 k points: 0.77 ms per linearisation against 0.055 for runs); here every block of S
// is summed where it is written:
//   ba_pp_points  PP_LANES lanes per point (the observations dealt over them, the sums met by DPP): C_p, its inverse factor, T_o = (Jc_o^T Jp_o) C_p^-1/2 of every observation
//                 stored (18 doubles), t_f and u of the point stored, the focal / cost scalars reduced per workgroup;
//   ba_pp_pairs   one wave per camera pair that some point sees together: S[a][b] -= sum over the pair's entries
//                 (host-built list of (observation of a, observation of b)) of T_a T_b^T, 36 atomics per PAIR;
//   ba_cam_blocks one workgroup per (camera, slice) over the camera-major list: F^T F as before, and with the stored
//                 T, t_f, u the Schur terms of the camera's own blocks: S[c][c] -= T T^T, S[c][f] -= T t_f, g[c] -= T u.
// (one thread per point left 20 k points on 313 waves, each walking its observations twice: 39 us of pure latency)
constexpr int PP_LANES = 8;
__device__ __forceinline__ double pp_group_sum(double v) {  // over the 8 lanes of a point, every lane gets the total
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  return v;
}
__global__ __launch_bounds__(256) void ba_pp_points(BaDev d, const int* __restrict__ plist, const int* __restrict__ obase,
                                                    int n_list, double radius, double lm_lo, double lm_hi, int rank,
                                                    double* __restrict__ T, double* __restrict__ tfu, int norms,
                                                    double* __restrict__ part /* 8 per workgroup */) {
  lm_view(d, radius);
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gt / PP_LANES, sub = gt % PP_LANES;
  double sff = 0, gf = 0, jf2 = 0, jfr = 0, rr = 0, gmax = 0, nfail = 0;
  if (i < n_list) {
    const int p = plist[i];
    const int k0 = d.optr[p], n = d.optr[p + 1] - k0;
    const double X[3] = {d.pts[3 * p], d.pts[3 * p + 1], d.pts[3 * p + 2]};
    double sp[3] = {1.0, 1.0, 1.0};
    if (!norms) {
      sp[0] = d.scale_p[3 * p];
      sp[1] = d.scale_p[3 * p + 1];
      sp[2] = d.scale_p[3 * p + 2];
    }
    const double sf = norms ? 1.0 : *d.scale_f, focal = *d.focal;
    double C[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, wf[3] = {0, 0, 0};
    for (int o = sub; o < n; o += PP_LANES) {
      const int c = d.ocam[k0 + o];
      const double2 xy = d.oxy[k0 + o];
      ObsLin ol;
      obs_linearize(d.camd + (size_t)CAMD * c, X, focal, xy.x, xy.y, norms ? nullptr : d.scale_c + 6 * c, sp, sf, ol);
      C[0] += ol.Jp[0] * ol.Jp[0] + ol.Jp[3] * ol.Jp[3];
      C[1] += ol.Jp[1] * ol.Jp[0] + ol.Jp[4] * ol.Jp[3];
      C[2] += ol.Jp[1] * ol.Jp[1] + ol.Jp[4] * ol.Jp[4];
      C[3] += ol.Jp[2] * ol.Jp[0] + ol.Jp[5] * ol.Jp[3];
      C[4] += ol.Jp[2] * ol.Jp[1] + ol.Jp[5] * ol.Jp[4];
      C[5] += ol.Jp[2] * ol.Jp[2] + ol.Jp[5] * ol.Jp[5];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        gp[a] += ol.Jp[a] * ol.r0 + ol.Jp[3 + a] * ol.r1;
        wf[a] += ol.Jp[a] * ol.Jf[0] + ol.Jp[3 + a] * ol.Jf[1];
      }
      jf2 += ol.Jf[0] * ol.Jf[0] + ol.Jf[1] * ol.Jf[1];
      jfr += ol.Jf[0] * ol.r0 + ol.Jf[1] * ol.r1;
      rr += ol.r0 * ol.r0 + ol.r1 * ol.r1;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) C[a] = pp_group_sum(C[a]);
#pragma unroll
    for (int a = 0; a < 3; ++a) gp[a] = pp_group_sum(gp[a]), wf[a] = pp_group_sum(wf[a]);
    // (jf2, jfr, rr stay per lane: the workgroup sums below add them up)
    if (!norms) {
      C[0] += fmin(fmax(C[0], lm_lo), lm_hi) / radius;
      C[2] += fmin(fmax(C[2], lm_lo), lm_hi) / radius;
      C[5] += fmin(fmax(C[5], lm_lo), lm_hi) / radius;
      double Li[6];
      const bool pd = chol3_inv(C, Li);
      if (!pd) {
        Li[0] = Li[1] = Li[2] = Li[3] = Li[4] = Li[5] = 0;
        if (sub == 0) nfail = 1;
      }
      const double tf[3] = {Li[0] * wf[0], Li[1] * wf[0] + Li[2] * wf[1], Li[3] * wf[0] + Li[4] * wf[1] + Li[5] * wf[2]};
      const double u[3] = {Li[0] * gp[0], Li[1] * gp[0] + Li[2] * gp[1], Li[3] * gp[0] + Li[4] * gp[1] + Li[5] * gp[2]};
      if (sub == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          tfu[6 * (size_t)i + a] = tf[a];
          tfu[6 * (size_t)i + 3 + a] = u[a];
        }
        sff = tf[0] * tf[0] + tf[1] * tf[1] + tf[2] * tf[2];
        gf = tf[0] * u[0] + tf[1] * u[1] + tf[2] * u[2];
        gmax = fmax(fabs(gp[0] / sp[0]), fmax(fabs(gp[1] / sp[1]), fabs(gp[2] / sp[2])));
      }
      double* Tp = T + 18 * (size_t)obase[i];
      for (int o = sub; o < n; o += PP_LANES) {  // (linearised again: cheaper than keeping 12 + 6 doubles per observation)
        const int c = d.ocam[k0 + o];
        const double2 xy = d.oxy[k0 + o];
        ObsLin ol;
        obs_linearize(d.camd + (size_t)CAMD * c, X, focal, xy.x, xy.y, d.scale_c + 6 * c, sp, sf, ol);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const double w0 = ol.Jc[r] * ol.Jp[0] + ol.Jc[6 + r] * ol.Jp[3];
          const double w1 = ol.Jc[r] * ol.Jp[1] + ol.Jc[6 + r] * ol.Jp[4];
          const double w2 = ol.Jc[r] * ol.Jp[2] + ol.Jc[6 + r] * ol.Jp[5];
          Tp[18 * o + 3 * r + 0] = w0 * Li[0];
          Tp[18 * o + 3 * r + 1] = w0 * Li[1] + w1 * Li[2];
          Tp[18 * o + 3 * r + 2] = w0 * Li[3] + w1 * Li[4] + w2 * Li[5];
        }
      }
    }
  }
  // workgroup sums: the focal's diagonal / rhs / column norm, the cost, failures, the gradient maximum
  __shared__ double sh[4][8];
  double v[7] = {jf2 - sff, jfr - gf, jf2, jfr, rr, nfail, gmax};
#pragma unroll
  for (int e = 0; e < 7; ++e)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[e] = e == 6 ? fmax(v[e], __shfl_down(v[e], off)) : v[e] + __shfl_down(v[e], off);
  if ((threadIdx.x & 63) == 0)
    for (int e = 0; e < 7; ++e) sh[threadIdx.x >> 6][e] = v[e];
  __syncthreads();
  // (no atomics here: a few hundred workgroups adding to the same seven addresses drain at ~44 ns each, 27 us for
  // 625 workgroups; ba_cam_blocks, which always follows, adds the partial sums up and issues the seven atomics once)
  if (threadIdx.x < 7) {
    const int e = threadIdx.x;
    part[8 * (size_t)blockIdx.x + e] =
        e == 6 ? fmax(fmax(sh[0][e], sh[1][e]), fmax(sh[2][e], sh[3][e])) : sh[0][e] + sh[1][e] + sh[2][e] + sh[3][e];
  }
}

// one wave per camera pair: S[a][b] (a < b: the block right of the diagonal; a == b: a point that sees camera a twice,
// the symmetric sum into the diagonal block's upper part) -= sum_e T(oa_e) T(ob_e)^T
__global__ __launch_bounds__(64) void ba_pp_pairs(BaDev d, const int* __restrict__ pair_ptr, const int2* __restrict__ pair_cams,
                                                  const int2* __restrict__ entries, const double* __restrict__ T) {
  const int pr = blockIdx.x, lane = threadIdx.x;
  const int e0 = pair_ptr[pr], e1 = pair_ptr[pr + 1];
  const int2 cc = pair_cams[pr];
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  for (int e = e0 + lane; e < e1; e += 64) {
    const int2 en = entries[e];
    const double* Ta = T + 18 * (size_t)en.x;
    const double* Tb = T + 18 * (size_t)en.y;
    double a[18], bq[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      a[k] = Ta[k];
      bq[k] = Tb[k];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[6 * i + j] += a[3 * i] * bq[3 * j] + a[3 * i + 1] * bq[3 * j + 1] + a[3 * i + 2] * bq[3 * j + 2];
  }
  // (the wave's sums by reduce-scatter: every one of 36 lanes ends up with one entry and issues its own atomic)
  double tot1[1];
  int k = 0, nk = 36;
  wave_reduce_scatter<36, 32, 1>(acc, lane, k, nk, tot1);
  const double tot = tot1[0];
  if (nk > 0) {
    double* S = red_S(d);
    const int ra = 6 * cc.x, rb = 6 * cc.y, i = k / 6, j = k % 6;
    if (cc.x != cc.y) {
      atomic_add_f64(S + (size_t)(ra + i) * d.ld + rb + j, -tot);
    } else {
      // (a point that sees camera a twice: the symmetric sum into the diagonal block's upper part)
      if (i == j) atomic_add_f64(S + (size_t)(ra + i) * d.ld + ra + i, -2.0 * tot);
      else atomic_add_f64(S + (size_t)(ra + (i < j ? i : j)) * d.ld + ra + (i < j ? j : i), -tot);
    }
  }
}

// LM diagonal of the camera/focal columns onto S; gradient max over those columns
__global__ __launch_bounds__(1024) void ba_finalize(BaDev d, double radius, double lm_lo, double lm_hi, int world,
                                                    int add_diag) {
  __shared__ double sh[16];
  if (d.lm) radius = d.lm->radius;
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
  for (int i = threadIdx.x; i < d.ld; i += blockDim.x) {
    d.xinv[(size_t)i * d.ld + i] = 1.0;  // (zeroed with the rest of the buffer at the linearisation)
    d.z[i] = 0.0;
  }
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
// r >= c, so a column of L is contiguous.  ld = dim rounded up to 64 (identity on the padded
// diagonal), so no tile needs a bounds check and the 32-column panels come in pairs.  The rhs g
// is carried as one extra tile row (row 0 of tile row nt, kept in y) so that y = L^-1 g falls out
// of the factorisation.  The identity rides along the same way as nt more tile rows, X <- X L^-T, so
// that the backward substitution L^T z = y -- a second 1216-step dependency chain, 80 us of grouped
// kernels before -- is one product z = X y (chol_apply_inverse, ~6 us).  Right-looking, one launch per two
// panels (chol_step2 below), no dependency between workgroups inside a launch.
#ifdef SFM_CHOL_STAMPS
// diagnostic build only (scripts/chol_stamps.py): s_memtime at the phase boundaries of one panel
// workgroup, written to a buffer nothing else reads
__device__ unsigned long long g_chol_stamps[32];
#endif
constexpr int CB = 32;
constexpr int CBP = 34;  // LDS row pitch in doubles: 16-byte aligned rows, conflict-free tile writes

// P[c][r] -= sum_kk Lc[c][kk] * Lr[r][kk] for the 16x16 sub-tiles (ci, ri) of a 32x32 tile.
// MFMA roles: A operand = Lc (lane: row c = lane&15, k = lane>>4), B operand = Lr (col r =
// lane&15), so that the result's lane index is the memory-contiguous tile row r and its 4
// registers are tile columns c = (lane>>4) + 4g.
#define CHOL_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0)

// Bounded LDS spin, until *flag >= need: one opaque asm block (a C loop splits the unrolled block
// steps that call it into basic blocks, and the register allocator spills).  The counters of a workgroup are one
// 64-byte block of LDS; a spin that runs out of budget (a bug, not a data condition) leaves a mark in the block's
// last word (F_TIMEOUT), which every wave looks at when it is done (chol2_report_timeout): the launch then reports
// a failed factorisation and the host discards the step instead of taking a wrong one.
__device__ __forceinline__ int lds_wait_ge(const int* flag, int need) {
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) int*)flag;
  const unsigned mark = (addr & ~63u) | 60u;
  int seen_, budget_ = 1 << 20;
  asm volatile(
      "1:\n\t"
      "ds_read_b32 %0, %2\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_lt_i32 vcc, %0, %3\n\t"
      "s_cbranch_vccz 2f\n\t"
      "s_sub_u32 %1, %1, 1\n\t"
      "s_cmp_eq_u32 %1, 0\n\t"
      "s_cbranch_scc1 3f\n\t"
      "s_sleep 1\n\t"
      "s_branch 1b\n\t"
      "3:\n\t"
      "ds_write_b32 %4, %3\n\t"  // (need >= 1: any non-zero value marks the timeout)
      "2:\n\t"
      : "=&v"(seen_), "+s"(budget_)
      : "v"(addr), "v"(need), "v"(mark)
      : "vcc", "scc", "memory");
  return seen_;  // (what the counter read: callers catching up on several steps need not poll again)
}

// ------------------------------------------------------------------------------------------
// chol_step2: TWO 32-column panels (a = 2*k2, b = a+1) per launch.  Every tile is kept up to date
// except for the two panels of the previous launch ("pending", K = 64), which each launch folds in
// as it reads a tile.  A launch is latency-bound: ~4k cycles of global-memory round trip, then two
// dependent 32x32 factorisations of ~8k cycles each on ONE wave, with the solves of the workgroup's
// tile row trailing them; everything else is arranged around that chain.
//
// Inside the panel workgroup everything is a rank-4 block step on tiles held in REGISTERS in the
// MFMA accumulator layout (lane = tile row mod 16 + 16*(column mod 4), 4 registers = columns
// q, q+4, q+8, q+12 of a 16x16 sub-tile), because v_mfma_f64_16x16x4_f64 has K = 4:
//   block step of POTRF  the block's four columns go through LDS into row-per-lane registers,
//                        are factored there (pivots by v_readlane), written back as L, re-read
//                        in operand layout, and folded into the columns still to come with one
//                        MFMA per live sub-tile (chol2_potrf_blocks, ~1k cycles);
//   block step of TRSM   the same against a finished block of L (published through LDS with a
//                        progress counter), X = T L^-T (chol2_trsm_blocks, ~0.8k cycles);
//   U2                   D_bb -= Y Y^T and T_b -= X_a Y^T, one MFMA per sub-tile per finished block
//                        of Y (and X_a), so that the second panel's tiles are final ~one block
//                        step after the first panel's solves end.
// Panel workgroup (11 waves; roles and their SIMDs in chol2_panel), for tile row r of the block
// column (the owner: rows a and b; every workgroup factors D_aa and D_bb redundantly):
//   D_aa <- global minus pending, operands straight from global memory (the chain starts here);
//   the pending panels' rows of block rows a, b and r are staged in LDS once (C2_OFF_PAB/PRW) and
//   every other fold (D_ba = tile(b,a), T_a = tile(r,a), D_bb, T_b = tile(r,b)) takes its operands
//   from there; POTRF(D_aa) | Y = D_ba L_aa^-T (= L_ba) and X_a = T_a L_aa^-T trailing it | U2 |
//   POTRF(D_bb) | X_b = T_b L_bb^-T trailing it.  The owner has no tile row of its own and stores nothing:
//   the diagonal tiles stay as they were read (no later launch reads them, and the workgroups of this launch may
//   start at any time -- a launch that shares the device with another stream or process does not start as one
//   round); it reports a failed pivot.  The row blocks 0..b of X (see chol_step2) are further tile rows like any other.
// f64 MFMA throughput (16 FMA/clock/SIMD on gfx950, no more than the vector ALU) is what the
// first half of a launch is short of: the folds are ordered per SIMD by hand (priorities do not
// order the MFMAs of co-resident waves) and the right column halves of D_bb and T_b, which the
// second factorisation and solve need four block steps later, are folded by wave 0 after its own
// factorisation.
// All hand-offs are LDS counters (one writer lane, bounded spins); LDS operations of one wave
// execute in order, so data written before a counter update is visible to whoever saw the update.
// Trailing workgroups (one wave per tile, four tiles each) fold the pending panels into the tiles
// right of the block column.
constexpr int C2_WAVES = 11;
enum { F_PROG_A = 0, F_PROG_B, F_CNT_DAA, F_CNT_TA, F_CNT_DBA, F_YPROG, F_XPROG, F_STAGED, F_DBB_HI, F_TB_HI, F_DBA_S3, F_COUNT, F_TIMEOUT = 15 };
// dynamic LDS of chol_step2 (doubles): five tiles | 1/diag of both panels | the pending panels' rows
// of the block rows a and b (64 rows x 64 columns, [column][row], pitch PAB) and of the own tile row
// (32 x 64, pitch PRW) | the counters.  The pitches put the four k-slices of an MFMA operand read
// (lanes 16 apart) on disjoint banks.
constexpr int C2_PAB = 80, C2_PRW = 48;
constexpr int C2_OFF_SDI = 5 * CB * CBP, C2_OFF_PAB = C2_OFF_SDI + 2 * CB, C2_OFF_PRW = C2_OFF_PAB + 64 * C2_PAB,
              C2_OFF_FLAG = C2_OFF_PRW + 64 * C2_PRW, C2_LDS_BYTES = (C2_OFF_FLAG + 8) * 8;
static_assert(F_COUNT <= F_TIMEOUT, "the counters have 8 doubles of LDS, the last word is the timeout mark");
static_assert((C2_OFF_FLAG * 8) % 64 == 0, "lds_wait_ge finds the timeout mark in the counters' 64-byte block");
typedef double v2d __attribute__((ext_vector_type(2)));

// (explicitly LDS-typed: through a generic pointer these become flat, system-scope operations)
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void lds_flag_set(int* flag, int v) {
  asm volatile("" ::: "memory");
  *(volatile lds_int*)flag = v;
}
__device__ __forceinline__ void lds_flag_add(int* flag, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add((lds_int*)flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// d[k] = p[k * st], k = 0..N-1, by pointer increments: written as p[k * st] every address costs a
// 64-bit multiply-add (quarter rate), and a wave spends longer forming addresses than the loads
// take to come back.  The pin keeps the optimiser from re-deriving the products.
typedef __attribute__((address_space(1))) double gbl_double;  // (through the pin a generic pointer would load flat)
template <int N>
__device__ __forceinline__ void ld_strided(double* d, const double* p_, size_t st) {
  const gbl_double* p = (const gbl_double*)p_;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    d[k] = *p;
    p += st;
    asm volatile("" : "+v"(p));
  }
}

// acc -= sum_ks a[ks] (x) b[ks], four partial accumulators
__device__ __forceinline__ void chol2_fold_regs(v4d& acc, const double (&a)[16], const double (&b)[16]) {
  v4d p1 = {0.0, 0.0, 0.0, 0.0}, p2 = p1, p3 = p1;
#pragma unroll
  for (int ks = 0; ks < 16; ks += 4) {
    CHOL_MFMA(acc, -a[ks], b[ks]);
    CHOL_MFMA(p1, -a[ks + 1], b[ks + 1]);
    CHOL_MFMA(p2, -a[ks + 2], b[ks + 2]);
    CHOL_MFMA(p3, -a[ks + 3], b[ks + 3]);
  }
  acc += (p1 + p2) + p3;
}
// acc -= sum over the 64 pending columns of (column operand) x (row operand), both staged in LDS:
// pa / pb point at the lane's element of k-slice 0 (row index + lane&15 added, + (lane>>4) * pitch)
// (four partial accumulators: a dependent v_mfma_f64_16x16x4_f64 issues every ~120 cycles, an
// independent one every 64)
__device__ __forceinline__ void chol2_fold(v4d& acc, const double* pa, int pitch_a, const double* pb, int pitch_b) {
  double a[16], b[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    a[ks] = pa[4 * ks * pitch_a];
    b[ks] = pb[4 * ks * pitch_b];
  }
  chol2_fold_regs(acc, a, b);
}
// 16x16 sub-tile s (ci = s&1 column half, ri = s>>1 row half) of the tile at M(rb.., cb..), from global memory
__device__ __forceinline__ v4d chol2_tile(const double* __restrict__ A, int ld, int rb, int cb, int s, int lane) {
  const int j16 = lane & 15, q = lane >> 4, ci = s & 1, ri = s >> 1;
  double t[4];
  ld_strided<4>(t, A + (size_t)(cb + 16 * ci + q) * ld + rb + 16 * ri + j16, 4 * (size_t)ld);
  return v4d{t[0], t[1], t[2], t[3]};
}
// the same of the rhs tile row: y[cb..] on tile row 0, zero elsewhere
__device__ __forceinline__ v4d chol2_tile_rhs(const double* __restrict__ y, int cb, int s, int lane) {
  const int j16 = lane & 15, q = lane >> 4, ci = s & 1, ri = s >> 1;
  v4d t;
#pragma unroll
  for (int g = 0; g < 4; ++g) t[g] = ri == 0 && j16 == 0 ? y[cb + 16 * ci + q + 4 * g] : 0.0;
  return t;
}

__device__ __forceinline__ void chol2_put(double* dst, int s, int lane, v4d acc) {
  const int j16 = lane & 15, q = lane >> 4, ci = s & 1, ri = s >> 1;
#pragma unroll
  for (int g = 0; g < 4; ++g) dst[(16 * ri + j16) * CBP + 16 * ci + q + 4 * g] = acc[g];
}
__device__ __forceinline__ v4d chol2_get(const double* src, int s, int lane) {
  const int j16 = lane & 15, q = lane >> 4, ci = s & 1, ri = s >> 1;
  v4d acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = src[(16 * ri + j16) * CBP + 16 * ci + q + 4 * g];
  return acc;
}

#ifdef SFM_CHOL_STAMPS
// branch-free (a conditional stamp splits the unrolled chains into basic blocks and the register
// allocator spills): every panel workgroup stamps, all but one into a dump buffer
__device__ unsigned long long g_chol_dump[32];
#define C2_STAMP(slot)                                                                          \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
    ((k2 == 2 && blockIdx.x == 1) ? g_chol_stamps : g_chol_dump)[slot] = t_;                    \
  } while (0)
#else
#define C2_STAMP(slot)
#endif
// keeps the optimiser from enumerating which of the LDS tiles an offset selects (it would
// materialise every tile address of the unrolled loops as a scalar constant, and spill them)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// POTRF of a 32x32 tile whose lower sub-tiles are in acc (MFMA layout: [0] = (rows 0-15, cols 0-15),
// [1] = (rows 16-31, cols 0-15), [2] = (rows 16-31, cols 16-31)), eight block steps of four columns.
// sD receives L (row-major; above the diagonal: unspecified), sdi 1/diag, *prog the number of finished
// blocks; gL (owner only, else null) the copy in global memory.
// late (if any): sub-tile [2] arrives at block step 4 -- *late counts to 1 when another wave has left the
// part of it that does not depend on this factorisation in the tile's LDS home, to be added.
__device__ __forceinline__ void chol2_potrf_blocks(v4d (&acc)[3], double* sD, double* sdi, int* prog, bool& bad,
                                                   int lane, double* gL, int ld, const int* late) {
  const int i = lane & 31, j16 = lane & 15, q = lane >> 4;
  double* const rowp = sD + i * CBP;       // row-per-lane view (both half-waves alike)
  double* const op0 = sD + j16 * CBP + q;  // operand / accumulator view, rows 0..15
  double* const op1 = op0 + 16 * CBP;      //                             rows 16..31
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int c = 4 * b, g = b & 3;
    if (b == 4 && late) {
      lds_wait_ge(late, 1);
      const v4d hi4 = chol2_get(sD, 3, lane);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) acc[2][g4] += hi4[g4];
    }
    // the block's columns: accumulator layout -> LDS -> row per lane
    if (b < 4) {
      op0[c] = acc[0][g];
      op1[c] = acc[1][g];
    } else {
      op1[c] = acc[2][g];
    }
    const v2d lo = *(const v2d*)(rowp + c), hi = *(const v2d*)(rowp + c + 2);
    double v[4] = {lo.x, lo.y, hi.x, hi.y}, l[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double djj = readlane_f64(v[k], c + k);
      bad |= !(djj > 0.0);
      r[k] = rsqrt_f64_h(djj);
      // (rows above the diagonal are not zeroed: what they hold only ever reaches entries of the
      // trailing tiles and solves that belong to finished columns -- nobody reads those again)
      l[k] = v[k] * r[k];  // lane c+k: djj * r = sqrt(djj)
#pragma unroll
      for (int m = k + 1; m < 4; ++m) v[m] -= l[k] * readlane_f64(l[k], c + m);
    }
    *(v2d*)(rowp + c) = v2d{l[0], l[1]};
    *(v2d*)(rowp + c + 2) = v2d{l[2], l[3]};
    *(v2d*)(sdi + c) = v2d{r[0], r[1]};
    *(v2d*)(sdi + c + 2) = v2d{r[2], r[3]};
    lds_flag_set(prog, b + 1);
    if (gL && lane < CB) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i >= c + k) gL[(size_t)(c + k) * ld + i] = l[k];
    }
    if (b < 7) {
      // fold the block into the columns still to come
      const double L0 = op0[c], L1 = op1[c];
      if (b < 3) {
        CHOL_MFMA(acc[0], -L0, L0);
        CHOL_MFMA(acc[1], -L0, L1);
      }
      CHOL_MFMA(acc[2], -L1, L1);
    }
  }
}

// X = T L^-T for a 32x32 tile T in acc (MFMA layout, [2*ci + ri]), trailing the factorisation of L
// block by block through *prog.  sT is the tile's LDS home: scratch for the layout changes, and
// X when done (row-major); *oflag (if any) counts its finished blocks; out/stride/on: the copy in
// global memory (column c of the lane's row goes to out[c * stride]); late: see chol2_potrf_blocks
// (here sub-tiles [2] and [3], the right column half).
__device__ __forceinline__ void chol2_trsm_blocks(v4d (&acc)[4], double* sT, const double* sL, const double* sdi,
                                                  const int* prog, int* oflag, double* out, size_t stride, bool on,
                                                  int lane, const int* late) {
  const int i = lane & 31, j16 = lane & 15, q = lane >> 4;
  double* const rowp = sT + i * CBP;
  double* const op0 = sT + j16 * CBP + q;
  double* const op1 = op0 + 16 * CBP;
  const double* const lop0 = sL + j16 * CBP + q;
  const double* const lop1 = lop0 + 16 * CBP;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int c = 4 * b, g = b & 3, cb = b >> 2;
    if (b == 4 && late) {  // (as in chol2_potrf_blocks: the right column half arrives late)
      lds_wait_ge(late, 1);
      const v4d h2 = chol2_get(sT, 1, lane), h3 = chol2_get(sT, 3, lane);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        acc[2][g4] += h2[g4];
        acc[3][g4] += h3[g4];
      }
    }
    op0[c] = acc[2 * cb][g];
    op1[c] = acc[2 * cb + 1][g];
    const v2d lo = *(const v2d*)(rowp + c), hi = *(const v2d*)(rowp + c + 2);
    lds_wait_ge(prog, b + 1);
    // the 4x4 diagonal block of L and 1/diag (wave-uniform addresses: LDS broadcasts)
    const double l10 = sL[(c + 1) * CBP + c];
    const v2d l2 = *(const v2d*)(sL + (c + 2) * CBP + c);
    const v2d l3 = *(const v2d*)(sL + (c + 3) * CBP + c);
    const double l32 = sL[(c + 3) * CBP + c + 2];
    const v2d r01 = *(const v2d*)(sdi + c), r23 = *(const v2d*)(sdi + c + 2);
    double La0 = 0.0, La1 = 0.0;
    if (b < 7) {
      if (b < 3) La0 = -lop0[c];
      La1 = -lop1[c];
    }
    const double x0 = lo.x * r01.x;
    const double x1 = (lo.y - x0 * l10) * r01.y;
    const double x2 = (hi.x - x0 * l2.x - x1 * l2.y) * r23.x;
    const double x3 = (hi.y - x0 * l3.x - x1 * l3.y - x2 * l32) * r23.y;
    *(v2d*)(rowp + c) = v2d{x0, x1};
    *(v2d*)(rowp + c + 2) = v2d{x2, x3};
    if (oflag) lds_flag_set(oflag, b + 1);
    if (on) {
      out[(size_t)(c + 0) * stride] = x0;
      out[(size_t)(c + 1) * stride] = x1;
      out[(size_t)(c + 2) * stride] = x2;
      out[(size_t)(c + 3) * stride] = x3;
    }
    if (b < 7) {
      const double X0 = op0[c], X1 = op1[c];
      if (b < 3) {
        CHOL_MFMA(acc[0], La0, X0);
        CHOL_MFMA(acc[1], La0, X1);
      }
      CHOL_MFMA(acc[2], La1, X0);
      CHOL_MFMA(acc[3], La1, X1);
    }
  }
}

template <bool RHS>
__device__ __forceinline__ void chol2_panel(double* __restrict__ A, double* __restrict__ y, double* __restrict__ Arow,
                                            int ld, int k2, int* __restrict__ info, bool owner, int r0, double* sAll) {
  constexpr int T = CB * CBP;
  double* const sdi_all = sAll + C2_OFF_SDI;
  double* const sPab = sAll + C2_OFF_PAB;
  double* const sPrw = sAll + C2_OFF_PRW;
  int* const s_flag = (int*)(sAll + C2_OFF_FLAG);
  double* const sDa = sAll;          // D_aa -> L_aa   [row][col]
  double* const sTa = sAll + T;      // tile(r,a) -> X_a
  double* const sYb = sAll + 2 * T;  // tile(b,a) -> Y = L_ba
  //                  sAll + 3 * T      L_bb
  //                  sAll + 4 * T      scratch of the second solve -> X_b
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, j16 = lane & 15, q = lane >> 4;
  const int a0 = 2 * k2 * CB, b0 = a0 + CB, p0 = a0 - 2 * CB;
  const bool pend = k2 > 0;
  // Roles (wave -> SIMD is wave mod 4, and a SIMD that is folding -- back-to-back f64 MFMAs, 64
  // cycles each -- lets a co-resident wave issue one VALU instruction per MFMA: the chain (wave 0)
  // keeps SIMD 0 to itself, the two first-panel solves share SIMD 2 with D_ba folds only, which
  // they wait for anyway, and the bulk of the folding goes to SIMDs 1 and 3):
  //   0  D_aa s0, POTRF(D_aa)      1  D_aa s3, T_a s2 s3        2  D_aa s2, Y = D_ba L_aa^-T
  //   3  D_bb, U2, POTRF(D_bb)     5  T_b, U2, X_b              6  D_ba s0, X_a = T_a L_aa^-T
  //   7  D_ba s2, T_a s0 s1        9  D_ba s3                   10 D_ba s1          (4, 8: staging only)
  const int dba_s = wave == 6 ? 0 : wave == 10 ? 1 : wave == 7 ? 2 : wave == 9 ? 3 : -1;
  const int ta_s = owner ? -1 : wave == 7 ? 0 : wave == 1 ? 2 : -1;  // folds T_a sub-tiles ta_s, ta_s + 1
  auto own_tile = [&](int cb, int s) -> v4d {  // sub-tile s of the own tile row at columns cb..
    return RHS ? chol2_tile_rhs(y, cb, s, lane) : chol2_tile(Arow, ld, r0, cb, s, lane);
  };
  if (threadIdx.x <= F_TIMEOUT) s_flag[threadIdx.x] = 0;
  __syncthreads();
  if (wave == 0) C2_STAMP(0);
  const double* const opab = sPab + q * C2_PAB + j16;  // the lane's element of k-slice 0, row 0
  const double* const oprw = sPrw + q * C2_PRW + j16;
  v4d t0, t1, t2, t3;  // the tiles go straight into the accumulators of the waves that fold them
  if (wave < 3) {
    // ---- the diagonal tile's lower sub-tiles, operands straight from global memory: the chain
    // starts on them, one round trip and no hand-off
    const int s = wave == 0 ? 0 : wave == 1 ? 3 : 2, ci = s & 1, ri = s >> 1;
    const size_t st = 4 * (size_t)ld;
    __builtin_amdgcn_s_setprio(3);
    t0 = chol2_tile(A, ld, a0, a0, s, lane);
    double a[16], b[16];
    if (pend) {
      ld_strided<16>(a, A + (size_t)(p0 + q) * ld + a0 + 16 * ci + j16, st);
      ld_strided<16>(b, A + (size_t)(p0 + q) * ld + a0 + 16 * ri + j16, st);
    }
    if (ta_s >= 0) {
      t1 = own_tile(a0, ta_s);
      t2 = own_tile(a0, ta_s + 1);
    }
    // (what waves 0 and 2 fold once their first-panel work is done, see below)
    if (wave == 0 && !owner) {
      t2 = own_tile(b0, 1);  // T_b (ci 1, ri 0)
      t3 = own_tile(b0, 3);  // T_b (ci 1, ri 1)
    }
    if (wave == 0) t1 = chol2_tile(A, ld, b0, b0, 3, lane);  // D_bb (ci 1, ri 1)
    if (pend) chol2_fold_regs(t0, a, b);
    chol2_put(sDa, s, lane, t0);
    lds_flag_add(&s_flag[F_CNT_DAA], lane);
    if (wave == 0) C2_STAMP(1);
  } else {
    // ---- waves 3..10 stage the pending panels' rows once per workgroup (every other job needs two
    // of the three row blocks; loaded per job from global memory the workgroup moves 4x the bytes
    // and the vector memory pipe, not the factorisation, sets the pace).  16 bytes per lane,
    // coalesced, all loads in flight before the first LDS write.
    const int tid = threadIdx.x - 3 * 64;  // 0..511
    v2d pab[4], prw[2];
    if (pend) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + 512 * it, col = c >> 5, pr = c & 31;
        pab[it] = *(const v2d*)(A + (size_t)(p0 + col) * ld + a0 + 2 * pr);
      }
      if (RHS) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int c = tid + 512 * it, col = c >> 4, pr = c & 15;
          prw[it] = v2d{pr ? 0.0 : y[p0 + col], 0.0};
        }
      } else if (!owner) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int c = tid + 512 * it, col = c >> 4, pr = c & 15;
          prw[it] = *(const v2d*)(Arow + (size_t)(p0 + col) * ld + r0 + 2 * pr);
        }
      }
    }
    // (the tile loads are issued before the waits of the LDS writes)
    if (wave == 3) {
      t0 = chol2_tile(A, ld, b0, b0, 0, lane);
      t1 = chol2_tile(A, ld, b0, b0, 2, lane);
    } else if (wave == 5) {
      if (!owner) {
        t0 = own_tile(b0, 0);  // (ci 0, ri 0)
        t1 = own_tile(b0, 2);  // (ci 0, ri 1)
      }
    } else if (dba_s >= 0) {
      t0 = chol2_tile(A, ld, b0, a0, dba_s, lane);
      if (ta_s >= 0) {
        t1 = own_tile(a0, ta_s);
        t2 = own_tile(a0, ta_s + 1);
      }
    }
    if (pend) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + 512 * it, col = c >> 5, pr = c & 31;
        *(v2d*)(sPab + col * C2_PAB + 2 * pr) = pab[it];
      }
      if (RHS || !owner) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int c = tid + 512 * it, col = c >> 4, pr = c & 15;
          *(v2d*)(sPrw + col * C2_PRW + 2 * pr) = prw[it];
        }
      }
    }
    lds_flag_add(&s_flag[F_STAGED], lane);
    if (wave == 4 || wave == 8) return;
  }

  // ---- the folds that go through LDS tiles: D_ba (the solves spin on it), then T_a
  if (dba_s >= 0 || ta_s >= 0) {
    lds_wait_ge(&s_flag[F_STAGED], 8);
    if (dba_s >= 0) {
      __builtin_amdgcn_s_setprio(3);
      if (pend) chol2_fold(t0, opab + 16 * (dba_s & 1), C2_PAB, opab + 32 + 16 * (dba_s >> 1), C2_PAB);
      chol2_put(sYb, dba_s, lane, t0);
      lds_flag_add(&s_flag[F_CNT_DBA], lane);
#ifndef SFM_CHOL_BREAK_HANDOFF  // (diagnostic build: the hand-off never arrives, the waiting wave must report it)
      if (dba_s == 3) lds_flag_add(&s_flag[F_DBA_S3], lane);
#endif
      C2_STAMP(16 + dba_s);
    }
    if (ta_s >= 0) {
      __builtin_amdgcn_s_setprio(2);
      // (priorities do not order MFMAs: D_ba, which Y waits for, goes first on each SIMD -- wave 7
      // has just done its own, wave 1 lets wave 9 finish)
      if (wave == 1) lds_wait_ge(&s_flag[F_DBA_S3], 1);
      if (pend) {
        // (ta_s is even: the two sub-tiles are the column halves of row half ta_s >> 1)
        chol2_fold(t1, opab, C2_PAB, oprw + 16 * (ta_s >> 1), C2_PRW);
        chol2_fold(t2, opab + 16, C2_PAB, oprw + 16 * (ta_s >> 1), C2_PRW);
      }
      chol2_put(sTa, ta_s, lane, t1);
      chol2_put(sTa, ta_s + 1, lane, t2);
      if (lane == 0) __hip_atomic_fetch_add((lds_int*)&s_flag[F_CNT_TA], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      C2_STAMP(20 + ta_s);
    }
  }

  if (wave == 0 || wave == 3) {
    // ---- the factorisations: wave 0 the first panel's, wave 3 the second panel's
    const int pass = wave == 3;
    if (pass) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(3);
    v4d acc[3];
    if (pass) {
      // D_bb minus the pending panels (the three lower sub-tiles share their operands) ...
      // (only the left column half here: the factorisation needs sub-tile (1,1) four block steps
      // later, and wave 2 folds it when Y is done -- f64 MFMA throughput, 16 FMA/clock/SIMD, is
      // what the first half of the launch is short of)
      acc[0] = t0, acc[1] = t1;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[2][g] = 0.0;
      if (pend) {
        // (priorities do not order the MFMAs of co-resident waves: the first panel's folds go first)
        lds_wait_ge(&s_flag[owner ? F_CNT_DBA : F_CNT_TA], 4);
        C2_STAMP(15);
        double Lb[2][16];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) Lb[h][ks] = opab[32 + 16 * h + 4 * ks * C2_PAB];
        v4d p0 = {0.0, 0.0, 0.0, 0.0}, p1 = p0;
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
          CHOL_MFMA(acc[0], -Lb[0][ks], Lb[0][ks]);
          CHOL_MFMA(acc[1], -Lb[0][ks], Lb[1][ks]);
          CHOL_MFMA(p0, -Lb[0][ks + 1], Lb[0][ks + 1]);
          CHOL_MFMA(p1, -Lb[0][ks + 1], Lb[1][ks + 1]);
        }
        acc[0] += p0;
        acc[1] += p1;
      }
      C2_STAMP(23);
      // ... minus Y Y^T, block by block as wave 2 finishes them
      const double* const y0 = sYb + j16 * CBP + q;
      const double* const y1 = y0 + 16 * CBP;
      int seen = 0;
#pragma unroll 1
      for (int b = 0; b < 8; ++b) {
        if (seen <= b) seen = __builtin_amdgcn_readfirstlane(lds_wait_ge(&s_flag[F_YPROG], b + 1));
        const double Y0 = y0[4 * b], Y1 = y1[4 * b];
        CHOL_MFMA(acc[0], -Y0, Y0);
        CHOL_MFMA(acc[1], -Y0, Y1);
      }
      __builtin_amdgcn_s_setprio(3);
      C2_STAMP(3);
    } else {
      lds_wait_ge(&s_flag[F_CNT_DAA], 3);
      acc[0] = chol2_get(sDa, 0, lane);
      acc[1] = chol2_get(sDa, 2, lane);
      acc[2] = chol2_get(sDa, 3, lane);
    }
    bool bad = false;
    const int c0 = pass ? b0 : a0;
    chol2_potrf_blocks(acc, sAll + opaque(pass * 3 * T), sdi_all + opaque(pass * CB), s_flag + opaque(F_PROG_A + pass),
                       bad, lane, nullptr, ld, pass ? &s_flag[F_DBB_HI] : nullptr);
    C2_STAMP(pass ? 4 : 2);
    if (owner && bad && lane == 0) atomicExch(info, c0 + 1);  // the host discards the step
    if (!pass) {
      // ---- wave 0, its factorisation done (SIMD 0 has nothing else to do, and folding anywhere else
      // would slow a solve down): sub-tile (1,1) of D_bb (minus the pending panels, minus Y Y^T),
      // left in L_bb's LDS home for wave 3 to pick up at block step 4
      __builtin_amdgcn_s_setprio(1);
      if (pend) {
        double Lb[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) Lb[ks] = opab[32 + 16 + 4 * ks * C2_PAB];
        chol2_fold_regs(t1, Lb, Lb);
      }
      lds_wait_ge(&s_flag[F_YPROG], 8);
      const double* const yh = sYb + (16 + j16) * CBP + q;
      v4d p0 = {0.0, 0.0, 0.0, 0.0}, p1 = p0, p2 = p0;
#pragma unroll
      for (int b = 0; b < 8; b += 4) {
        const double Ya = yh[4 * b], Yb = yh[4 * b + 4], Yc = yh[4 * b + 8], Yd = yh[4 * b + 12];
        CHOL_MFMA(t1, -Ya, Ya);
        CHOL_MFMA(p0, -Yb, Yb);
        CHOL_MFMA(p1, -Yc, Yc);
        CHOL_MFMA(p2, -Yd, Yd);
      }
      t1 += (p0 + p1) + p2;
      chol2_put(sAll + 3 * T, 3, lane, t1);
      lds_flag_add(&s_flag[F_DBB_HI], lane);
      C2_STAMP(14);
    }
    if (!pass && !owner) {
      // ---- ... then the right column half of T_b (minus the pending panels, minus X_a Y^T), left in
      // the second solve's LDS home for wave 5 to pick up at block step 4
      if (pend) {
        double Lb[16], Lr[2][16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          Lb[ks] = opab[32 + 16 + 4 * ks * C2_PAB];
          Lr[0][ks] = oprw[4 * ks * C2_PRW];
          Lr[1][ks] = oprw[16 + 4 * ks * C2_PRW];
        }
        v4d p0 = {0.0, 0.0, 0.0, 0.0}, p1 = p0;
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
          CHOL_MFMA(t2, -Lb[ks], Lr[0][ks]);
          CHOL_MFMA(t3, -Lb[ks], Lr[1][ks]);
          CHOL_MFMA(p0, -Lb[ks + 1], Lr[0][ks + 1]);
          CHOL_MFMA(p1, -Lb[ks + 1], Lr[1][ks + 1]);
        }
        t2 += p0;
        t3 += p1;
      }
      const double* const y1 = sYb + (16 + j16) * CBP + q;
      const double* const x0 = sTa + j16 * CBP + q;
      int seen_y = 0, seen_x = 0;
#pragma unroll 1
      for (int b = 0; b < 8; ++b) {
        if (seen_y <= b) seen_y = __builtin_amdgcn_readfirstlane(lds_wait_ge(&s_flag[F_YPROG], b + 1));
        if (seen_x <= b) seen_x = __builtin_amdgcn_readfirstlane(lds_wait_ge(&s_flag[F_XPROG], b + 1));
        const double Y1 = -y1[4 * b], X0 = x0[4 * b], X1 = x0[16 * CBP + 4 * b];
        CHOL_MFMA(t2, Y1, X0);
        CHOL_MFMA(t3, Y1, X1);
      }
      chol2_put(sAll + 4 * T, 1, lane, t2);
      chol2_put(sAll + 4 * T, 3, lane, t3);
      lds_flag_add(&s_flag[F_TB_HI], lane);
      C2_STAMP(13);
    }
  } else if (wave == 2 || wave == 6 || wave == 5) {
    // ---- the solves: wave 2 tile (b,a) against L_aa (-> Y), wave 6 the own tile against L_aa, wave 5
    // the own tile against L_bb after U2
    const int pass = wave == 5;
    if (pass) __builtin_amdgcn_s_setprio(0);
    else if (wave == 2) __builtin_amdgcn_s_setprio(3);  // Y gates the second factorisation
    else __builtin_amdgcn_s_setprio(2);
    if (owner && wave != 2) return;  // (the owner has no tile row of its own: rows a, b are the panels)
    v4d acc[4];
    if (pass) {
      // T_b minus the pending panels ...
      // (the left column half; wave 0 folds the right one when its factorisation is done)
      acc[0] = t0, acc[1] = t1;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[2][g] = acc[3][g] = 0.0;
      if (pend) {
        lds_wait_ge(&s_flag[F_CNT_TA], 4);  // (as for D_bb)
        double Lb[16], Lr[2][16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          Lb[ks] = opab[32 + 4 * ks * C2_PAB];
          Lr[0][ks] = oprw[4 * ks * C2_PRW];
          Lr[1][ks] = oprw[16 + 4 * ks * C2_PRW];
        }
        v4d p0 = {0.0, 0.0, 0.0, 0.0}, p1 = p0;
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
          CHOL_MFMA(acc[0], -Lb[ks], Lr[0][ks]);
          CHOL_MFMA(acc[1], -Lb[ks], Lr[1][ks]);
          CHOL_MFMA(p0, -Lb[ks + 1], Lr[0][ks + 1]);
          CHOL_MFMA(p1, -Lb[ks + 1], Lr[1][ks + 1]);
        }
        acc[0] += p0;
        acc[1] += p1;
      }
      // ... minus X_a Y^T, block by block as waves 6 and 2 finish them
      const double* const y0 = sYb + j16 * CBP + q;
      const double* const x0 = sTa + j16 * CBP + q;
      int seen_y = 0, seen_x = 0;
#pragma unroll 1
      for (int b = 0; b < 8; ++b) {
        if (seen_y <= b) seen_y = __builtin_amdgcn_readfirstlane(lds_wait_ge(&s_flag[F_YPROG], b + 1));
        if (seen_x <= b) seen_x = __builtin_amdgcn_readfirstlane(lds_wait_ge(&s_flag[F_XPROG], b + 1));
        const double Y0 = -y0[4 * b], X0 = x0[4 * b], X1 = x0[16 * CBP + 4 * b];
        CHOL_MFMA(acc[0], Y0, X0);
        CHOL_MFMA(acc[1], Y0, X1);
      }
      __builtin_amdgcn_s_setprio(2);
    } else {
      const double* src = wave == 2 ? sYb : sTa;
      lds_wait_ge(s_flag + opaque(wave == 2 ? F_CNT_DBA : F_CNT_TA), 4);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[2 * (s & 1) + (s >> 1)] = chol2_get(src, s, lane);
    }
    const int c0 = pass ? b0 : a0;
    double* out;  // base and column stride of the copy in global memory
    size_t stride;
    bool on = lane < CB;
    if (wave == 2) {
      out = A + (size_t)a0 * ld + b0 + i, stride = ld, on = false;  // (Y stays in LDS, see chol_step2)
    } else if (RHS) {
      out = y + c0, stride = 1, on = lane == 0;
    } else {
      out = Arow + (size_t)c0 * ld + r0 + i, stride = ld;
    }
    C2_STAMP(wave == 6 ? 6 : wave == 2 ? 11 : 9);
    chol2_trsm_blocks(acc, sAll + opaque(wave == 2 ? 2 * T : T + pass * 3 * T), sAll + opaque(pass * 3 * T),
                      sdi_all + opaque(pass * CB), s_flag + opaque(F_PROG_A + pass),
                      wave == 5 ? nullptr : s_flag + opaque(wave == 2 ? F_YPROG : F_XPROG), out, stride, on, lane,
                      pass && !owner ? &s_flag[F_TB_HI] : nullptr);
    C2_STAMP(wave == 6 ? 7 : wave == 2 ? 12 : 10);
  }
}

__device__ __forceinline__ void chol2_report_timeout(double* sAll, int* __restrict__ info) {
  const int* s_flag = (const int*)(sAll + C2_OFF_FLAG);
  if (*(volatile const lds_int*)(s_flag + F_TIMEOUT) != 0 && (threadIdx.x & 63) == 0) atomicExch(info, -1);
}

// (bid: the workgroup's index within its matrix -- blockIdx.x when a launch factors one matrix)
// (nxc: the tile columns of X that are wanted -- nt for a whole matrix, the interior tiles for a chain, whose X is
// only used up to there: the trailing tiles of X right of that are not formed)
// Deferred trailing updates (dfr = D > 1, large matrices): a launch rewrites every trailing tile for 64 columns of update, and at
// 100+ tile rows that traffic -- not the MFMAs -- is what a launch waits for.  A tile column is only needed up to date when it becomes
// the panel, so launch k2 touches the columns whose distance to the panels, counted in column pairs, is a multiple of D, and folds
// the min(D, k2) pairs of panels they have missed (K = 64 D per visit): the columns next to the panels are among them every time.
// chol_trail_tiles: the tiles of those columns (tc counted from the first column right of the panels), column by column.
__host__ __device__ inline int chol_trail_tiles(int m2, int D) {
  int n = 0;
  for (int tc = 0; tc < m2; tc += (tc & 1) ? 2 * D - 1 : 1) n += m2 - tc;
  return n;
}

// one trailing tile (t: its index in the launch's list -- the tiles of S, the rhs row, the tiles of X), one wave
__device__ __forceinline__ void chol2_trailing_tile(double* __restrict__ A, double* __restrict__ y, double* __restrict__ X, const int ld,
                                                    const int k2, const int m2, const int xlo, const int mx, const int ntile,
                                                    const int dfr, int t, const int lane, const int catchup = 0) {
  const int ntrail = ntile + m2;
  int ti_rel = 0;
  const double* Rrow = A;  // the tile's row space
  double* Wrow = A;
  int rb_x = -1;
  int npend = 1;  // pairs of panels to fold
  if (t >= ntrail) {
    t -= ntrail;
    rb_x = (xlo + t / mx) * CB;
    t %= mx;
    Rrow = X;
    Wrow = X;
  } else if (dfr > 1) {
    if (t >= ntile) {
      t -= ntile;  // the rhs row: every column, every launch
      ti_rel = m2;
    } else {
      int tc = 0;
      while (t >= m2 - tc) {
        t -= m2 - tc;
        tc += (tc & 1) ? 2 * dfr - 1 : 1;
      }
      ti_rel = tc + t;
      t = tc;
      npend = min(dfr, k2);
    }
  } else {
    while (true) {
      const int w = ti_rel < m2 ? ti_rel + 1 : m2;
      if (t < w) break;
      t -= w;
      ++ti_rel;
    }
    // (the launch that ends a run of deferred updates visits every column and folds what each has missed: a column whose
    // distance to the panels is r pairs was last visited when that distance was the next multiple of `catchup` above r)
    if (catchup > 1 && ti_rel < m2) npend = min(catchup - (t / 2) % catchup, k2);
  }
  const int c0 = (2 * k2 + 2) * CB;
  const int rb = rb_x >= 0 ? rb_x : c0 + ti_rel * CB, cb = c0 + t * CB;
  int p0 = (2 * k2 - 2 * npend) * CB;
  const int j16 = lane & 15, q = lane >> 4;
  double a[2][16], b[2][16];
  v4d acc[4];  // [2 * ci + ri]
  if (rb_x < 0 && ti_rel == m2) {
    // the rhs row: y[cb..] -= y[pending] L(cb.., pending)^T, tile row 0 only
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int kk = p0 + 4 * ks + q;
      a[0][ks] = -A[(size_t)kk * ld + cb + j16];
      a[1][ks] = -A[(size_t)kk * ld + cb + 16 + j16];
      b[0][ks] = j16 == 0 ? y[kk] : 0.0;
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[ci][g] = j16 == 0 ? y[cb + 16 * ci + q + 4 * g] : 0.0;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      CHOL_MFMA(acc[0], a[0][ks], b[0][ks]);
      CHOL_MFMA(acc[1], a[1][ks], b[0][ks]);
    }
    if (j16 == 0) {
#pragma unroll
      for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int g = 0; g < 4; ++g) y[cb + 16 * ci + q + 4 * g] = acc[ci][g];
    }
    return;
  }
  const size_t st = 4 * (size_t)ld;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
      double t4[4];
      ld_strided<4>(t4, Rrow + (size_t)(cb + 16 * ci + q) * ld + rb + 16 * ri + j16, st);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[2 * ci + ri][g] = t4[g];
    }
#pragma unroll 1
  for (; npend > 0; --npend, p0 += 2 * CB) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ld_strided<16>(a[h], A + (size_t)(p0 + q) * ld + cb + 16 * h + j16, st);
      ld_strided<16>(b[h], Rrow + (size_t)(p0 + q) * ld + rb + 16 * h + j16, st);
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      CHOL_MFMA(acc[0], -a[0][ks], b[0][ks]);
      CHOL_MFMA(acc[1], -a[0][ks], b[1][ks]);
      CHOL_MFMA(acc[2], -a[1][ks], b[0][ks]);
      CHOL_MFMA(acc[3], -a[1][ks], b[1][ks]);
    }
  }
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
      gbl_double* pt = (gbl_double*)(Wrow + (size_t)(cb + 16 * ci + q) * ld + rb + 16 * ri + j16);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *pt = acc[2 * ci + ri][g];
        pt += st;
        asm volatile("" : "+v"(pt));
      }
    }
}

// (xb: 0, or the width in tile columns (even) of the diagonal blocks of X that are wanted: the inverse of a diagonal block of L
// is made of that block alone, so the rows and columns of X outside it need not be formed -- chol_back_block)
__device__ __forceinline__ void chol_step2_body(double* __restrict__ A, double* __restrict__ y, double* __restrict__ X,
                                                int ld, int nt, int nxc, int k2, int tiles_per_wg, int* __restrict__ info,
                                                const int bid, double* sAll, const int xb = 0, const int dfr = 1, const int catchup = 0) {
  const int m2 = nt - 2 * k2 - 2;  // tile rows below the two panels (the rhs row comes on top)
  const int npanel = m2 + 2;       // owner, m2 tile rows, rhs
  const int xlo = xb ? 2 * k2 / xb * xb : 0;  // first row block of X that takes part
  if (xb) nxc = min(nxc, xlo + xb);
  const int nx = 2 * k2 + 2 - xlo;  // rows xlo..b of the identity block X (see below) take part as tile rows
  if (bid < npanel + nx) {
    if (bid >= npanel) {
      // X starts as the identity and rides along as nt more tile rows: X <- X L^-T, i.e. L^-T when the
      // factorisation ends, and the backward substitution becomes the product z = X y.  (Row block
      // r' is all zero left of column block r' and untouched until its own panel: rows 0..b here.)
      chol2_panel<false>(A, y, X, ld, k2, info, false, (xlo + bid - npanel) * CB, sAll);
      chol2_report_timeout(sAll, info);
      return;
    }
    const bool owner = bid == 0;
    const int r0 = (2 * k2 + 1 + bid) * CB;
    if (bid == npanel - 1)
      chol2_panel<true>(A, y, A, ld, k2, info, false, r0, sAll);
    else
      chol2_panel<false>(A, y, A, ld, k2, info, owner, r0, sAll);
    chol2_report_timeout(sAll, info);
    return;
  }
  // ---- trailing tiles (k2 >= 1), one wave each (its four sub-tiles share the operands): tile row
  // ti_rel in [0, m2] (m2 = rhs) has min(ti_rel + 1, m2) tiles; then the tiles of X: rows 0..a-1 (the
  // rows that are nonzero in the pending panels) x the m2 column blocks right of the panels
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ntile = dfr > 1 ? chol_trail_tiles(m2, dfr) : m2 * (m2 + 1) / 2;
  const int mx = max(0, nxc - 2 * k2 - 2);  // column blocks of X right of the panels
  const int total = ntile + m2 + (2 * k2 - xlo) * mx;
  // (four tiles per workgroup -- one wave per SIMD: a tile is 64 MFMAs -- while that fits one round of
  // workgroups on the device; every workgroup of this kernel holds a CU's LDS)
  if (wave >= tiles_per_wg) return;
  const int t = (bid - npanel - nx) * tiles_per_wg + wave;
  if (t >= total) return;
  chol2_trailing_tile(A, y, X, ld, k2, m2, xlo, mx, ntile, dfr, t, lane, catchup);
}

__global__ __launch_bounds__(C2_WAVES * 64) void chol_step2(double* __restrict__ A, double* __restrict__ y,
                                                            double* __restrict__ X, int ld, int nt, int k2,
                                                            int tiles_per_wg, int* __restrict__ info, int xb, int dfr, int catchup) {
  extern __shared__ __attribute__((aligned(16))) double sAll[];  // C2_LDS_BYTES, see C2_OFF_*
  chol_step2_body(A, y, X, ld, nt, nt, k2, tiles_per_wg, info, (int)blockIdx.x, sAll, xb, dfr, catchup);
}

// Several independent matrices ("chains": the interiors of a dissected camera graph, see NdPlan) at the same
// panel pair k2 in ONE launch.  ALL panel workgroups come first in the grid, the trailing workgroups after them
// (the panel workgroups are the long ones: they start in the first round; the host keeps their number within the
// device's CUs).
constexpr int ND_MAX = 8;
struct ChainSet {
  int n;
  double* A[ND_MAX];
  double* y[ND_MAX];
  double* X[ND_MAX];
  int ld[ND_MAX], nt[ND_MAX], nxc[ND_MAX], tpw[ND_MAX];
  int pan0[ND_MAX + 1];  // first panel workgroup of chain c (pan0[n]: all panel workgroups)
  int trl0[ND_MAX + 1];  // first trailing workgroup of chain c, counted from pan0[n]
};
__global__ __launch_bounds__(C2_WAVES * 64) void chol_step2_chains(ChainSet cs, int k2, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sAll[];
  int c = 0, bid;
  if ((int)blockIdx.x < cs.pan0[cs.n]) {
    while (c + 1 < cs.n && (int)blockIdx.x >= cs.pan0[c + 1]) ++c;
    bid = (int)blockIdx.x - cs.pan0[c];
  } else {
    const int t = (int)blockIdx.x - cs.pan0[cs.n];
    while (c + 1 < cs.n && t >= cs.trl0[c + 1]) ++c;
    bid = cs.pan0[c + 1] - cs.pan0[c] + (t - cs.trl0[c]);
  }
  chol_step2_body(cs.A[c], cs.y[c], cs.X[c], cs.ld[c], cs.nt[c], cs.nxc[c], k2, cs.tpw[c], info, bid, sAll);
}
#undef CHOL_MFMA

// z = L^-T y = X y.  X (column-major like A: X(i, j) = X[j*ld + i]) is upper triangular by 32x32
// tiles; one workgroup per tile row: thread (row i of the tile, one of 32 column slices) sums its columns in
// ascending order, the slices are added through LDS in slice order -- no atomics: the same S and g give the same z
// bit for bit, run after run and rank after rank (the replicated cameras of the sharded solver stay identical).
__global__ __launch_bounds__(1024) void chol_apply_inverse(const double* __restrict__ X, const double* __restrict__ y,
                                                          double* __restrict__ z, int ld, int nt) {
  __shared__ double s_part[32][CB + 1];
  const int tr = blockIdx.x;
  const int i = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const double* xr = X + tr * CB + i;
  double acc = 0.0;
  for (int j = tr * CB + sl; j < nt * CB; j += 32) acc += xr[(size_t)j * ld] * y[j];
  s_part[sl][i] = acc;
  __syncthreads();
  if (sl == 0) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) v += s_part[k][i];
    z[tr * CB + i] = v;
  }
}

// ---------------------------------------------------------------- backward substitution by diagonal blocks
// Large dense systems (no dissection, nt >= DENSE_XB_MIN_NT -- 250 cameras: +3 % there, +5 % at 330, +55 % at 640): the rows of X outside the diagonal blocks of xb tile columns are
// half of all trailing tiles of the factorisation and buy only the one-product backward substitution; with X kept inside the
// diagonal blocks (chol_step2_body, xb) the substitution walks the blocks from the last to the first, one launch each:
//   z_K = (L_KK)^-T w_K,   w_J -= L(K, J)^T z_K for every block J < K       (w = y on entry)
// Launch K applies z_{K+1} to every tile column left of block K+1 (one workgroup per tile column, sums in a fixed order) and the
// workgroups of block K's own columns then write their part X(., j) w_j of z_K; the parts are added, in column order, by every
// workgroup of the next launch (one more launch for z_0): no counters, and the same S and g give the same z bit for bit.
constexpr int DENSE_XB = 8, DENSE_XB_MIN_NT = 48, DENSE_DEFER4_MIN_NT = 100, DENSE_SWITCH_M2 = 40;
__global__ __launch_bounds__(256) void chol_x_reset(double* __restrict__ X, int ld, int nt, int xb) {
  const int j = blockIdx.x, r0 = j / xb * xb * CB, r1 = min(nt, (j / xb + 1) * xb) * CB;
  for (int e = threadIdx.x; e < CB * (r1 - r0); e += 256) {
    const int c = e / (r1 - r0), r = e % (r1 - r0);
    X[(size_t)(j * CB + c) * ld + r0 + r] = 0.0;
  }
}

__global__ __launch_bounds__(256) void chol_back_block(const double* __restrict__ A, const double* __restrict__ X,
                                                       double* __restrict__ w, double* __restrict__ z, int ld, int nt, int xb, int K,
                                                       const double* __restrict__ part_in, double* __restrict__ part_out) {
  __shared__ double s_z[DENSE_XB * CB];
  __shared__ double s_w[CB];
  __shared__ double s_red[CB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nk = K >= 0 ? min(xb, nt - K * xb) : 0;  // tile columns of block K
  // the workgroups of block K's columns come first: tile column j
  const bool own = (int)blockIdx.x < nk;
  const int j = own ? K * xb + (int)blockIdx.x : (int)blockIdx.x - nk;
  const int rb = (K + 1) * xb * CB, nkn = max(0, min(xb, nt - (K + 1) * xb)), nr = nkn * CB;  // block K+1: first row, tile columns, rows
  // every load first (none depends on another): L(rows of block K+1, the column's 32), the column's part of X_KK, w_j, the parts of z_{K+1}
  double a[8][4], x[CB], wj = 0.0, zt = 0.0;
  if (K >= 0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const double* col = A + (size_t)(j * CB + 8 * wave + c) * ld + rb;
#pragma unroll
      for (int m = 0; m < 4; ++m) a[c][m] = lane + 64 * m < nr ? col[lane + 64 * m] : 0.0;
    }
    const int jj = (int)blockIdx.x, i0 = K * xb * CB;
#pragma unroll
    for (int c = 0; c < CB; ++c) x[c] = own && t < (jj + 1) * CB ? X[(size_t)(j * CB + c) * ld + i0 + t] : 0.0;
    if (t < CB) wj = w[j * CB + t];
  }
  if (t < nr) {
    // z_{K+1} = the sum of its columns' parts (written by the launch before), in column order: every workgroup forms it
#pragma unroll
    for (int q = 0; q < DENSE_XB; ++q)
      if (q >= t / CB && q < nkn) zt += part_in[q * (DENSE_XB * CB) + t];
    if (blockIdx.x == 0) z[rb + t] = zt;
  }
  if (K < 0) return;
  s_z[t] = zt;
  __syncthreads();
  if (nr > 0) {
    // lanes along the rows (contiguous in memory), a wave per eight columns
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      acc[c] = 0.0;
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[c] += a[c][m] * s_z[lane + 64 * m];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) s_red[8 * wave + c] = acc[c];
    }
    __syncthreads();
    if (t < CB) wj -= s_red[t];
  }
  if (!own) {
    if (t < CB && nr > 0) w[j * CB + t] = wj;
    return;
  }
  // ---- block K's own columns: this column's part of z_K = X_KK w_K, the rows of the block up to the column's own tile
  if (t < CB) s_w[t] = wj;
  __syncthreads();
  double p = 0.0;
#pragma unroll
  for (int c = 0; c < CB; ++c) p += x[c] * s_w[c];
  part_out[(int)blockIdx.x * (DENSE_XB * CB) + t] = p;
}

// ---------------------------------------------------------------- dissected reduced system
// S is dense by storage only: its block pattern is the camera co-visibility graph.  When that graph has small
// vertex separators (an ordered capture: cfg4's ring, banded plus a cyclic corner), the cameras are ordered
// [interior 1 | interior 2 | ... | separator + focal] with no edge between two interiors (NdPlan, host), and
//   [ D_1          C_1^T ]        L_ii = chol(D_i),  L_Si = C_i L_ii^-T   (P independent "chains", one launch
//   [      D_2     C_2^T ]                                                 of chol_step2_chains per panel pair)
//   [ C_1  C_2 ... D_S   ]        D_S' = D_S - sum_i L_Si L_Si^T,  L_SS = chol(D_S')
// The dependency chain is max_i |interior i| + |separator| columns instead of all of them.  Every chain is a
// dense square matrix of its own, M_i = [D_i, . ; C_i, 0] (interior tiles, then the separator's), on which the
// dense kernel runs its first |interior i| / 64 launches: what is then in the lower right block is chain i's
// part of the Schur complement (but for the last launch's pending panels, nd_combine adds both), its rhs row
// holds y_i = L_ii^-1 g_i and -L_Si y_i, its X holds L_ii^-T.  Solution: z_S = L_SS^-T y_S,
// z_i = L_ii^-T (y_i - L_Si^T z_S).
struct NdChain {
  double* M;       // (32 N)^2 column-major lower
  double* y;       // 32 N
  double* X;       // (32 N)^2
  const int* inv;  // 32 N: chain index -> index in S (the parameter's column), -1 = padding
  int ld, ni, N;   // 32 N; interior tiles (the separator: all of them); tiles
};
struct NdSet {
  int n;  // chains; c[n] is the separator
  NdChain c[ND_MAX + 1];
};

// fills the chains from S (after ba_finalize: the LM diagonal is on it), g and the identity.
// job = (chain, tile row, tile column, kind): kind 0 a 32x32 tile of M, kind 1 tile row of y and of X's diagonal
// fin != 0: ba_finalize's work rides along (one launch less per LM iteration) -- the LM diagonal is added to the
// diagonal elements as the tiles are copied (S itself stays undamped), and one more workgroup (job kind 4) leaves the
// clamped column norms, the gradient maximum and a zeroed z.
constexpr int ND_ZERO_SLICE = 8192;  // doubles per workgroup of nd_gather's zeroing role
__global__ __launch_bounds__(256) void nd_gather(NdSet ns, const int4* __restrict__ jobs, int n_jobs, const double* __restrict__ S,
                                                 const double* __restrict__ g, int ldS, BaDev d, int fin, double radius,
                                                 double lm_lo, double lm_hi, int world, double* __restrict__ zero_ptr, long long zero_n) {
  if ((int)blockIdx.x >= n_jobs) {
    // the workgroups behind the jobs zero the OTHER reduced-system buffer, slice by slice: the next linearisation starts on it
    // without a memset of its own (an 11.8 MB fill is 5 us as a launch in the stream, nothing beside this kernel's jobs)
    const long long lo = (long long)((int)blockIdx.x - n_jobs) * ND_ZERO_SLICE;
    double2* p2 = (double2*)(zero_ptr + lo);
    const long long n2 = (zero_n - lo < ND_ZERO_SLICE ? zero_n - lo : ND_ZERO_SLICE) / 2;
    for (long long i = threadIdx.x; i < n2; i += 256) p2[i] = make_double2(0.0, 0.0);
    return;
  }
  if (d.lm) radius = d.lm->radius;
  const int4 job = jobs[blockIdx.x];
  if (job.w == 4) {
    if (!fin) return;
    __shared__ double shm[4];
    const double* gF = red_gF(d);
    const double* dc = red_dc(d);
    double* scv = red_sc(d);
    double gm = 0;
    for (int i = threadIdx.x; i < d.dim; i += 256) {
      d.diag[i] = fmin(fmax(dc[i], lm_lo), lm_hi);
      const double sc = i < 6 * d.nc ? d.scale_c[i] : *d.scale_f;
      gm = fmax(gm, fabs(gF[i] / sc));
    }
    for (int i = threadIdx.x; i < d.ld; i += 256) d.z[i] = 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_down(gm, o));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0) {
      gm = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
      for (int r = 0; r < world; ++r) gm = fmax(gm, scv[SC + r]);
      scv[3] = gm;  // gradient max norm (unscaled), all parameter blocks, all ranks
    }
    return;
  }
  const NdChain ch = ns.c[job.x];
  const int tr = job.y, tc = job.z;
  if (job.w == 1) {
    if (threadIdx.x < 32) {
      const int idx = tr * 32 + threadIdx.x;
      const int l = ch.inv[idx];
      ch.y[idx] = (tr < ch.ni && l >= 0) ? g[l] : 0.0;
    }
    return;
  }
  if (job.w >= 2) {  // 2: a zero tile of M (the chain's part of the Schur complement starts at 0); 3: a tile of X = I
    double* dst = (job.w == 2 ? ch.M : ch.X) + (size_t)(tc * 32) * ch.ld + tr * 32;
    const int r_ = threadIdx.x & 31;
    for (int j = threadIdx.x >> 5; j < 32; j += 8) dst[(size_t)j * ch.ld + r_] = (job.w == 3 && tr == tc && j == r_) ? 1.0 : 0.0;
    return;
  }
  __shared__ double sh[32][33];
  const int r = threadIdx.x & 31, cg = threadIdx.x >> 5;
  // S holds its upper triangle row-major: element (a, b), a <= b, at S[a*ldS + b].  Read with the lanes along
  // the contiguous index, transpose through LDS where the tile lies on the other side.
  const int lr_lane = ch.inv[tr * 32 + r], lc_lane = ch.inv[tc * 32 + r];
  for (int j = cg; j < 32; j += 8) {
    // lanes along the tile's rows: value (row r, column j)
    const int lc = ch.inv[tc * 32 + j];
    double v = 0.0;
    if (lr_lane >= 0 && lc >= 0) {
      if (lr_lane >= lc) v = S[(size_t)lc * ldS + lr_lane];
      if (fin && lr_lane == lc) v += fmin(fmax(red_dc(d)[lc], lm_lo), lm_hi) / radius;
    } else if (lr_lane < 0 && lc < 0 && tr == tc && r == j) v = 1.0;
    sh[j][r] = v;
  }
  __syncthreads();
  for (int j = cg; j < 32; j += 8) {
    // lanes along the tile's columns: value (row j, column r), for the pairs stored the other way round
    const int lr = ch.inv[tr * 32 + j];
    if (lr >= 0 && lc_lane >= 0 && lr < lc_lane) sh[r][j] = S[(size_t)lr * ldS + lc_lane];
  }
  __syncthreads();
  for (int j = cg; j < 32; j += 8) ch.M[(size_t)(tc * 32 + j) * ch.ld + tr * 32 + r] = sh[j][r];
}

// D_S' and its rhs: what the chains accumulated in their lower right blocks, and the rank-64 update of each
// chain's last launch that nothing has folded in yet.  One workgroup per tile (tr, tc) of the separator's lower
// triangle (tr == NS: the rhs), one wave per chain: the tile and the pending panels' rows go straight from global
// memory into the MFMA operand layout (as in chol_step2's trailing tiles), the waves' tiles are summed through LDS.
__global__ __launch_bounds__(256) void nd_combine(NdSet ns) {
  const NdChain sp = ns.c[ns.n];
  int tr = 0, t = blockIdx.x;  // -> (tr, tc): rows 0..NS-1 of the lower triangle, then the NS rhs jobs
  while (tr < sp.N && t > tr) {
    t -= tr + 1;
    ++tr;
  }
  const int tc = t;
  __shared__ double s_red[4][16][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j16 = lane & 15, q = lane >> 4;
  if (tr == sp.N) {
    // the rhs row: y_S[c] += y_i[o + c] - sum_k y_i[k0 + k] L_i(o + c, k0 + k); lane = (c, half of the 64 k)
    double acc = 0.0;
    const int c = lane & 31, kh = lane >> 5;
    for (int ci = wave; ci < ns.n; ci += 4) {
      const NdChain ch = ns.c[ci];
      const int o = ch.ni * 32, k0 = o - 64 + 32 * kh;
      double lv[32], yv[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        lv[k] = ch.M[(size_t)(k0 + k) * ch.ld + o + tc * 32 + c];
        yv[k] = ch.y[k0 + k];
      }
      if (kh == 0) acc += ch.y[o + tc * 32 + c];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc -= yv[k] * lv[k];
    }
    acc += __shfl_down(acc, 32);
    if (lane < 32) s_red[wave][0][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 32) sp.y[tc * 32 + c] += s_red[0][0][c] + s_red[1][0][c] + s_red[2][0][c] + s_red[3][0][c];
    return;
  }
  v4d acc[4];  // [2 * ci + ri]: 16x16 sub-tiles, lane = row j16 of the sub-tile, registers = columns q + 4g
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
  // (wave 0 adds the sums to the separator's tile at the end: its values are asked for now, not after the reduction)
  double spv[16];
  if (wave == 0) {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          spv[4 * (2 * c2 + ri) + g] = sp.M[(size_t)(tc * 32 + 16 * c2 + q + 4 * g) * sp.ld + tr * 32 + 16 * ri + j16];
  }
  for (int ci = wave; ci < ns.n; ci += 4) {
    const NdChain ch = ns.c[ci];
    const int o = ch.ni * 32, p0 = o - 64, rb = o + tr * 32, cb = o + tc * 32;
    const size_t st = 4 * (size_t)ch.ld;
    double a[2][16], bb[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ld_strided<16>(a[h], ch.M + (size_t)(p0 + q) * ch.ld + cb + 16 * h + j16, st);
      ld_strided<16>(bb[h], ch.M + (size_t)(p0 + q) * ch.ld + rb + 16 * h + j16, st);
    }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri) {
        double t4[4];
        ld_strided<4>(t4, ch.M + (size_t)(cb + 16 * c2 + q) * ch.ld + rb + 16 * ri + j16, st);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[2 * c2 + ri][g] += t4[g];
      }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[0][ks], bb[0][ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[0][ks], bb[1][ks], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[1][ks], bb[0][ks], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[1][ks], bb[1][ks], acc[3], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) s_red[wave][4 * i + g][lane] = acc[i][g];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int e = 4 * (2 * c2 + ri) + g;
          double* pt = sp.M + (size_t)(tc * 32 + 16 * c2 + q + 4 * g) * sp.ld + tr * 32 + 16 * ri + j16;
          *pt = spv[e] + (s_red[0][e][lane] + s_red[1][e][lane] + s_red[2][e][lane] + s_red[3][e][lane]);
        }
  }
}

// The three-step solve z_S = L_SS^-T y_S; w_i = y_i - L_Si^T z_S; z_i = L_ii^-T w_i as three small launches, each
// one global round trip deep (a fused kernel had to recompute z_S and most of w in every workgroup: 35 MB of
// reads, 23 us; these three: 10 us).
// nd_xy: out = X y over the interior tiles of chains [c_lo, c_lo + gridDim.y): workgroup = 32 rows, thread =
// (row, one of 32 column slices); X(i, j) = X[j*ld + i] upper triangular.  The result goes to z where the
// parameter lives in S's index space and, for the separator, to its zs in chain order.
__global__ __launch_bounds__(1024) void nd_xy(NdSet ns, int c_lo, double* __restrict__ z) {
  __shared__ double s_red[32][33];
  const NdChain ch = ns.c[c_lo + blockIdx.y];
  const int tr = blockIdx.x;
  if (tr >= ch.ni) return;
  const int n = ch.ni * 32, r = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double a = 0.0;
  for (int jb = tr * 32; jb < n; jb += 256) {
    double x[8], yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = jb + sl + 32 * u, jj = j < n ? j : n - 1;
      x[u] = ch.X[(size_t)jj * ch.ld + tr * 32 + r];
      yv[u] = ch.y[jj] * (j < n ? 1.0 : 0.0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) a += x[u] * yv[u];
  }
  s_red[sl][r] = a;
  __syncthreads();
  if (threadIdx.x < 32) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) v += s_red[k][threadIdx.x];
    const int l = ch.inv[tr * 32 + threadIdx.x];
    if (l >= 0) z[l] = v;
    if (c_lo == ns.n) ch.M[tr * 32 + threadIdx.x] = v;  // z_S in chain order: column 0 of L_SS is no longer needed
  }
}
// nd_w: y_i[c] -= sum_s L_i(o + s, c) z_S[s]; a wave per interior column (flat index over the chains)
struct NdCols {
  int col0[ND_MAX + 1];
};
__global__ __launch_bounds__(256) void nd_w(NdSet ns, NdCols cols) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + wave;
  if (col >= cols.col0[ns.n]) return;
  int c = 0;
  while (col >= cols.col0[c + 1]) ++c;
  const NdChain ch = ns.c[c], sp = ns.c[ns.n];
  const int k = col - cols.col0[c], nS = sp.N * 32;
  const double* Lc = ch.M + (size_t)k * ch.ld + ch.ni * 32;
  const double yk = ch.y[k];
  double acc = 0.0;
  for (int s0 = 4 * lane; s0 < nS; s0 += 256) {
    const double4 l4 = *(const double4*)(Lc + s0), z4 = *(const double4*)(sp.M + s0);
    acc += l4.x * z4.x + l4.y * z4.y + l4.z * z4.z + l4.w * z4.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) ch.y[k] = yk - acc;
}

#include "ba_front.h"

// ---------------------------------------------------------------- step application
// candidate cameras / focal: x + (-z)*scale, their tables, and the camera part of the norms
__global__ void ba_cand_cams(BaDev d, const unsigned char* __restrict__ cam_used, int rank) {
  if (lm_stopped(d)) return;
  {
    double radius_unused;
    lm_view(d, radius_unused);
  }
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
  // (the camera part of the norms: a pair of sums per wave -- one workgroup is one wave here -- that step_finish adds in wave order)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sn2 += __shfl_down(sn2, o);
    cn2 += __shfl_down(cn2, o);
  }
  if ((threadIdx.x & 63) == 0) {
    double* cp = d.step_part + 4 * (size_t)d.step_total + 2 * (size_t)blockIdx.x;
    cp[0] = sn2, cp[1] = cn2;
  }
}

// per point: back-substitute, model cost change, candidate point + candidate cost.
// One linearisation per observation: with a_o = -Jc_o z_c - Jf_o z_f (known before the point's step) and
// m_o = a_o + Jp_o s the model cost change  sum_o m_o.(r_o + m_o/2)  is
//   sum a.r + s.(sum Jp^T r) + (sum |a|^2)/2 + s.(sum Jp^T a) + s^T (sum Jp^T Jp) s / 2,
// all of them sums the first pass forms next to C_p and e = sum Jp^T (r + a); the second pass only evaluates the
// candidate's residuals.
// WPP waves per block of 64 points: a lane is a point, wave `sub` of the block takes the observations sub, sub + WPP,
// ... -- the lanes of a wave still look at the same observation index, so inside a run they agree on the camera and
// its table comes through scalar loads (below) -- and the 14 sums meet through LDS, added in wave order by every
// wave alike.  A thread walks a chain of dependent loads per observation; with one wave per block a SIMD has 1.5
// waves and nothing to overlap them with.
constexpr int BS_SUMS = 14;

// A record on its way to the host: LmDev into slot seq % LM_RING of the pinned ring, its seq word last (the host spins on the
// slot of the decision it waits for, and reads the ring for the log).
__device__ __forceinline__ void lm_publish(const BaDev& d, const LmDev& s) {
  constexpr int NW = (int)(sizeof(LmDev) / 4);
  static_assert(sizeof(LmDev) % 8 == 0 && offsetof(LmDev, seq) % 4 == 0, "LmDev is copied word by word");
  static_assert(NW <= 64, "a lane per word");
  volatile unsigned* dst = (volatile unsigned*)(d.lm_host) + (size_t)(s.seq % LM_RING) * NW;
  const unsigned* src = (const unsigned*)d.lm;  // (the record in device memory: what `s` was read from)
  constexpr int SEQ_W = (int)(offsetof(LmDev, seq) / 4);
  const int w = (int)threadIdx.x;
  if (w < NW && w != SEQ_W) dst[w] = src[w];  // (one store instruction of the wave; a word at a time took 24 us over the link)
  __threadfence_system();
  if (w == 0) dst[SEQ_W] = s.seq;
  __threadfence_system();
}

// the decision itself, by ONE thread, once every sum it reads is final (the last workgroup of the step evaluation, or
// ba_decide behind the all-reduce): the linearisation's scalars, the step evaluation's sums, the reduced solve's status
// (what the decision reads that is final BEFORE the step evaluation ends -- the linearisation's scalars, the record -- can be on
// its way while the finisher still waits for the last slots: lm_inputs_early / lm_decide_with)
struct LmEarly {
  double lin0, lin2, lin3;
  LmDev s;
};
__device__ __forceinline__ void lm_inputs_early(const BaDev& d, LmEarly& e) {
  const double* scv = red_sc(d);
  e.lin0 = scv[0], e.lin2 = scv[2], e.lin3 = scv[3];
  e.s = *d.lm;
}
__device__ __forceinline__ void lm_decide_with(const BaDev& d, LmEarly& e, const double sums[4], double peer_timeout = 0.0) {
  LmIn in;
  in.lin_cost = 0.5 * e.lin0;
  in.lin_nfail = e.lin2;
  in.lin_gmax = e.lin3;
  in.cost_c = 0.5 * sums[0];
  in.mcc = -sums[1];
  in.step_n2 = sums[2];
  in.cand_n2 = sums[3];
  in.info = *(volatile int*)d.info;
  // (another rank's spin ran out: this rank stops with it -- and with the same kind of time-out, so that every rank takes the
  // same way out: the level-by-level repeat, or SFMHIP_ERR_TIMEOUT when a finisher's slot never arrived anywhere)
  if (peer_timeout >= 1024.0) in.info = INFO_FINISHER_TIMEOUT;
  else if (peer_timeout > 0.0 && in.info >= 0) in.info = -1;
  lm_decide(e.s, in);
  *d.lm = e.s;
}
__device__ __forceinline__ void lm_decide_here(const BaDev& d) {
  LmEarly e;
  lm_inputs_early(d, e);
  const double sums[4] = {d.red2[0], d.red2[1], d.red2[2], d.red2[3]};
  lm_decide_with(d, e, sums, d.red2[RED2_TIMEOUT]);  // (behind the all-reduce: the ranks' flags summed)
}

__global__ void ba_decide(BaDev d) {
  if (threadIdx.x == 0) lm_decide_here(d);
}

// the record as it stands, into the host's ring: behind the last iteration of a batch (the host waits for it there), behind
// every iteration when the log is on -- not inside the decision, whose kernel every next kernel waits for
__global__ __launch_bounds__(64) void ba_lm_publish(BaDev d) {
  const LmDev s = *d.lm;
  lm_publish(d, s);
}

// End of a step-evaluation workgroup: its four sums (thread 0 holds them) go to the workgroup's own slot, fire and forget.
// The FINISHER -- the last workgroup of the step evaluation's last kernel -- then adds all slots, and the cameras' parts, in
// a fixed order (a tree over the slot index) and leaves the totals in red2[0..3] (what the all-reduce, the decision and the
// host read): the sums -- hence rho, the radius, the whole trajectory -- are the same bits every run, which atomics on
// shared words were not.  The data is its own flag (as in the down-sweep's mailbox): a slot holds a NaN no sum produces
// until its workgroup has written it, the finisher polls a slot until it is something else and puts the NaN back.  No
// fence, no counter, no wait on the writers' side: slots are written and read with agent-scope accesses (they bypass the
// non-coherent cache levels); the polls are bounded (a spin that runs out reports info = -1).  With the loop on the device
// and one rank the finisher takes the LM decision too.
#define STEP_PENDING 0x7FF8DEADBEEF0001ull
template <bool DECIDE>
__device__ __forceinline__ void step_finish(const BaDev& d, int slot, double cost_c, double mcc, double sn2, double cn2, int rank) {
  __shared__ double s_fin[4][4];
  __shared__ int s_to;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) {
    gbl_double* my = (gbl_double*)(d.step_part + 4 * (size_t)slot);
    double v[4] = {cost_c, mcc, sn2, cn2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if ((unsigned long long)__double_as_longlong(v[k]) == STEP_PENDING) v[k] = __longlong_as_double(0x7FF8000000000000ll);
      __hip_atomic_store(my + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_to = 0;
  }
  if (!(d.step_last && blockIdx.x == gridDim.x - 1)) return;
  __syncthreads();
  const bool decide = DECIDE && d.lm && d.decide_here;
  LmEarly early;
  if (decide && tid == 0) lm_inputs_early(d, early);  // (their round trip runs under the polls below)
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = tid; i < d.step_total; i += (int)blockDim.x) {
    gbl_double* q = (gbl_double*)(d.step_part + 4 * (size_t)i);
    double v[4];
    int budget = 1 << 16;
    for (;;) {
      bool pending = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending |= (unsigned long long)__double_as_longlong(v[k]) == STEP_PENDING;
      }
      if (!pending) break;
      if (--budget == 0) {
        s_to = 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = 0.0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __hip_atomic_store(q + k, __longlong_as_double((long long)STEP_PENDING), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next step evaluation)
      a[k] += v[k];
    }
  }
  if (rank == 0) {  // (the cameras and the focal are every rank's: counted once)
    const double* cp = d.step_part + 4 * (size_t)d.step_total;  // (written by an earlier kernel of the stream)
    for (int i = tid; i < d.cam_parts; i += (int)blockDim.x) a[2] += cp[2 * i], a[3] += cp[2 * i + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += __shfl_down(a[k], off);
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) s_fin[k][wave] = a[k];
  __syncthreads();
  if (tid == 0) {
    const int nw = (int)blockDim.x >> 6;
    double sums[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = s_fin[k][0];
      for (int w = 1; w < nw; ++w) v += s_fin[k][w];
      d.red2[k] = sums[k] = v;
    }
    if (s_to) atomicExch(d.info, INFO_FINISHER_TIMEOUT);  // (a workgroup of the step evaluation never wrote its slot: not a state the data can cause)
    // several ranks: a spin that ran out is this rank's alone (a scheduling artefact), the decision must be every rank's --
    // the flag travels with the sums through the all-reduce and lm_decide_here reads the total (RED2_TIMEOUT)
    // (1 per rank whose reduced solve timed out, 1024 per rank whose finisher did: the total tells every rank which case it is)
    {
      const int inf = *(volatile int*)d.info;
      d.red2[RED2_TIMEOUT] = inf == INFO_FINISHER_TIMEOUT ? 1024.0 : inf < 0 ? 1.0 : 0.0;
    }
    if (decide) lm_decide_with(d, early, sums);
  }
}

// a step evaluation enqueued behind a stop: nothing is evaluated; the decision is still counted (the host waits for its number)
template <bool DECIDE>
__device__ __forceinline__ void step_skip(const BaDev& d) {
  if (DECIDE && d.step_last && d.decide_here && blockIdx.x == 0 && threadIdx.x == 0) lm_decide_here(d);
}

template <int WPP>
__global__ __launch_bounds__(256) void ba_backsub(BaDev d, double radius, double lm_lo, double lm_hi, const int* __restrict__ plist, int npl,
                                                  int slot0 /* this kernel's first slot of step_part */) {
  if (lm_stopped(d)) return;  // (the decision of a launch of this kernel is ba_decide's: the kernel sits at its register limit)
  lm_view(d, radius);
  constexpr int BLOCKS = 4 / WPP;  // blocks of 64 points per workgroup
  __shared__ double s_part[WPP > 1 ? 4 * BS_SUMS * 64 : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wb = wave / WPP, sub = wave % WPP;
  // (plist: the points that ba_backsub_runs does not take -- those of the pair path; null: all of them)
  const int pidx = (blockIdx.x * BLOCKS + wb) * 64 + lane;
  const bool active = pidx < npl;
  const int p = active ? (plist ? plist[pidx] : pidx) : 0;
  double mcc = 0, cost_c = 0, sn2 = 0, cn2 = 0;
  double X[3] = {0, 0, 0}, sp[3] = {1, 1, 1};
  double sm[BS_SUMS];  // C (6: 00 10 11 20 21 22), Jp^T r (3), Jp^T a (3), a.r, |a|^2
#pragma unroll
  for (int e = 0; e < BS_SUMS; ++e) sm[e] = 0.0;
  int k0 = 0, k1 = 0;
  const double sf = *d.scale_f, focal = *d.focal, focal_c = *d.focal_c;
  const double zf = d.z[6 * d.nc];
  if (active) {
#pragma unroll
    for (int j = 0; j < 3; ++j) X[j] = d.pts[3 * p + j], sp[j] = d.scale_p[3 * p + j];
    k0 = d.optr[p], k1 = d.optr[p + 1];
    for (int k = k0 + sub; k < k1; k += WPP) {
      const int c = d.ocam[k];
      const double2 xy = d.oxy[k];
      ObsLin o;
      double a0, a1;
      // (the points of a run see the same cameras: the wave's lanes then want the same table, which goes through
      // the scalar cache -- with 51 doubles per observation the kernel is otherwise bound by the number of vector
      // memory instructions, 16 cycles each whatever the lanes ask for)
      if (__builtin_amdgcn_ballot_w64(c != __builtin_amdgcn_readfirstlane(c)) == 0) {
        cst_double* const zc = wave_uniform_ptr(d.z + 6 * c);
        obs_linearize_g(wave_uniform_ptr(d.camd + (size_t)CAMD * c), X, focal, xy.x, xy.y,
                        wave_uniform_ptr(d.scale_c + 6 * c), sp, sf, o);
        a0 = -o.Jf[0] * zf, a1 = -o.Jf[1] * zf;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const double zj = zc[j];
          a0 -= o.Jc[j] * zj;
          a1 -= o.Jc[6 + j] * zj;
        }
      } else {
        obs_linearize(d.camd + (size_t)CAMD * c, X, focal, xy.x, xy.y, d.scale_c + 6 * c, sp, sf, o);
        a0 = -o.Jf[0] * zf, a1 = -o.Jf[1] * zf;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const double zj = d.z[6 * c + j];
          a0 -= o.Jc[j] * zj;
          a1 -= o.Jc[6 + j] * zj;
        }
      }
      sm[0] += o.Jp[0] * o.Jp[0] + o.Jp[3] * o.Jp[3];
      sm[1] += o.Jp[1] * o.Jp[0] + o.Jp[4] * o.Jp[3];
      sm[2] += o.Jp[1] * o.Jp[1] + o.Jp[4] * o.Jp[4];
      sm[3] += o.Jp[2] * o.Jp[0] + o.Jp[5] * o.Jp[3];
      sm[4] += o.Jp[2] * o.Jp[1] + o.Jp[5] * o.Jp[4];
      sm[5] += o.Jp[2] * o.Jp[2] + o.Jp[5] * o.Jp[5];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        sm[6 + a] += o.Jp[a] * o.r0 + o.Jp[3 + a] * o.r1;
        sm[9 + a] += o.Jp[a] * a0 + o.Jp[3 + a] * a1;
      }
      sm[12] += a0 * o.r0 + a1 * o.r1;
      sm[13] += a0 * a0 + a1 * a1;
    }
  }
  if (WPP > 1) {
#pragma unroll
    for (int e = 0; e < BS_SUMS; ++e) s_part[(wave * BS_SUMS + e) * 64 + lane] = sm[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < BS_SUMS; ++e) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < WPP; ++w) v += s_part[((wb * WPP + w) * BS_SUMS + e) * 64 + lane];
      sm[e] = v;
    }
  }
  if (active) {
    double C[6] = {sm[0], sm[1], sm[2], sm[3], sm[4], sm[5]};
    const double* pr = sm + 6;
    const double* pa = sm + 9;
    const double ar = sm[12], aa = sm[13];
    const double e[3] = {pr[0] + pa[0], pr[1] + pa[1], pr[2] + pa[2]};
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
      if (sub == 0) {
        d.pts_c[3 * p + j] = Xc[j];
        sn2 += dl * dl;
        cn2 += Xc[j] * Xc[j];
      }
    }
    if (sub == 0) {
      // s^T C0 s (C0 = the undamped sums, lower triangle: 00, 10, 11, 20, 21, 22)
      const double q = stp[0] * (sm[0] * stp[0] + 2.0 * (sm[1] * stp[1] + sm[3] * stp[2])) +
                       stp[1] * (sm[2] * stp[1] + 2.0 * sm[4] * stp[2]) + stp[2] * sm[5] * stp[2];
      mcc = ar + (stp[0] * pr[0] + stp[1] * pr[1] + stp[2] * pr[2]) +
            0.5 * aa + (stp[0] * pa[0] + stp[1] * pa[1] + stp[2] * pa[2]) + 0.5 * q;
    }
    for (int k = k0 + sub; k < k1; k += WPP) {
      const int c = d.ocam[k];
      const double2 xy = d.oxy[k];
      double r0, r1;
      if (__builtin_amdgcn_ballot_w64(c != __builtin_amdgcn_readfirstlane(c)) == 0)
        obs_residual(wave_uniform_ptr(d.camd_c + (size_t)CAMD * c), Xc, focal_c, xy.x, xy.y, r0, r1);
      else
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
  if (lane == 0) {
    sh[0][wave] = mcc;
    sh[1][wave] = cost_c;
    sh[2][wave] = sn2;
    sh[3][wave] = cn2;
  }
  __syncthreads();
  step_finish<false>(d, slot0 + (int)blockIdx.x, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3],
              sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3], sh[3][0] + sh[3][1] + sh[3][2] + sh[3][3], d.rank);
}

// The same step for the points in runs (ba_eliminate_mfma's chunks: contiguous points that see the same ascending camera list, at most
// 10 cameras), still a thread per point -- 64 points per wave is what keeps the arithmetic per observation small; a row of 16 lanes per
// point, the elimination's layout, was measured at 36 us: 4 points per wave-instruction -- but with the run's structure used for the
// memory side, which is what bounds ba_backsub (K7): every lane of the workgroup wants the SAME camera at the same time, so the run's
// tables, scales and steps sit in LDS and are read by broadcast (no vector-memory instruction per table entry, no scalar-cache misses),
// and the observations, point-major in memory (a wave's load of "observation k of my point" touches 64 lines), are staged through LDS by
// coalesced loads, every line fetched once, and read back transposed ([k][point]: conflict-free).
constexpr int BSR_PTS = 256;  // points per block of a workgroup (a thread each)
// bs_desc: 16 ints per run, large runs first: n, p0, cnt, the first observation's index, the n <= 10 cameras -- everything the
// workgroup's loads depend on in ONE record (chunk id -> chunk -> camera list -> tables was four dependent round trips)
__global__ __launch_bounds__(256) void ba_backsub_runs(BaDev d, const int4* __restrict__ bs_desc, double radius, double lm_lo, double lm_hi,
                                                      int split) {
  if (lm_stopped(d)) {
    step_skip<true>(d);
    return;
  }
  lm_view(d, radius);
  __shared__ __attribute__((aligned(16))) double s_tab[10 * 12];    // the run's cameras: R, t ...
  __shared__ __attribute__((aligned(16))) double s_tabc[10 * 12];   // ... the candidates' R, t ...
  __shared__ __attribute__((aligned(16))) double s_zs[10 * 12];     // ... and per camera: M = sum_j scale_j z_j dR/dw_j (9), scale z of the translation (3)
  __shared__ __attribute__((aligned(16))) double2 s_xy[10 * BSR_PTS];
  __shared__ double sh[4][4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pt = tid;  // (a thread per point: every lane of a wave is at the same observation index, hence the same camera)
  __shared__ int s_cams[12];
  const int4* const rec = bs_desc + 4 * (size_t)(blockIdx.x / split);
  const int4 hd = rec[0];  // n, p0, cnt, first observation
  struct { int n, p0, cnt; } ch = {hd.x, hd.y, hd.z};
  const int kobs0 = hd.w;
  if (tid < 12) s_cams[tid] = ((const int*)(rec + 1))[tid];
  const int part = blockIdx.x % split;
  const int pt_lo = (int)((long long)ch.cnt * part / split), pt_hi = (int)((long long)ch.cnt * (part + 1) / split);
  const int n = ch.n;
  const double sf = *d.scale_f, focal = *d.focal, focal_c = *d.focal_c, zf = d.z[6 * d.nc];
  __syncthreads();
  for (int idx = tid; idx < n * 12; idx += 256) {
    const int o = idx / 12, e = idx - o * 12;
    s_tab[idx] = d.camd[(size_t)CAMD * s_cams[o] + e];
    s_tabc[idx] = d.camd_c[(size_t)CAMD * s_cams[o] + e];
  }
  // The model cost change needs a = -Jc z_c - Jf z_f per observation, not Jc itself: Jc z_c = dr/dP (M X + t_z) with, per
  // camera, M = sum_j scale_j z_j dR/dw_j and t_z = (scale z) of the translation -- formed once here, so that an observation
  // reads 12 doubles of them where it read the 27 of the three derivative matrices and 12 of scale and step, and forms
  // one 3 x 3 product where it formed three (the kernel is bound by instructions per observation, K7).
  for (int idx = tid; idx < n * 12; idx += 256) {
    const int o = idx / 12, e = idx - o * 12, c = s_cams[o];
    double v;
    if (e < 9) {
      v = 0.0;
#pragma unroll
      for (int j = 0; j < 3; ++j) v += d.camd[(size_t)CAMD * c + 12 + 9 * j + e] * (d.scale_c[6 * c + j] * d.z[6 * c + j]);
    } else {
      v = d.scale_c[6 * c + e - 6] * d.z[6 * c + e - 6];
    }
    s_zs[idx] = v;
  }
  double mcc = 0, cost_c = 0, sn2 = 0, cn2 = 0;
  for (int blk = pt_lo; blk < pt_hi; blk += BSR_PTS) {
    const int npts = min(BSR_PTS, pt_hi - blk);
    const bool active = pt < npts;
    const int p = ch.p0 + blk + (active ? pt : 0);
    // (the point, its scale and the block's observations are asked for together: one round trip, not three)
    double X[3], sp[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) X[j] = d.pts[3 * p + j], sp[j] = d.scale_p[3 * p + j];
    const double2* src = d.oxy + kobs0 + (size_t)blk * n;  // npts x n records, point-major: coalesced
    double2 stage[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int e = tid + 256 * u;
      stage[u] = e < npts * n ? src[e] : make_double2(0.0, 0.0);
    }
    __syncthreads();  // (the previous block's observations are read)
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int e = tid + 256 * u;
      if (e < npts * n) {
        const int pl = e / n, k = e - pl * n;
        s_xy[k * BSR_PTS + pl] = stage[u];
      }
    }
    __syncthreads();  // (and, the first time round, the tables are in)
    double sm[BS_SUMS];  // C (6: 00 10 11 20 21 22), Jp^T r (3), Jp^T a (3), a.r, |a|^2 -- as in ba_backsub
#pragma unroll
    for (int e = 0; e < BS_SUMS; ++e) sm[e] = 0.0;
    for (int k = 0; k < n; ++k) {
      const double2 xy = s_xy[k * BSR_PTS + pt];
      // (the same address for every lane: LDS broadcasts.  Scalar loads instead -- operands in SGPRs -- were measured slower here,
      // 25.7 us against 21.0: K7)
      const lds_double* const tab = (const lds_double*)s_tab + k * 12;
      const lds_double* const mz = (const lds_double*)s_zs + k * 12;
      struct { double Jp[6], r0, r1; } o;
      double a0, a1;
      {
        const double px = tab[0] * X[0] + tab[1] * X[1] + tab[2] * X[2] + tab[9];
        const double py = tab[3] * X[0] + tab[4] * X[1] + tab[5] * X[2] + tab[10];
        const double pz = tab[6] * X[0] + tab[7] * X[1] + tab[8] * X[2] + tab[11];
        const double iz = rcp_f64(pz);
        const double xp = px * iz, yp = py * iz;
        o.r0 = focal * xp - xy.x;
        o.r1 = focal * yp - xy.y;
        const double d00 = focal * iz, d02 = -focal * xp * iz, d12 = -focal * yp * iz;  // dr/dP rows
        // (Jp without the point's column scale: the sums below are scaled once per point, not every entry per observation)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          o.Jp[j] = d00 * tab[j] + d02 * tab[6 + j];
          o.Jp[3 + j] = d00 * tab[3 + j] + d12 * tab[6 + j];
        }
        const double qx = mz[0] * X[0] + mz[1] * X[1] + mz[2] * X[2] + mz[9];   // dP = M X + t_z
        const double qy = mz[3] * X[0] + mz[4] * X[1] + mz[5] * X[2] + mz[10];
        const double qz = mz[6] * X[0] + mz[7] * X[1] + mz[8] * X[2] + mz[11];
        const double fz = sf * zf;
        a0 = -(d00 * qx + d02 * qz) - xp * fz;
        a1 = -(d00 * qy + d12 * qz) - yp * fz;
      }
      sm[0] += o.Jp[0] * o.Jp[0] + o.Jp[3] * o.Jp[3];
      sm[1] += o.Jp[1] * o.Jp[0] + o.Jp[4] * o.Jp[3];
      sm[2] += o.Jp[1] * o.Jp[1] + o.Jp[4] * o.Jp[4];
      sm[3] += o.Jp[2] * o.Jp[0] + o.Jp[5] * o.Jp[3];
      sm[4] += o.Jp[2] * o.Jp[1] + o.Jp[5] * o.Jp[4];
      sm[5] += o.Jp[2] * o.Jp[2] + o.Jp[5] * o.Jp[5];
      // (Jp^T r and Jp^T a only ever appear as their sum: in the step and in the model cost change)
      const double u0 = o.r0 + a0, u1 = o.r1 + a1;
#pragma unroll
      for (int a = 0; a < 3; ++a) sm[6 + a] += o.Jp[a] * u0 + o.Jp[3 + a] * u1;
      sm[12] += a0 * o.r0 + a1 * o.r1;
      sm[13] += a0 * a0 + a1 * a1;
    }
    sm[0] *= sp[0] * sp[0], sm[1] *= sp[1] * sp[0], sm[2] *= sp[1] * sp[1];
    sm[3] *= sp[2] * sp[0], sm[4] *= sp[2] * sp[1], sm[5] *= sp[2] * sp[2];
    double C[6] = {sm[0], sm[1], sm[2], sm[3], sm[4], sm[5]};
    const double ar = sm[12], aa = sm[13];
    const double e3[3] = {sm[6] * sp[0], sm[7] * sp[1], sm[8] * sp[2]};
    C[0] += fmin(fmax(C[0], lm_lo), lm_hi) / radius;
    C[2] += fmin(fmax(C[2], lm_lo), lm_hi) / radius;
    C[5] += fmin(fmax(C[5], lm_lo), lm_hi) / radius;
    double Li[6];
    if (!chol3_inv(C, Li)) Li[0] = Li[1] = Li[2] = Li[3] = Li[4] = Li[5] = 0;
    const double t0 = Li[0] * e3[0], t1 = Li[1] * e3[0] + Li[2] * e3[1], t2 = Li[3] * e3[0] + Li[4] * e3[1] + Li[5] * e3[2];
    const double stp[3] = {-(Li[0] * t0 + Li[1] * t1 + Li[3] * t2), -(Li[2] * t1 + Li[4] * t2), -(Li[5] * t2)};
    double Xc[3];
    const bool head = active;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double dl = stp[j] * sp[j];
      Xc[j] = X[j] + dl;
      if (head) {
        d.pts_c[3 * p + j] = Xc[j];
        sn2 += dl * dl;
        cn2 += Xc[j] * Xc[j];
      }
    }
    if (head) {
      const double qd = stp[0] * (sm[0] * stp[0] + 2.0 * (sm[1] * stp[1] + sm[3] * stp[2])) +
                        stp[1] * (sm[2] * stp[1] + 2.0 * sm[4] * stp[2]) + stp[2] * sm[5] * stp[2];
      mcc += ar + 0.5 * aa + (stp[0] * e3[0] + stp[1] * e3[1] + stp[2] * e3[2]) + 0.5 * qd;
    }
    for (int k = 0; k < n; ++k) {
      const double2 xy = s_xy[k * BSR_PTS + pt];
      double r0, r1;
      obs_residual((const lds_double*)s_tabc + k * 12, Xc, focal_c, xy.x, xy.y, r0, r1);
      if (active) cost_c += r0 * r0 + r1 * r1;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mcc += __shfl_down(mcc, off);
    cost_c += __shfl_down(cost_c, off);
    sn2 += __shfl_down(sn2, off);
    cn2 += __shfl_down(cn2, off);
  }
  if (lane == 0) {
    sh[0][wave] = mcc;
    sh[1][wave] = cost_c;
    sh[2][wave] = sn2;
    sh[3][wave] = cn2;
  }
  __syncthreads();
  step_finish<true>(d, (int)blockIdx.x, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3],
              sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3], sh[3][0] + sh[3][1] + sh[3][2] + sh[3][3], d.rank);
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
  bool started = false;
  bool have_lin = false;  // the reduced system in `red` is the linearisation at the current x with the current radius
  LmDev s{};              // the trust-region record (the host's copy; the device's own while a device loop runs)
};

struct sfmhip_ba {
  sfmhip_ctx* ctx = nullptr;
  LmState lm;
  // the loop on the device (round 5): the record, the host's ring of copies, the step evaluation's slots
  LmDev* d_lm = nullptr;
  LmDev* h_ring = nullptr;      // pinned, LM_RING records
  LmDev* h_ring_dev = nullptr;  // the same as the device sees it
  unsigned lm_seq = 0;          // decisions made on this problem so far (monotonic: a stale ring slot never matches)
  double* d_step_part = nullptr;
  size_t step_part_n = 0;
  bool tree_by_level = false;   // a hand-off between fronts timed out once: one launch per tree level from then on
  int spin_timeouts = 0;
  int nc = 0, np_in = 0, no_in = 0;  // as given
  int np = 0, no = 0;                // with >= 1 observation, sorted order
  int dim = 0, ld = 0;
  size_t ssz = 0;  // ld*ld: doubles of the S part of `red`
  BaDev d{};
  bool use_arena = false;     // device memory from the context's grow-only block (sfmhip_ba_solve's problems)
  size_t arena_off = 0, arena_need = 0;
  bool pinned_shared = false;  // h_sc / h_ring are the context's pinned block, not this problem's
  std::vector<int> perm;  // sorted point -> input point
  std::vector<int> obs_src;  // sorted observation -> input observation (ba_set_observations: the same structure, new measurements)
  std::vector<int> cxy_src;  // entry of the pair path's camera-major list -> sorted observation
  double2* d_oxy_w = nullptr;  // (the writable view of d.oxy)
  int* d_obs_src = nullptr;    // observation w of the plan's order is observation d_obs_src[w] of the caller's
  std::vector<unsigned char> h_cam_used;
  unsigned char* d_cam_used = nullptr;
  bool cam_used_known = false;
  double* d_flag = nullptr;  // one double: rank-consistent decisions (ba_agree_flag)
  bool chol_attr_set = false;
  // plan
  Chunk* d_chunks = nullptr;
  bool elim_deterministic = !(getenv("SFMHIP_BA_DETERMINISTIC") && atoi(getenv("SFMHIP_BA_DETERMINISTIC")) == 0);  // (the default since round 3)
  // slab epilogue of ba_eliminate_mfma + ba_gather_slabs: [0] the full linearisation, [1] the norms-only mode
  double* d_slab = nullptr;
  int* d_gth_ptr[2] = {nullptr, nullptr};
  unsigned* d_gth_src[2] = {nullptr, nullptr};
  int* d_gth_dest[2] = {nullptr, nullptr};
  int n_gth[2] = {0, 0};
  // the row lists of ba_gather_rows (full linearisation): rows of S the MFMA path writes, their (chunk, local row) sources
  int4* d_grow_hdr = nullptr;
  int4* d_grow_head = nullptr;
  int4* d_grow_src = nullptr;
  int* d_grow_colmap = nullptr;
  int n_grow = 0, grow_waves = 4, grow_accw = 64, n_chunks = 0;  // grow_accw: a wave's accumulator (doubles)
  int* d_chunk_ids[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int n_chunk_ids[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [NB-1]: 4-wave workgroups (long runs), [4 + NB-1]: 1-wave (short runs)
  int* d_sig_cams = nullptr;
  int* d_cptr = nullptr;  // camera-major observation list: cptr[nc+1], cpt[no], cxy[no]
  int* d_cpt = nullptr;
  double2* d_cxy = nullptr;
  int cam_split = 1;
  int dense_xb = 0;              // dense factorisation: X kept inside diagonal blocks of this many tile columns (0: all of X)
  double* d_back_part = nullptr;  // chol_back_block: the columns' parts of z_K
  double* d_cb_part = nullptr;  // ba_cam_blocks with cam_split > 1: 66 sums per (camera, slice); d_cb_cnt: arrivals per camera
  int* d_cb_cnt = nullptr;
  int* d_fb_points = nullptr;  // the pair path's points (sorted indices), their first T row
  int* d_pp_obase = nullptr;
  int n_fb = 0;
  int2* d_cslot = nullptr;     // camera-major list: (T row, point slot)
  double* d_ppT = nullptr;     // 18 doubles per observation of the pair path
  double* d_tfu = nullptr;     // t_f, u per point of the pair path
  double* d_pp_part = nullptr;  // 8 partial sums per workgroup of ba_pp_points
  int n_pp_part = 0;
  int* d_pair_ptr = nullptr;
  int2* d_pair_cams = nullptr;
  int2* d_pair_ent = nullptr;
  int n_pairs_pp = 0;
  int tree_xoff = 0;  // which of every `tree_stride` workgroups holds a front: a different one for every problem of the process
  long long tree_dbg_ints = 0, tree_dbg_doubles = 0;  // (sizes of the front tree's tables and pool: diagnostic builds)
  int* d_bs_ids = nullptr;  // ba_backsub_runs' records, 16 ints per chunk, large chunks first
  int elim_waves = 4;  // waves per workgroup of the long-run class of ba_eliminate_mfma (8, 4 or 2)
  // dissected reduced system (NdPlan below): built at the first solve (with world > 1 the camera graph is the
  // union over the ranks, which needs the all-reduce)
  std::vector<unsigned long long> h_adj;  // camera co-visibility, nc x ceil(nc/64) bit rows (this rank's points)
  bool nd_ready = false, nd_on = false, nd_kept = false;  // nd_kept: the front plan came from the context's last one-shot problem
  NdSet nd{};
  NdCols nd_cols{};
  int nd_max_ni = 0;
  double* nd_buf = nullptr;  // all chain matrices, vectors and X blocks (nd_gather writes what the factorisation reads)
  size_t nd_buf_count = 0;
  int4* nd_gather_jobs = nullptr;
  int nd_n_gather = 0;
  bool chol_chains_attr_set = false;
  // front tree (ba_front_plan.h / ba_front.h): the multifrontal factorisation, one workgroup per front
  bool tree_on = false, tree_attr_set = false, solve_cand = false;
  FrontSet tree_fs{};
  unsigned tree_epoch = 0;
  int tree_stride = 1, tree_levels = 0, tree_chain_tiles = 0, tree_chain_blocks = 0, tree_max_T = 0;
  // ba_finalize deferred to the next nd_gather (the LM loop's linearisations, when the dissected solve follows)
  bool fin_pending = false, defer_fin = false;
  double fin_radius = 0, fin_lo = 0, fin_hi = 0;
  // device storage owned
  std::vector<void*> allocs;
  size_t red_count = 0;
  // dissected solve: a second [S | g | ... | red2] buffer that nd_gather's spare workgroups zero while they are at it, so
  // that the next linearisation starts on a clean buffer instead of behind a memset launch
  double* red_alt = nullptr;
  bool alt_clean = false, red_is_alt = false;
  bool prezero = !(getenv("SFMHIP_BA_PREZERO") && atoi(getenv("SFMHIP_BA_PREZERO")) == 0);  // (read when the problem is created)
  double* d_red_pack = nullptr;  // world > 1: the all-reduce payload (packed upper triangle of S + tail)
  int2* d_xblocks = nullptr;     // world > 1, sparse camera graph: the co-visible camera pairs (a <= b) that are exchanged
  int n_xblocks = 0;             // 0: the dense exchange
  double* h_sc = nullptr;  // pinned: scalars read back per iteration (+ the sequence number of ba_publish)
  double* h_sc_dev = nullptr;  // the same buffer as the device sees it
  double h_seq = 0.0;
  // comm
  sfmhip_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int rank = 0, world = 1;
  // state
  bool scale_ready = false;
  bool camd_valid = false;  // d.camd holds the tables of d.cams (set_params invalidates; an accepted step swaps in
                            // the candidate's tables, which ba_cand_cams built in full)
  std::vector<double> h_pts_unused;  // input points without observations keep their values
  std::vector<double> h_pts_in;
  double x_norm = 0;
  // timing
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double t_acc[4] = {0, 0, 0, 0};
  bool ev_on[5] = {false, false, false, false, false};  // recorded since the last ba_acc_timing
  int launches = 0;
};

// fn(lo, hi) over [0, n) on up to 8 host threads (problem set-up only)
static int host_threads(int n) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int nth = (int)std::max(1u, std::min(16u, hw ? hw : 1u));
  return n < 20000 ? 1 : nth;
}
// The host threads of the set-up's passes: a pool that lives as long as the process (starting and joining sixteen threads is
// 0.4-0.5 ms, and a set-up has five such passes).  One job at a time; a caller that finds the pool busy (another host thread is
// setting a problem up) starts threads of its own, as every pass did before.
class HostPool {
 public:
  static HostPool& get() {
    static HostPool* p = new HostPool();  // (never destroyed: its threads wait on a condition variable until the process ends)
    return *p;
  }
  // f(t) for t in [0, nth): the caller is t = 0.  Returns false when the pool is taken (nothing has run).
  bool run(int nth, const std::function<void(int)>& f) {
    if (getpid() != pid_) return false;  // (a forked child has this object but none of its threads: it starts its own)
    std::unique_lock<std::mutex> job_lock(job_m_, std::try_to_lock);
    if (!job_lock.owns_lock()) return false;
    {
      std::lock_guard<std::mutex> lk(m_);
      while ((int)workers_.size() < nth - 1) {
        const int id = (int)workers_.size() + 1;
        workers_.emplace_back([this, id]() { work(id); });
        workers_.back().detach();
      }
      f_ = &f;
      nth_ = nth;
      pending_ = nth - 1;
      ++gen_;
    }
    cv_.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this]() { return pending_ == 0; });
    f_ = nullptr;
    return true;
  }

 private:
  void work(int id) {
    unsigned seen = 0;
    for (;;) {
      const std::function<void(int)>* f = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (id < nth_) f = f_;
      }
      if (!f) continue;
      (*f)(id);
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  const pid_t pid_ = getpid();
  std::mutex job_m_, m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* f_ = nullptr;
  int nth_ = 0, pending_ = 0;
  unsigned gen_ = 0;
};

// fn(t, lo, hi): thread t of host_threads(n) takes [lo, hi)
template <typename F>
static void host_parallel_for_t(int n, int nth, F fn) {
  if (nth <= 1) {
    fn(0, 0, n);
    return;
  }
  const std::function<void(int)> job = [&](int t) {
    const int lo = (int)((long long)n * t / nth), hi = (int)((long long)n * (t + 1) / nth);
    fn(t, lo, hi);
  };
  if (HostPool::get().run(nth, job)) return;
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t) th.emplace_back([&job, t]() { job(t); });
  for (auto& x : th) x.join();
}
template <typename F>
static void host_parallel_for(int n, F fn) {
  host_parallel_for_t(n, host_threads(n), [&](int, int lo, int hi) { fn(lo, hi); });
}

template <typename T>
static int ba_alloc(sfmhip_ba* b, T** p, size_t n) {
  const size_t bytes = ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
  b->arena_need += bytes;
  if (b->use_arena && b->arena_off + bytes <= b->ctx->ba_arena_bytes) {
    // (a problem of the one-shot entry point: carved from the context's block, given back as a whole when the problem goes)
    void* v = (char*)b->ctx->ba_arena + b->arena_off;
    b->arena_off += bytes;
    static const bool poison = getenv("SFMHIP_POISON") != nullptr;  // (tests: as sfm_dev_alloc does for fresh allocations)
    if (poison) {
      hipMemset(v, 0xFF, bytes);
      hipDeviceSynchronize();
    }
    *p = (T*)v;
    return SFMHIP_OK;
  }
  SFM_TRY(sfm_dev_alloc(p, n));
  b->allocs.push_back((void*)*p);
  return SFMHIP_OK;
}

#ifdef SFM_FRONT_STAMPS
extern "C" int sfmhip_debug_front_stamps(unsigned long long* out, int n_fronts) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_front_stamps), sizeof(unsigned long long) * 128 * (size_t)std::min(n_fronts, 128)) == hipSuccess ? 0 : -2;
}
#endif
#ifdef SFM_FRONT_STAMPS
// the front tree's tables and its pool (L | y | contribution tiles per front) as they are after the last solve
extern "C" int sfmhip_debug_tree_dump(sfmhip_ba* b, int* ints, long long n_ints, double* pool, long long n_doubles, long long sizes[2]) {
  if (!b || !b->tree_on) return -1;
  sizes[0] = b->tree_dbg_ints, sizes[1] = b->tree_dbg_doubles;
  hipDeviceSynchronize();
  if (ints && n_ints >= sizes[0]) hipMemcpy(ints, b->tree_fs.ints, sizes[0] * sizeof(int), hipMemcpyDeviceToHost);
  if (pool && n_doubles >= sizes[1]) hipMemcpy(pool, b->tree_fs.pool, sizes[1] * sizeof(double), hipMemcpyDeviceToHost);
  return 0;
}
extern "C" int sfmhip_debug_front_ubench(int mode, unsigned long long* out24) {
  unsigned long long* d_out = nullptr;
  double* d_sink = nullptr;
  if (hipMalloc(&d_out, 24 * 8) != hipSuccess || hipMalloc(&d_sink, 768 * 8) != hipSuccess) return -1;
  hipMemset(d_out, 0, 24 * 8);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)front_ubench, hipFuncAttributeMaxDynamicSharedMemorySize, FR_LDS_BYTES);
    attr = true;
  }
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(front_ubench, dim3(1), dim3(FR_WAVES * 64), FR_LDS_BYTES, 0, d_out, d_sink, mode);
  const hipError_t e = hipMemcpy(out24, d_out, 24 * 8, hipMemcpyDeviceToHost);
  hipFree(d_out);
  hipFree(d_sink);
  return e == hipSuccess ? 0 : -2;
}
#endif
#ifdef SFM_FRONT_STAMPS
extern "C" int sfmhip_debug_down_stamps(unsigned long long* out, int n_fronts) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_down_stamps), sizeof(unsigned long long) * 8 * (size_t)std::min(n_fronts, 128)) == hipSuccess ? 0 : -2;
}
#endif
#ifdef SFM_ELIM_STAMPS
extern "C" int sfmhip_debug_elim_stamps(unsigned long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_elim_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
}
extern "C" int sfmhip_debug_elim_wg(unsigned long long* out) {  // 2048 x 5
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_elim_wg), sizeof(unsigned long long) * 2048 * 5) == hipSuccess ? 0 : -2;
}
#endif
#ifdef SFM_CHOL_STAMPS
extern "C" int sfmhip_debug_chol_stamps(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_chol_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
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
  o->verbose = getenv("SFMHIP_BA_VERBOSE") ? 1 : 0;  // (diagnosis: the LM log on stderr)
}

// Host memory of the one-shot entry point's set-ups, kept with the context between calls: the passes below fill some 50 MB of
// vectors at cfg4, and fresh ones cost their page faults on the way in (3-4 ms) and their unmapping on the way out (3.9 ms of a
// 33 ms call) -- the allocator of the host process decides which, the library should not depend on it.
struct BaHostScratch {
  std::vector<int> cnt, slot, scam, run_of, order, optr, ocam, table, obs_src, cxy_src;
  std::vector<uint64_t> sig_hash;
  std::vector<double> h_pts_in;
  // the front tree of the last one-shot problem: the per-view call pattern (src/Sfm.cpp:996) changes the tracks from call to
  // call and the camera graph hardly ever -- a problem with the same graph (compared bit for bit) and the same planning
  // switches takes the kept plan instead of dissecting again (2.5 ms of a 23 ms call at cfg4)
  bool nd_valid = false;
  int nd_nc = 0, nd_key[5] = {0, 0, 0, 0, 0};
  std::vector<unsigned long long> nd_adj;
  fplan::Plan nd_P;
  fplan::Flat nd_fl;
};
static void ba_host_scratch_free(void* p) { delete (BaHostScratch*)p; }
static BaHostScratch* ba_host_scratch(sfmhip_ctx* ctx) {
  if (!ctx->ba_host_scratch) {
    ctx->ba_host_scratch = new BaHostScratch();
    ctx->ba_host_scratch_free = ba_host_scratch_free;
  }
  return (BaHostScratch*)ctx->ba_host_scratch;
}

// the observations' coordinates from the caller's order into the plan's (a million 16-byte records: 1.1 ms of a set-up's host time
// as a gather over sixteen host threads, 10 us here)
__global__ __launch_bounds__(256) void ba_permute_xy(const double2* __restrict__ src, const int* __restrict__ perm, double2* __restrict__ dst, int n) {
  for (int w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) dst[w] = src[perm[w]];
}

// the caller's coordinates -> the context's device scratch -> the plan's order (d.oxy).  Synchronous: the scratch is free again
// at the return.
static int ba_upload_xy(sfmhip_ba* b, const double* obs_xy, int n_obs) {
  if (!b->no) return SFMHIP_OK;
  void* raw = nullptr;
  SFM_TRY(sfm_ctx_dev_scratch(b->ctx, 1, sizeof(double) * 2 * (size_t)n_obs, &raw));
  SFM_HIP_TRY(hipMemcpy(raw, obs_xy, sizeof(double) * 2 * (size_t)n_obs, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ba_permute_xy, dim3((unsigned)std::min(1024, (b->no + 255) / 256)), dim3(256), 0, b->ctx->stream, (const double2*)raw,
                     (const int*)b->d_obs_src, b->d_oxy_w, b->no);
  SFM_HIP_TRY(hipGetLastError());
  SFM_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
  return SFMHIP_OK;
}

static int ba_create_impl(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, const int32_t* obs_cam, const int32_t* obs_pt,
                          const double* obs_xy, bool arena, sfmhip_ba** out);
extern "C" int sfmhip_ba_create(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, const int32_t* obs_cam,
                                const int32_t* obs_pt, const double* obs_xy, sfmhip_ba** out) {
  return ba_create_impl(ctx, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_xy, false, out);
}
static int ba_create_impl(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, const int32_t* obs_cam, const int32_t* obs_pt,
                          const double* obs_xy, bool arena, sfmhip_ba** out) {
  if (!ctx || !out || n_cam <= 0 || n_pt < 0 || n_obs < 0) return SFMHIP_ERR_ARG;
  if (n_obs && (!obs_cam || !obs_pt || !obs_xy)) return SFMHIP_ERR_ARG;
  // (one pass: the range check, and whether the observations already come grouped by point -- the order the reference adds
  // residual blocks in, src/BundleAdjustment.cpp:83-110 -- in which case the counting sort's scatter below is the identity)
  bool grouped = true;
  {
    int bad = 0;
    for (int o = 0; o < n_obs; ++o) {
      bad |= (obs_cam[o] < 0) | (obs_cam[o] >= n_cam) | (obs_pt[o] < 0) | (obs_pt[o] >= n_pt);
      grouped &= o == 0 || obs_pt[o - 1] <= obs_pt[o];
    }
    if (bad) return SFMHIP_ERR_ARG;
  }
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
  b->use_arena = arena;
  b->nc = n_cam;
  b->np_in = n_pt;
  b->no_in = n_obs;
  b->dim = 6 * n_cam + 1;
  b->ld = (b->dim + 2 * CB - 1) / (2 * CB) * (2 * CB);  // chol_step2 takes two 32-column panels per launch
  b->ssz = (size_t)b->ld * b->ld;
  // ---- group observations by point, ascending camera inside a point (std::map order of
  //      Point3D::idxImage, reference src/BundleAdjustment.cpp:87)
  // (a one-shot problem takes the context's vectors -- their capacity survives the call -- and gives its own back when it goes)
  BaHostScratch own_scratch, *hs = arena ? ba_host_scratch(ctx) : &own_scratch;
  if (arena) {
    b->obs_src.swap(hs->obs_src);
    b->cxy_src.swap(hs->cxy_src);
    b->h_pts_in.swap(hs->h_pts_in);
  }
  std::vector<int>& cnt = hs->cnt;
  cnt.assign((size_t)n_pt + 1, 0);
  for (int o = 0; o < n_obs; ++o) cnt[obs_pt[o] + 1]++;
  for (int p = 0; p < n_pt; ++p) cnt[p + 1] += cnt[p];
  std::vector<int>& slot = hs->slot;
  slot.resize(n_obs);
  if (grouped) {
    host_parallel_for(n_obs, [&](int lo, int hi) {
      for (int o = lo; o < hi; ++o) slot[o] = o;
    });
  } else {
    std::vector<int> fill(n_pt, 0);
    for (int o = 0; o < n_obs; ++o) slot[cnt[obs_pt[o]] + fill[obs_pt[o]]++] = o;
  }
  // per point (a few host threads: every pass over a million observations is a cache-miss chain on
  // one core): stable insertion sort -- a point has a handful of observations --, then the point's
  // ascending camera list, flat, and a hash of it: the signature grouping below compares
  // (length, hash) first and walks the lists only on equal hashes
  std::vector<int>& scam = hs->scam;
  scam.resize(n_obs);
  std::vector<uint64_t>& sig_hash = hs->sig_hash;
  sig_hash.assign(n_pt, 0);
  host_parallel_for(n_pt, [&](int plo, int phi) {
    for (int p = plo; p < phi; ++p) {
      for (int i = cnt[p] + 1; i < cnt[p + 1]; ++i) {
        const int v = slot[i], cv = obs_cam[v];
        int j = i - 1;
        for (; j >= cnt[p] && obs_cam[slot[j]] > cv; --j) slot[j + 1] = slot[j];
        slot[j + 1] = v;
      }
      uint64_t h = 1469598103934665603ull;
      for (int k = cnt[p]; k < cnt[p + 1]; ++k) {
        scam[k] = obs_cam[slot[k]];
        h = (h ^ (uint64_t)(uint32_t)scam[k]) * 1099511628211ull;
      }
      sig_hash[p] = h;
    }
  });
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
  // (round 6: the three passes run on the host threads.  Every thread groups the points of ITS block with a table of its own --
  // local run ids in the block's order of first appearance --, the blocks' runs then meet one table in block order, which IS the
  // points' order of first appearance, and the counting sort scatters block by block from per-block start positions: the same
  // run ids, the same order as one thread produces, whatever the number of threads)
  std::vector<int>& run_of = hs->run_of;
  run_of.resize(n_pt);
  std::vector<int> run_rep, run_cnt;
  const int sig_threads = host_threads(n_pt);
  std::vector<std::vector<int>> loc_rep((size_t)sig_threads), loc_cnt((size_t)sig_threads);
  auto probe = [&](int* table, size_t cap, std::vector<int>& rep, int p) -> int {  // the run of point p among `rep`, entered when new
    size_t slot_i = (size_t)(sig_hash[p] ^ (sig_hash[p] >> 29)) & (cap - 1);
    for (;; slot_i = (slot_i + 1) & (cap - 1)) {
      const int r = table[slot_i];
      if (r < 0) {
        table[slot_i] = (int)rep.size();
        rep.push_back(p);
        return (int)rep.size() - 1;
      }
      if (sig_equal(rep[r], p)) return r;
    }
  };
  {
    const size_t blk = ((size_t)n_pt + sig_threads - 1) / sig_threads;
    size_t cap = 64;
    while (cap < 2 * blk + 16) cap <<= 1;
    std::vector<int>& table = hs->table;  // open addressing: run id, keyed by the signature hash; a slice per thread
    table.resize(cap * (size_t)sig_threads);
    std::vector<char> too_many((size_t)sig_threads, 0);
    host_parallel_for_t(n_pt, sig_threads, [&](int t, int lo, int hi) {
      int* tab = table.data() + cap * (size_t)t;
      std::fill(tab, tab + cap, -1);
      for (int p = lo; p < hi; ++p) {
        const int n = cnt[p + 1] - cnt[p];
        if (n > FB_MAXN) {
          too_many[t] = 1;
          return;
        }
        run_of[p] = n == 0 ? -1 : probe(tab, cap, loc_rep[t], p);
      }
    });
    for (char c : too_many)
      if (c) {
        delete b;
        return SFMHIP_ERR_UNSUPPORTED;  // more than FB_MAXN observations of one point
      }
    // the blocks' runs, block by block in their local order, into one table: global run ids in the points' order of first appearance
    size_t total = 0;
    for (const auto& r : loc_rep) total += r.size();
    size_t gcap = 64;
    while (gcap < 2 * total + 16) gcap <<= 1;
    std::vector<int> gtab(gcap, -1);
    for (int t = 0; t < sig_threads; ++t)
      for (int& lp : loc_rep[t]) lp = probe(gtab.data(), gcap, run_rep, lp);  // (the block's representative -> the global run)
    // global ids for the points, and every block's count per run
    const size_t R = run_rep.size();
    host_parallel_for_t(n_pt, sig_threads, [&](int t, int lo, int hi) {
      std::vector<int>& c = loc_cnt[t];
      c.assign(R, 0);
      const int* l2g = loc_rep[t].data();
      for (int p = lo; p < hi; ++p)
        if (run_of[p] >= 0) ++c[run_of[p] = l2g[run_of[p]]];
    });
    run_cnt.assign(R, 0);
    for (int t = 0; t < sig_threads; ++t)
      for (size_t r = 0; r < R; ++r) run_cnt[r] += loc_cnt[t][r];
  }
  lap_("  sig: hash table");
  std::vector<int> run_start(run_rep.size() + 1, 0);
  for (size_t r = 0; r < run_rep.size(); ++r) run_start[r + 1] = run_start[r] + run_cnt[r];
  std::vector<int>& order = hs->order;
  order.resize(run_start.back());
  {
    // block t starts run r behind the points of the blocks before it (loc_cnt becomes the blocks' write positions)
    for (size_t r = 0; r < run_rep.size(); ++r) {
      int at = run_start[r];
      for (int t = 0; t < sig_threads; ++t) {
        const int c = loc_cnt[t][r];
        loc_cnt[t][r] = at;
        at += c;
      }
    }
    host_parallel_for_t(n_pt, sig_threads, [&](int t, int lo, int hi) {
      int* at = loc_cnt[t].data();
      for (int p = lo; p < hi; ++p)
        if (run_of[p] >= 0) order[at[run_of[p]]++] = p;
    });
  }
  lap_("  sig: counting sort");
  b->np = (int)order.size();
  b->perm = order;
  std::vector<int>& optr = hs->optr;
  optr.resize((size_t)b->np + 1);
  {
    // prefix sums of the sorted points' observation counts: block sums first, then every block from its own start
    const int nth = host_threads(b->np);
    std::vector<long long> bsum((size_t)nth + 1, 0);
    host_parallel_for_t(b->np, nth, [&](int t, int lo, int hi) {
      long long sacc = 0;
      for (int sp = lo; sp < hi; ++sp) sacc += cnt[order[sp] + 1] - cnt[order[sp]];
      bsum[t + 1] = sacc;
    });
    for (int t = 0; t < nth; ++t) bsum[t + 1] += bsum[t];
    host_parallel_for_t(b->np, nth, [&](int t, int lo, int hi) {
      int at = (int)bsum[t];
      for (int sp = lo; sp < hi; ++sp) {
        optr[sp] = at;
        at += cnt[order[sp] + 1] - cnt[order[sp]];
      }
    });
    optr[b->np] = (int)bsum[nth];
  }
  if (getenv("SFMHIP_BA_CHECK_SETUP")) {
    // (tests: the grouping of one thread, pass by pass as it was written before round 6, must be what the threads produced)
    std::vector<int> rep1, of1(n_pt, -1), cnt1;
    size_t cap1 = 64;
    while (cap1 < 2 * (size_t)n_pt + 16) cap1 <<= 1;
    std::vector<int> tab1(cap1, -1);
    for (int p = 0; p < n_pt; ++p) {
      if (cnt[p + 1] == cnt[p]) continue;
      const size_t before = rep1.size();
      of1[p] = probe(tab1.data(), cap1, rep1, p);
      if (rep1.size() != before) cnt1.push_back(0);
      ++cnt1[of1[p]];
    }
    std::vector<int> start1(rep1.size() + 1, 0);
    for (size_t r = 0; r < rep1.size(); ++r) start1[r + 1] = start1[r] + cnt1[r];
    std::vector<int> order1((size_t)start1.back()), optr1((size_t)start1.back() + 1, 0);
    for (int p = 0; p < n_pt; ++p)
      if (of1[p] >= 0) order1[start1[of1[p]]++] = p;
    for (size_t sp = 0; sp < order1.size(); ++sp) optr1[sp + 1] = optr1[sp] + (cnt[order1[sp] + 1] - cnt[order1[sp]]);
    if (rep1 != run_rep || of1 != run_of || order1 != order || optr1 != optr) {
      fprintf(stderr, "sfmhip_ba_create: the threads' grouping differs from one thread's (SFMHIP_BA_CHECK_SETUP)\n");
      delete b;
      return SFMHIP_ERR_STATE;
    }
  }
  b->no = optr[b->np];
  std::vector<int>& ocam = hs->ocam;
  ocam.resize(b->no);
  b->obs_src.resize(b->no);
  b->h_cam_used.assign(n_cam, 0);
  lap_("  sig: optr + resizes");
  // the gather of a million observations is a cache-miss chain on one core: split it over a few (each marks the cameras it
  // meets in a list of its own; the lists are merged behind the threads)
  {
    const int nth = host_threads(b->np);
    std::vector<std::vector<unsigned char>> used((size_t)nth, std::vector<unsigned char>(n_cam, 0));
    host_parallel_for_t(b->np, nth, [&](int t, int lo, int hi) {
      unsigned char* mine = used[t].data();
      for (int sp = lo; sp < hi; ++sp) {
        const int p = order[sp];
        int w = optr[sp];
        for (int k = cnt[p]; k < cnt[p + 1]; ++k, ++w) {
          const int o = slot[k];
          b->obs_src[w] = o;
          ocam[w] = obs_cam[o];
          if (!mine[ocam[w]]) mine[ocam[w]] = 1;  // (written once: the threads' lists are neighbours in memory, and a store per
                                                  //  observation to a line another thread's list shares kept that line travelling)
          // (the coordinates are put in this order on the device: ba_permute_xy)
        }
      }
    });
    for (const auto& u : used)
      for (int c = 0; c < n_cam; ++c) b->h_cam_used[c] |= u[c];
  }
  lap_("signature sort + csr");
  // ---- chunks: runs of equal signature with strictly ascending cameras, n <= 10 -> MFMA path,
  //      classed by the width of the local Gram matrix: NB = ceil((6n+2)/16) column blocks
  std::vector<Chunk> chunks;
  std::vector<int> ids[8], sig_cams, fb;
  constexpr int SHORT_RUN = 12;  // runs of at most this many points go to the pair path
  // points per workgroup: 2 workgroups of 4 waves are resident per CU (register-bound), so the launch runs in
  // rounds of 512 workgroups; a wave takes 4 points per iteration (~3.7 us at n = 10) and a fixed ~7 iterations'
  // worth of prologue, reductions and scatter (s_memtime stamps, scripts/elim_stamps.py).  Pick the run length
  // that minimises rounds x (iterations per wave + fixed).  Measured at cfg4 (scripts/gpu_ba_elim_ab.py, stage
  // time per LM iteration): 400 workgroups of 4 waves 100 us; 800 of 2 waves 106 us; 200 of 8 waves 150 us (not a
  // matter of the two waves of a SIMD running in step: starting waves 4..7 up to 8 k cycles late changes nothing,
  // 148-151 us); 800 of 4 waves (2 rounds) 136 us.
  int target = 64;
  const std::vector<int>& gstart = run_start;  // first sorted point of every run, + np
  {
    std::vector<int> gsz;
    for (size_t gi = 0; gi + 1 < gstart.size(); ++gi)
      if (gstart[gi + 1] - gstart[gi] > SHORT_RUN) gsz.push_back(gstart[gi + 1] - gstart[gi]);
    double best = 1e300;
    for (int t = 32; t <= 1024; t += 4) {
      long long w = 0;
      for (int g : gsz) w += (g + t - 1) / t;
      // (a workgroup's iteration takes 4 waves x 4 points, or x 6 with ten lanes per point, whose iterations are a third longer:
      // the fixed part counts for fewer of them)
      const int ppi = elim_lp10() ? 24 : 16;
      const double cost = (double)((w + 511) / 512) * ((double)((t + ppi - 1) / ppi) + (elim_lp10() ? 5.5 : 7.0));
      if (cost < best) {
        best = cost;
        target = t;
      }
    }
    if (const char* e = getenv("SFMHIP_BA_ELIM")) {  // "waves,target" (measurement)
      int w_ = 0, t_ = 0;
      if (sscanf(e, "%d,%d", &w_, &t_) == 2 && (w_ == 2 || w_ == 4 || w_ == 8) && t_ >= 8) {
        b->elim_waves = w_;
        target = t_;
      }
    }
  }
  // One round of workgroups: when the runs cut at `target` leave resident slots empty (cfg4: 400 workgroups on 512
  // slots, so 112 CUs hold one workgroup and idle half the launch while 144 hold two), the largest pieces are cut
  // once more until the slots are full, and the launch lists the large pieces first: the dispatcher deals the first
  // n_cu workgroups one per CU, so every CU ends up with a large and a small piece or two small ones.  The busiest
  // SIMD then has 250 + 167 points instead of 500 (SFMHIP_BA_ELIM_FILL=0: the plain cut).
  const int slots = (getenv("SFMHIP_BA_ELIM_SLOTS") ? atoi(getenv("SFMHIP_BA_ELIM_SLOTS")) : 2) * std::max(b->ctx->n_cu, 1);  // (resident workgroups per CU: a measurement knob)
  const bool fill_env = !(getenv("SFMHIP_BA_ELIM_FILL") && atoi(getenv("SFMHIP_BA_ELIM_FILL")) == 0);
  std::vector<int> parts_of(gstart.size(), 0);
  {
    long long w = 0;
    for (size_t gi = 0; gi + 1 < gstart.size(); ++gi) {
      const int g = gstart[gi + 1] - gstart[gi];
      if (g > SHORT_RUN) w += (parts_of[gi] = (g + target - 1) / target);
    }
    if (fill_env && w > slots / 2 && w < slots) {
      // (a max-heap on the current piece size; a piece of fewer than 64 points is not worth another workgroup)
      std::vector<std::pair<double, size_t>> heap;
      for (size_t gi = 0; gi + 1 < gstart.size(); ++gi)
        if (parts_of[gi]) heap.push_back({(double)(gstart[gi + 1] - gstart[gi]) / parts_of[gi], gi});
      std::make_heap(heap.begin(), heap.end());
      while (w < slots && !heap.empty()) {
        std::pop_heap(heap.begin(), heap.end());
        const size_t gi = heap.back().second;
        heap.pop_back();
        const int g = gstart[gi + 1] - gstart[gi];
        if (g / (parts_of[gi] + 1) < 64) continue;
        ++parts_of[gi];
        ++w;
        heap.push_back({(double)g / parts_of[gi], gi});
        std::push_heap(heap.begin(), heap.end());
      }
    }
  }
  // Short runs: the pair path sums per camera pair instead of per run, which is what a camera list shared by a dozen points
  // wants -- when there are thousands of such lists.  The path itself costs four launches behind the elimination (33 us of a
  // 220 us iteration at cfg4, measured with ONE such point), so while nothing else needs it (no ragged, unsorted or > 10-camera
  // point) and the short runs are few, each becomes a small piece of the elimination: a workgroup among 512
  // (SFMHIP_BA_SHORT_PIECES = the most short runs that are turned into pieces, 0: none; scripts/gpu_short_runs_ab.py)
  bool short_as_pieces = false;
  {
    const int max_pieces = getenv("SFMHIP_BA_SHORT_PIECES") ? atoi(getenv("SFMHIP_BA_SHORT_PIECES")) : 512;  // (read per problem)
    int n_short = 0;
    bool pair_path_needed = false;
    for (size_t gi = 0; gi + 1 < gstart.size() && !pair_path_needed; ++gi) {
      const int sp = gstart[gi], n = optr[sp + 1] - optr[sp];
      bool strict = true;
      for (int k = 1; k < n; ++k) strict = strict && ocam[optr[sp] + k - 1] < ocam[optr[sp] + k];
      if (!(n <= 10 && strict)) pair_path_needed = true;
      else if (gstart[gi + 1] - sp <= SHORT_RUN) ++n_short;
    }
    short_as_pieces = !pair_path_needed && n_short > 0 && n_short <= max_pieces;
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
      if (e - sp <= SHORT_RUN && !short_as_pieces) {
        sig_cams.resize(so);  // (a camera list shared by few points: the pair path, per-pair instead of per-run sums)
        for (int q = sp; q < e; ++q) fb.push_back(q);
      } else {
        const int parts = std::max(parts_of[gi], 1);
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
  if (fill_env)
    for (auto& l : ids)  // large pieces first (stable: equal sizes keep the point order)
      std::stable_sort(l.begin(), l.end(), [&](int a, int c) { return chunks[a].cnt > chunks[c].cnt; });
  lap_("chunks");
  // ---- the gather lists of the slab epilogue (ba_gather_slabs): for every destination in `red` the slab entries that
  // add to it, in chunk order; list 0 for a full linearisation, list 1 for the norms-only mode (diagonal only)
  std::vector<int> gth_ptr[2], gth_dest[2], grow_ptr, grow_id, grow_colmap;
  std::vector<unsigned> gth_src[2];
  std::vector<int4> grow_src, grow_hdr, grow_head, grow_over;
  if (b->elim_deterministic && chunks.size() * (size_t)ELIM_SLAB >= ((size_t)1 << 31)) {
    // the gather lists address a slab entry with 31 bits (bit 31 carries the sign): past ~740 000 chunks the elimination goes
    // back to the atomic epilogue -- said out loud, because the sums are then no longer the same bit patterns run after run
    fprintf(stderr, "sfmhip_ba: %zu chunks exceed the slab epilogue's 31-bit offsets; atomic epilogue (not run-to-run identical)\n",
            chunks.size());
    b->elim_deterministic = false;
  }
  if (b->elim_deterministic && !chunks.empty()) {
    const int ld = b->ld, fo = 6 * n_cam;
    const long long ssz = (long long)b->ssz, o_g = ssz, o_gF = ssz + ld, o_dc = ssz + 2LL * ld, o_sc = ssz + 3LL * ld;
    std::vector<std::pair<long long, unsigned>> ent[2];  // (destination, source | sign)
    // rows of S by their own kernel role while a wave's accumulator fits the default LDS limit (until round 6 the accumulator was
    // a whole row, ld entries zeroed and scanned whatever the row held: at 640 cameras that outweighed what the row-wise reads
    // save, and rows of more than 3072 columns went through the per-destination lists -- scripts/gpu_gather_bits.py: 640 cameras
    // 3305 -> 3619 it/s, 1000: 2805 -> 3267, the same bits as the whole-row form wherever that ran).
    // SFMHIP_BA_GATHER_ROWS = 0: never (the per-destination lists: measurement, and the check of the row lists)
    const int rows_env = getenv("SFMHIP_BA_GATHER_ROWS") ? atoi(getenv("SFMHIP_BA_GATHER_ROWS")) : 1;
    // (round 6: a row's accumulator holds only the columns the row can have -- the cameras that share a run with the row's camera,
    // the focal column, g's / the diagonal's / F^T b's entries --, not all ld of them: a wave zeroed and scanned ld entries whatever
    // the row held, which is what kept rows of 640 cameras and more on the per-destination lists)
    // per camera: the cameras of the runs it is in, ascending (tl_flat[tl_off[c] .. tl_off[c + 1])): a bit row per camera first
    std::vector<int> tl_off(n_cam + 1, 0), tl_flat;
    {
      const int wpr_ = (n_cam + 63) / 64;
      std::vector<unsigned long long> bits((size_t)n_cam * wpr_, 0ull);
      std::vector<char> seen_sig(sig_cams.size() + 1, 0);
      for (const Chunk& ch : chunks) {
        if (seen_sig[ch.sig_off]) continue;
        seen_sig[ch.sig_off] = 1;
        for (int a = 0; a < ch.n; ++a) {
          unsigned long long* row = bits.data() + (size_t)sig_cams[ch.sig_off + a] * wpr_;
          for (int c2 = 0; c2 < ch.n; ++c2) row[sig_cams[ch.sig_off + c2] >> 6] |= 1ull << (sig_cams[ch.sig_off + c2] & 63);
        }
      }
      for (int c = 0; c < n_cam; ++c) {
        for (int w = 0; w < wpr_; ++w)
          for (unsigned long long m = bits[(size_t)c * wpr_ + w]; m; m &= m - 1) tl_flat.push_back(64 * w + __builtin_ctzll(m));
        tl_off[c + 1] = (int)tl_flat.size();
      }
    }
    size_t accw = 64;  // a wave's accumulator: 6 entries per camera of the longest list + 4, in whole 64s
    for (int c = 0; c < n_cam; ++c) accw = std::max(accw, (6 * (size_t)(tl_off[c + 1] - tl_off[c]) + 4 + 63) / 64 * 64);
    const bool use_rows = accw * 8 <= 65536 && rows_env != 0;
    b->grow_waves = accw * 8 * 4 <= 65536 ? 4 : accw * 8 * 2 <= 65536 ? 2 : 1;
    b->grow_accw = (int)accw;
    if (use_rows) {
      std::vector<int> cntr((size_t)fo + 1, 0);
      for (const Chunk& ch : chunks)
        for (int sl = 0; sl < ch.n; ++sl)
          for (int i = 0; i < 6; ++i) ++cntr[(size_t)6 * sig_cams[ch.sig_off + sl] + i + 1];
      for (int r = 0; r < fo; ++r) cntr[r + 1] += cntr[r];
      grow_src.resize((size_t)cntr[fo]);
      std::vector<int> pos(cntr.begin(), cntr.end() - 1);
      // the cameras' lists, each behind its length: a row's header points at its camera's
      std::vector<int> clist_of(n_cam, 0);
      for (int c = 0; c < n_cam; ++c) {
        if (tl_off[c + 1] == tl_off[c]) continue;
        grow_colmap.push_back(tl_off[c + 1] - tl_off[c]);
        clist_of[c] = (int)grow_colmap.size();
        grow_colmap.insert(grow_colmap.end(), tl_flat.begin() + tl_off[c], tl_flat.begin() + tl_off[c + 1]);
      }
      // (signature = offset of its camera list, the row's camera's place in it) -> offset of the column map (64 ints)
      std::vector<int> cmap_of(sig_cams.size() + 1, -1);
      for (size_t c = 0; c < chunks.size(); ++c) {  // chunk order inside every row
        const Chunk& ch = chunks[c];
        const int n = ch.n, NBc = (6 * n + 2 + 15) / 16;
        for (int sl = 0; sl < n; ++sl) {
          int& cm = cmap_of[ch.sig_off + sl];
          if (cm < 0) {
            // local column lc < 6 n of the signature -> its place in the accumulator of a row of camera sig[sl]: 6 * (the rank of
            // camera sig[lc / 6] in that camera's list) + lc % 6; behind the nT = 6 * |list| columns of S: the focal column, g's
            // entry, the diagonal's and F^T b's (ba_gather_rows)
            cm = (int)grow_colmap.size();
            const int row_cam = sig_cams[ch.sig_off + sl];
            const int* tl = tl_flat.data() + tl_off[row_cam];
            const int nT = 6 * (tl_off[row_cam + 1] - tl_off[row_cam]);
            grow_colmap.resize((size_t)cm + 64);
            int* out = grow_colmap.data() + cm;
            for (int a = 0, t = 0; a < n; ++a) {  // (both lists ascend: one walk gives every camera's rank)
              while (tl[t] != sig_cams[ch.sig_off + a]) ++t;
              for (int i = 0; i < 6; ++i) out[6 * a + i] = 6 * t + i;
            }
            for (int lc = 6 * n; lc < 62; ++lc) out[lc] = lc == 6 * n ? nT : nT + 1;
            out[62] = nT + 2, out[63] = nT + 3;
          }
          for (int i = 0; i < 6; ++i) {
            const int lr = 6 * sl + i, ti = lr >> 4;
            const int t0 = ti * NBc - ti * (ti - 1) / 2;  // tile (ti, ti)
            const int roff = (t0 * 4 + ((lr & 15) >> 2)) * 64 + (lr & 3) * 16 - 256 * ti;
            const int dcr = (i * 6 - i * (i - 1) / 2) * FP + sl, gfr = (27 + i) * FP + sl;
            grow_src[(size_t)pos[(size_t)6 * sig_cams[ch.sig_off + sl] + i]++] =
                make_int4((int)(unsigned)(c * (size_t)ELIM_SLAB), lr | (n << 8) | (roff << 16), cm, dcr | (gfr << 16));
          }
        }
      }
      for (int r = 0; r < fo; ++r)
        if (cntr[r + 1] > cntr[r]) {
          grow_ptr.push_back(cntr[r]);
          grow_id.push_back(r);
        }
      grow_ptr.push_back(cntr[fo]);
      // a row's first 32 records in a table of their own (fixed stride), the rest in one overflow list
      for (size_t r = 0; r < grow_id.size(); ++r) {
        const int k0 = grow_ptr[r], cnt = grow_ptr[r + 1] - k0;
        grow_hdr.push_back(make_int4(grow_id[r], cnt, (int)grow_over.size(), clist_of[grow_id[r] / 6]));
        for (int k = 0; k < 32; ++k) grow_head.push_back(k < cnt ? grow_src[(size_t)k0 + k] : make_int4(0, 0, 0, 0));
        for (int k = 32; k < cnt; ++k) grow_over.push_back(grow_src[(size_t)k0 + k]);
      }
      if (grow_over.empty()) grow_over.push_back(make_int4(0, 0, 0, 0));
    }
    const long long GMAX = -1;                            // (sorts first; the kernel takes the rank's slot as an argument)
    for (size_t c = 0; c < chunks.size(); ++c) {
      const Chunk& ch = chunks[c];
      const int n = ch.n, NBc = (6 * n + 2 + 15) / 16, NTc = NBc * (NBc + 1) / 2;
      const int* cams = sig_cams.data() + ch.sig_off;
      const unsigned base = (unsigned)(c * (size_t)ELIM_SLAB);
      auto gidx = [&](int l) { return l < 6 * n ? 6 * cams[l / 6] + l % 6 : l == 6 * n ? fo : l == 6 * n + 1 ? -2 : -1; };
      // (ba_gather_rows: the cameras' rows from the row lists above, the focal row chunk by chunk -- nothing of the Gram block
      // goes through the destination lists then, and walking its NTc * 256 entries per chunk was half of this stage's time)
      for (int idx = 0; !use_rows && idx < NTc * 256; ++idx) {  // the Gram block, as the kernel lays it out
        int t = idx >> 8, ti = 0;
        while (t >= NBc - ti) {
          t -= NBc - ti;
          ++ti;
        }
        const int tj = ti + t, gg = (idx >> 6) & 3, ln = idx & 63;
        const int lr = 16 * ti + (ln >> 4) + 4 * gg, lc = 16 * tj + (ln & 15);
        const int gr = gidx(lr), gc = gidx(lc);
        if (gr < 0 || lr > lc || gc == -1) continue;
        ent[0].push_back({gc >= 0 ? (long long)gr * ld + gc : o_g + gr, (base + idx) | 0x80000000u});  // S -= Gram (F^T F folded in)
      }
      for (int e = 0; e < 33; ++e)
        for (int slot = 0; slot < n; ++slot) {
          const unsigned sidx = base + ELIM_SLAB_FF + e * FP + slot;
          const int r0 = 6 * cams[slot];
          if (e < 21) {
            int i = 0, rem = e;
            while (rem >= 6 - i) {
              rem -= 6 - i;
              ++i;
            }
            if (rem == 0) {
              if (!use_rows) ent[0].push_back({o_dc + r0 + i, sidx});
              ent[1].push_back({o_dc + r0 + i, sidx});
            }
          } else if (e >= 27) {
            if (!use_rows) ent[0].push_back({o_gF + r0 + e - 27, sidx});
          }
        }
      const unsigned tail = base + ELIM_SLAB_FF + 36 * FP;
      ent[1].push_back({o_dc + fo, tail});
      if (!use_rows) {
        ent[0].push_back({o_dc + fo, tail});
        ent[0].push_back({o_gF + fo, tail + 1});
        ent[0].push_back({o_sc + 0, tail + 2});
        ent[0].push_back({GMAX, tail + 3});
        ent[0].push_back({o_sc + 2, tail + 4});
      }
    }
    for (int m = 0; m < 2; ++m) {
      const size_t range = (size_t)(o_sc + SC + 64) + 2;   // destinations + the GMAX key shifted to 0
      if (ent[m].size() * 16 < range) {
        // few entries for the range (the row lists carry S: what is left are the diagonal's entries): a stable sort of the
        // entries instead of three passes over ld^2 counters (cfg4: 31 k entries, 1.5 M destinations)
        std::stable_sort(ent[m].begin(), ent[m].end(), [](const std::pair<long long, unsigned>& a, const std::pair<long long, unsigned>& c) { return a.first < c.first; });
        gth_src[m].resize(ent[m].size());
        for (size_t k = 0; k < ent[m].size(); ++k) {
          gth_src[m][k] = ent[m][k].second;
          if (k == 0 || ent[m][k].first != ent[m][k - 1].first) {
            gth_ptr[m].push_back((int)k);
            gth_dest[m].push_back((int)ent[m][k].first);
          }
        }
        gth_ptr[m].push_back((int)ent[m].size());
        continue;
      }
      // counting sort by destination (stable: a destination's sources stay in chunk order)
      std::vector<int> cnt(range + 1, 0);
      for (const auto& e : ent[m]) ++cnt[(size_t)(e.first + 1) + 1];
      for (size_t k = 0; k < range; ++k) cnt[k + 1] += cnt[k];
      gth_src[m].resize(ent[m].size());
      {
        std::vector<int> pos(cnt.begin(), cnt.end() - 1);
        for (const auto& e : ent[m]) gth_src[m][(size_t)pos[(size_t)(e.first + 1)]++] = e.second;
      }
      for (size_t k = 0; k < range; ++k)
        if (cnt[k + 1] > cnt[k]) {
          gth_ptr[m].push_back(cnt[k]);
          gth_dest[m].push_back((int)((long long)k - 1));
        }
      gth_ptr[m].push_back((int)ent[m].size());
    }
  }
  lap_("gather lists");
  // ---- camera co-visibility (one bit row per camera) for the dissection of the reduced system
  if (n_cam <= 4096) {  // (from one camera on: a small system is one front)
    const int wpr = (n_cam + 63) / 64;
    b->h_adj.assign((size_t)n_cam * wpr, 0ull);
    auto add_clique = [&](const int* cs, int n) {
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) b->h_adj[(size_t)cs[i] * wpr + (cs[j] >> 6)] |= 1ull << (cs[j] & 63);
    };
    std::vector<char> is_fb(b->np, 0);
    for (int q : fb) is_fb[q] = 1;
    for (size_t gi = 0; gi + 1 < gstart.size(); ++gi) {
      const int sp = gstart[gi];
      if (!is_fb[sp]) add_clique(&ocam[optr[sp]], optr[sp + 1] - optr[sp]);  // one signature per run
    }
    for (int q : fb) add_clique(&ocam[optr[q]], optr[q + 1] - optr[q]);
    lap_("camera graph");
  }
  // ---- the pair path's lists (ba_pp_points / ba_pp_pairs / ba_cam_blocks): its points in ascending sorted order, a
  //      row of T per observation, the camera-major list of those observations, and per camera pair that a point
  //      sees together the (row of a, row of b) entries
  std::sort(fb.begin(), fb.end());
  std::vector<int> cptr(n_cam + 1, 0), cpt, pp_obase(fb.size() + 1, 0), pair_ptr(1, 0);
  std::vector<int2> cslot, pair_cams, pair_ent;
  std::vector<double> cxy;
  {
    for (size_t i = 0; i < fb.size(); ++i) pp_obase[i + 1] = pp_obase[i] + (optr[fb[i] + 1] - optr[fb[i]]);
    const size_t nfo = (size_t)pp_obase[fb.size()];
    cpt.resize(nfo);
    cslot.resize(nfo);
    cxy.resize(2 * nfo);
    b->cxy_src.resize(nfo);
    for (int sp : fb)
      for (int k = optr[sp]; k < optr[sp + 1]; ++k) cptr[ocam[k] + 1]++;
    for (int c = 0; c < n_cam; ++c) cptr[c + 1] += cptr[c];
    std::vector<int> fill(cptr.begin(), cptr.end() - 1);
    struct PE {
      long long key;
      int a, b;
    };
    std::vector<PE> pes;
    for (size_t i = 0; i < fb.size(); ++i) {  // ascending sorted point index: the order inside a camera is the stable one
      const int sp = fb[i], k0 = optr[sp], n = optr[sp + 1] - k0;
      for (int o = 0; o < n; ++o) {
        const int dst = fill[ocam[k0 + o]]++;
        cpt[dst] = sp;
        cslot[dst] = make_int2(pp_obase[i] + o, (int)i);
        b->cxy_src[dst] = k0 + o;
        cxy[2 * (size_t)dst] = obs_xy[2 * (size_t)b->obs_src[k0 + o]];
        cxy[2 * (size_t)dst + 1] = obs_xy[2 * (size_t)b->obs_src[k0 + o] + 1];
        for (int o2 = o + 1; o2 < n; ++o2) {
          int ca = ocam[k0 + o], cb = ocam[k0 + o2], ra = pp_obase[i] + o, rb = pp_obase[i] + o2;
          if (ca > cb) std::swap(ca, cb), std::swap(ra, rb);
          pes.push_back(PE{(long long)ca * n_cam + cb, ra, rb});
        }
      }
    }
    std::sort(pes.begin(), pes.end(), [](const PE& x, const PE& y) { return x.key != y.key ? x.key < y.key : (x.a != y.a ? x.a < y.a : x.b < y.b); });
    for (size_t e = 0; e < pes.size(); ++e) {
      if (e == 0 || pes[e].key != pes[e - 1].key) {
        if (e) pair_ptr.push_back((int)e);
        pair_cams.push_back(make_int2((int)(pes[e].key / n_cam), (int)(pes[e].key % n_cam)));
      }
      pair_ent.push_back(make_int2(pes[e].a, pes[e].b));
    }
    if (!pes.empty()) pair_ptr.push_back((int)pes.size());
    // (a workgroup per (camera, slice): ~1024 entries each, so that the 69-value block reduction is paid once per four
    // entries of a thread: 57 -> 25 us at 200 cameras x 1000 observations)
    b->cam_split = (int)std::max<size_t>(1, std::min<size_t>(64, nfo / (size_t)std::max(n_cam, 1) / 1024));
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
  // [S | g | F^T b | diag | SC scalars + one slot per rank (<= 64) | the step evaluation's 8 sums, a
  // scratch double, the factorisation's status]: the tail past the all-reduced part is zeroed with
  // the rest at every linearisation and comes back to the host in the same copy as the scalars
  b->red_count = b->ssz + 3 * (size_t)b->ld + SC + 64 + RED2_N + b->ssz;  // ... | X (chol_step2), zeroed with the rest
#define BA_A(ptr, n)                         \
  if (rc == SFMHIP_OK) rc = ba_alloc(b, &(ptr), (size_t)(n))
  BA_A(d_optr, b->np + 1);
  BA_A(d_ocam, b->no);
  BA_A(d_oxy, b->no);
  BA_A(b->d_obs_src, b->no);
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
  BA_A(d.iscale_p, 3 * (size_t)b->np);
  BA_A(d.scale_f, 1);
  BA_A(d.diag, b->ld);
  BA_A(d.red, b->red_count);
  BA_A(d.z, b->ld);
  if (rc == SFMHIP_OK) {
    d.red2 = d.red + b->ssz + 3 * (size_t)b->ld + SC + 64;
    d.info = (int*)(d.red2 + RED2_INFO);
    d.xinv = d.red2 + RED2_N;
  }
  BA_A(b->d_cam_used, n_cam);
  BA_A(b->d_flag, 2);
  BA_A(b->d_chunks, chunks.size());
  if (!gth_dest[0].empty() || !gth_dest[1].empty() || !grow_id.empty()) {
    BA_A(b->d_slab, chunks.size() * (size_t)ELIM_SLAB);
    for (int m = 0; m < 2; ++m) {
      BA_A(b->d_gth_ptr[m], gth_ptr[m].size());
      BA_A(b->d_gth_src[m], gth_src[m].size());
      BA_A(b->d_gth_dest[m], gth_dest[m].size());
    }
    if (!grow_id.empty()) {
      BA_A(b->d_grow_hdr, grow_hdr.size());
      BA_A(b->d_grow_head, grow_head.size());
      BA_A(b->d_grow_src, grow_over.size());
      BA_A(b->d_grow_colmap, grow_colmap.size());
    }
  }
  for (int c = 0; c < 8; ++c) BA_A(b->d_chunk_ids[c], ids[c].size());
  BA_A(b->d_sig_cams, sig_cams.size());
  BA_A(b->d_cptr, cptr.size());
  BA_A(b->d_cpt, cpt.size());
  BA_A(b->d_cxy, cpt.size());
  BA_A(b->d_fb_points, fb.size());
  BA_A(b->d_pp_obase, pp_obase.size());
  BA_A(b->d_cslot, cslot.size());
  BA_A(b->d_ppT, 18 * cslot.size());
  BA_A(b->d_tfu, 6 * fb.size());
  b->n_pp_part = (int)((fb.size() * PP_LANES + 255) / 256);
  BA_A(b->d_pp_part, 8 * (size_t)b->n_pp_part);
  if (b->cam_split > 1) {
    BA_A(b->d_cb_part, 66 * (size_t)n_cam * b->cam_split);
    BA_A(b->d_cb_cnt, n_cam);
    if (rc == SFMHIP_OK && hipMemset(b->d_cb_cnt, 0, sizeof(int) * (size_t)n_cam) != hipSuccess) rc = SFMHIP_ERR_HIP;
  }
  {
    // (measurement: 0 = all of X, DENSE_XB = blocks even below the size that switches them on.  The block width itself is not a
    // knob: chol_back_block's registers and LDS and the sizes of d_back_part are built for DENSE_XB tile columns)
    static const int xb_env = [] {
      const char* e = getenv("SFMHIP_BA_DENSE_XB");
      if (!e) return -1;
      const int v = atoi(e);
      if (v != 0 && v != DENSE_XB) {
        fprintf(stderr, "[sfmhip-ba] SFMHIP_BA_DENSE_XB=%d ignored: 0 (all of X) or %d (diagonal blocks) are the choices\n", v, DENSE_XB);
        return -1;
      }
      return v;
    }();
    const int nt = b->ld / CB;
    b->dense_xb = xb_env == 0 ? 0 : (xb_env > 0 || nt >= DENSE_XB_MIN_NT) ? DENSE_XB : 0;
    if (b->dense_xb) {
      BA_A(b->d_back_part, 2 * DENSE_XB * DENSE_XB * CB);  // (two sets: a launch reads the one the launch before wrote)
      // X outside its diagonal blocks is never written -- but the panel workgroups of a block's first pair of panels read the
      // block's rows at the pending columns, which lie in the block before: zero once and for all (chol_x_reset clears the
      // blocks themselves at every linearisation)
      if (rc == SFMHIP_OK && hipMemset(d.xinv, 0, sizeof(double) * b->ssz) != hipSuccess) rc = SFMHIP_ERR_HIP;
    }
  }
  BA_A(b->d_pair_ptr, pair_ptr.size());
  BA_A(b->d_pair_cams, pair_cams.size());
  BA_A(b->d_pair_ent, pair_ent.size());
#undef BA_A
  if (rc != SFMHIP_OK) {
    sfmhip_ba_destroy(b);
    return rc;
  }
  d.optr = d_optr;
  d.ocam = d_ocam;
  d.oxy = d_oxy;
  b->d_oxy_w = d_oxy;
  for (int c = 0; c < 8; ++c) b->n_chunk_ids[c] = (int)ids[c].size();
  b->n_fb = (int)fb.size();
  auto up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  lap_("hipMalloc x27");
  SFM_HIP_TRY(up(d_optr, optr.data(), optr.size() * 4));
  SFM_HIP_TRY(up(d_ocam, ocam.data(), ocam.size() * 4));
  SFM_HIP_TRY(up(b->d_obs_src, b->obs_src.data(), b->obs_src.size() * 4));
  SFM_TRY(ba_upload_xy(b, obs_xy, n_obs));
  SFM_HIP_TRY(up(b->d_cam_used, b->h_cam_used.data(), n_cam));
  SFM_HIP_TRY(up(b->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk)));
  b->n_chunks = (int)chunks.size();
  if (!chunks.empty()) {
    std::vector<int> all(chunks.size());
    for (size_t i = 0; i < all.size(); ++i) all[i] = (int)i;
    std::stable_sort(all.begin(), all.end(), [&](int a, int c) { return chunks[a].cnt > chunks[c].cnt; });
    std::vector<int> desc(16 * all.size(), 0);
    for (size_t i = 0; i < all.size(); ++i) {
      const Chunk& c = chunks[all[i]];
      int* r = &desc[16 * i];
      r[0] = c.n, r[1] = c.p0, r[2] = c.cnt, r[3] = optr[c.p0];
      for (int k = 0; k < c.n && k < 10; ++k) r[4 + k] = sig_cams[c.sig_off + k];
    }
    SFM_TRY(ba_alloc(b, &b->d_bs_ids, desc.size()));
    SFM_HIP_TRY(hipMemcpy(b->d_bs_ids, desc.data(), desc.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  for (int m = 0; m < 2 && b->d_slab; ++m) {
    SFM_HIP_TRY(up(b->d_gth_ptr[m], gth_ptr[m].data(), gth_ptr[m].size() * 4));
    SFM_HIP_TRY(up(b->d_gth_src[m], gth_src[m].data(), gth_src[m].size() * 4));
    SFM_HIP_TRY(up(b->d_gth_dest[m], gth_dest[m].data(), gth_dest[m].size() * 4));
    b->n_gth[m] = (int)gth_dest[m].size();
  }
  if (b->d_slab && !grow_id.empty()) {
    SFM_HIP_TRY(up(b->d_grow_hdr, grow_hdr.data(), grow_hdr.size() * sizeof(int4)));
    SFM_HIP_TRY(up(b->d_grow_head, grow_head.data(), grow_head.size() * sizeof(int4)));
    SFM_HIP_TRY(up(b->d_grow_src, grow_over.data(), grow_over.size() * sizeof(int4)));
    SFM_HIP_TRY(up(b->d_grow_colmap, grow_colmap.data(), grow_colmap.size() * 4));
    b->n_grow = (int)grow_id.size();
  }
  for (int c = 0; c < 8; ++c) SFM_HIP_TRY(up(b->d_chunk_ids[c], ids[c].data(), ids[c].size() * 4));
  SFM_HIP_TRY(up(b->d_sig_cams, sig_cams.data(), sig_cams.size() * 4));
  SFM_HIP_TRY(up(b->d_cptr, cptr.data(), cptr.size() * 4));
  SFM_HIP_TRY(up(b->d_cpt, cpt.data(), cpt.size() * 4));
  SFM_HIP_TRY(up(b->d_cxy, cxy.data(), cxy.size() * 8));
  SFM_HIP_TRY(up(b->d_fb_points, fb.data(), fb.size() * 4));
  SFM_HIP_TRY(up(b->d_pp_obase, pp_obase.data(), pp_obase.size() * 4));
  SFM_HIP_TRY(up(b->d_cslot, cslot.data(), cslot.size() * sizeof(int2)));
  SFM_HIP_TRY(up(b->d_pair_ptr, pair_ptr.data(), pair_ptr.size() * 4));
  SFM_HIP_TRY(up(b->d_pair_cams, pair_cams.data(), pair_cams.size() * sizeof(int2)));
  SFM_HIP_TRY(up(b->d_pair_ent, pair_ent.data(), pair_ent.size() * sizeof(int2)));
  b->n_pairs_pp = (int)pair_cams.size();
  lap_("uploads");
  const size_t sc_bytes = (sizeof(double) * (SC + 64 + RED2_N + 1) + 255) & ~(size_t)255, ring_bytes = sizeof(LmDev) * LM_RING;
  if (arena) {
    // (the records the device writes to the host: the context's pinned block, made once -- 0.5 ms per problem otherwise)
    if (ctx->ba_pinned_bytes < sc_bytes + ring_bytes) {
      if (ctx->ba_pinned) hipHostFree(ctx->ba_pinned);
      ctx->ba_pinned = nullptr, ctx->ba_pinned_bytes = 0;
      SFM_HIP_TRY(hipHostMalloc(&ctx->ba_pinned, sc_bytes + ring_bytes, hipHostMallocDefault));
      ctx->ba_pinned_bytes = sc_bytes + ring_bytes;
    }
    b->pinned_shared = true;
    b->h_sc = (double*)ctx->ba_pinned;
    b->h_ring = (LmDev*)((char*)ctx->ba_pinned + sc_bytes);
  } else {
    SFM_HIP_TRY(hipHostMalloc((void**)&b->h_sc, sizeof(double) * (SC + 64 + RED2_N + 1), hipHostMallocDefault));
    SFM_HIP_TRY(hipHostMalloc((void**)&b->h_ring, ring_bytes, hipHostMallocDefault));
  }
  SFM_HIP_TRY(hipHostGetDevicePointer((void**)&b->h_sc_dev, b->h_sc, 0));
  b->h_sc[SC + 64 + RED2_N] = 0.0;
  // the trust-region record, the host's ring of copies of it, the step evaluation's slots (runs x split <= 8, blocks of 64
  // points of ba_backsub, then a pair per front / per wave of ba_cand_cams)
  memset(b->h_ring, 0, sizeof(LmDev) * LM_RING);  // (a shared block: the previous problem's records must not match this one's sequence numbers)
  SFM_HIP_TRY(hipHostGetDevicePointer((void**)&b->h_ring_dev, b->h_ring, 0));
  SFM_TRY(ba_alloc(b, &b->d_lm, 1));
  b->step_part_n = 4 * ((size_t)b->n_chunks * 8 + ((size_t)b->np + 63) / 64 + 8) + 2 * ((size_t)n_cam + 64);
  SFM_TRY(ba_alloc(b, &b->d_step_part, b->step_part_n));
  {
    std::vector<unsigned long long> pend(b->step_part_n, STEP_PENDING);  // (a slot is a NaN no sum produces until its workgroup has written it)
    SFM_HIP_TRY(hipMemcpy(b->d_step_part, pend.data(), sizeof(double) * b->step_part_n, hipMemcpyHostToDevice));
  }
  d.lm = nullptr;
  d.step_part = b->d_step_part;
  d.step_total = 0, d.cam_parts = 0, d.decide_here = 0, d.rank = 0;
  d.lm_host = (double*)b->h_ring_dev;
  for (auto& e : b->ev) SFM_HIP_TRY(hipEventCreate(&e));
  b->h_pts_in.assign(3 * (size_t)n_pt, 0.0);
  lap_("pinned + events");
  *out = b;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_set_allreduce(sfmhip_ba* b, sfmhip_allreduce_fn fn, void* user, int rank, int world) {
  if (!b || world < 1 || rank < 0 || rank >= world || world > 64) return SFMHIP_ERR_ARG;
  if (world > 1 && !fn) return SFMHIP_ERR_ARG;
  if (world != b->world || rank != b->rank) {
    // the dissection plan and the sparse exchange list are built from the UNION of the ranks' camera graphs, and the
    // Jacobi scale from the sum of their column norms: both belong to the (rank, world) they were made for
    b->nd_ready = false;
    b->nd_on = false;
    b->n_xblocks = 0;
    b->scale_ready = false;
    b->lm = LmState();  // (an LM run in progress belongs to the old partition: the next iterate starts over)
  }
  b->allreduce = fn;
  b->allreduce_user = user;
  b->rank = rank;
  b->d.rank = rank;
  b->world = world;
  if (world > 1 && !b->d_red_pack) {
    SFM_HIP_TRY(hipSetDevice(b->ctx->device));
    SFM_TRY(ba_alloc(b, &b->d_red_pack, (size_t)b->ld * (b->ld + 1) / 2 + 3 * (size_t)b->ld + SC + 64));
  }
  return SFMHIP_OK;
}

// device -> the context's pinned block by stores of a kernel (what the LM loop's records do): a blit out of a fresh allocation
// cost 8 ms the first time (the runtime's set-up for that block of memory; measured in whichever call met the block first)
__global__ __launch_bounds__(256) void ba_export_params(const double* __restrict__ cams, size_t n_c, const double* __restrict__ pts, size_t n_p,
                                                        const double* __restrict__ focal, double* __restrict__ host) {
  const size_t n = n_c + n_p + 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    host[i] = i < n_c ? cams[i] : i < n_c + n_p ? pts[i - n_c] : *focal;
}

extern "C" int sfmhip_ba_set_params(sfmhip_ba* b, const double* cams6, const double* pts3, double focal) {
  if (!b || !cams6 || (!pts3 && b->np_in)) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  hipStream_t st = b->ctx->stream;
  if (b->np_in) memcpy(b->h_pts_in.data(), pts3, sizeof(double) * 3 * (size_t)b->np_in);
  // through the context's pinned block (synchronous use: free again at the return): a copy from pageable memory goes through the
  // runtime's own staging, whose first use in a process costs 8-27 ms -- measured inside whichever call met it first
  const size_t n_c = 6 * (size_t)b->nc, n_p = 3 * (size_t)b->np;
  void* pin = nullptr;
  std::vector<double> unpinned;  // (pinning refused: the runtime's own staging)
  if (sfm_ctx_pinned(b->ctx, sizeof(double) * (n_c + n_p + 1), &pin) != SFMHIP_OK) {
    (void)hipGetLastError();
    unpinned.resize(n_c + n_p + 1);
    pin = unpinned.data();
  }
  double* const hc = (double*)pin;
  double* const sorted = hc + n_c;
  memcpy(hc, cams6, sizeof(double) * n_c);
  host_parallel_for(b->np, [&](int lo, int hi) {
    for (int sp = lo; sp < hi; ++sp)
      for (int j = 0; j < 3; ++j) sorted[3 * (size_t)sp + j] = pts3[3 * (size_t)b->perm[sp] + j];
  });
  sorted[n_p] = focal;
  SFM_HIP_TRY(hipMemcpyAsync(b->d.cams, hc, sizeof(double) * n_c, hipMemcpyHostToDevice, st));
  if (b->np) SFM_HIP_TRY(hipMemcpyAsync(b->d.pts, sorted, sizeof(double) * n_p, hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemcpyAsync(b->d.focal, sorted + n_p, sizeof(double), hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  b->scale_ready = false;
  b->camd_valid = false;
  b->lm.started = false;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_get_params(sfmhip_ba* b, double* cams6, double* pts3, double* focal) {
  if (!b) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  hipStream_t st = b->ctx->stream;
  const size_t n_c = 6 * (size_t)b->nc, n_p = 3 * (size_t)b->np;
  void* pin = nullptr;
  std::vector<double> unpinned;
  const bool pinned = sfm_ctx_pinned(b->ctx, sizeof(double) * (n_c + n_p + 1), &pin) == SFMHIP_OK;  // (as in sfmhip_ba_set_params)
  if (!pinned) {
    (void)hipGetLastError();
    unpinned.resize(n_c + n_p + 1);
    pin = unpinned.data();
  }
  double* const hc = (double*)pin;
  double* const sorted = hc + n_c;
  if (pinned) {
    double* hdev = nullptr;
    SFM_HIP_TRY(hipHostGetDevicePointer((void**)&hdev, hc, 0));
    hipLaunchKernelGGL(ba_export_params, dim3((unsigned)std::min<size_t>((n_c + n_p) / 1024 + 1, 512)), dim3(256), 0, st, b->d.cams, n_c, b->d.pts,
                       n_p, b->d.focal, hdev);
    SFM_HIP_TRY(hipGetLastError());
  } else {
    SFM_HIP_TRY(hipMemcpyAsync(hc, b->d.cams, sizeof(double) * n_c, hipMemcpyDeviceToHost, st));
    if (n_p) SFM_HIP_TRY(hipMemcpyAsync(sorted, b->d.pts, sizeof(double) * n_p, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(sorted + n_p, b->d.focal, sizeof(double), hipMemcpyDeviceToHost, st));
  }
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (cams6) memcpy(cams6, hc, sizeof(double) * n_c);
  if (focal) *focal = sorted[n_p];
  if (pts3) {
    memcpy(pts3, b->h_pts_in.data(), sizeof(double) * 3 * (size_t)b->np_in);  // points without observations
    host_parallel_for(b->np, [&](int lo, int hi) {
      for (int sp = lo; sp < hi; ++sp)
        for (int j = 0; j < 3; ++j) pts3[3 * (size_t)b->perm[sp] + j] = sorted[3 * (size_t)sp + j];
    });
  }
  return SFMHIP_OK;
}

// -------- building blocks of one LM iteration (all asynchronous on the context stream)
// Only the upper triangle of S (row-major: row r, columns r..ld-1) is ever written before the
// all-reduce, so only that crosses xGMI: [packed triangle | g | F^T b | diag | scalars + rank slots],
// ld(ld+1)/2 + 3 ld + SC + world doubles (6.0 MB instead of 11.9 MB at cfg4).
__global__ __launch_bounds__(256) void ba_pack_red(const double* __restrict__ red, double* __restrict__ packed, int ld,
                                                   int tail_n, int unpack) {
  const size_t tri = (size_t)ld * (ld + 1) / 2, ssz = (size_t)ld * ld;
  const int r = blockIdx.x;
  double* redw = const_cast<double*>(red);
  if (r == ld) {
    for (int i = threadIdx.x; i < tail_n; i += 256) {
      if (unpack) redw[ssz + i] = packed[tri + i];
      else packed[tri + i] = red[ssz + i];
    }
    return;
  }
  const size_t off = (size_t)r * ld - (size_t)r * (r - 1) / 2 - r;  // packed index of (r, c) = off + c
  for (int c = r + threadIdx.x; c < ld; c += 256) {
    if (unpack) redw[(size_t)r * ld + c] = packed[off + c];
    else packed[off + c] = red[(size_t)r * ld + c];
  }
}

// The same exchange for a camera graph with few edges (the union over the ranks, known since the set-up): only the
// 6 x 6 blocks of co-visible camera pairs (a <= b), the focal column and the tail cross xGMI --
// [n_blocks x 36 | S[0..dim)[focal] | g | F^T b | diag | scalars + rank slots]; cfg4: 0.65 MB instead of 6.0 MB,
// a latency-bound instead of a bandwidth-bound all-reduce on point-to-point links.
__global__ __launch_bounds__(64) void ba_pack_sparse(const double* __restrict__ red, double* __restrict__ packed, int ld, int dim,
                                                     const int2* __restrict__ blocks, int n_blocks, int tail_n, int unpack) {
  const size_t ssz = (size_t)ld * ld;
  double* redw = const_cast<double*>(red);
  const int bi = blockIdx.x;
  if (bi < n_blocks) {
    const int2 ab = blocks[bi];
    const int e = threadIdx.x;
    if (e < 36) {
      const size_t at = (size_t)(6 * ab.x + e / 6) * ld + 6 * ab.y + e % 6;
      if (unpack) redw[at] = packed[(size_t)bi * 36 + e];
      else packed[(size_t)bi * 36 + e] = red[at];
    }
    return;
  }
  const size_t base = (size_t)n_blocks * 36;
  // the focal column (rows 0..dim-1, column dim-1), then the tail
  for (int i = threadIdx.x + 64 * (bi - n_blocks); i < dim + tail_n; i += 64 * (int)(gridDim.x - n_blocks)) {
    if (i < dim) {
      const size_t at = (size_t)i * ld + dim - 1;
      if (unpack) redw[at] = packed[base + i];
      else packed[base + i] = red[at];
    } else {
      if (unpack) redw[ssz + i - dim] = packed[base + i];
      else packed[base + i] = red[ssz + i - dim];
    }
  }
}

static int ba_allreduce(sfmhip_ba* b, double* buf, size_t count) {
  if (b->world <= 1) return SFMHIP_OK;
  const int rc = b->allreduce(buf, count, b->allreduce_user);
  return rc == 0 ? SFMHIP_OK : SFMHIP_ERR_COMM;
}

// sum of a 0/1 flag over the ranks (host value in, host value out): a rank-consistent decision
static int ba_agree_flag(sfmhip_ba* b, int* flag) {
  hipStream_t st = b->ctx->stream;
  double* slot = b->d_flag;
  double v = *flag ? 1.0 : 0.0;
  SFM_HIP_TRY(hipMemcpyAsync(slot, &v, sizeof(double), hipMemcpyHostToDevice, st));
  SFM_TRY(ba_allreduce(b, slot, 1));
  SFM_HIP_TRY(hipMemcpyAsync(&v, slot, sizeof(double), hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  *flag = v > 0.5 ? 1 : 0;
  return SFMHIP_OK;
}

// the MFMA linearisation / elimination launches (one per Gram width and run class); norms = 1: the norms-only mode
static int ba_launch_eliminate(sfmhip_ba* b, double inv_radius, double lm_lo, double lm_hi, int norms) {
  hipStream_t st = b->ctx->stream;
  int nl = 0;
  // Two epilogues.  Default (since round 3): every workgroup stores its sums in a slab of its own and ba_gather_rows adds the
  // slabs in a fixed order -- S, g and the cost are the same bit patterns run after run.  SFMHIP_BA_DETERMINISTIC=0 (read when
  // the problem is created): the workgroups scatter their sums into S with f64 atomics (order-dependent in the last bits).
  // Measured at cfg4 (scripts/gpu_prof_elim_ab.sh): the elimination kernel 84.5 us with the scatter, 74.5 without; the gather
  // 22.5 us as a thread per destination (1 M scattered 8-byte sources), 10.9 us row by row (ba_gather_rows): the stage
  // 92.2 us against 91.3.
  double* slab = b->elim_deterministic ? b->d_slab : nullptr;
  // ten lanes per point (six points per wave and iteration) unless SFMHIP_BA_ELIM_LP=16 asks for the sixteen of rounds 2-5 (A/B)
  const bool lp10 = elim_lp10();
#define BA_ELIM(NB)                                                                                                   \
  for (int cls = 0; cls < 2; ++cls) {                                                                                 \
    const int li = 4 * cls + NB - 1, nthreads = cls ? 64 : 64 * b->elim_waves;                                        \
    if (!b->n_chunk_ids[li]) continue;                                                                                \
    /* wave panels | the cross-wave Gram reduction (NT x 4 x 64) | the F^T F reduction */                             \
    const size_t nw_ = nthreads / 64, gram_ = (size_t)(NB * (NB + 1) / 2) * 256;                                      \
    /* (four waves: the cross-wave sum takes four buffers of up to five tiles) */                                      \
    const size_t red_ = nw_ == 4 ? (size_t)4 * std::min(NB * (NB + 1) / 2, 5) * 256 : 0;                              \
    const size_t prows_ = lp10 ? ElimShape<10>::PROWS : ElimShape<16>::PROWS;                                          \
    const size_t lds = sizeof(double) * (nw_ * 36 * FP + 3 * FP + std::max(std::max(nw_ * prows_ * MP, gram_), red_)); \
    if (lp10)                                                                                                         \
      hipLaunchKernelGGL((ba_eliminate_mfma<NB, 10>), dim3(b->n_chunk_ids[li]), dim3(nthreads), lds, st, b->d,        \
                         b->d_chunks, b->d_chunk_ids[li],                                                             \
                         b->d_sig_cams, inv_radius, lm_lo, lm_hi, b->rank, norms, slab);                              \
    else                                                                                                              \
      hipLaunchKernelGGL((ba_eliminate_mfma<NB, 16>), dim3(b->n_chunk_ids[li]), dim3(nthreads), lds, st, b->d,        \
                         b->d_chunks, b->d_chunk_ids[li],                                                             \
                         b->d_sig_cams, inv_radius, lm_lo, lm_hi, b->rank, norms, slab);                              \
    ++nl;                                                                                                             \
  }
  BA_ELIM(1)
  BA_ELIM(2)
  BA_ELIM(3)
  BA_ELIM(4)
#undef BA_ELIM
  const int m = norms ? 1 : 0;
  if (slab && nl && m == 0 && b->n_grow) {
    const int rw = (b->n_grow + b->grow_waves - 1) / b->grow_waves, gw = (b->n_gth[0] * 16 + 64 * b->grow_waves - 1) / (64 * b->grow_waves);
    const size_t lds_g = sizeof(double) * std::max((size_t)b->grow_accw * b->grow_waves, (size_t)7 * 64 * b->grow_waves);
    hipLaunchKernelGGL(ba_gather_rows, dim3(rw + gw + 1), dim3(64 * b->grow_waves), lds_g, st,
                       (const double*)slab, (const int*)b->d_grow_colmap, (const int4*)b->d_grow_hdr,
                       (const int4*)b->d_grow_head, (const int4*)b->d_grow_src, b->n_grow, rw, b->ld, b->grow_accw, 6 * b->nc, b->n_chunks, (const int*)b->d_gth_ptr[0],
                       (const unsigned*)b->d_gth_src[0], (const int*)b->d_gth_dest[0], b->n_gth[0], b->d.red,
                       (long long)(b->ssz + 3 * (size_t)b->ld + SC + b->rank), b->n_fb ? 1 : 0, (const LmDev*)b->d.lm);
    ++nl;
  } else if (slab && nl && b->n_gth[m]) {
    hipLaunchKernelGGL(ba_gather_slabs, dim3((b->n_gth[m] + 15) / 16), dim3(256), 0, st, (const double*)slab,
                       (const int*)b->d_gth_ptr[m], (const unsigned*)b->d_gth_src[m], (const int*)b->d_gth_dest[m], b->n_gth[m],
                       b->d.red, (long long)(b->ssz + 3 * (size_t)b->ld + SC + b->rank), (const LmDev*)b->d.lm);
    ++nl;
  }
  return nl;
}

static int ba_prepare_scale(sfmhip_ba* b, int jacobi) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  const size_t tail = b->ssz + 2 * (size_t)b->ld;  // dc | sc
  SFM_HIP_TRY(hipMemsetAsync(d.red + tail, 0, sizeof(double) * ((size_t)b->ld + SC + 64), st));
  hipLaunchKernelGGL(ba_cam_prep, dim3((b->nc + 63) / 64), dim3(64), 0, st, d.cams, d.camd, b->nc, 1);
  b->camd_valid = true;
  if (b->np) hipLaunchKernelGGL(ba_point_norms, dim3((b->np + 255) / 256), dim3(256), 0, st, d, jacobi);
  if (b->no && jacobi) {
    ba_launch_eliminate(b, 1.0, 1e-6, 1e32, 1);
    if (b->n_fb) {
      hipLaunchKernelGGL(ba_pp_points, dim3((unsigned)(((size_t)b->n_fb * PP_LANES + 255) / 256)), dim3(256), 0, st, d, b->d_fb_points, b->d_pp_obase, b->n_fb, 1.0,
                         1e-6, 1e32, b->rank, b->d_ppT, b->d_tfu, 1, b->d_pp_part);
      hipLaunchKernelGGL(ba_cam_blocks, dim3(b->nc * b->cam_split), dim3(256), 0, st, d, b->d_cptr, b->d_cpt, b->d_cxy,
                         b->cam_split, 1, b->d_cslot, b->d_ppT, b->d_tfu, b->d_pp_part, b->n_pp_part, b->rank, b->d_cb_part, b->d_cb_cnt);
    }
  }
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
  double* tmp = d.red2 + RED2_TMP;
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
  if (b->ctx->timing) {
    SFM_HIP_TRY(hipEventRecord(b->ev[0], st));
    b->ev_on[0] = true;
  }
  // (X, the last ld*ld doubles, belongs to the dense factorisation only)
  auto swap_red = [&]() {
    std::swap(d.red, b->red_alt);
    b->red_is_alt = !b->red_is_alt;
    d.red2 = d.red + b->ssz + 3 * (size_t)b->ld + SC + 64;
    d.info = (int*)(d.red2 + RED2_INFO);
  };
  bool filled = true;
  if (b->nd_on && b->alt_clean) {
    swap_red();  // (the last nd_gather zeroed it; stream order is the only dependency)
    b->alt_clean = false;
    filled = false;
  } else {
    if (!b->nd_on && b->red_is_alt) swap_red();  // (the dense factorisation's X lives behind the first buffer)
    b->alt_clean = false;
    const bool all_x = !b->nd_on && !b->dense_xb;
    SFM_HIP_TRY(hipMemsetAsync(d.red, 0, sizeof(double) * (all_x ? b->red_count : b->red_count - b->ssz), st));
    if (!b->nd_on && b->dense_xb) {  // (only the diagonal blocks of X are formed and read)
      hipLaunchKernelGGL(chol_x_reset, dim3(b->ld / CB), dim3(256), 0, st, d.xinv, b->ld, b->ld / CB, b->dense_xb);
      b->launches += 1;
    }
  }
  if (!b->camd_valid) {
    hipLaunchKernelGGL(ba_cam_prep, dim3((b->nc + 63) / 64), dim3(64), 0, st, d.cams, d.camd, b->nc, 1);
    b->camd_valid = true;
    b->launches += 1;
  }
  const double inv_radius = 1.0 / radius;
  int nl = 0;
  nl += ba_launch_eliminate(b, inv_radius, o->min_lm_diagonal, o->max_lm_diagonal, 0);
  if (b->n_fb) {  // the pair path: points, then the blocks of camera pairs and of the cameras themselves
    hipLaunchKernelGGL(ba_pp_points, dim3((unsigned)(((size_t)b->n_fb * PP_LANES + 255) / 256)), dim3(256), 0, st, d, b->d_fb_points, b->d_pp_obase, b->n_fb,
                       radius, o->min_lm_diagonal, o->max_lm_diagonal, b->rank, b->d_ppT, b->d_tfu, 0, b->d_pp_part);
    if (b->n_pairs_pp)
      hipLaunchKernelGGL(ba_pp_pairs, dim3(b->n_pairs_pp), dim3(64), 0, st, d, b->d_pair_ptr, b->d_pair_cams, b->d_pair_ent,
                         (const double*)b->d_ppT);
    hipLaunchKernelGGL(ba_cam_blocks, dim3(b->nc * b->cam_split), dim3(256), 0, st, d, b->d_cptr, b->d_cpt, b->d_cxy,
                       b->cam_split, 0, b->d_cslot, b->d_ppT, b->d_tfu, b->d_pp_part, b->n_pp_part, b->rank, b->d_cb_part, b->d_cb_cnt);
    nl += 3;
  }
  SFM_HIP_TRY(hipGetLastError());
  b->launches += (filled ? 1 : 0) + nl;
  if (b->ctx->timing) {
    SFM_HIP_TRY(hipEventRecord(b->ev[1], st));
    b->ev_on[1] = true;
  }
  if (b->world > 1) {
    const int tail_n = 3 * b->ld + SC + b->world;
    const size_t tri = (size_t)b->ld * (b->ld + 1) / 2;
    if (b->n_xblocks > 0) {
      const int extra = (b->dim + tail_n + 63) / 64 < 64 ? (b->dim + tail_n + 63) / 64 : 64;
      hipLaunchKernelGGL(ba_pack_sparse, dim3(b->n_xblocks + extra), dim3(64), 0, st, d.red, b->d_red_pack, b->ld, b->dim,
                         (const int2*)b->d_xblocks, b->n_xblocks, tail_n, 0);
      SFM_HIP_TRY(hipGetLastError());
      SFM_TRY(ba_allreduce(b, b->d_red_pack, (size_t)b->n_xblocks * 36 + b->dim + tail_n));
      hipLaunchKernelGGL(ba_pack_sparse, dim3(b->n_xblocks + extra), dim3(64), 0, st, d.red, b->d_red_pack, b->ld, b->dim,
                         (const int2*)b->d_xblocks, b->n_xblocks, tail_n, 1);
      SFM_HIP_TRY(hipGetLastError());
    } else {
      hipLaunchKernelGGL(ba_pack_red, dim3(b->ld + 1), dim3(256), 0, st, d.red, b->d_red_pack, b->ld, tail_n, 0);
      SFM_HIP_TRY(hipGetLastError());
      SFM_TRY(ba_allreduce(b, b->d_red_pack, tri + tail_n));
      hipLaunchKernelGGL(ba_pack_red, dim3(b->ld + 1), dim3(256), 0, st, d.red, b->d_red_pack, b->ld, tail_n, 1);
      SFM_HIP_TRY(hipGetLastError());
    }
    b->launches += 2;
  }
  if (b->ctx->timing) {
    SFM_HIP_TRY(hipEventRecord(b->ev[2], st));
    b->ev_on[2] = true;
  }
  if (b->defer_fin && b->nd_on && add_diag) {
    b->fin_pending = true;
    b->fin_radius = radius, b->fin_lo = o->min_lm_diagonal, b->fin_hi = o->max_lm_diagonal;
    return SFMHIP_OK;
  }
  b->fin_pending = false;  // (this linearisation replaces one whose finalisation may still have been pending)
  hipLaunchKernelGGL(ba_finalize, dim3(1), dim3(1024), 0, st, d, radius, o->min_lm_diagonal, o->max_lm_diagonal,
                     b->world, add_diag ? 1 : 0);
  SFM_HIP_TRY(hipGetLastError());
  b->launches += 1;
  return SFMHIP_OK;
}

// (a linearisation whose finalisation was left to a gather that will not come: before anyone reads its scalars)
static int ba_finish_pending(sfmhip_ba* b) {
  if (!b->fin_pending) return SFMHIP_OK;
  b->fin_pending = false;
  hipLaunchKernelGGL(ba_finalize, dim3(1), dim3(1024), 0, b->ctx->stream, b->d, b->fin_radius, b->fin_lo, b->fin_hi, b->world, 1);
  SFM_HIP_TRY(hipGetLastError());
  b->launches += 1;
  return SFMHIP_OK;
}

// ---------------------------------------------------------------- NdPlan: dissection of the camera graph
// Order: reverse Cuthill-McKee positions pi; a cut at position p puts every camera at or behind p that sees a
// camera before p into the separator; what is left falls into connected components that do not see each
// other (interiors).  Cuts are chosen from a grid of positions (1..3 cuts) to minimise the number of
// two-panel launches on the dependency chain, max_i tiles_i / 2 + tiles_S / 2 + a constant for the gather /
// combine / three-step solve; the dense factorisation stays when that does not win by 20 %.
static int ba_nd_build(sfmhip_ba* b) {
  b->nd_ready = true;
  b->nd_on = false;
  const int nc = b->nc;
  const char* env = getenv("SFMHIP_BA_ND");  // "0": dense always; "1": dissect whenever a cut exists (tests)
  if (b->h_adj.empty()) return SFMHIP_OK;
  const bool force = env && env[0] == '1';
  const int wpr = (nc + 63) / 64;
  std::vector<unsigned long long> adj = b->h_adj;
  if (b->world > 1) {
    // union over the ranks: bits as doubles through the caller's sum all-reduce (set-up, once)
    std::vector<double> h((size_t)nc * nc);
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j < nc; ++j) h[(size_t)i * nc + j] = (adj[(size_t)i * wpr + (j >> 6)] >> (j & 63)) & 1ull ? 1.0 : 0.0;
    double* dbuf = nullptr;
    SFM_TRY(ba_alloc(b, &dbuf, h.size()));
    SFM_HIP_TRY(hipMemcpy(dbuf, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    SFM_TRY(ba_allreduce(b, dbuf, h.size()));
    SFM_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    SFM_HIP_TRY(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j < nc; ++j)
        if (h[(size_t)i * nc + j] > 0.5) adj[(size_t)i * wpr + (j >> 6)] |= 1ull << (j & 63);
    // ---- the exchange of a linearisation: only the blocks of camera pairs that some rank's points see together,
    // when that is less than half of the packed triangle (SFMHIP_BA_XSPARSE=0: the dense exchange always)
    std::vector<int2> xb;
    for (int a = 0; a < nc; ++a)
      for (int c = a; c < nc; ++c)
        if (c == a || ((adj[(size_t)a * wpr + (c >> 6)] >> (c & 63)) & 1ull) || ((adj[(size_t)c * wpr + (a >> 6)] >> (a & 63)) & 1ull))
          xb.push_back(make_int2(a, c));
    const size_t tri = (size_t)b->ld * (b->ld + 1) / 2;
    const char* xs = getenv("SFMHIP_BA_XSPARSE");
    if (!(xs && xs[0] == '0') && xb.size() * 36 + b->dim < tri / 2) {
      SFM_TRY(ba_alloc(b, &b->d_xblocks, xb.size()));
      SFM_HIP_TRY(hipMemcpy(b->d_xblocks, xb.data(), xb.size() * sizeof(int2), hipMemcpyHostToDevice));
      b->n_xblocks = (int)xb.size();
    }
  }
  if (env && env[0] == '0') return SFMHIP_OK;
  // ---- the front tree first (SFMHIP_BA_ND=2 or unset): every front on one CU; "1" keeps the chains + separator plan below
  if (!(env && env[0] == '1')) {
    // components up to this many columns become leaves: the largest size whose fronts fit (a leaf of three tiles under a
    // border of five does not); SFMHIP_BA_TREE_LEAF fixes it (experiments)
    const char* lc = getenv("SFMHIP_BA_TREE_LEAF");
    // helper workgroups per front (ba_front_plan.h, Front::nhelp; SFMHIP_BA_TREE_HELPERS = 1 ... 8): built in round 5 and OFF by
    // default -- measured slower at every setting (cfg4: 0.2146 ms per iteration without, 0.223-0.236 with 2-5 helpers keeping
    // 4-10 tiles in the front).  A helper's tile reaches the parent three trips through memory behind the front's last solve
    // (the front's stores acknowledged, its flag seen, its rows of L loaded: ~8 k cycles) where the front folds ALL its tiles
    // in 10-17 k; DESIGN.md appendix A has the counts.  Never more than leave every workgroup of the up-sweep a compute unit
    // of its own.
    const char* he = getenv("SFMHIP_BA_TREE_HELPERS");
    fplan::Plan P;
    fplan::Flat fl;
    const bool prof_ = getenv("SFMHIP_PROFILE_CREATE") != nullptr;
    auto tp_ = std::chrono::steady_clock::now();
    auto lap_ = [&](const char* what) {
      if (!prof_) return;
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[ba_nd_build] %-22s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp_).count());
      tp_ = now;
    };
    static const int keep_env = getenv("SFMHIP_BA_TREE_KEEP") ? atoi(getenv("SFMHIP_BA_TREE_KEEP")) : 8;  // (measurement)
    // (a one-shot problem: the plan of the last one, when the camera graph and the switches are the same -- BaHostScratch)
    static const bool plan_cache_on = !(getenv("SFMHIP_BA_PLAN_CACHE") && atoi(getenv("SFMHIP_BA_PLAN_CACHE")) == 0);
    BaHostScratch* const hs = plan_cache_on && b->use_arena && b->world == 1 ? ba_host_scratch(b->ctx) : nullptr;
    const int key[5] = {lc ? atoi(lc) : -1, he ? atoi(he) : -1, keep_env, b->ctx->n_cu, 1};
    const bool kept = hs && hs->nd_valid && hs->nd_nc == nc && memcmp(hs->nd_key, key, sizeof key) == 0 && hs->nd_adj == adj;
    b->nd_kept = kept;
    if (kept) {
      P = hs->nd_P;
      fl = hs->nd_fl;
    } else {
      for (int helpers = he ? std::max(0, std::min(8, atoi(he))) : 0; helpers >= 0; --helpers) {
        for (int leaf : {96, 64, 32}) {
          P = fplan::build_plan(nc, adj.data(), wpr, lc ? atoi(lc) : leaf, helpers, keep_env);
          if (P.ok || lc) break;
        }
        if (!P.ok) break;
        fl = fplan::flatten(P);
        if ((int)fl.up_roles.size() <= b->ctx->n_cu || he) break;
      }
      if (hs) {  // (a refused plan is kept as well: the next call does not search for it again)
        hs->nd_valid = true, hs->nd_nc = nc, hs->nd_adj = adj, hs->nd_P = P, hs->nd_fl = fl;
        memcpy(hs->nd_key, key, sizeof key);
      }
    }
    lap_(kept ? "front plan (kept)" : "front plan");
    if (P.ok) {
      int* d_ints = nullptr;
      int* d_up = nullptr;
      int* d_down = nullptr;
      unsigned* d_flags = nullptr;
      double* pool = nullptr;
      SFM_TRY(ba_alloc(b, &d_ints, fl.ints.size()));
      SFM_TRY(ba_alloc(b, &d_up, fl.up_roles.size()));
      SFM_TRY(ba_alloc(b, &d_down, fl.down_order.size()));
      const size_t n_flags = (size_t)fl.n_fronts + (size_t)fl.n_tflags;  // per front: z in place; per contribution tile
      SFM_TRY(ba_alloc(b, &d_flags, n_flags));
      SFM_TRY(ba_alloc(b, &pool, fl.n_doubles));
      SFM_HIP_TRY(hipMemcpy(d_ints, fl.ints.data(), fl.ints.size() * sizeof(int), hipMemcpyHostToDevice));
      SFM_HIP_TRY(hipMemcpy(d_up, fl.up_roles.data(), fl.up_roles.size() * sizeof(int), hipMemcpyHostToDevice));
      SFM_HIP_TRY(hipMemcpy(d_down, fl.down_order.data(), fl.down_order.size() * sizeof(int), hipMemcpyHostToDevice));
      SFM_HIP_TRY(hipMemset(d_flags, 0, n_flags * sizeof(unsigned)));
      SFM_HIP_TRY(hipMemset(pool, 0, fl.n_doubles * sizeof(double)));

      b->tree_fs.ints = d_ints;
      b->tree_fs.up_order = d_up;
      b->tree_fs.down_order = d_down;
      b->tree_fs.pool = pool;
      b->tree_dbg_ints = (long long)fl.ints.size(), b->tree_dbg_doubles = (long long)fl.n_doubles;
      {
        static std::atomic<int> n_trees{0};
        b->tree_xoff = n_trees.fetch_add(1) & 7;
      }
      b->tree_fs.flag_down = d_flags;
      b->tree_fs.tflag = d_flags + fl.n_fronts;
      double* zq = nullptr;
      SFM_TRY(ba_alloc(b, &zq, 2 * (size_t)b->ld));
      {
        std::vector<unsigned long long> pend(2 * (size_t)b->ld, FR_Z_PENDING);
        SFM_HIP_TRY(hipMemcpy(zq, pend.data(), pend.size() * 8, hipMemcpyHostToDevice));
      }
      lap_("plan uploads");
      b->tree_fs.zq = zq;
      b->tree_fs.zq_ld = b->ld;
      b->tree_fs.n_fronts = fl.n_fronts;
      b->tree_fs.n_roles = (int)fl.up_roles.size();

      b->tree_levels = fl.levels;
      b->tree_chain_tiles = P.chain_tiles;
      b->tree_chain_blocks = P.chain_blocks;
      b->tree_max_T = P.max_T;
      const char* se = getenv("SFMHIP_BA_TREE_STRIDE");
      b->tree_stride = se ? std::max(1, atoi(se)) : 1;
      b->tree_on = true;
      b->nd_on = true;  // (what the two share: the deferred ba_finalize, the pre-zeroed second buffer)
      if (getenv("SFMHIP_BA_ND_VERBOSE"))
        fprintf(stderr, "[sfmhip] reduced system as a front tree: %d fronts (+ %d helper workgroups), %d levels, %d tile steps (%d block steps) on the chain, fronts of up to %d tiles; dense %d tiles\n",
                fl.n_fronts, (int)fl.up_roles.size() - fl.n_fronts, fl.levels, P.chain_tiles, P.chain_blocks, P.max_T, b->ld / CB);
      return SFMHIP_OK;
    }
    if (getenv("SFMHIP_BA_ND_VERBOSE")) fprintf(stderr, "[sfmhip] no front tree: %s\n", P.why);
    if (env && env[0] == '2') return SFMHIP_OK;
  }
  std::vector<std::vector<int>> nb(nc);
  for (int i = 0; i < nc; ++i)
    for (int j = 0; j < nc; ++j)
      if (j != i && ((adj[(size_t)i * wpr + (j >> 6)] >> (j & 63)) & 1ull)) nb[i].push_back(j);
  // ---- RCM positions (every connected component from a pseudo-peripheral start)
  std::vector<int> order, pos(nc, -1), lvl(nc);
  order.reserve(nc);
  auto bfs = [&](int start, std::vector<int>& out) {
    out.clear();
    std::fill(lvl.begin(), lvl.end(), -1);
    out.push_back(start);
    lvl[start] = 0;
    for (size_t h = 0; h < out.size(); ++h) {
      const int u = out[h];
      std::vector<int> nx;
      for (int v : nb[u])
        if (lvl[v] < 0 && pos[v] < 0) {
          lvl[v] = lvl[u] + 1;
          nx.push_back(v);
        }
      std::sort(nx.begin(), nx.end(), [&](int a, int c) { return nb[a].size() != nb[c].size() ? nb[a].size() < nb[c].size() : a < c; });
      for (int v : nx) out.push_back(v);
    }
  };
  std::vector<int> comp;
  for (int s0 = 0; s0 < nc; ++s0) {
    if (pos[s0] >= 0) continue;
    int start = s0;
    for (int rep = 0; rep < 2; ++rep) {  // farthest vertex of the farthest vertex
      bfs(start, comp);
      start = comp.back();
    }
    bfs(start, comp);
    for (int v : comp) {
      pos[v] = (int)order.size();
      order.push_back(v);
    }
  }
  std::vector<int> minpos(nc);
  for (int v = 0; v < nc; ++v) {
    int m = pos[v];
    for (int u : nb[v]) m = std::min(m, pos[u]);
    minpos[v] = m;
  }
  // ---- evaluate a set of cuts: separator, components (chains by LPT when more than ND_MAX), launches
  struct Eval {
    double cost = 1e300;
    std::vector<int> sep;                  // cameras
    std::vector<std::vector<int>> chains;  // cameras of every chain, ascending
  };
  const int dense_tiles = b->ld / CB, n_cu_ = b->ctx->n_cu;
  auto evaluate = [&](const std::vector<int>& cuts, Eval& e) {
    std::vector<char> in_sep(nc, 0);
    for (int v = 0; v < nc; ++v)
      for (int p : cuts)
        if (pos[v] >= p && minpos[v] < p) in_sep[v] = 1;
    std::vector<int> root(nc);
    for (int v = 0; v < nc; ++v) root[v] = v;
    std::function<int(int)> find = [&](int x) {
      while (root[x] != x) x = root[x] = root[root[x]];
      return x;
    };
    for (int v = 0; v < nc; ++v)
      if (!in_sep[v])
        for (int u : nb[v])
          if (!in_sep[u]) root[find(u)] = find(v);
    std::map<int, std::vector<int>> comps;
    e.sep.clear();
    for (int v = 0; v < nc; ++v) {
      if (in_sep[v]) e.sep.push_back(v);
      else comps[find(v)].push_back(v);
    }
    if (comps.size() < 2) return;
    std::vector<std::vector<int>> cl;
    for (auto& kv : comps) cl.push_back(kv.second);
    std::sort(cl.begin(), cl.end(), [](const std::vector<int>& a, const std::vector<int>& c) {
      return a.size() != c.size() ? a.size() > c.size() : a[0] < c[0];
    });
    const int nch = (int)std::min<size_t>(cl.size(), ND_MAX);
    e.chains.assign(nch, {});
    for (auto& c : cl) {
      int best = 0;
      for (int k = 1; k < nch; ++k)
        if (e.chains[k].size() < e.chains[best].size()) best = k;
      e.chains[best].insert(e.chains[best].end(), c.begin(), c.end());
    }
    int max_t = 0;
    for (auto& c : e.chains) {
      std::sort(c.begin(), c.end());
      max_t = std::max(max_t, (int)((6 * c.size() + 63) / 64) * 2);
    }
    const int sep_t = (int)((6 * e.sep.size() + 1 + 63) / 64) * 2;
    e.cost = 0.5 * max_t + 0.5 * sep_t + 2.5;
    // the panel workgroups of a launch in one round of workgroups (a second round doubles the launch): launch 0
    // has N_i + 2 per chain
    int pan = 0;
    for (auto& c : e.chains) pan += (int)((6 * c.size() + 63) / 64) * 2 + sep_t + 2;
    if (pan > n_cu_) e.cost = 1e300;
  };
  Eval best;
  {
    const int G = nc > 1024 ? 12 : 24;
    std::vector<int> grid;
    for (int k = 1; k < G; ++k) grid.push_back((int)((long long)nc * k / G));
    Eval e;
    const char* ce = getenv("SFMHIP_BA_ND_CUTS");  // at most this many cuts (experiments)
    const int max_cuts = ce ? atoi(ce) : 3;
    for (size_t i = 0; i < grid.size(); ++i) {
      evaluate({grid[i]}, e);
      if (e.cost < best.cost) best = e;
      for (size_t j = i + 1; j < grid.size() && max_cuts >= 2; ++j) {
        evaluate({grid[i], grid[j]}, e);
        if (e.cost < best.cost) best = e;
        if (nc <= 1024 && max_cuts >= 3)
          for (size_t k = j + 1; k < grid.size(); k += 2) {
            evaluate({grid[i], grid[j], grid[k]}, e);
            if (e.cost < best.cost) best = e;
          }
      }
    }
  }
  if (best.chains.empty() || best.cost >= 1e299 || !(force || best.cost <= 0.8 * (0.5 * dense_tiles))) return SFMHIP_OK;
  {  // what nd_backsolve holds in LDS
    size_t max_c = 0;
    for (auto& c : best.chains) max_c = std::max(max_c, c.size());
    (void)max_c;
  }
  // ---- chains: index maps, buffers, job lists
  const int P = (int)best.chains.size();
  if (getenv("SFMHIP_BA_ND_VERBOSE")) {  // the plan's invariants: a partition of the cameras, no edge between two chains
    std::vector<int> owner(nc, -2);
    int bad = 0;
    for (int c : best.sep) owner[c] = -1;
    for (int i = 0; i < P; ++i)
      for (int c : best.chains[i]) {
        if (owner[c] != -2) ++bad;
        owner[c] = i;
      }
    for (int c = 0; c < nc; ++c) {
      if (owner[c] == -2) ++bad;
      for (int u : nb[c])
        if (owner[c] >= 0 && owner[u] >= 0 && owner[u] != owner[c]) ++bad;
    }
    fprintf(stderr, "[sfmhip] dissection check: %d violations\n", bad);
  }
  const int NS = (int)((6 * best.sep.size() + 1 + 63) / 64) * 2;
  std::vector<int> invS((size_t)NS * 32, -1);
  {
    int k = 0;
    for (int c : best.sep)
      for (int j = 0; j < 6; ++j) invS[k++] = 6 * c + j;
    invS[k++] = 6 * nc;  // the focal
  }
  NdSet& ns = b->nd;
  ns.n = P;
  std::vector<std::vector<int>> inv(P);
  size_t total = 0;
  std::vector<size_t> offM(P + 1), offy(P + 1), offX(P + 1);
  b->nd_max_ni = 0;
  for (int i = 0; i <= P; ++i) {
    const int ni = i < P ? (int)((6 * best.chains[i].size() + 63) / 64) * 2 : NS;
    const int N = i < P ? ni + NS : NS;
    if (i < P) {
      inv[i].assign((size_t)N * 32, -1);
      int k = 0;
      for (int c : best.chains[i])
        for (int j = 0; j < 6; ++j) inv[i][k++] = 6 * c + j;
      for (int k2 = 0; k2 < NS * 32; ++k2) inv[i][(size_t)ni * 32 + k2] = invS[k2];
      b->nd_max_ni = std::max(b->nd_max_ni, ni);
    }
    ns.c[i].ld = N * 32;
    ns.c[i].ni = ni;
    ns.c[i].N = N;
    offM[i] = total;
    total += (size_t)N * 32 * N * 32;
    offX[i] = total;
    total += (size_t)N * 32 * N * 32;
    offy[i] = total;
    total += (size_t)N * 32;
  }
  SFM_TRY(ba_alloc(b, &b->nd_buf, total));
  b->nd_buf_count = total;
  std::vector<int4> gj;
  b->nd_cols.col0[0] = 0;
  for (int i = 0; i <= P; ++i) {
    NdChain& c = ns.c[i];
    c.M = b->nd_buf + offM[i];
    c.X = b->nd_buf + offX[i];
    c.y = b->nd_buf + offy[i];
    int* dinv = nullptr;
    const std::vector<int>& hv = i < P ? inv[i] : invS;
    SFM_TRY(ba_alloc(b, &dinv, hv.size()));
    SFM_HIP_TRY(hipMemcpy(dinv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    c.inv = dinv;
    if (i < P) b->nd_cols.col0[i + 1] = b->nd_cols.col0[i] + c.ni * 32;
    for (int tr = 0; tr < c.N; ++tr) {
      for (int tc = 0; tc <= std::min(tr, c.ni - 1); ++tc) gj.push_back(make_int4(i, tr, tc, 0));
      gj.push_back(make_int4(i, tr, 0, 1));
      // what the factorisation reads before it writes: the lower right block of M (zero), the interior rows of X
      // (the identity; its tiles left of the diagonal are read too) -- no memset of the chain buffers
      for (int tc = c.ni; tc <= tr; ++tc) gj.push_back(make_int4(i, tr, tc, 2));
      // (a chain's X is only formed up to its interior columns: nxc in chol_step2_chains)
      if (tr < c.ni)
        for (int tc = 0; tc < (i < P ? c.ni : c.N); ++tc) gj.push_back(make_int4(i, tr, tc, 3));
    }
  }
  gj.push_back(make_int4(0, 0, 0, 4));  // (ba_finalize's part, when it is deferred to the gather)
  SFM_TRY(ba_alloc(b, &b->nd_gather_jobs, gj.size()));
  SFM_HIP_TRY(hipMemcpy(b->nd_gather_jobs, gj.data(), gj.size() * sizeof(int4), hipMemcpyHostToDevice));
  b->nd_n_gather = (int)gj.size();
  b->nd_on = true;
  if (getenv("SFMHIP_BA_ND_VERBOSE")) {
    fprintf(stderr, "[sfmhip] reduced system dissected: %d chains (", P);
    for (int i = 0; i < P; ++i) fprintf(stderr, "%s%d", i ? "," : "", ns.c[i].ni);
    fprintf(stderr, " tiles) + separator %d tiles (%zu cameras); dense %d tiles\n", NS, best.sep.size(), dense_tiles);
  }
  return SFMHIP_OK;
}

// trailing tiles per workgroup of a chol_step2 launch: the fewest that keep the launch to one round of workgroups
static void chol_launch_shape(int nt, int nxc, int k2, int* npan, int* ntrail) {
  const int m2 = nt - 2 * k2 - 2, mx = std::max(0, nxc - 2 * k2 - 2);
  *ntrail = k2 == 0 ? 0 : m2 * (m2 + 1) / 2 + m2 + 2 * k2 * mx;
  *npan = m2 + 2 + 2 * k2 + 2;
}

// diagnostic (SFMHIP_BA_ND_DEBUG): NaN / magnitude census of every chain's buffers after a stage
static void nd_census(sfmhip_ba* b, const char* stage) {
  hipStreamSynchronize(b->ctx->stream);
  for (int i = 0; i <= b->nd.n; ++i) {
    const NdChain& c = b->nd.c[i];
    const size_t n = (size_t)c.ld;
    std::vector<double> M(n * n), X(n * n), y(n);
    hipMemcpy(M.data(), c.M, n * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), c.X, n * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(y.data(), c.y, n * 8, hipMemcpyDeviceToHost);
    size_t nanL = 0, nanW = 0, nanX = 0, nany = 0;
    double maxL = 0, maxW = 0;
    const size_t o = (size_t)c.ni * 32;
    long first_nan_col = -1, first_nan_row = -1;
    for (size_t col = 0; col < n; ++col)
      for (size_t r = col; r < n; ++r) {
        const double v = M[col * n + r];
        const bool w = col >= o && i < b->nd.n;
        if (v != v) {
          (w ? nanW : nanL)++;
          if (first_nan_col < 0) first_nan_col = (long)col, first_nan_row = (long)r;
        } else (w ? maxW : maxL) = std::max(w ? maxW : maxL, std::fabs(v));
      }
    for (double v : X) nanX += v != v;
    for (double v : y) nany += v != v;
    fprintf(stderr, "[nd %s] chain %d (ni %d N %d): NaN L %zu W %zu X %zu y %zu  first (r %ld, c %ld)  max|L| %.3e max|W| %.3e\n", stage, i,
            c.ni, c.N, nanL, nanW, nanX, nany, first_nan_row, first_nan_col, maxL, maxW);
  }
}

// the front tree: one up-sweep launch (assembly, factorisation, forward substitution; its spare workgroups zero the other
// reduced-system buffer and do ba_finalize's bookkeeping), one down-sweep launch (backward substitution, candidate cameras)
static int ba_reduced_solve_tree(sfmhip_ba* b) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  if (!b->tree_attr_set) {
    SFM_HIP_TRY(hipFuncSetAttribute((const void*)front_up, hipFuncAttributeMaxDynamicSharedMemorySize, FR_LDS_BYTES));
    b->tree_attr_set = true;
  }
  const bool prezero = b->prezero;
  const size_t nz = b->red_count - b->ssz;
  if (prezero && !b->red_alt && nz % 2 == 0) SFM_TRY(ba_alloc(b, &b->red_alt, nz));
  const int zwg = prezero && b->red_alt ? (int)((nz + ND_ZERO_SLICE - 1) / ND_ZERO_SLICE) : 0;
  const int nF = b->tree_fs.n_roles, stride = b->tree_stride;  // (fronts and their helper workgroups)
  // grid: the fronts at multiples of `stride`; the zeroing workgroups fill the gaps and follow; one more for the bookkeeping
  const int gaps = (stride - 1) * nF;
  const int grid = stride * nF + std::max(0, zwg + 1 - gaps);
  const unsigned epoch = ++b->tree_epoch;
  static const bool by_level_env = getenv("SFMHIP_BA_TREE_BY_LEVEL") != nullptr;  // (diagnostic: one launch per tree level)
  const bool by_level = by_level_env || b->tree_by_level;  // (... and what a solve falls back to once a hand-off between fronts has timed out)
  int nl = 0;
  if (by_level) {
    for (int l = b->tree_levels - 1; l >= 0; --l, ++nl)
      hipLaunchKernelGGL(front_up, dim3(l == b->tree_levels - 1 ? grid : stride * nF), dim3(FR_WAVES * 64), FR_LDS_BYTES, st, b->tree_fs, d.red,
                         d.red + b->ssz, d.ld, d, b->fin_pending ? 1 : 0, b->fin_radius, b->fin_lo, b->fin_hi, b->world, epoch, stride, l, l,
                         b->red_alt, (long long)nz, l == b->tree_levels - 1 ? zwg : 0, b->tree_xoff % stride);
  } else {
    hipLaunchKernelGGL(front_up, dim3(grid), dim3(FR_WAVES * 64), FR_LDS_BYTES, st, b->tree_fs, d.red, d.red + b->ssz, d.ld, d,
                       b->fin_pending ? 1 : 0, b->fin_radius, b->fin_lo, b->fin_hi, b->world, epoch, stride, 0, 1 << 30, b->red_alt,
                       (long long)nz, zwg, b->tree_xoff % stride);
    nl = 1;
  }
  if (zwg) b->alt_clean = true;
  b->fin_pending = false;
  hipLaunchKernelGGL(front_down, dim3(b->tree_fs.n_fronts), dim3(FD_THREADS), 0, st, b->tree_fs, d, b->d_cam_used, b->rank, epoch, b->solve_cand ? 1 : 0);
  SFM_HIP_TRY(hipGetLastError());
  b->launches += nl + 1;
  return SFMHIP_OK;
}

static int ba_reduced_solve_nd(sfmhip_ba* b) {
  if (b->tree_on) return ba_reduced_solve_tree(b);
  const bool dbg = getenv("SFMHIP_BA_ND_DEBUG") != nullptr;
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  const NdSet& ns = b->nd;
  const int P = ns.n;
  const NdChain& sp = ns.c[P];
  if (!b->chol_chains_attr_set) {
    SFM_HIP_TRY(hipFuncSetAttribute((const void*)chol_step2_chains, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS_BYTES));
    SFM_HIP_TRY(hipFuncSetAttribute((const void*)chol_step2, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS_BYTES));
    b->chol_chains_attr_set = true;
  }
  const bool prezero = b->prezero;
  const size_t nz = b->red_count - b->ssz;  // everything but X (an even number of doubles: ld is a multiple of 64, SC + 64 + RED2_N even)
  if (prezero && !b->red_alt && nz % 2 == 0) SFM_TRY(ba_alloc(b, &b->red_alt, nz));
  const int zwg = prezero && b->red_alt ? (int)((nz + ND_ZERO_SLICE - 1) / ND_ZERO_SLICE) : 0;
  hipLaunchKernelGGL(nd_gather, dim3(b->nd_n_gather + zwg), dim3(256), 0, st, ns, b->nd_gather_jobs, b->nd_n_gather, d.red,
                     d.red + b->ssz, d.ld, d, b->fin_pending ? 1 : 0, b->fin_radius, b->fin_lo, b->fin_hi, b->world, b->red_alt,
                     (long long)nz);
  if (zwg) b->alt_clean = true;
  b->fin_pending = false;
  int nl = 1;
  if (dbg) nd_census(b, "gather");
  for (int k2 = 0; 2 * k2 < b->nd_max_ni; ++k2, ++nl) {
    ChainSet cs{};
    int tpw = 4, total = 0;
    for (;; ++tpw) {
      cs.n = 0;
      int pan = 0, trl = 0;
      for (int i = 0; i < P; ++i) {
        if (2 * k2 >= ns.c[i].ni) continue;
        int npan, ntrail;
        chol_launch_shape(ns.c[i].N, ns.c[i].ni, k2, &npan, &ntrail);
        const int j = cs.n++;
        cs.A[j] = ns.c[i].M;
        cs.y[j] = ns.c[i].y;
        cs.X[j] = ns.c[i].X;
        cs.ld[j] = ns.c[i].ld;
        cs.nt[j] = ns.c[i].N;
        cs.nxc[j] = ns.c[i].ni;
        cs.tpw[j] = tpw;
        cs.pan0[j] = pan;
        cs.trl0[j] = trl;
        pan += npan;
        trl += (ntrail + tpw - 1) / tpw;
      }
      cs.pan0[cs.n] = pan;
      cs.trl0[cs.n] = trl;
      total = pan + trl;
      if (total <= b->ctx->n_cu || tpw >= C2_WAVES) break;
    }
    hipLaunchKernelGGL(chol_step2_chains, dim3(total), dim3(C2_WAVES * 64), C2_LDS_BYTES, st, cs, k2, d.info);
    if (dbg) {
      char nm[32];
      snprintf(nm, sizeof nm, "chains k2=%d wg=%d tpw=%d", k2, total, tpw);
      nd_census(b, nm);
    }
  }
  hipLaunchKernelGGL(nd_combine, dim3(sp.N * (sp.N + 1) / 2 + sp.N), dim3(256), 0, st, ns);
  ++nl;
  for (int k2 = 0; 2 * k2 < sp.N; ++k2, ++nl) {
    int npan, ntrail, tpw = 4;
    chol_launch_shape(sp.N, sp.N, k2, &npan, &ntrail);
    while (tpw < C2_WAVES && npan + (ntrail + tpw - 1) / tpw > b->ctx->n_cu) ++tpw;
    hipLaunchKernelGGL(chol_step2, dim3(npan + (ntrail + tpw - 1) / tpw), dim3(C2_WAVES * 64), C2_LDS_BYTES, st, sp.M, sp.y,
                       sp.X, sp.ld, sp.N, k2, tpw, d.info, 0, 1, 0);
  }
  // z_S = L_SS^-T y_S;  w_i = y_i - L_Si^T z_S;  z_i = L_ii^-T w_i
  hipLaunchKernelGGL(nd_xy, dim3(sp.N, 1), dim3(1024), 0, st, ns, P, d.z);
  hipLaunchKernelGGL(nd_w, dim3((b->nd_cols.col0[P] + 3) / 4), dim3(256), 0, st, ns, b->nd_cols);
  hipLaunchKernelGGL(nd_xy, dim3(b->nd_max_ni, P), dim3(1024), 0, st, ns, 0, d.z);
  SFM_HIP_TRY(hipGetLastError());
  b->launches += nl + 3;
  return SFMHIP_OK;
}

static int ba_reduced_solve(sfmhip_ba* b) {
  if (b->nd_on) return ba_reduced_solve_nd(b);
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  double* A = d.red;
  double* y = d.red + b->ssz;  // g becomes y = L^-1 g
  const int nt = b->ld / CB;
  int nchol = 0;
  {
    // (once per problem object, not once per process: the attribute belongs to the device's code object,
    // and contexts on several devices may share a process)
    if (!b->chol_attr_set) {
      SFM_HIP_TRY(hipFuncSetAttribute((const void*)chol_step2, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS_BYTES));
      b->chol_attr_set = true;
    }
    static const int dfr_env = getenv("SFMHIP_BA_DENSE_DEFER") ? atoi(getenv("SFMHIP_BA_DENSE_DEFER")) : 0;  // (measurement)
    const int xb = b->dense_xb, dfr0 = !xb ? 1 : dfr_env > 0 ? dfr_env : nt >= DENSE_DEFER4_MIN_NT ? 4 : 2;
    static const int sw_env = getenv("SFMHIP_BA_DENSE_SWITCH") ? atoi(getenv("SFMHIP_BA_DENSE_SWITCH")) : -1;  // (measurement)
    // (measured, scripts/gpu_dense_sizes.py: worth it only behind visits of four pairs -- 640 cameras 624 -> 643 it/s at 40 tile
    // rows, 618 at 72; behind visits of two pairs the undeferred tail is slower: 400 cameras 1433 -> 1407)
    const int sw_m2 = sw_env >= 0 ? sw_env : dfr0 >= 4 ? DENSE_SWITCH_M2 : 0;
    bool deferred = false;  // some earlier launch of this factorisation left columns behind
    for (int k2 = 0; 2 * k2 < nt; ++k2, ++nchol) {
      const int m2 = nt - 2 * k2 - 2;
      // the updates are deferred while the trailing matrix is large; once a launch's visits of 64 dfr MFMAs would outlast its
      // panel chain (few tiles left: m2 <= DENSE_SWITCH_M2 tile rows) one launch catches every column up and the rest run undeferred
      const int dfr = m2 > sw_m2 ? dfr0 : 1;
      const int catchup = dfr == 1 && deferred ? dfr0 : 0;
      if (k2 > 0) deferred = dfr > 1;
      // launch 0 has no pending update; later launches: the tiles right of the panels, and those of X
      const int xlo = xb ? 2 * k2 / xb * xb : 0, mx = xb ? std::max(0, std::min(nt, xlo + xb) - 2 * k2 - 2) : m2;
      const int ntrail = k2 == 0 ? 0 : (dfr > 1 ? chol_trail_tiles(m2, dfr) : m2 * (m2 + 1) / 2) + m2 + (2 * k2 - xlo) * mx;
      const int npan = m2 + 2 + 2 * k2 + 2 - xlo;
      int tpw = 4;  // trailing tiles per workgroup: the fewest that keep the launch to one round of workgroups
      while (tpw < C2_WAVES && npan + (ntrail + tpw - 1) / tpw > b->ctx->n_cu) ++tpw;
      // (deferred updates, K = 256 per visit: a workgroup of eleven such visits outlasts the panel workgroups twice over and
      // the launch ends on the stragglers of a second round -- four visits, one per SIMD, measured best: scripts/gpu_dense_sizes.py)
      if (dfr >= 4 || catchup >= 4) tpw = 4;
      // (... unless that makes many rounds of workgroups: then two visits per SIMD, one's loads under the other's MFMAs -- 1400
      // cameras 118.7 -> 125 it/s; at 640 cameras, under two rounds, the same choice loses 1-2 %)
      static const int rounds_env = getenv("SFMHIP_BA_DENSE_TPW8_ROUNDS") ? atoi(getenv("SFMHIP_BA_DENSE_TPW8_ROUNDS")) : 4;  // (measurement)
      if (dfr >= 4 && (ntrail + 3) / 4 > rounds_env * b->ctx->n_cu) tpw = 8;
      hipLaunchKernelGGL(chol_step2, dim3(npan + (ntrail + tpw - 1) / tpw), dim3(C2_WAVES * 64), C2_LDS_BYTES, st, A, y, d.xinv, d.ld, nt, k2,
                         tpw, d.info, xb, dfr, catchup);
    }
  }
  int nbs = 1;
  if (b->dense_xb) {
    // z block by block from the last (chol_back_block)
    const int xb = b->dense_xb, nB = (nt + xb - 1) / xb;
    const size_t pn = (size_t)DENSE_XB * DENSE_XB * CB;
    for (int K = nB - 1; K >= -1; --K)
      hipLaunchKernelGGL(chol_back_block, dim3(K < 0 ? 1 : K == nB - 1 ? nt - K * xb : (K + 1) * xb), dim3(256), 0, st, A, d.xinv, y, d.z,
                         d.ld, nt, xb, K, b->d_back_part + ((K + 1) & 1) * pn, b->d_back_part + (K & 1) * pn);
    nbs = nB + 1;
  } else {
    // z = L^-T y = X y
    hipLaunchKernelGGL(chol_apply_inverse, dim3(nt), dim3(1024), 0, st, d.xinv, y, d.z, d.ld, nt);
  }
  SFM_HIP_TRY(hipGetLastError());
  b->launches += nchol + nbs;
  return SFMHIP_OK;
}

// (no point carries an observation: the step evaluation has no kernel of its own, the camera parts are still summed -- and the
// decision still taken -- by the one workgroup of this)
__global__ __launch_bounds__(256) void ba_step_nopoints(BaDev d) {
  if (lm_stopped(d)) {
    step_skip<true>(d);
    return;
  }
  step_finish<true>(d, 0, 0.0, 0.0, 0.0, 0.0, d.rank);
}

// the slots of the step evaluation's sums: a slot per workgroup of its kernels, then a pair per front of the down-sweep (or per
// wave of ba_cand_cams).  Fixed per problem; set before the reduced solve, whose down-sweep writes the camera parts.
struct StepLayout {
  bool runs;
  bool fits;  // the slots fit step_part (checked BEFORE the reduced solve, whose down-sweep writes the camera parts behind them)
  int n_runs_wg, npl, n_bs_wg, split, wpp;
  const int* plist;
};
static StepLayout ba_step_layout(sfmhip_ba* b) {
  static const bool runs_env = !(getenv("SFMHIP_BA_BACKSUB_RUNS") && atoi(getenv("SFMHIP_BA_BACKSUB_RUNS")) == 0);
  static const int wpp_env = getenv("SFMHIP_BA_BACKSUB_WPP") ? atoi(getenv("SFMHIP_BA_BACKSUB_WPP")) : 2;  // (measurement)
  static const int split_env = getenv("SFMHIP_BA_BACKSUB_SPLIT") ? std::min(8, std::max(1, atoi(getenv("SFMHIP_BA_BACKSUB_SPLIT")))) : 1;  // (measured at cfg4: 21.0 us / 33 / 53 for 1 / 2 / 4 workgroups per run)
  StepLayout L;
  // the points in runs: a thread per point, the run's cameras in LDS (ba_backsub_runs); the pair path's points (or,
  // SFMHIP_BA_BACKSUB_RUNS=0, all of them): ba_backsub
  L.runs = runs_env && b->n_chunks > 0 && b->d_bs_ids && b->np > 0;
  L.split = split_env, L.wpp = wpp_env;
  L.n_runs_wg = L.runs ? b->n_chunks * split_env : 0;
  L.plist = L.runs ? b->d_fb_points : nullptr;
  L.npl = b->np ? (L.runs ? b->n_fb : b->np) : 0;
  const size_t nblk = ((size_t)L.npl + 63) / 64;  // blocks of 64 points
  L.n_bs_wg = L.npl <= 0 ? 0 : wpp_env == 1 ? (int)((nblk + 3) / 4) : wpp_env == 2 ? (int)((nblk + 1) / 2) : (int)nblk;
  b->d.step_total = std::max(1, L.n_runs_wg + L.n_bs_wg);
  b->d.cam_parts = b->tree_on ? b->tree_fs.n_fronts : (b->nc + 1 + 63) / 64;
  L.fits = 4 * (size_t)b->d.step_total + 2 * (size_t)b->d.cam_parts <= b->step_part_n;
  return L;
}

static int ba_step_eval(sfmhip_ba* b, double radius, const sfmhip_ba_opts* o) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  const StepLayout L = ba_step_layout(b);
  const bool runs = L.runs;
  const int n_runs_wg = L.n_runs_wg, npl = L.npl, n_bs_wg = L.n_bs_wg, split_env = L.split, wpp_env = L.wpp;
  const int* plist = L.plist;
  // (the decision: by the finisher of ba_backsub_runs when that is the step evaluation's last kernel and there is one
  // rank; else by ba_decide behind the all-reduce)
  d.decide_here = d.lm && b->world == 1 && npl <= 0 ? 1 : 0;
  if (!L.fits) return SFMHIP_ERR_STATE;
  if (!b->tree_on) hipLaunchKernelGGL(ba_cand_cams, dim3((b->nc + 1 + 63) / 64), dim3(64), 0, st, d, b->d_cam_used, b->rank);
  int nbs = 0;
  if (runs) {
    d.step_last = npl > 0 ? 0 : 1;
    hipLaunchKernelGGL(ba_backsub_runs, dim3(n_runs_wg), dim3(256), 0, st, d, (const int4*)b->d_bs_ids, radius, o->min_lm_diagonal,
                       o->max_lm_diagonal, split_env);
    ++nbs;
  }
  d.step_last = 1;
  if (npl > 0) {
    if (wpp_env == 1)
      hipLaunchKernelGGL(ba_backsub<1>, dim3(n_bs_wg), dim3(256), 0, st, d, radius, o->min_lm_diagonal, o->max_lm_diagonal, plist, npl, n_runs_wg);
    else if (wpp_env == 2)
      hipLaunchKernelGGL(ba_backsub<2>, dim3(n_bs_wg), dim3(256), 0, st, d, radius, o->min_lm_diagonal, o->max_lm_diagonal, plist, npl, n_runs_wg);
    else
      hipLaunchKernelGGL(ba_backsub<4>, dim3(n_bs_wg), dim3(256), 0, st, d, radius, o->min_lm_diagonal, o->max_lm_diagonal, plist, npl, n_runs_wg);
    ++nbs;
  }
  if (!nbs) {
    hipLaunchKernelGGL(ba_step_nopoints, dim3(1), dim3(256), 0, st, d);
    ++nbs;
  }
  SFM_HIP_TRY(hipGetLastError());
  b->launches += (b->tree_on ? 0 : 1) + nbs;
  SFM_TRY(ba_allreduce(b, d.red2, 8));
  if (d.lm && !d.decide_here) {  // (several ranks: every rank takes the same decision from the same all-reduced sums)
    hipLaunchKernelGGL(ba_decide, dim3(1), dim3(64), 0, st, d);
    SFM_HIP_TRY(hipGetLastError());
    b->launches += 1;
  }
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

// The iteration's scalars (the linearisation's and, past the per-rank slots, the step evaluation's)
// go to the host in one piece: a one-workgroup kernel writes them into the pinned buffer and then a
// sequence number behind them; the host spins on the sequence number.  (A blit + hipStreamSynchronize
// costs ~15 us more per iteration: copy-kernel launch, completion interrupt, wake-up.)
constexpr int H_SC_N = SC + 64 + RED2_N;
static_assert(H_SC_N <= 256, "ba_publish copies one double per thread");
__global__ __launch_bounds__(256) void ba_publish(const double* __restrict__ src, double* __restrict__ host, double seq) {
  const int i = threadIdx.x;
  if (i < H_SC_N) host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (i == 0) {
    *(volatile double*)(host + H_SC_N) = seq;
    __threadfence_system();
  }
}

static int ba_read_scalars(sfmhip_ba* b, IterScalars* s, bool with_step) {
  hipStream_t st = b->ctx->stream;
  BaDev& d = b->d;
  const size_t sc_off = b->ssz + 3 * (size_t)b->ld;
  b->h_seq += 1.0;
  hipLaunchKernelGGL(ba_publish, dim3(1), dim3(256), 0, st, d.red + sc_off, b->h_sc_dev, b->h_seq);
  SFM_HIP_TRY(hipGetLastError());
  {
    volatile double* flag = b->h_sc + H_SC_N;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*flag != b->h_seq) {
      if ((++spins & 0xFFF) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.02) {
        // long-running iteration (or a fault on the stream): fall back to the blocking wait, which
        // also surfaces asynchronous errors
        SFM_HIP_TRY(hipStreamSynchronize(st));
        if (*flag != b->h_seq) return SFMHIP_ERR_HIP;
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  s->cost = 0.5 * b->h_sc[0];
  s->nfail = b->h_sc[2];
  s->gmax = b->h_sc[3];
  if (with_step) {
    const double* step = b->h_sc + SC + 64;  // (red2[0..3]: the totals step_finish left, all-reduced)
    s->cost_c = 0.5 * step[0];
    s->mcc = -step[1];
    s->step_n2 = step[2];
    s->cand_n2 = step[3];
    memcpy(&s->info, step + RED2_INFO, sizeof(int));
    if (step[RED2_TIMEOUT] >= 1024.0) s->info = INFO_FINISHER_TIMEOUT;  // (a finisher's slot never arrived on some rank: every rank reports it)
    else if (step[RED2_TIMEOUT] > 0.0 && s->info >= 0) s->info = -1;     // (another rank's spin ran out: every rank repeats the solve)
  }
  return SFMHIP_OK;
}

// (after the iteration's stream synchronisation; only events recorded since the last call -- timing
// can be switched on between a linearisation and the iteration that uses it)
static void ba_acc_timing(sfmhip_ba* b) {
  float ms;
  for (int k = 0; k < 4; ++k)
    if (b->ev_on[k] && b->ev_on[k + 1] && hipEventElapsedTime(&ms, b->ev[k], b->ev[k + 1]) == hipSuccess)
      b->t_acc[k] += ms * 1e-3;
  for (bool& on : b->ev_on) on = false;
}

// TrustRegionMinimizer::Minimize (Ceres 1.13) + LevenbergMarquardtStrategy.  The decision itself is lm_decide (above),
// taken on the device by default; the loop state lives in the sfmhip_ba object so that sfmhip_ba_iterate can be called a
// few iterations at a time (bench.py interleaves it with matching sweeps).
static void lm_set_options(LmDev& s, const sfmhip_ba_opts* o, bool timing_only) {
  s.gtol = o->gradient_tolerance, s.ptol = o->parameter_tolerance, s.ftol = o->function_tolerance;
  s.min_rel_dec = o->min_relative_decrease, s.max_radius = o->max_radius, s.min_radius = o->min_radius;
  s.max_invalid = o->max_consecutive_invalid;
  s.max_iter = timing_only ? 0x7FFFFFFF : o->max_iterations;
  s.timing_only = timing_only ? 1 : 0;
  s.stop = LM_RUNNING;
}

static int ba_begin(sfmhip_ba* b, const sfmhip_ba_opts* o) {
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  for (double& t : b->t_acc) t = 0;
  b->launches = 0;
  if (!b->nd_ready) SFM_TRY(ba_nd_build(b));
  SFM_TRY(ba_prepare_scale(b, o->jacobi_scaling));
  b->lm = LmState();
  LmDev& s = b->lm.s;
  s.radius = o->initial_radius;
  s.dec_factor = 2.0;
  s.x_norm = b->x_norm;
  // iteration 0: cost + gradient at x0 (the same pass also forms the first reduced system)
  IterScalars sc{};
  SFM_TRY(ba_linearize_eliminate(b, s.radius, o, true));
  SFM_TRY(ba_read_scalars(b, &sc, false));
  s.cost = s.initial_cost = sc.cost;
  s.gmax = sc.gmax;
  b->lm.have_lin = true;
  b->lm.started = true;
  return SFMHIP_OK;
}

static void lm_log(const sfmhip_ba* b, const LmDev& r) {
  if (r.log_kind == LM_KIND_NONE) return;
  if (r.log_kind == LM_KIND_INVALID)
    fprintf(stderr, "[sfmhip-ba] it %d invalid step (info %d, point blocks not PD %.0f, model cost change %.6e, candidate cost %.6e, |step|^2 %.6e), radius %.3e\n",
            r.iter, r.log_info, r.log_nfail, r.log_mcc, r.log_cost_c, r.log_step_norm, r.radius);
  else if (r.log_kind == LM_KIND_STOP && r.stop == LM_STOP_TIMEOUT)
    fprintf(stderr, "[sfmhip-ba] a bounded spin of the reduced solve ran out (info %d): the solve is repeated level by level\n", r.log_info);
  else
    fprintf(stderr, "[sfmhip-ba] it %d cost %.9e -> %.9e rho %.3e radius %.3e |step| %.3e%s\n", r.iter, r.log_cost0, r.log_cost_c, r.log_rho,
            r.radius, r.log_step_norm, r.log_kind == LM_KIND_ACCEPTED ? "" : r.log_kind == LM_KIND_REJECTED ? " (rejected)" : " (stop)");
  static const bool bits = getenv("SFMHIP_BA_VERBOSE_BITS") != nullptr;  // (diagnostics: the decision's inputs to the last bit)
  if (bits) fprintf(stderr, "[sfmhip-ba-bits] it %d cost %a cost_c %a mcc %a rho %a radius %a |step| %a\n", r.iter, r.log_cost0, r.log_cost_c, r.log_mcc, r.log_rho, r.radius, r.log_step_norm);
  (void)b;
}

// the linearisation at the current x with the current radius, if the reduced-system buffer does not hold it
static int ba_ensure_lin(sfmhip_ba* b, const sfmhip_ba_opts* o) {
  if (b->lm.have_lin) return SFMHIP_OK;
  b->defer_fin = true;  // (the dissected solve follows: its gather does ba_finalize's part)
  const int rc = ba_linearize_eliminate(b, b->lm.s.radius, o, true);
  b->defer_fin = false;
  SFM_TRY(rc);
  b->lm.have_lin = true;
  return SFMHIP_OK;
}

// one iteration body, enqueued: [the linearisation, unless the buffer holds it] + reduced solve + step evaluation
// (+ the decision, when the loop runs on the device)
static int ba_enqueue_body(sfmhip_ba* b, const sfmhip_ba_opts* o) {
  hipStream_t st = b->ctx->stream;
  SFM_TRY(ba_ensure_lin(b, o));
  b->lm.have_lin = false;
  b->solve_cand = true;  // (the front tree's down-sweep leaves the candidate cameras and their tables)
  if (!ba_step_layout(b).fits) return SFMHIP_ERR_STATE;  // (... and the camera parts of the step's norms, in the slots behind the step evaluation's)
  const int rc_solve = ba_reduced_solve(b);
  b->solve_cand = false;
  SFM_TRY(rc_solve);
  if (b->ctx->timing) {
    SFM_HIP_TRY(hipEventRecord(b->ev[3], st));
    b->ev_on[3] = true;
  }
  SFM_TRY(ba_step_eval(b, b->lm.s.radius, o));
  if (b->ctx->timing) {
    SFM_HIP_TRY(hipEventRecord(b->ev[4], st));
    b->ev_on[4] = true;
  }
  return SFMHIP_OK;
}

// the record of decision `seq`, as soon as the device has written it into the host's ring
static int ba_wait_record(sfmhip_ba* b, unsigned seq, LmDev* out) {
  volatile LmDev* slot = b->h_ring + seq % LM_RING;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (slot->seq != seq) {
    if ((++spins & 0xFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.02) {
      SFM_HIP_TRY(hipStreamSynchronize(b->ctx->stream));  // (a long batch, or a fault on the stream, which this surfaces)
      if (slot->seq != seq) return SFMHIP_ERR_HIP;
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  memcpy(out, (const void*)slot, sizeof(LmDev));
  return SFMHIP_OK;
}

// A reduced solve reported a spin that ran out (info < 0).  In the front tree that is a child front that did not get a
// compute unit while its parent polled (a busy device; dispatch order is not a promise): from now on the tree runs one
// launch per level, where every child has finished before its parent starts.  Anywhere else (the spins inside a
// workgroup) it is a bug: the caller gets SFMHIP_ERR_TIMEOUT, not a silently different trajectory.
static int ba_handle_timeout(sfmhip_ba* b, int info) {
  b->spin_timeouts += 1;
  {
    // the finisher re-arms the slots it has read; one that ran out left a slot that its late writer may still fill: every slot
    // is armed again, on the stream, before anything is repeated
    const std::vector<unsigned long long> pend(b->step_part_n, STEP_PENDING);
    SFM_HIP_TRY(hipMemcpyAsync(b->d_step_part, pend.data(), sizeof(double) * b->step_part_n, hipMemcpyHostToDevice, b->ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(b->ctx->stream));  // (pend leaves scope; a time-out is no hot path)
  }
  if (info == INFO_FINISHER_TIMEOUT) return SFMHIP_ERR_TIMEOUT;  // (not a hand-off between fronts: one launch per level would not cure it)
  if (b->tree_on && !b->tree_by_level) {
    b->tree_by_level = true;
    return SFMHIP_OK;
  }
  return SFMHIP_ERR_TIMEOUT;
}

// The LM loop.  `iters` < 0: until a stopping rule fires (sfmhip_ba_run); else exactly that many iterations without
// convergence tests (sfmhip_ba_iterate).  *term receives the termination type.
static int ba_lm_loop(sfmhip_ba* b, const sfmhip_ba_opts* o, int iters, const std::function<double()>& elapsed, int* term) {
  const bool t_only = iters >= 0;
  hipStream_t st = b->ctx->stream;
  LmDev& s = b->lm.s;
  lm_set_options(s, o, t_only);
  *term = SFMHIP_BA_NO_CONVERGENCE;
  // the host decides when it is asked to (SFMHIP_BA_HOST_LOOP=1, read per call) and whenever stage timing is on (its events
  // want a synchronisation per iteration anyway)
  const char* hl = getenv("SFMHIP_BA_HOST_LOOP");
  const bool host_loop = (hl && atoi(hl) != 0) || b->ctx->timing;
  // what TrustRegionMinimizer tests before its first iteration
  if (!t_only) {
    if (s.iter >= s.max_iter) return SFMHIP_OK;
    if (s.radius < s.min_radius) {
      *term = SFMHIP_BA_CONVERGENCE;
      return SFMHIP_OK;
    }
  }
  auto time_is_up = [&](int* up) -> int {
    *up = 0;
    if (t_only || !(o->max_time_s > 0)) return SFMHIP_OK;
    // Every other stop rule reads all-reduced scalars; a wall clock is per rank.  With world > 1 the ranks agree on it
    // first (one 1-double sum: anybody's limit reached stops everybody), or one would leave the loop while the others
    // wait for it in the next all-reduce.
    *up = elapsed() >= o->max_time_s ? 1 : 0;
    if (b->world > 1) SFM_TRY(ba_agree_flag(b, up));
    return SFMHIP_OK;
  };
  int done = 0;
  if (host_loop) {
    b->d.lm = nullptr;
    s.seq = b->lm_seq;
    for (;;) {
      if (t_only && done >= iters) break;
      int up = 0;
      SFM_TRY(time_is_up(&up));
      if (up) break;
      SFM_TRY(ba_enqueue_body(b, o));
      IterScalars sc{};
      SFM_TRY(ba_read_scalars(b, &sc, true));
      ba_acc_timing(b);
      LmIn in{sc.cost, sc.nfail, sc.gmax, sc.cost_c, sc.mcc, sc.step_n2, sc.cand_n2, sc.info};
      lm_decide(s, in);
      b->lm_seq = s.seq;
      if (o->verbose) lm_log(b, s);
      if (s.parity) {  // an accepted step: the candidate buffers become x
        ba_swap_candidate(b);
        s.parity = 0;
      }
      ++done;
      if (s.stop == LM_STOP_TIMEOUT) {
        SFM_TRY(ba_handle_timeout(b, s.log_info));
        s.stop = LM_RUNNING;
        --done;
        continue;
      }
      if (s.stop != LM_RUNNING) {
        *term = s.stop;
        break;
      }
    }
    return SFMHIP_OK;
  }
  // ---- the loop on the device: iterations are enqueued a batch ahead, the records are read behind the GPU
  static const int batch_env = getenv("SFMHIP_BA_LM_BATCH") ? std::max(1, std::min(LM_RING / 2, atoi(getenv("SFMHIP_BA_LM_BATCH")))) : 0;
  int rc = SFMHIP_OK;
  s.seq = b->lm_seq;
  s.parity = 0;
  SFM_HIP_TRY(hipMemcpyAsync(b->d_lm, &s, sizeof(LmDev), hipMemcpyHostToDevice, st));
  b->d.lm = b->d_lm;
  for (;;) {
    if (t_only && done >= iters) break;
    int up = 0;
    if ((rc = time_is_up(&up)) != SFMHIP_OK || up) break;
    // (several ranks: the batch is a function of the iteration count only, so that every rank issues the same all-reduces)
    // (how far ahead: 4500 it/s with 4 iterations enqueued ahead, 4590 with 12-20; with 32 -- 160 launches outstanding per stream --
    // the runtime's enqueue slows down once several streams are alive: 3970 it/s; scripts/gpu_ba_two_problems.py)
    int B = batch_env ? batch_env : t_only ? 20 : 4;
    if (t_only) B = std::min(B, iters - done);
    else B = std::max(1, std::min(B, s.max_iter - s.iter));
    const unsigned seq0 = b->lm_seq, epoch0 = b->tree_epoch;
    for (int i = 0; i < B && rc == SFMHIP_OK; ++i) {
      rc = ba_enqueue_body(b, o);
      if (rc == SFMHIP_OK && (i == B - 1 || o->verbose)) hipLaunchKernelGGL(ba_lm_publish, dim3(1), dim3(64), 0, st, b->d);
    }
    if (rc == SFMHIP_OK && hipGetLastError() != hipSuccess) rc = SFMHIP_ERR_HIP;
    if (rc != SFMHIP_OK) break;
    b->lm_seq = seq0 + (unsigned)B;
    if ((rc = ba_wait_record(b, b->lm_seq, &s)) != SFMHIP_OK) break;
    if (o->verbose)
      for (unsigned q = seq0 + 1; q <= b->lm_seq; ++q) lm_log(b, b->h_ring[q % LM_RING]);
    done += B;
    if (s.stop != LM_RUNNING) {
      // The bodies enqueued behind the stop did nothing (their kernels return at once); the reduced-system buffer the last of
      // them would have filled holds no linearisation, the other one was not zeroed, and the front tree's epochs go on from
      // the last solve that ran (the down-sweep's mailbox alternates by epoch).
      const int ran = (int)(s.stop_seq - seq0);
      b->lm.have_lin = false;
      b->alt_clean = false;
      b->fin_pending = false;
      if (b->tree_on) b->tree_epoch = epoch0 + (unsigned)ran;
      if (s.stop == LM_STOP_TIMEOUT) {
        if ((rc = ba_handle_timeout(b, s.log_info)) != SFMHIP_OK) break;
        done -= B - ran + 1;  // (the solve that timed out is repeated)
        s.stop = LM_RUNNING;
        s.seq = b->lm_seq;
        if (hipMemcpyAsync(b->d_lm, &s, sizeof(LmDev), hipMemcpyHostToDevice, st) != hipSuccess) {
          rc = SFMHIP_ERR_HIP;
          break;
        }
        continue;
      }
      *term = s.stop;
      break;
    }
  }
  b->d.lm = nullptr;
  if (rc != SFMHIP_OK) {
    hipStreamSynchronize(st);
    b->lm.started = false;
    return rc;
  }
  // the host takes over: its pointers follow the device's view of which parameter set is x
  if (s.parity) {
    ba_swap_candidate(b);
    s.parity = 0;
  }
  return SFMHIP_OK;
}

// cost / gradient of the linearisation behind the last accepted step, which no decision has read yet
static int ba_flush_lin(sfmhip_ba* b, const sfmhip_ba_opts* o) {
  LmDev& s = b->lm.s;
  if (!s.lin_unread) return SFMHIP_OK;
  IterScalars sc{};
  SFM_TRY(ba_ensure_lin(b, o));
  SFM_TRY(ba_finish_pending(b));
  SFM_TRY(ba_read_scalars(b, &sc, false));
  s.cost = sc.cost;
  s.gmax = sc.gmax;
  s.lin_unread = 0;
  return SFMHIP_OK;
}

static void ba_fill_summary(sfmhip_ba* b, int term, double time_s, sfmhip_ba_summary* sum) {
  const LmDev& s = b->lm.s;
  b->x_norm = s.x_norm;
  sum->termination = term;
  sum->iterations = s.iter;
  sum->successful_steps = s.nsucc;
  sum->initial_cost = s.initial_cost;
  sum->final_cost = s.cost;
  sum->final_radius = s.radius;
  sum->gradient_max_norm = s.gmax;
  sum->time_s = time_s;
  sum->spin_timeouts = b->spin_timeouts;
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
  const bool prof_ = getenv("SFMHIP_PROFILE_CREATE") != nullptr;
  SFM_TRY(ba_begin(b, opts));
  if (prof_) fprintf(stderr, "[sfmhip_ba_run] begin %7.2f ms\n", elapsed() * 1e3);
  LmDev& s = b->lm.s;
  int term = SFMHIP_BA_NO_CONVERGENCE;
  if (s.gmax <= opts->gradient_tolerance) term = SFMHIP_BA_CONVERGENCE;
  else SFM_TRY(ba_lm_loop(b, opts, -1, elapsed, &term));
  if (prof_) fprintf(stderr, "[sfmhip_ba_run] loop  %7.2f ms (%d iterations)\n", elapsed() * 1e3, s.iter);
  SFM_TRY(ba_flush_lin(b, opts));
  if (prof_) fprintf(stderr, "[sfmhip_ba_run] flush %7.2f ms\n", elapsed() * 1e3);
  if (term == SFMHIP_BA_NO_CONVERGENCE && s.gmax <= opts->gradient_tolerance) term = SFMHIP_BA_CONVERGENCE;
  b->lm.started = false;  // a finished solve is not resumable
  ba_fill_summary(b, term, elapsed(), &sm);
  if (summary) *summary = sm;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_iterate(sfmhip_ba* b, int iters, sfmhip_ba_summary* summary) {
  if (!b || iters < 0) return SFMHIP_ERR_ARG;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(clk::now() - t0).count(); };
  sfmhip_ba_opts o;
  sfmhip_ba_default_opts(&o);
  static const bool verbose_env = getenv("SFMHIP_BA_VERBOSE") != nullptr;  // (diagnostics: a line per iteration on stderr)
  if (verbose_env) o.verbose = 1;
  sfmhip_ba_summary sm;
  memset(&sm, 0, sizeof sm);
  if (!b->lm.started) SFM_TRY(ba_begin(b, &o));
  int term = SFMHIP_BA_NO_CONVERGENCE;
  SFM_TRY(ba_lm_loop(b, &o, iters, elapsed, &term));
  SFM_TRY(ba_flush_lin(b, &o));
  ba_fill_summary(b, SFMHIP_BA_NO_CONVERGENCE, elapsed(), &sm);
  if (summary) *summary = sm;
  return SFMHIP_OK;
}

// Test hook: one decision of the trust-region loop on the host, by the function the device runs (include/sfmhip.h)
extern "C" int sfmhip_ba_lm_decide(sfmhip_lm_state* st, const sfmhip_lm_inputs* in) {
  if (!st || !in) return SFMHIP_ERR_ARG;
  LmDev s{};
  s.gtol = st->gradient_tolerance, s.ptol = st->parameter_tolerance, s.ftol = st->function_tolerance;
  s.min_rel_dec = st->min_relative_decrease, s.max_radius = st->max_radius, s.min_radius = st->min_radius;
  s.max_invalid = st->max_consecutive_invalid, s.max_iter = st->max_iterations, s.timing_only = st->timing_only;
  s.radius = st->radius, s.dec_factor = st->decrease_factor, s.cost = st->cost, s.gmax = st->gradient_max_norm, s.x_norm = st->x_norm;
  s.iter = st->iterations, s.nsucc = st->successful_steps, s.invalid = st->invalid_steps, s.lin_unread = st->lin_unread;
  s.parity = 0, s.stop = st->stop;
  LmIn li{in->lin_cost, in->lin_failed_blocks, in->lin_gradient_max, in->candidate_cost, in->model_cost_change, in->step_norm2,
          in->candidate_norm2, in->solve_info};
  lm_decide(s, li);
  st->radius = s.radius, st->decrease_factor = s.dec_factor, st->cost = s.cost, st->gradient_max_norm = s.gmax, st->x_norm = s.x_norm;
  st->iterations = s.iter, st->successful_steps = s.nsucc, st->invalid_steps = s.invalid, st->lin_unread = s.lin_unread;
  st->accepted = s.parity, st->stop = s.stop;
  return SFMHIP_OK;
}

// Test hook: residual and Jacobian of n observations exactly as the solver's kernels linearise them
// (cam_table + obs_linearize: the analytic derivative of the branch of AngleAxisRotatePoint that autodiff
// takes, reference src/BundleAdjustment.cpp:10-35), one thread per observation.
namespace {
__global__ void ba_linearize_obs_kernel(const double* __restrict__ cams6, const double* __restrict__ pts3, double focal,
                                        const double* __restrict__ obs_xy, int n, double* __restrict__ r,
                                        double* __restrict__ Jc, double* __restrict__ Jp, double* __restrict__ Jf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double cd[CAMD];
  cam_table(cams6 + 6 * (size_t)i, cd, true);
  const double X[3] = {pts3[3 * (size_t)i], pts3[3 * (size_t)i + 1], pts3[3 * (size_t)i + 2]};
  ObsLin o;
  obs_linearize((const double*)cd, X, focal, obs_xy[2 * (size_t)i], obs_xy[2 * (size_t)i + 1], nullptr, nullptr, 1.0, o);
  r[2 * (size_t)i] = o.r0;
  r[2 * (size_t)i + 1] = o.r1;
  for (int k = 0; k < 12; ++k) Jc[12 * (size_t)i + k] = o.Jc[k];
  for (int k = 0; k < 6; ++k) Jp[6 * (size_t)i + k] = o.Jp[k];
  Jf[2 * (size_t)i] = o.Jf[0];
  Jf[2 * (size_t)i + 1] = o.Jf[1];
}
}  // namespace

extern "C" int sfmhip_ba_linearize_obs(sfmhip_ctx* ctx, int n, const double* cams6, const double* pts3, double focal,
                                       const double* obs_xy, double* r, double* Jc, double* Jp, double* Jf) {
  if (!ctx || n < 0 || (n && (!cams6 || !pts3 || !obs_xy || !r || !Jc || !Jp || !Jf))) return SFMHIP_ERR_ARG;
  if (n == 0) return SFMHIP_OK;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  double *d_in = nullptr, *d_out = nullptr;
  const size_t nin = (size_t)n * 11, nout = (size_t)n * 22;
  int rc = sfm_dev_alloc(&d_in, nin);
  if (rc == SFMHIP_OK) rc = sfm_dev_alloc(&d_out, nout);
  if (rc != SFMHIP_OK) {
    hipFree(d_in);
    hipFree(d_out);
    return rc;
  }
  hipStream_t st = ctx->stream;
  hipError_t e = hipMemcpyAsync(d_in, cams6, sizeof(double) * 6 * n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_in + 6 * (size_t)n, pts3, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_in + 9 * (size_t)n, obs_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ba_linearize_obs_kernel, dim3((n + 127) / 128), dim3(128), 0, st, d_in, d_in + 6 * (size_t)n, focal,
                       d_in + 9 * (size_t)n, n, d_out, d_out + 2 * (size_t)n, d_out + 14 * (size_t)n, d_out + 20 * (size_t)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(r, d_out, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(Jc, d_out + 2 * (size_t)n, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(Jp, d_out + 14 * (size_t)n, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(Jf, d_out + 20 * (size_t)n, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(d_in);
  hipFree(d_out);
  if (e != hipSuccess) {
    g_sfmhip_last_hip_error = (int)e;
    return SFMHIP_ERR_HIP;
  }
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_reduced_system(sfmhip_ba* b, double radius, double* S, double* g, double* cost) {
  if (!b || !(radius > 0)) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  sfmhip_ba_opts o;
  sfmhip_ba_default_opts(&o);
  if (!b->scale_ready) SFM_TRY(ba_prepare_scale(b, o.jacobi_scaling));
  // (a test hook on a live object: the LM loop's pending linearisation is read out first, and what the hook leaves in
  // the reduced-system buffer is not that linearisation -- the next iterate linearises again)
  SFM_TRY(ba_flush_lin(b, &o));
  b->lm.have_lin = false;
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

extern "C" int sfmhip_ba_reduced_step(sfmhip_ba* b, double radius, double* z, int* chol_failed) {
  if (!b || !z || !(radius > 0)) return SFMHIP_ERR_ARG;
  if (b->world != 1) return SFMHIP_ERR_UNSUPPORTED;
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  sfmhip_ba_opts o;
  sfmhip_ba_default_opts(&o);
  if (!b->nd_ready) SFM_TRY(ba_nd_build(b));
  if (!b->scale_ready) SFM_TRY(ba_prepare_scale(b, o.jacobi_scaling));
  SFM_TRY(ba_flush_lin(b, &o));  // (as in sfmhip_ba_reduced_system: the hook's system is not the LM loop's)
  b->lm.have_lin = false;
  SFM_TRY(ba_linearize_eliminate(b, radius, &o, true));
  SFM_TRY(ba_reduced_solve(b));
  hipStream_t st = b->ctx->stream;
  int info = 0;
  SFM_HIP_TRY(hipMemcpyAsync(z, b->d.z, sizeof(double) * b->dim, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipMemcpyAsync(&info, b->d.info, sizeof(int), hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (chol_failed) *chol_failed = info;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_reduced_layout(sfmhip_ba* b, int32_t layout[4]) {
  if (!b || !layout) return SFMHIP_ERR_ARG;
  const bool chains = b->nd_on && !b->tree_on;
  layout[0] = chains ? b->nd.n : 0;
  layout[1] = chains ? b->nd_max_ni : 0;
  layout[2] = chains ? b->nd.c[b->nd.n].N : 0;
  layout[3] = b->nd_ready ? b->ld / CB : 0;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_reduced_tree(sfmhip_ba* b, int32_t tree[4]) {
  if (!b || !tree) return SFMHIP_ERR_ARG;
  tree[0] = b->tree_on ? b->tree_fs.n_fronts : 0;
  tree[1] = b->tree_on ? b->tree_levels : 0;
  tree[2] = b->tree_on ? b->tree_chain_tiles : 0;
  tree[3] = b->tree_on ? b->tree_max_T : 0;
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
  const bool prof_ = getenv("SFMHIP_PROFILE_CREATE") != nullptr;
  auto tp_ = std::chrono::steady_clock::now();
  auto lap_ = [&](const char* what) {
    if (!prof_) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[sfmhip_ba_destroy] %-21s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp_).count());
    tp_ = now;
  };
  hipSetDevice(b->ctx->device);
  const size_t n_free = b->allocs.size();
  for (void* p : b->allocs) hipFree(p);
  if (prof_) fprintf(stderr, "[sfmhip_ba_destroy] %zu hipFree\n", n_free);
  lap_("hipFree");
  if (b->h_sc && !b->pinned_shared) hipHostFree(b->h_sc);
  if (b->h_ring && !b->pinned_shared) hipHostFree(b->h_ring);
  lap_("hipHostFree");
  for (auto& e : b->ev)
    if (e) hipEventDestroy(e);
  lap_("events");
  if (b->use_arena && b->ctx->ba_host_scratch) {  // (a one-shot problem: its large vectors go back to the context's)
    BaHostScratch* hs = (BaHostScratch*)b->ctx->ba_host_scratch;
    hs->obs_src.swap(b->obs_src);
    hs->cxy_src.swap(b->cxy_src);
    hs->h_pts_in.swap(b->h_pts_in);
  }
  delete b;
  lap_("host memory");
}

// New measurements for a problem of unchanged structure (the same obs_cam / obs_pt arrays as at its creation): the sorted
// copies are re-gathered through the maps the set-up left and uploaded; nothing of the plan changes.
static int ba_set_observations(sfmhip_ba* b, const double* obs_xy) {
  SFM_HIP_TRY(hipSetDevice(b->ctx->device));
  if (!b->no) return SFMHIP_OK;
  SFM_TRY(ba_upload_xy(b, obs_xy, b->no_in));
  const size_t nfo = b->cxy_src.size();
  if (nfo) {  // (the pair path's camera-major copy: few points, or none)
    std::vector<double> cxy(2 * nfo);
    for (size_t k = 0; k < nfo; ++k) {
      const size_t o = (size_t)b->obs_src[b->cxy_src[k]];
      cxy[2 * k] = obs_xy[2 * o];
      cxy[2 * k + 1] = obs_xy[2 * o + 1];
    }
    SFM_HIP_TRY(hipMemcpy(b->d_cxy, cxy.data(), sizeof(double) * 2 * nfo, hipMemcpyHostToDevice));
  }
  return SFMHIP_OK;
}

// The one-shot entry point keeps the LAST problem it built with the context (BundleAdjustment::adjustBundle is a static one-shot
// function, include/BundleAdjustment.h:19-20, that the reference means to call again and again -- src/Sfm.cpp:883-888, :996):
// a call whose observation structure (n_cam, n_pt, obs_cam[], obs_pt[]) equals the kept one's skips the whole set-up -- grouping,
// signature sort, chunking, gather lists, camera graph and dissection, allocations, uploads -- and only takes the new
// measurements and parameters.  The arrays are compared element for element (no hash to collide).  SFMHIP_BA_PLAN_CACHE=0
// switches it off; the kept problem (its device buffers) lives until another structure replaces it or the context shuts down.
struct BaPlanCache {
  sfmhip_ba* b = nullptr;
  int n_cam = 0, n_pt = 0, n_obs = 0;
  std::vector<int32_t> obs_cam, obs_pt;
};
static void ba_plan_cache_free(void* p) {
  BaPlanCache* c = (BaPlanCache*)p;
  if (!c) return;
  sfmhip_ba_destroy(c->b);
  delete c;
}
static bool ba_same_i32(const int32_t* a, const int32_t* c, int n) {
  bool same = true;
  std::vector<char> part((size_t)host_threads(n) + 1, 1);
  host_parallel_for_t(n, host_threads(n), [&](int t, int lo, int hi) {
    part[t] = memcmp(a + lo, c + lo, sizeof(int32_t) * (size_t)(hi - lo)) == 0;
  });
  for (char v : part) same = same && v;
  return same;
}

extern "C" int sfmhip_ba_solve(sfmhip_ctx* ctx, int n_cam, int n_pt, int n_obs, double* cams6, double* pts3,
                               double* focal, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                               const sfmhip_ba_opts* opts, sfmhip_ba_summary* summary) {
  if (!ctx || !cams6 || !focal) return SFMHIP_ERR_ARG;
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  const auto tstart = t0;
  auto lap = [&]() {
    const auto now = clk::now();
    const double ms = std::chrono::duration<double, std::milli>(now - t0).count();
    t0 = now;
    return ms;
  };
  sfmhip_ba_solve_profile& pr = ctx->ba_profile;
  pr = sfmhip_ba_solve_profile{};
  static const bool cache_on = !(getenv("SFMHIP_BA_PLAN_CACHE") && atoi(getenv("SFMHIP_BA_PLAN_CACHE")) == 0);
  BaPlanCache* cache = (BaPlanCache*)ctx->ba_cache;
  sfmhip_ba* b = nullptr;
  int rc = SFMHIP_OK;
  if (cache_on && cache && cache->b && cache->n_cam == n_cam && cache->n_pt == n_pt && cache->n_obs == n_obs && n_obs > 0 &&
      obs_cam && obs_pt && obs_xy && ba_same_i32(cache->obs_cam.data(), obs_cam, n_obs) &&
      ba_same_i32(cache->obs_pt.data(), obs_pt, n_obs)) {
    b = cache->b;
    pr.plan_reused = 1;
    b->spin_timeouts = 0;
    rc = ba_set_observations(b, obs_xy);
  } else {
    if (cache) {  // another structure: the kept problem goes first (the block it was carved from is the new one's)
      sfmhip_ba_destroy(cache->b);
      cache->b = nullptr;
    }
    // the block: as large as the last problem turned out to need, and a quarter more (a problem whose needs exceed it takes
    // the rest by hipMalloc, and the next call's block is larger)
    static const bool arena_on = !(getenv("SFMHIP_BA_ARENA") && atoi(getenv("SFMHIP_BA_ARENA")) == 0);
    if (arena_on && ctx->ba_arena_need > ctx->ba_arena_bytes) {
      hipSetDevice(ctx->device);
      if (ctx->ba_arena) hipFree(ctx->ba_arena);
      ctx->ba_arena = nullptr, ctx->ba_arena_bytes = 0;
      const size_t want = ctx->ba_arena_need + ctx->ba_arena_need / 4;
      if (hipMalloc(&ctx->ba_arena, want) == hipSuccess) ctx->ba_arena_bytes = want;
      else ctx->ba_arena = nullptr;  // (no block: plain allocations)
    }
    rc = ba_create_impl(ctx, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_xy, arena_on, &b);
  }
  pr.create_ms = lap();
  if (rc == SFMHIP_OK) rc = sfmhip_ba_set_params(b, cams6, pts3, *focal);
  pr.set_params_ms = lap();
  if (rc == SFMHIP_OK) rc = sfmhip_ba_run(b, opts, summary);
  pr.run_ms = lap();
  pr.front_plan_reused = b && !pr.plan_reused && b->nd_kept ? 1 : 0;
  if (rc == SFMHIP_OK) rc = sfmhip_ba_get_params(b, cams6, pts3, focal);
  pr.get_params_ms = lap();
  if (b && !pr.plan_reused) ctx->ba_arena_need = std::max(ctx->ba_arena_need, b->arena_need);
  if (rc == SFMHIP_OK && cache_on && n_obs > 0) {
    if (!pr.plan_reused) {
      if (!cache) {
        cache = new BaPlanCache();
        ctx->ba_cache = cache;
        ctx->ba_cache_free = ba_plan_cache_free;
      }
      cache->b = b;
      cache->n_cam = n_cam, cache->n_pt = n_pt, cache->n_obs = n_obs;
      cache->obs_cam.assign(obs_cam, obs_cam + n_obs);
      cache->obs_pt.assign(obs_pt, obs_pt + n_obs);
    }
  } else {
    if (cache && cache->b == b) cache->b = nullptr;  // (a failed solve is not kept)
    sfmhip_ba_destroy(b);
  }
  pr.keep_ms = lap();
  pr.total_ms = std::chrono::duration<double, std::milli>(clk::now() - tstart).count();
  return rc;
}

extern "C" int sfmhip_host_parallel_for(int n, void (*fn)(int lo, int hi, void* user), void* user) {
  if (n < 0 || !fn) return SFMHIP_ERR_ARG;
  if (n == 0) return SFMHIP_OK;
  host_parallel_for(n, [&](int lo, int hi) { fn(lo, hi, user); });
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_last_solve_profile(sfmhip_ctx* ctx, sfmhip_ba_solve_profile* out) {
  if (!ctx || !out) return SFMHIP_ERR_ARG;
  *out = ctx->ba_profile;
  return SFMHIP_OK;
}
