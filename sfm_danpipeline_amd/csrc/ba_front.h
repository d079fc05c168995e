// ba_front.h -- device side of the front tree (included by ba.hip inside its anonymous namespace, behind the blocked
// Cholesky helpers it builds on: v4d, lds_wait_ge, lds_flag_set/add, chol2_put/get, rsqrt_f64_h, readlane_f64).
//
// front_up    one workgroup per front, children before parents in the grid.  The workgroup assembles its front from S
//             (+ the LM diagonal), waits for its children's contribution blocks (one lane polls their flags, payload
//             stored and loaded write-through: MI355X_MICROARCH "visibility", hand-off form R1), adds them, and factors
//             its own columns WITHOUT LEAVING THE CU: the chain wave factors the diagonal tile of step j four columns
//             at a time in the MFMA accumulator layout, the tile waves solve the tiles of column j against it as its
//             blocks appear (panel j, kept in LDS) and fold panel j into every tile they hold in registers; the next
//             diagonal tile reaches the chain wave through LDS one update short and gets the last one from it.  The
//             right-hand side rides along on a wave of its own.  L (row-major, for the down-sweep), the contribution
//             block U and the rhs go to memory; one flag store tells the parent.
// front_down  parents before children: z_v = L_vv^-T (y_v - L_bv^T z_b), the rows of L prefetched into registers before
//             the parent's flag arrives; the front's cameras then get their candidate parameters and tables
//             (what ba_cand_cams does for the other factorisations).
// Replaces Eigen's LLT / triangular solves behind ceres::Solve(DENSE_SCHUR), reference src/BundleAdjustment.cpp:116,123.
#pragma once

// acc -= a b: the f64 MFMA reads its blgp field as neg:[a, b, c].  Negating an operand with a vector instruction instead ties
// the MFMA to the LDS read that produced it -- the compiler put the next block's negations (and the wait for their loads)
// ahead of this block's MFMAs in every update loop.
#define FR_MFMA_SUB(acc, a, b) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 1)

#ifdef SFM_FRONT_STAMPS
// diagnostic build only (scripts/front_stamps.py): shader-clock stamps of the chain wave and a few others, per front
__device__ unsigned long long g_front_stamps[128][128];  // [32 + 8 wave + ...]: the tile waves' deferred phase
#define FR_STAMP(slot)                                                            \
  do {                                                                            \
    unsigned long long t_;                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
    if (f < 128 && lane == 0) g_front_stamps[f][slot] = t_;                       \
  } while (0)
#define FR_STAMP_REAL(slot)                                                       \
  do {                                                                            \
    unsigned long long t_;                                                        \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    if (f < 128 && lane == 0) g_front_stamps[f][slot] = t_;                       \
  } while (0)
__device__ unsigned long long g_down_stamps[128][8];
#define FD_STAMP(slot)                                                            \
  do {                                                                            \
    unsigned long long t_;                                                        \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    if (f < 128 && threadIdx.x == 0) g_down_stamps[f][slot] = t_;                 \
  } while (0)
#else
#define FR_STAMP(slot)
#define FR_STAMP_REAL(slot)
#define FD_STAMP(slot)
#endif

constexpr int FR_WAVES = fplan::FP_WAVES, FR_SLOTS = fplan::FP_SLOTS, FR_TMAX = fplan::FP_T_MAX;
constexpr int FR_TILE = CB * CBP;  // doubles of a 32 x 32 tile in LDS ([row][col], pitch CBP)
// dynamic LDS of front_up (doubles): two generations of panel tiles (rows 1..T-1 of the step's block column) | the
// factored diagonal tile of two steps | the staging tile of the next diagonal tile | 1/diag of two steps | the rhs |
// the front's index map | the counters (two 64-byte blocks, each ending in a timeout mark)
constexpr int FR_OFF_PANEL = 0;
constexpr int FR_OFF_SD = FR_OFF_PANEL + 2 * (FR_TMAX - 1) * FR_TILE;
constexpr int FR_OFF_SDG = FR_OFF_SD + 2 * FR_TILE;
constexpr int FR_OFF_SDI = FR_OFF_SDG + FR_TILE;
constexpr int FR_OFF_Y = FR_OFF_SDI + 2 * CB;
constexpr int FR_OFF_INV = FR_OFF_Y + CB * FR_TMAX;
constexpr int FR_OFF_FLAG = FR_OFF_INV + CB * FR_TMAX / 2;
constexpr int FR_LDS_BYTES = (FR_OFF_FLAG + 16) * 8;
static_assert((FR_OFF_FLAG * 8) % 64 == 0, "lds_wait_ge finds the timeout mark in the counter's 64-byte block");
static_assert(FR_LDS_BYTES <= 160 * 1024, "one front per CU");
// counters: block 0 = prog[gen][row] (14) + progL[0], mark at 15; block 1 = progL[1], cons[2], dready, dtaken, turn[4 SIMDs], mark at 31
enum { FRC_PROG = 0, FRC_PROGL0 = 14, FRC_MARK0 = 15, FRC_PROGL1 = 16, FRC_CONS = 17, FRC_DREADY = 19, FRC_DTAKEN = 20, FRC_TURN = 21, FRC_MARK1 = 31 };
static_assert(2 * FR_TMAX <= FRC_PROGL0, "panel progress counters");

// (a NaN no solve produces: the mailbox entries are polled until they stop being it -- the data is its own flag)
#define FR_Z_PENDING 0x7FF8DEADBEEF0000ull
struct FrontSet {
  const int* ints;        // fplan::Flat::ints: descriptors, then the pools
  const int* up_order;    // front of the k-th up-sweep workgroup (deepest level first)
  const int* down_order;  // front of the k-th down-sweep workgroup (root first)
  double* pool;           // per front: L | U | y
  unsigned* tflag;        // per tile of a front's contribution block (+ one for its rhs): the epoch of the solve that finished it
  unsigned* flag_down;    // per front: the epoch of the solve whose z is in place
  double* zq;             // the down-sweep's mailbox, two generations of ld doubles (solve e uses e & 1): z per parameter,
                          // FR_Z_PENDING until its front has solved it; a front resets its entries of the OTHER generation
  int zq_ld;
  int n_fronts;
  int n_roles;            // workgroups of the up-sweep with a front's or a helper's role (up_order: front | helper << 16)
};

struct FrDesc {
  int no, ns, T, nb_last, parent, nchild, child_off, inv_off, ptinv_off, sched_off, ncam, cam_off, has_focal, offL, offy, own_cols, off_pbuf, live,
      focal_pos, tflag_off, nhelp, help_off, pflag_off;
};
__device__ __forceinline__ FrDesc fr_desc(const int* __restrict__ ints, int f) {
  const int* p = ints + (size_t)fplan::FD_INTS * f;
  FrDesc D;
  D.no = p[fplan::FD_NO], D.ns = p[fplan::FD_NS], D.T = p[fplan::FD_T], D.nb_last = p[fplan::FD_NB_LAST];
  D.parent = p[fplan::FD_PARENT], D.nchild = p[fplan::FD_NCHILD], D.child_off = p[fplan::FD_CHILD_OFF];
  D.inv_off = p[fplan::FD_INV_OFF], D.ptinv_off = p[fplan::FD_PTINV_OFF], D.sched_off = p[fplan::FD_SCHED_OFF];
  D.ncam = p[fplan::FD_NCAM], D.cam_off = p[fplan::FD_CAM_OFF], D.has_focal = p[fplan::FD_HAS_FOCAL];
  D.offL = p[fplan::FD_OFF_L], D.offy = p[fplan::FD_OFF_Y], D.own_cols = p[fplan::FD_OWN_COLS];
  D.off_pbuf = p[fplan::FD_OFF_PBUF], D.live = p[fplan::FD_LIVE], D.focal_pos = p[fplan::FD_FOCAL_POS], D.tflag_off = p[fplan::FD_TFLAG_OFF];
  D.nhelp = p[fplan::FD_NHELP], D.help_off = p[fplan::FD_HELP_OFF], D.pflag_off = p[fplan::FD_PFLAG_OFF];
  return D;
}

// write-through / L1-bypassing accesses of what another workgroup of the SAME launch reads or wrote
__device__ __forceinline__ double fr_ld_sc1(const double* p) {
  return __hip_atomic_load((gbl_double*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fr_st_sc1(double* p, double v) {
  __hip_atomic_store((gbl_double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef __attribute__((address_space(1))) unsigned gbl_u32;
// until *flag == epoch (every lane polls the same word; the branch is uniform); bounded (~tens of ms), a spin that runs out
// leaves a timeout mark in LDS and the launch reports a failed factorisation
__device__ __forceinline__ void fr_poll_flag(const unsigned* flag, unsigned epoch, int* s_mark) {
  int budget = 1 << 17;
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load((const gbl_u32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != epoch) {
    if (--budget == 0) {
      *(volatile lds_int*)s_mark = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  asm volatile("" ::: "memory");  // (compiler only: no load of the flagged data may be moved above the loop's exit; the hardware order is the sender's s_waitcnt + the sc1 loads)
}
// the other side: this wave's stores have left, then the word
__device__ __forceinline__ void fr_raise_flag(unsigned* flag, unsigned epoch, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store((gbl_u32*)flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tile (r, c) of the front in the accumulator layout ([2 * ci + ri]; lane = row 16 ri + j16, registers = columns
// 16 ci + q + 4 g): entries of S for the own columns (S holds its upper triangle row-major), the LM diagonal, the
// identity on padded own columns, zero in the border x border block.  Every load is issued whatever the entry turns out
// to be (a clamped address, the value selected afterwards): loads inside branches go out one round trip at a time.
__device__ __forceinline__ void fr_assemble(v4d (&t)[4], int r, int c, int no, const int* sInv, const double* __restrict__ S, int ldS,
                                            const double* __restrict__ dc, int fin, double inv_radius, double lm_lo, double lm_hi, int lane) {
  const int j16 = lane & 15, q = lane >> 4;
  if (c >= no) {
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = v4d{0.0, 0.0, 0.0, 0.0};
    return;
  }
  int ga[2];
  double dg[2];  // the LM diagonal of the lane's two rows (used where row == column)
#pragma unroll
  for (int ri = 0; ri < 2; ++ri) {
    ga[ri] = sInv[CB * r + 16 * ri + j16];
    dg[ri] = (fin && ga[ri] >= 0) ? fmin(fmax(dc[ga[ri]], lm_lo), lm_hi) * inv_radius : 0.0;
  }
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {  // (eight loads in flight at a time: the wave's three tiles leave few registers)
    int gb[4];
    double v[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) gb[g] = sInv[CB * c + 16 * ci + q + 4 * g];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int x = ga[ri], y = gb[g];
        const bool ok = x >= 0 && y >= 0;
        v[4 * ri + g] = S[ok ? (size_t)min(x, y) * ldS + max(x, y) : 0];
      }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int x = ga[ri], y = gb[g];
        const int a = CB * r + 16 * ri + j16, b = CB * c + 16 * ci + q + 4 * g;
        double val = (x >= 0 && y >= 0) ? v[4 * ri + g] : (a == b ? 1.0 : 0.0);
        if (x >= 0 && x == y) val += dg[ri];
        t[2 * ci + ri][g] = val;
      }
  }
}
// A contribution tile crosses from a child to its parent as 8 KB in the accumulator layout by register pairs: element
// (register idx = 4 sub + g, lane) at 16-byte chunk (idx >> 1) * 64 + lane, half idx & 1 -- a wave's store or load is 1 KB
// contiguous, 16 bytes per lane, write-through / L1-bypassing (sc1).  The borders are whole tiles of ancestors' layouts
// (ba_front_plan.h), so tile (i, j) of the child's border x border block IS tile (ptile[i], ptile[j]) of the parent.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u fr_pack2(double a, double b) {
  return v4u{(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
}
// live: bits (2 ci + ri) of the sub-tiles that hold anything (dead row halves of the border are neither sent nor read)
__device__ __forceinline__ void fr_send(const v4d (&t)[4], __amdgpu_buffer_rsrc_t pool, int tile_off_bytes, unsigned live, int lane) {
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
    if ((live >> sub) & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_buffer_store_b128(fr_pack2(t[sub][2 * h], t[sub][2 * h + 1]), pool,
                                               tile_off_bytes + ((2 * sub + h) * 64 + lane) * 16, 0, 16 /* sc1 */);
    }
}
// (Round 5 measured the alternative in which the data is its own flag -- the buffer holds a NaN until the sender has stored the
// tile, the receiver re-arms it: one round trip and the sender's wait for its stores less per hand-off on paper.  Slower by
// 10 us per solve as built: the polling loop around eight 16-byte loads per lane costs front_up the registers it does not have
// (24 bytes of scratch per lane at its 168-register limit), and a parent polls for as long as its children work.)
__device__ __forceinline__ void fr_recv(v4d (&t)[4], __amdgpu_buffer_rsrc_t pool, int tile_off_bytes, unsigned live, int lane) {
  v4u v[8];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
    if ((live >> sub) & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        v[2 * sub + h] = __builtin_amdgcn_raw_buffer_load_b128(pool, tile_off_bytes + ((2 * sub + h) * 64 + lane) * 16, 0, 16 /* sc1 */);
    }
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
    if ((live >> sub) & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        t[sub][2 * h] += __hiloint2double((int)v[2 * sub + h].y, (int)v[2 * sub + h].x);
        t[sub][2 * h + 1] += __hiloint2double((int)v[2 * sub + h].w, (int)v[2 * sub + h].z);
      }
    }
}
// two contribution tiles at once (the chain wave: 64 registers of loads in flight)
__device__ __forceinline__ void fr_recv2(v4d (&t)[4], __amdgpu_buffer_rsrc_t pool, int off0, unsigned live0, int off1, unsigned live1, int lane) {
  v4u v[8], w[8];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    if ((live0 >> sub) & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) v[2 * sub + h] = __builtin_amdgcn_raw_buffer_load_b128(pool, off0 + ((2 * sub + h) * 64 + lane) * 16, 0, 16 /* sc1 */);
    }
    if ((live1 >> sub) & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) w[2 * sub + h] = __builtin_amdgcn_raw_buffer_load_b128(pool, off1 + ((2 * sub + h) * 64 + lane) * 16, 0, 16 /* sc1 */);
    }
  }
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {  // (child 0's first, then child 1's: the order fr_recv twice adds them in)
    if ((live0 >> sub) & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        t[sub][2 * h] += __hiloint2double((int)v[2 * sub + h].y, (int)v[2 * sub + h].x);
        t[sub][2 * h + 1] += __hiloint2double((int)v[2 * sub + h].w, (int)v[2 * sub + h].z);
      }
    }
  }
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    if ((live1 >> sub) & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        t[sub][2 * h] += __hiloint2double((int)w[2 * sub + h].y, (int)w[2 * sub + h].x);
        t[sub][2 * h + 1] += __hiloint2double((int)w[2 * sub + h].w, (int)w[2 * sub + h].z);
      }
    }
  }
}

// which sub-tiles [2 ci + ri] of tile (r, c) hold anything: row half ri of r and row half ci of c live, not above the diagonal
__device__ __forceinline__ unsigned fr_live_subs(unsigned live, int r, int c) {
  const unsigned lr = (live >> (2 * r)) & 3, lc = (live >> (2 * c)) & 3;
  unsigned m = 0;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
      if (((lc >> ci) & 1) && ((lr >> ri) & 1) && !(r == c && ci == 1 && ri == 0)) m |= 1u << (2 * ci + ri);
  return m;
}

// (keeps the optimiser from cloning a step's body per value: the kernel is larger than the instruction cache as it is)
__device__ __forceinline__ int opaque_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
// POTRF of the diagonal tile in acc ([0] = rows 0-15 x cols 0-15, [1] = rows 16-31 x cols 0-15, [2] = rows 16-31 x cols
// 16-31), nb blocks of four columns (the last own tile of a front stops at its last real column).  sD receives L
// (row-major; above the diagonal and right of column 4 nb: unspecified), sdi 1/diag, *prog = base + finished blocks;
// gL_bytes: where the copy in memory starts in the pool (row-major, leading dimension ldk; the down-sweep, a later
// launch, reads it).
__device__ __forceinline__ void fr_potrf(v4d (&acc)[3], double* sD, double* sdi, int* prog, int base, int nb, bool& bad, int lane,
                                         __amdgpu_buffer_rsrc_t pool, int gL_bytes, int ldk) {
  const int i = lane & 31, j16 = lane & 15, q = lane >> 4;
  double* const rowp = sD + i * CBP;
  double* const op0 = sD + j16 * CBP + q;
  double* const op1 = op0 + 16 * CBP;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if (b < nb) {
      const int c = 4 * b, g = b & 3;
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
        l[k] = v[k] * r[k];
#pragma unroll
        for (int m = k + 1; m < 4; ++m) v[m] -= l[k] * readlane_f64(l[k], c + m);
      }
      *(v2d*)(rowp + c) = v2d{l[0], l[1]};
      *(v2d*)(rowp + c + 2) = v2d{l[2], l[3]};
      *(v2d*)(sdi + c) = v2d{r[0], r[1]};
      *(v2d*)(sdi + c + 2) = v2d{r[2], r[3]};
      lds_flag_set(prog, base + b + 1);
      if (lane < CB) {
        // (the row's four entries of this block; above the diagonal: what the registers hold, never read)
        __builtin_amdgcn_raw_buffer_store_b128(fr_pack2(l[0], l[1]), pool, gL_bytes + (i * ldk + c) * 8, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(fr_pack2(l[2], l[3]), pool, gL_bytes + (i * ldk + c + 2) * 8, 0, 0);
      }
      if (b + 1 < nb) {
        const double L0 = op0[c], L1 = op1[c];
        if (b < 3) {
          FR_MFMA_SUB(acc[0], L0, L0);
          FR_MFMA_SUB(acc[1], L0, L1);
        }
        FR_MFMA_SUB(acc[2], L1, L1);
      }
    }
  }
}

// X = T L^-T for the tile in acc ([2 * ci + ri]), trailing the factorisation of L block by block through *prog.  sT: the
// tile's LDS home (scratch for the layout changes, X when done, row-major); *oflag = base + finished blocks; out_bytes:
// the lane's row of the copy in memory, as an offset into the pool (its columns of this block column are contiguous).
__device__ __forceinline__ void fr_trsm(v4d (&acc)[4], double* sT, const double* sL, const double* sdi, const int* prog, int* oflag,
                                        int base, int nb, __amdgpu_buffer_rsrc_t pool, int out_bytes, int lane,
                                        bool through /* helper workgroups of this launch read the rows: write-through */) {
  const int i = lane & 31, j16 = lane & 15, q = lane >> 4;
  double* const rowp = sT + i * CBP;
  double* const op0 = sT + j16 * CBP + q;
  double* const op1 = op0 + 16 * CBP;
  const double* const lop0 = sL + j16 * CBP + q;
  const double* const lop1 = lop0 + 16 * CBP;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if (b < nb) {
      const int c = 4 * b, g = b & 3, cb = b >> 2;
      op0[c] = acc[2 * cb][g];
      op1[c] = acc[2 * cb + 1][g];
      const v2d lo = *(const v2d*)(rowp + c), hi = *(const v2d*)(rowp + c + 2);
      lds_wait_ge(prog, base + b + 1);
      const double l10 = sL[(c + 1) * CBP + c];
      const v2d l2 = *(const v2d*)(sL + (c + 2) * CBP + c);
      const v2d l3 = *(const v2d*)(sL + (c + 3) * CBP + c);
      const double l32 = sL[(c + 3) * CBP + c + 2];
      const v2d r01 = *(const v2d*)(sdi + c), r23 = *(const v2d*)(sdi + c + 2);
      double La0 = 0.0, La1 = 0.0;
      if (b + 1 < nb) {
        if (b < 3) La0 = lop0[c];
        La1 = lop1[c];
      }
      const double x0 = lo.x * r01.x;
      const double x1 = (lo.y - x0 * l10) * r01.y;
      const double x2 = (hi.x - x0 * l2.x - x1 * l2.y) * r23.x;
      const double x3 = (hi.y - x0 * l3.x - x1 * l3.y - x2 * l32) * r23.y;
      *(v2d*)(rowp + c) = v2d{x0, x1};
      *(v2d*)(rowp + c + 2) = v2d{x2, x3};
      lds_flag_set(oflag, base + b + 1);
      if (lane < CB) {
        if (through) {
          __builtin_amdgcn_raw_buffer_store_b128(fr_pack2(x0, x1), pool, out_bytes + c * 8, 0, 16 /* sc1 */);
          __builtin_amdgcn_raw_buffer_store_b128(fr_pack2(x2, x3), pool, out_bytes + (c + 2) * 8, 0, 16);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(fr_pack2(x0, x1), pool, out_bytes + c * 8, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(fr_pack2(x2, x3), pool, out_bytes + (c + 2) * 8, 0, 0);
        }
      }
      if (b + 1 < nb) {
        const double X0 = op0[c], X1 = op1[c];
        if (b < 3) {
          FR_MFMA_SUB(acc[0], La0, X0);
          FR_MFMA_SUB(acc[1], La0, X1);
        }
        FR_MFMA_SUB(acc[2], La1, X0);
        FR_MFMA_SUB(acc[3], La1, X1);
      }
    }
  }
}

// T(r, c) -= X_r X_c^T over the nb blocks of a panel, block by block as the two solves finish them; subs: the sub-tiles
// [2 ci + ri] that hold anything (fr_live_subs: dead row halves, and above the diagonal of a diagonal tile, are skipped).
// The operands of block kb + 1 are read while the MFMAs of block kb run (a wave that folds alone on its SIMD has nobody
// else to hide the LDS latency behind).
__device__ __forceinline__ void fr_update(v4d (&acc)[4], const double* Xr, const int* progr, const double* Xc, const int* progc, int base,
                                          int nb, unsigned subs, int lane) {
  const int j16 = lane & 15, q = lane >> 4;
  const double* const r0 = Xr + j16 * CBP + q;
  const double* const c0 = Xc + j16 * CBP + q;
  int seen = 0;
  auto ready = [&](int kb) {
    if (seen < base + kb + 1) {
      const int a = __builtin_amdgcn_readfirstlane(lds_wait_ge(progr, base + kb + 1));
      const int b = __builtin_amdgcn_readfirstlane(lds_wait_ge(progc, base + kb + 1));
      seen = min(a, b);
    }
  };
  if (nb <= 0) return;
  ready(0);
  double R0 = r0[0], R1 = r0[16 * CBP], C0 = c0[0], C1 = c0[16 * CBP];
#pragma unroll 1  // (instantiated per register slot: code size, not loop overhead, is what counts here)
  for (int kb = 0; kb < nb; ++kb) {
    double nR0 = 0, nR1 = 0, nC0 = 0, nC1 = 0;
    if (kb + 1 < nb) {
      ready(kb + 1);
      nR0 = r0[4 * (kb + 1)], nR1 = r0[16 * CBP + 4 * (kb + 1)];
      nC0 = c0[4 * (kb + 1)], nC1 = c0[16 * CBP + 4 * (kb + 1)];
    }
    if (subs & 1) FR_MFMA_SUB(acc[0], C0, R0);
    if (subs & 2) FR_MFMA_SUB(acc[1], C0, R1);
    if (subs & 4) FR_MFMA_SUB(acc[2], C1, R0);
    if (subs & 8) FR_MFMA_SUB(acc[3], C1, R1);
    R0 = nR0, R1 = nR1, C0 = nC0, C1 = nC1;
  }
}

// The same update for a panel that is COMPLETE (the deferred border tiles): four blocks' operands are read, then their sixteen
// MFMAs issued back to back.  Measured on one wave (scripts/front_ubench.py): an f64 MFMA leaves the pipe every 64 cycles when
// its operands sit in registers, but LDS reads do not complete under a stream of them -- the one-block-ahead loop above takes
// 470 cycles per block of four MFMAs (256 of matrix pipe), four blocks at a time 294, eight at a time 272 (64 operand registers).
__device__ __forceinline__ void fr_fold(v4d (&acc)[4], const double* Xr, const double* Xc, int nb, unsigned subs, int lane) {
  typedef const __attribute__((address_space(3))) double* ldsp;
  const int j16 = lane & 15, q = lane >> 4;
  const ldsp xr = (ldsp)(Xr + j16 * CBP + q);
  const ldsp xc = (ldsp)(Xc + j16 * CBP + q);
  int kb = 0;
  if (subs == 15u || subs == 11u) {
#pragma unroll 1
    for (; kb + 4 <= nb; kb += 4) {
      double R0[4], R1[4], C0[4], C1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) R0[k] = xr[4 * (kb + k)], R1[k] = xr[16 * CBP + 4 * (kb + k)], C0[k] = xc[4 * (kb + k)], C1[k] = xc[16 * CBP + 4 * (kb + k)];
      __builtin_amdgcn_sched_barrier(0);
      if (subs == 15u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          FR_MFMA_SUB(acc[0], C0[k], R0[k]);
          FR_MFMA_SUB(acc[1], C0[k], R1[k]);
          FR_MFMA_SUB(acc[2], C1[k], R0[k]);
          FR_MFMA_SUB(acc[3], C1[k], R1[k]);
        }
      } else {  // a diagonal tile: nothing above the diagonal
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          FR_MFMA_SUB(acc[0], C0[k], R0[k]);
          FR_MFMA_SUB(acc[1], C0[k], R1[k]);
          FR_MFMA_SUB(acc[3], C1[k], R1[k]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll 1
  for (; kb < nb; ++kb) {  // what is left of a short last panel, and tiles with dead row halves
    const double R0 = xr[4 * kb], R1 = xr[16 * CBP + 4 * kb], C0 = xc[4 * kb], C1 = xc[16 * CBP + 4 * kb];
    if (subs & 1) FR_MFMA_SUB(acc[0], C0, R0);
    if (subs & 2) FR_MFMA_SUB(acc[1], C0, R1);
    if (subs & 4) FR_MFMA_SUB(acc[2], C1, R0);
    if (subs & 8) FR_MFMA_SUB(acc[3], C1, R1);
  }
}

__device__ __forceinline__ void fr_report_timeout(const int* s_flag, int* __restrict__ info);

// the children's contributions to tile (R, C) of front D (a flag per tile; the front and its helpers alike).  Two children at
// a time: their two flags are polled TOGETHER (one round trip through memory when both are up, which they usually are by the time
// a tile is wanted -- polled one after the other, each in front of its own loads, a tile with two contributions cost four round
// trips, ~10 k cycles: profiles/round4_front_stamps.txt, "children done" and every "fetch<"), and a wave with registers to spare
// (WIDE: the chain wave, whose tile (0, 0) gates the level's first POTRF) has both tiles' loads in flight at once.
template <bool WIDE>
__device__ __forceinline__ void fr_recv_children(const FrontSet& fs, const FrDesc& D, v4d (&tt)[4], int R, int C, unsigned epoch,
                                                 __amdgpu_buffer_rsrc_t pool_rs, int* s_mark, int lane) {
  for (int k0 = 0; k0 < D.nchild; k0 += 2) {
    int off[2] = {0, 0};
    unsigned subs[2] = {0u, 0u};
    const unsigned* flg[2] = {nullptr, nullptr};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + u;
      if (k >= D.nchild) continue;
      const int* const chd = fs.ints + (size_t)fplan::FD_INTS * fs.ints[D.child_off + k];
      const int* const ptinv = fs.ints + D.ptinv_off + k * (FR_TMAX + 1);  // tile of this front -> border tile of the child, or -1
      const int i = ptinv[R], j = ptinv[C];
      if (i < 0 || j < 0) continue;
      const int cno = chd[fplan::FD_NO];
      subs[u] = fr_live_subs((unsigned)chd[fplan::FD_LIVE] >> (2 * cno), i, j);  // (the child's border tiles' live halves)
      if (!subs[u]) continue;
      flg[u] = fs.tflag + chd[fplan::FD_TFLAG_OFF] + i * (i + 1) / 2 + j;
      off[u] = (chd[fplan::FD_OFF_PBUF] + (i * (i + 1) / 2 + j) * (CB * CB)) * 8;
    }
    if (!subs[0] && !subs[1]) continue;
    {
      // (every lane polls the same words; the branch is uniform; bounded like fr_poll_flag)
      const unsigned* const f0 = subs[0] ? flg[0] : flg[1];
      const unsigned* const f1 = subs[1] ? flg[1] : flg[0];
      int budget = 1 << 17;
      for (;;) {
        const unsigned a = __hip_atomic_load((const gbl_u32*)f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned b = __hip_atomic_load((const gbl_u32*)f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(a) == epoch && __builtin_amdgcn_readfirstlane(b) == epoch) break;
        if (--budget == 0) {
          *(volatile lds_int*)s_mark = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      asm volatile("" ::: "memory");
    }
    if (WIDE && subs[0] && subs[1]) {
      fr_recv2(tt, pool_rs, off[0], subs[0], off[1], subs[1], lane);
    } else {
      if (subs[0]) fr_recv(tt, pool_rs, off[0], subs[0], lane);
      if (subs[1]) fr_recv(tt, pool_rs, off[1], subs[1], lane);
    }
  }
}

// A helper workgroup of front D (ba_front_plan.h, Front::nhelp): some of the front's border x border tiles are this
// workgroup's -- zero, plus the children's contributions, minus the front's panels -- folded on this compute unit's four
// matrix pipes while the front folds its own share.  The panels are the border rows of L that the front stores for the
// down-sweep anyway (write-through when it has helpers), a flag per solved tile; they are copied into LDS in the layout the
// front's own fold reads, every wave of the workgroup taking some, and folded from there.  The tiles go to the parent as the
// front's do: a contribution tile has a flag of its own, whoever computed it.
__device__ __forceinline__ void fr_helper(const FrontSet& fs, const FrDesc& D, int hlp, unsigned epoch, double* sAll, int* __restrict__ info) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int no = D.no, ns = D.ns, ldk = CB * no;
  double* const sPanel = sAll + FR_OFF_PANEL;
  int* const s_flag = (int*)(sAll + FR_OFF_FLAG);
  const __amdgpu_buffer_rsrc_t pool_rs = __builtin_amdgcn_make_buffer_rsrc((void*)fs.pool, 0, 0x7FFFFFFF, 0x00020000);
  const unsigned live = (unsigned)D.live;
  if (threadIdx.x < 32) s_flag[threadIdx.x] = 0;
  __syncthreads();
  // (one tile per wave -- ba_front_plan.h deals a helper at most FP_WAVES tiles: three register tiles per wave, as the front's
  // own waves hold them, cost this path 28 bytes of scratch per lane at front_up's 168-register limit)
  const int hv = __builtin_amdgcn_readfirstlane(fs.ints[D.help_off + (hlp - 1) * (FR_WAVES * FR_SLOTS) + wave * FR_SLOTS]);
  const int hr = hv < 0 ? -1 : (hv & 255), hc = hv < 0 ? -1 : ((hv >> 8) & 255);
  // the children's parts of this wave's tile (they may be there long before the front's panels)
  v4d t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = v4d{0.0, 0.0, 0.0, 0.0};
  if (hr >= 0) fr_recv_children<false>(fs, D, t, hr, hc, epoch, pool_rs, s_flag + FRC_MARK1, lane);
  // the front's panels: tile (r, j) of L, r a border row, as its solve finishes it
  auto panel_tile = [&](int gen, int r) -> double* { return sPanel + (size_t)(gen * (FR_TMAX - 1) + (r - 1)) * FR_TILE; };
#pragma unroll 1
  for (int idx = wave; idx < no * ns; idx += FR_WAVES) {
    const int j = idx / ns, i = idx - j * ns, r = no + i;
    fr_poll_flag(fs.tflag + D.pflag_off + j * ns + i, epoch, s_flag + FRC_MARK1);
    const int row = lane >> 1, half = lane & 1;
    const int src = (D.offL + (CB * r + row) * ldk + CB * j + 16 * half) * 8;
    v4u v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(pool_rs, src + 16 * k, 0, 16 /* sc1 */);
    double* const dst = panel_tile(j, r) + row * CBP + 16 * half;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *(v2d*)(dst + 2 * k) = v2d{__hiloint2double((int)v[k].y, (int)v[k].x), __hiloint2double((int)v[k].w, (int)v[k].z)};
  }
  __syncthreads();
  if (hr >= 0) {
    const unsigned subs = fr_live_subs(live, hr, hc);
#pragma unroll 1
    for (int j = 0; j < no; ++j) fr_fold(t, panel_tile(j, hr), panel_tile(j, hc), j == no - 1 ? D.nb_last : 8, subs, lane);
    const int i = hr - no, j = hc - no;
    fr_send(t, pool_rs, (D.off_pbuf + (i * (i + 1) / 2 + j) * (CB * CB)) * 8, subs, lane);
    fr_raise_flag(fs.tflag + D.tflag_off + i * (i + 1) / 2 + j, epoch, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  fr_report_timeout(s_flag, info);
}

__device__ __forceinline__ void fr_report_timeout(const int* s_flag, int* __restrict__ info) {
  if ((*(volatile const lds_int*)(s_flag + FRC_MARK0) != 0 || *(volatile const lds_int*)(s_flag + FRC_MARK1) != 0) && (threadIdx.x & 63) == 0)
    atomicExch(info, -1);
}

__global__ __launch_bounds__(FR_WAVES * 64) void front_up(FrontSet fs, const double* __restrict__ S, const double* __restrict__ g, int ldS,
                                                          BaDev d, int fin, double radius, double lm_lo, double lm_hi, int world,
                                                          unsigned epoch, int stride, int lvl_lo, int lvl_hi, double* __restrict__ zero_ptr,
                                                          long long zero_n, int n_zero, int xoff /* the fronts' place in every `stride` workgroups */) {
  extern __shared__ __attribute__((aligned(16))) double sAll[];
  const int nF = fs.n_roles;  // (fronts and their helper workgroups)
  if (lm_stopped(d)) return;  // (an iteration enqueued behind a stop of the device's LM loop: every role of the launch returns)
  {
    // ---- which role: the fronts sit at multiples of `stride` (stride 8: one XCD's L2 under round-robin placement,
    // speed only), the workgroups between and behind them zero the other reduced-system buffer slice by slice (the next
    // linearisation starts on it without a memset of its own), one more does ba_finalize's bookkeeping
    // (xoff: problems that run side by side -- one per stream -- put their fronts on different XCDs)
    const int b = (int)blockIdx.x;
    const bool is_front = b < stride * nF && b % stride == xoff;
    if (!is_front) {
      const int before = b > xoff ? (b - xoff - 1) / stride + 1 : 0;  // fronts among the workgroups before this one
      const int zi = b < stride * nF ? b - before : b - nF;
      if (zi < n_zero) {
        const long long lo = (long long)zi * ND_ZERO_SLICE;
        double2* p2 = (double2*)(zero_ptr + lo);
        const long long n2 = (zero_n - lo < ND_ZERO_SLICE ? zero_n - lo : ND_ZERO_SLICE) / 2;
        for (long long i = threadIdx.x; i < n2; i += FR_WAVES * 64) p2[i] = make_double2(0.0, 0.0);
      } else if (zi == n_zero && fin) {
        __shared__ double shm[FR_WAVES];
        const double* gF = red_gF(d);
        const double* dc = red_dc(d);
        double* scv = red_sc(d);
        double gm = 0;
        for (int i = threadIdx.x; i < d.dim; i += FR_WAVES * 64) {
          d.diag[i] = fmin(fmax(dc[i], lm_lo), lm_hi);
          const double sc = i < 6 * d.nc ? d.scale_c[i] : *d.scale_f;
          gm = fmax(gm, fabs(gF[i] / sc));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_down(gm, o));
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = gm;
        __syncthreads();
        if (threadIdx.x == 0) {
          for (int w = 1; w < FR_WAVES; ++w) gm = fmax(gm, shm[w]);
          for (int r = 0; r < world; ++r) gm = fmax(gm, scv[SC + r]);
          scv[3] = gm;
        }
      }
      return;
    }
  }
  const int role = fs.up_order[blockIdx.x / stride];
  const int f = role & 0xFFFF;
  const FrDesc D = fr_desc(fs.ints, f);
  {
    const int lvl = fs.ints[(size_t)fplan::FD_INTS * f + fplan::FD_LEVEL];
    if (lvl < lvl_lo || lvl > lvl_hi) return;  // (level-by-level launches: the fallback behind a timed-out hand-off, and a diagnostic mode)
  }
  if (role >> 16) {  // a helper workgroup of front f
    fr_helper(fs, D, role >> 16, epoch, sAll, d.info);
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int no = D.no, T = D.T, nrow = CB * T, ldk = CB * no;
  double* const sPanel = sAll + FR_OFF_PANEL;
  double* const sD = sAll + FR_OFF_SD;
  double* const sDg = sAll + FR_OFF_SDG;
  double* const sdi = sAll + FR_OFF_SDI;
  double* const sy = sAll + FR_OFF_Y;
  int* const sInv = (int*)(sAll + FR_OFF_INV);
  int* const s_flag = (int*)(sAll + FR_OFF_FLAG);
  const __amdgpu_buffer_rsrc_t pool_rs = __builtin_amdgcn_make_buffer_rsrc((void*)fs.pool, 0, 0x7FFFFFFF, 0x00020000);
  const unsigned live = (unsigned)D.live;
  double* const yg = fs.pool + D.offy;
  if (d.lm) radius = d.lm->radius;
  const double inv_radius = 1.0 / radius;
  const double* const dcv = red_dc(d);
  if (threadIdx.x < 32) s_flag[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < nrow; i += FR_WAVES * 64) {
    const int gi = fs.ints[D.inv_off + i];
    sInv[i] = gi;
  }
  if (wave == 0) {
    FR_STAMP(0);
    FR_STAMP_REAL(30);
  }
  __syncthreads();

  // ---- the tiles this wave holds, assembled from S while the children are still at work
  int sr[FR_SLOTS], sc[FR_SLOTS], sturn[FR_SLOTS];
#pragma unroll
  for (int s = 0; s < FR_SLOTS; ++s) {
    const int v = __builtin_amdgcn_readfirstlane(fs.ints[D.sched_off + wave * FR_SLOTS + s]);
    sr[s] = v < 0 ? -1 : (v & 255);
    sc[s] = v < 0 ? -1 : ((v >> 8) & 255);
    sturn[s] = v < 0 ? 0 : (v >> 16);
  }
  v4d t[FR_SLOTS][4];
  const bool chain = wave == 0, rhs = wave == 4;
  if (chain) {
    fr_assemble(t[0], 0, 0, no, sInv, S, ldS, dcv, fin, inv_radius, lm_lo, lm_hi, lane);
  } else if (rhs) {
    for (int a = lane; a < nrow; a += 64) {
      const int ga = sInv[a];
      sy[a] = (a < ldk && ga >= 0) ? g[ga] : 0.0;
    }
  } else {
#pragma unroll
    for (int s = 0; s < FR_SLOTS; ++s)
      if (sr[s] >= 0) fr_assemble(t[s], sr[s], sc[s], no, sInv, S, ldS, dcv, fin, inv_radius, lm_lo, lm_hi, lane);
  }
  if (wave == 0) FR_STAMP(1);
  if (wave == 1) FR_STAMP(16);
  // ---- the children's contribution blocks, tile by tile as the children finish them: every tile of a child's block has
  // a flag of its own, and a wave fetches a tile only when it is about to use it -- the chain wave starts on the first
  // diagonal tile, and the solve next to it on its tile, while the children are still folding and sending the rest
  auto recv_tile = [&](v4d (&tt)[4], int R, int C) { fr_recv_children<false>(fs, D, tt, R, C, epoch, pool_rs, s_flag + FRC_MARK1, lane); };
  if (chain) {
    fr_recv_children<false>(fs, D, t[0], 0, 0, epoch, pool_rs, s_flag + FRC_MARK1, lane);
    FR_STAMP(2);
  } else if (rhs) {
    for (int k = 0; k < D.nchild; ++k) {
      const int* const chd = fs.ints + (size_t)fplan::FD_INTS * fs.ints[D.child_off + k];
      const int* const ptinv = fs.ints + D.ptinv_off + k * (FR_TMAX + 1);
      const int cno = chd[fplan::FD_NO], cns = chd[fplan::FD_NS];
      const double* const yc = fs.pool + chd[fplan::FD_OFF_Y] + CB * cno;
      fr_poll_flag(fs.tflag + chd[fplan::FD_TFLAG_OFF] + cns * (cns + 1) / 2, epoch, s_flag + FRC_MARK1);
      for (int a = lane; a < nrow; a += 64) {
        const int i = ptinv[a >> 5];
        const double v = fr_ld_sc1(yc + CB * max(i, 0) + (a & 31));
        if (i >= 0) sy[a] += v;
      }
    }
  }
  bool got[FR_SLOTS] = {false, false, false};  // (which slots have the children's parts already)
  const bool defer = no <= 2;  // (ba_front_plan.h: both panel generations are intact at the end; the border tiles wait for it)
  auto need = [&](int s) {  // (s is a compile-time constant at every call: the loops over the slots are unrolled)
    if (!got[s]) {
      recv_tile(t[s], sr[s], sc[s]);
      got[s] = true;
    }
  };

  if (wave == 0) FR_STAMP(3);
  // ---- the factorisation: no barrier from here to the end, LDS counters only
  auto panel_tile = [&](int gen, int r) -> double* { return sPanel + (size_t)(gen * (FR_TMAX - 1) + (r - 1)) * FR_TILE; };
  auto prog_of = [&](int gen, int r) -> int* { return s_flag + FRC_PROG + gen * FR_TMAX + (r - 1); };
  auto progL_of = [&](int gen) -> int* { return s_flag + (gen ? FRC_PROGL1 : FRC_PROGL0); };
  if (chain) {
    __builtin_amdgcn_s_setprio(3);
    v4d dacc[3] = {t[0][0], t[0][1], t[0][3]};
    bool bad = false;
#pragma unroll 1
    for (int j = 0; j < no; ++j) {
      const int gen = j & 1, base = 8 * (j >> 1), nbj = opaque_s(j == no - 1 ? D.nb_last : 8);
      if (j >= 2) lds_wait_ge(s_flag + FRC_CONS + gen, FR_WAVES * (j >> 1));  // (everybody is done with the tiles of step j - 2)
      FR_STAMP(4 + 3 * j);
      fr_potrf(dacc, sD + gen * FR_TILE, sdi + gen * CB, progL_of(gen), base, nbj, bad, lane, pool_rs, (D.offL + (CB * j) * ldk + CB * j) * 8, ldk);
      FR_STAMP(5 + 3 * j);
      if (j + 1 < no) {
        // the next diagonal tile: its owner left it in the staging tile with every panel but this one folded in
        lds_wait_ge(s_flag + FRC_DREADY, j + 1);
        dacc[0] = chol2_get(sDg, 0, lane);
        dacc[1] = chol2_get(sDg, 2, lane);
        dacc[2] = chol2_get(sDg, 3, lane);
        lds_flag_set(s_flag + FRC_DTAKEN, j + 1);
        const double* const y0 = panel_tile(gen, j + 1) + (lane & 15) * CBP + (lane >> 4);
        const int* const pr = prog_of(gen, j + 1);
        int seen = 0;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          if (kb < nbj) {
            if (seen < base + kb + 1) seen = __builtin_amdgcn_readfirstlane(lds_wait_ge(pr, base + kb + 1));
            const double Y0 = y0[4 * kb], Y1 = y0[16 * CBP + 4 * kb];
            FR_MFMA_SUB(dacc[0], Y0, Y0);
            FR_MFMA_SUB(dacc[1], Y0, Y1);
            FR_MFMA_SUB(dacc[2], Y1, Y1);
          }
        }
      }
      FR_STAMP(6 + 3 * j);
      lds_flag_add(s_flag + FRC_CONS + gen, lane);
    }
    if (bad && lane == 0) atomicExch(d.info, 1 + sInv[0]);  // the host discards the step
  } else if (rhs) {
    // ---- the right-hand side: y_j = L_jj^-1 y_j four columns at a time behind the factorisation (lane = column),
    // then y_r -= X_rj y_j for the rows below
#pragma unroll 1
    for (int j = 0; j < no; ++j) {
      const int gen = j & 1, base = 8 * (j >> 1), nbj = opaque_s(j == no - 1 ? D.nb_last : 8);
      if (j >= 2) lds_wait_ge(s_flag + FRC_CONS + gen, FR_WAVES * (j >> 1));  // (see the tile waves' loop)
      const double* const sL = sD + gen * FR_TILE;
      const double* const sdj = sdi + gen * CB;
      double* const yj = sy + CB * j;
      const int col = lane & 31;
      double yv = yj[col];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b < nbj) {
          const int c = 4 * b;
          lds_wait_ge(progL_of(gen), base + b + 1);
          const double y0 = readlane_f64(yv, c), y1 = readlane_f64(yv, c + 1), y2 = readlane_f64(yv, c + 2), y3 = readlane_f64(yv, c + 3);
          const double l10 = sL[(c + 1) * CBP + c];
          const v2d l2 = *(const v2d*)(sL + (c + 2) * CBP + c);
          const v2d l3 = *(const v2d*)(sL + (c + 3) * CBP + c);
          const double l32 = sL[(c + 3) * CBP + c + 2];
          const v2d r01 = *(const v2d*)(sdj + c), r23 = *(const v2d*)(sdj + c + 2);
          const double x0 = y0 * r01.x;
          const double x1 = (y1 - x0 * l10) * r01.y;
          const double x2 = (y2 - x0 * l2.x - x1 * l2.y) * r23.x;
          const double x3 = (y3 - x0 * l3.x - x1 * l3.y - x2 * l32) * r23.y;
          const v2d la = *(const v2d*)(sL + col * CBP + c), lb = *(const v2d*)(sL + col * CBP + c + 2);
          if (col > c + 3) yv -= la.x * x0 + la.y * x1 + lb.x * x2 + lb.y * x3;
          yv = col == c ? x0 : col == c + 1 ? x1 : col == c + 2 ? x2 : col == c + 3 ? x3 : yv;
        }
      }
      if (lane < CB) yj[lane] = yv;
      asm volatile("" ::: "memory");
      const int h = lane >> 5, kn = 4 * nbj;
      for (int r = j + 1; r < T; ++r) {
        lds_wait_ge(prog_of(gen, r), base + nbj);
        const double* const xr = panel_tile(gen, r) + col * CBP;
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          const int kk = 16 * h + k;
          if (kk < kn) {  // (four-column blocks: kk and kk + 1 are on the same side of kn)
            const v2d x = *(const v2d*)(xr + kk), yy = *(const v2d*)(yj + kk);
            a += x.x * yy.x + x.y * yy.y;
          }
        }
        a += __shfl_xor(a, 32);
        if (lane < CB) sy[CB * r + lane] -= a;
      }
      asm volatile("" ::: "memory");
      lds_flag_add(s_flag + FRC_CONS + gen, lane);
    }
    // the solved own part (the down-sweep reads it) and the border part (the parent adds it), write-through
    FR_STAMP(20);
    for (int a = lane; a < nrow; a += 64) fr_st_sc1(yg + a, sy[a]);
#ifdef SFM_FRONT_BREAK_HANDOFF
    // diagnostic build only (tests/test_gpu_geometry.py::test_a_timed_out_hand_off_...): in the one-launch form the first front of
    // the up-sweep never raises its right-hand side's flag -- what a child that got no compute unit in time looks like to its
    // parent; launched level by level the hand-off is whole
    if (!(lvl_lo == 0 && lvl_hi == (1 << 30) && (int)blockIdx.x / stride == 0))
#endif
    fr_raise_flag(fs.tflag + D.tflag_off + D.ns * (D.ns + 1) / 2, epoch, lane);
  } else {
    // ---- tile waves (wave 8 holds border tiles only, and only where they are folded at the end)
    auto publish_diag = [&](v4d (&tt)[4], int c) {
      lds_wait_ge(s_flag + FRC_DTAKEN, c - 1);  // (the chain wave has taken the tile before this one out of the staging tile)
      chol2_put(sDg, 0, lane, tt[0]);
      chol2_put(sDg, 2, lane, tt[1]);
      chol2_put(sDg, 3, lane, tt[3]);
      lds_flag_set(s_flag + FRC_DREADY, c);
    };
#pragma unroll
    for (int s = 0; s < FR_SLOTS; ++s)
      if (sr[s] == 1 && sc[s] == 1 && no > 1) {
        need(s);
        publish_diag(t[s], 1);
      }
#pragma unroll 1
    for (int j = 0; j < no; ++j) {
      const int gen = j & 1, base = 8 * (j >> 1), nbj = opaque_s(j == no - 1 ? D.nb_last : 8);
      // Step j reuses the panel generation of step j - 2: nobody enters it before EVERY wave has left step j - 2.  Every
      // wave waits here, also one with nothing to do in this step -- it would otherwise add its count for step j to the
      // generation's counter at once, the counter would reach a full round with a slow wave (the right-hand side's, when the
      // children's y came late) still reading step j - 2's panel, and the solves of step j would overwrite it under its hands.
      if (j >= 2) lds_wait_ge(s_flag + FRC_CONS + gen, FR_WAVES * (j >> 1));
#pragma unroll
      for (int s = 0; s < FR_SLOTS; ++s)
        if (sc[s] == j && sr[s] > j) {
          need(s);
          fr_trsm(t[s], panel_tile(gen, sr[s]), sD + gen * FR_TILE, sdi + gen * CB, progL_of(gen), prog_of(gen, sr[s]), base, nbj,
                  pool_rs, (D.offL + (CB * sr[s] + (lane & 31)) * ldk + CB * j) * 8, lane, D.nhelp > 0 && sr[s] >= no);
        }
#pragma unroll
      for (int s = 0; s < FR_SLOTS; ++s)
        if (sc[s] > j && !(sr[s] == sc[s] && sc[s] < no && j >= sc[s] - 1) && !(defer && sc[s] >= no)) {  // (a diagonal own tile's last update is the chain wave's)
          need(s);
          fr_update(t[s], panel_tile(gen, sr[s]), prog_of(gen, sr[s]), panel_tile(gen, sc[s]), prog_of(gen, sc[s]), base, nbj,
                    fr_live_subs(live, sr[s], sc[s]), lane);
          if (sr[s] == sc[s] && sc[s] == j + 2 && sc[s] < no) publish_diag(t[s], sc[s]);
        }
      lds_flag_add(s_flag + FRC_CONS + gen, lane);
    }
    // helper workgroups fold some of the border tiles: the border rows of L this wave has solved are in memory (write-through),
    // a flag per tile tells them.  A flag may only follow the wave's stores, and waiting for stores that have just been issued
    // costs ~2 us -- on every level, when it sat here between the steps and the wave's first border tile (measured: 11 us per
    // solve, more than the helpers gave back).  So the flags go out where the wait is free: while the wave waits for its turn
    // anyway, or behind the wait its first tile's own flag needs, or -- a wave with no border tile -- here.
    bool panels_told = D.nhelp <= 0;
    auto tell_panels = [&]() {
      if (!panels_told) {
#pragma unroll
        for (int s = 0; s < FR_SLOTS; ++s)
          if (sc[s] >= 0 && sc[s] < no && sr[s] >= no) fr_raise_flag(fs.tflag + D.pflag_off + sc[s] * D.ns + (sr[s] - no), epoch, lane);
        panels_told = true;
      }
    };
    // the border tiles.  A front that folded them as it went sends them; a front of one or two own tiles folds both panels into
    // them now that nothing solves beside them, a tile at a time per SIMD in the plan's order (ba_front_plan.h: the order in
    // which the ancestors need them), sends each and raises its flag
    if (wave == 1) FR_STAMP(17);
    if (wave == 11) FR_STAMP(18);
    FR_STAMP(32 + 8 * wave + 7);
#pragma unroll
    for (int s = 0; s < FR_SLOTS; ++s)
      if (sc[s] >= no) {
        const unsigned subs = fr_live_subs(live, sr[s], sc[s]);
        need(s);
        if (defer) {
          // none before the last diagonal tile is factored, and on the SIMDs that hold the last step's solves (2 and 3) none
          // before those are through (an MFMA stream beside a solve lets it issue one vector instruction per MFMA)
          int* const turn = s_flag + FRC_TURN + (wave & 3);
          lds_wait_ge(progL_of((no - 1) & 1), D.nb_last);
          if ((wave & 3) >= 2)
            for (int r = no; r < T; ++r) lds_wait_ge(prog_of((no - 1) & 1, r), D.nb_last);
          if (sturn[s] > 0) tell_panels();
          lds_wait_ge(turn, sturn[s]);
          FR_STAMP(32 + 8 * wave + 2 * s);
          for (int j = 0; j < no; ++j) {
            const int nbj = j == no - 1 ? D.nb_last : 8;
            lds_wait_ge(prog_of(j, sr[s]), nbj);  // (both rows of the panel complete: the fold reads them four blocks at a time)
            lds_wait_ge(prog_of(j, sc[s]), nbj);
            fr_fold(t[s], panel_tile(j, sr[s]), panel_tile(j, sc[s]), nbj, subs, lane);
          }
          lds_flag_set(turn, sturn[s] + 1);
          FR_STAMP(32 + 8 * wave + 2 * s + 1);
        }
        const int i = sr[s] - no, j = sc[s] - no;
        fr_send(t[s], pool_rs, (D.off_pbuf + (i * (i + 1) / 2 + j) * (CB * CB)) * 8, subs, lane);
        fr_raise_flag(fs.tflag + D.tflag_off + i * (i + 1) / 2 + j, epoch, lane);
        tell_panels();
        if (wave == 8 && sturn[s] == 0) FR_STAMP(29);
      }
    tell_panels();
  }
  // ---- every store of this workgroup has left before the flag does
  if (wave == 1) FR_STAMP(19);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wave == 1) FR_STAMP(21);
  __syncthreads();
  if (wave == 0) {
    FR_STAMP(22);
    FR_STAMP_REAL(31);
  }
  fr_report_timeout(s_flag, d.info);
}

// ---------------------------------------------------------------- down-sweep + candidate cameras
constexpr int FD_THREADS = 256;  // one wave per SIMD: 512 registers each (the rows of L wait in them for the parent's flag)
// candidate cameras / focal of a front: x + (-z) * scale, their tables, the camera part of the norms (every thread of the
// down-sweep workgroup calls it; szv: the front's z in front order)
__device__ __forceinline__ void fd_candidates(const FrontSet& fs, const FrDesc& D, const BaDev& d_in, const unsigned char* __restrict__ cam_used,
                                              int rank, int cand, const double* szv, double (*sred)[4]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (!cand) return;
  BaDev d = d_in;  // (which parameter set is x: the record's word when the loop runs on the device)
  {
    double radius_unused;
    lm_view(d, radius_unused);
  }
  double sn2 = 0, cn2 = 0;
  if ((int)threadIdx.x < D.ncam) {
    const int c = fs.ints[D.cam_off + threadIdx.x], cpos = fs.ints[D.cam_off + D.ncam + threadIdx.x];
    double cam[6];
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      const double dl = -szv[cpos + jj] * d.scale_c[6 * c + jj];
      cam[jj] = d.cams[6 * c + jj] + dl;
      d.cams_c[6 * c + jj] = cam[jj];
      if (cam_used[c]) {
        sn2 += dl * dl;
        cn2 += cam[jj] * cam[jj];
      }
    }
    cam_table(cam, d.camd_c + (size_t)CAMD * c, true);
  }
  if (D.has_focal && (int)threadIdx.x == D.ncam) {
    const double dl = -szv[D.focal_pos] * (*d.scale_f);
    const double fc = *d.focal + dl;
    *d.focal_c = fc;
    sn2 += dl * dl;
    cn2 += fc * fc;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sn2 += __shfl_down(sn2, off);
    cn2 += __shfl_down(cn2, off);
  }
  if (lane == 0) sred[0][wave] = sn2, sred[1][wave] = cn2;
  __syncthreads();
  if (threadIdx.x == 0) {
    // the camera part of the norms, a pair of sums per front: step_finish adds them in front order (the same bits every run;
    // atomics on two words landed in the order the fronts happened to finish)
    for (int w = 1; w < FD_THREADS / 64; ++w) sn2 += sred[0][w], cn2 += sred[1][w];
    double* cp = d.step_part + 4 * (size_t)d.step_total + 2 * (size_t)blockIdx.x;
    cp[0] = sn2, cp[1] = cn2;
  }
}

__global__ __launch_bounds__(FD_THREADS) void front_down(FrontSet fs, BaDev d, const unsigned char* __restrict__ cam_used, int rank,
                                                         unsigned epoch, int cand) {
  __shared__ int sInv[CB * FR_TMAX];
  __shared__ double sz[CB * FR_TMAX];      // z of the border
  __shared__ double sw[CB * fplan::FP_NO_MAX];  // y_v - L_bv^T z_b
  __shared__ double szv[CB * fplan::FP_NO_MAX];  // z of the own columns
  __shared__ double spart[4][CB * fplan::FP_NO_MAX];
  __shared__ double sred[2][FD_THREADS / 64];
  __shared__ double sLd[fplan::FP_NO_MAX][CB * CB];  // the diagonal tiles of L (wave j: tile j), [row][col]
  __shared__ double sRd[fplan::FP_NO_MAX][CB];        // 1 / their diagonals
  __shared__ double sT[fplan::FP_NO_MAX][CB];         // t_j, for the product with the inverse
  __shared__ int s_to;
  if (lm_stopped(d)) return;
  const int f = fs.down_order[blockIdx.x];
  const FrDesc D = fr_desc(fs.ints, f);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int no = D.no, o = CB * no, nsr = CB * D.ns, ldk = o;
  const int valid = CB * (no - 1) + 4 * D.nb_last;  // columns of L that were written
  const double* const Lt = fs.pool + D.offL;
  const double* const yg = fs.pool + D.offy;
  const double* const zq_cur = fs.zq + (size_t)(epoch & 1) * fs.zq_ld;
  double* const zq_mine = fs.zq + (size_t)(epoch & 1) * fs.zq_ld;
  double* const zq_next = fs.zq + (size_t)((epoch + 1) & 1) * fs.zq_ld;
  for (int i = threadIdx.x; i < CB * D.T; i += FD_THREADS) sInv[i] = fs.ints[D.inv_off + i];
  if (threadIdx.x == 0) s_to = 0;
  FD_STAMP(0);
  // ---- the rows of L this wave needs, before the parent's flag: a quarter of the border rows (lane = columns lane and
  // lane + 64), and, waves 0..no-1, their column tile of the own block (lane = column k + 32 h: half h of the rows of
  // every tile)
  constexpr int QR = CB * (FR_TMAX - 1) / 4;  // border rows per wave, at most
  double RB[2 * QR], RC[48];
  double Y[CB];  // waves 0..no-1, lane i < 32: row i of L_jj^-T (= column i of L_jj^-1); the root: its column of L_jj
  double rdi = 1.0;
  const int qrows = nsr / 4;  // (nsr is a multiple of 32)
  const int kc = lane & 31, hc = lane >> 5;
  {
    const int a0 = wave * qrows;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int k = lane + 64 * hh;
#pragma unroll
      for (int a = 0; a < QR; ++a) RB[hh * QR + a] = (a < qrows && k < valid) ? Lt[(size_t)(o + a0 + a) * ldk + k] : 0.0;
    }
  }
  if (wave < no) {
    // RC[16 rr + a], rr < 3: L(32 (j + 1 + rr) + 16 h + a, col); Ld[i]: L(32 j + i, col) below the diagonal
    const int j = wave, col = CB * j + kc;
    const bool live = col < valid;
#pragma unroll
    for (int rr = 0; rr < fplan::FP_NO_MAX - 1; ++rr)
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int r = j + 1 + rr;
        RC[16 * rr + a] = (live && r < no) ? Lt[(size_t)(CB * r + 16 * hc + a) * ldk + col] : 0.0;
      }
    // ---- the inverse of the diagonal tile, while the parent's z is still on its way: lane i solves L y = e_i column by
    // column (every lane its own right-hand side, L broadcast from LDS: no traffic between lanes), which leaves row i of
    // L^-T in its registers -- the backward substitution of the tile is then ONE product z_i = sum_k y_k t_k instead of a
    // chain of 32 dependent steps.  Rows / columns past the last real column count as the identity.
    // (the ROOT waits for nobody: the inversion would be time on the critical path -- it keeps its column below the
    // diagonal instead, Y[i] = L(32 j + i, col), and substitutes column by column)
    if (D.parent < 0) {
#pragma unroll
      for (int i = 0; i < CB; ++i) Y[i] = (live && i > kc && CB * j + i < valid) ? Lt[(size_t)(CB * j + i) * ldk + col] : 0.0;
      rdi = live ? rcp_f64(Lt[(size_t)col * ldk + col]) : 1.0;
    } else {
    // (staged TRANSPOSED, [column][row]: the rows below the diagonal of a column are contiguous, two per LDS read)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = lane + 64 * u, i = e >> 5, k = e & 31;  // coalesced along a row of L
      const bool in = CB * j + i < valid && CB * j + k < valid;
      sLd[j][k * CB + i] = in ? (k <= i ? Lt[(size_t)(CB * j + i) * ldk + CB * j + k] : 0.0) : (i == k ? 1.0 : 0.0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane < CB) sRd[j][lane] = rcp_f64(sLd[j][lane * CB + lane]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < CB; ++k) Y[k] = k == kc ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < CB; ++m) {
      Y[m] *= sRd[j][m];
      const double* const colm = &sLd[j][m * CB];  // L(., m)
      if ((m + 1) & 1) Y[m + 1 < CB ? m + 1 : m] -= (m + 1 < CB ? colm[m + 1] : 0.0) * Y[m];  // (an odd first row, then pairs)
#pragma unroll
      for (int k = (m + 2) & ~1; k < CB; k += 2) {
        const v2d l = *(const v2d*)(colm + k);
        Y[k] -= l.x * Y[m];
        Y[k + 1] -= l.y * Y[m];
      }
    }
    }
  }
  const double y_own = (int)threadIdx.x < o ? yg[threadIdx.x] : 0.0;  // (o <= 128 < FD_THREADS)
  __syncthreads();
  FD_STAMP(1);
  // the border's z out of the mailbox: every thread polls its own entry until the front that owns it has written it
  // (one round trip once it is there: the data is its own flag); bounded
  for (int a = threadIdx.x; a < nsr; a += FD_THREADS) {
    const int gi = sInv[o + a];
    double v = 0.0;
    if (gi >= 0) {
      int budget = 1 << 17;
      for (;;) {
        v = fr_ld_sc1(zq_cur + gi);
        if ((unsigned long long)__double_as_longlong(v) != FR_Z_PENDING) break;
        if (--budget == 0) {
          s_to = 1;
          v = 0.0;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    sz[a] = v;
  }
  FD_STAMP(2);
  __syncthreads();
  {
    const int a0 = wave * qrows;
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int a = 0; a < QR; ++a) {
      const double zz = a < qrows ? sz[a0 + a] : 0.0;
      p0 += RB[a] * zz;
      p1 += RB[QR + a] * zz;
    }
    spart[wave][lane] = p0;
    spart[wave][lane + 64] = p1;
  }
  __syncthreads();
  if ((int)threadIdx.x < o) sw[threadIdx.x] = y_own - (spart[0][threadIdx.x] + spart[1][threadIdx.x] + spart[2][threadIdx.x] + spart[3][threadIdx.x]);
  __syncthreads();
  FD_STAMP(3);
  for (int j = no - 1; j >= 0; --j) {
    if (wave == j) {
      double tv = hc == 0 ? sw[CB * j + kc] : 0.0;
#pragma unroll
      for (int rr = 0; rr < fplan::FP_NO_MAX - 1; ++rr)
        if (j + 1 + rr < no) {
#pragma unroll
          for (int a = 0; a < 16; ++a) tv -= RC[16 * rr + a] * szv[CB * (j + 1 + rr) + 16 * hc + a];
        }
      tv += __shfl_xor(tv, 32);  // (both halves now hold the whole sum)
      if (lane < CB) sT[j][lane] = tv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double zv = 0.0;
      if (D.parent < 0) {
        // L_jj^T z = t, last column first: z_i = t_i / L_ii, t_k -= L_ik z_i for k < i
#pragma unroll
        for (int i = CB - 1; i >= 0; --i) {
          const double zi = readlane_f64(tv * rdi, i);
          if (kc == i) zv = zi;
          tv -= Y[i] * zi;
        }
      } else {
        // z_j = L_jj^-T t_j: lane i holds row i of the inverse
#pragma unroll
        for (int k = 0; k < CB; ++k) zv += Y[k] * sT[j][k];
      }
      if (lane < CB) szv[CB * j + lane] = zv;
    }
    __syncthreads();
  }
  FD_STAMP(4);
  for (int k = threadIdx.x; k < o; k += FD_THREADS) {
    const int gi = sInv[k];
    if (gi >= 0) {
      fr_st_sc1(zq_mine + gi, szv[k]);  // the children poll this one
      d.z[gi] = szv[k];                 // ba_backsub (a later launch) reads this one
      ((unsigned long long*)zq_next)[gi] = FR_Z_PENDING;  // (the next solve's generation: nobody reads it before the next launch)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store((gbl_u32*)(fs.flag_down + f), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_to) atomicExch(d.info, -1);
  }
  FD_STAMP(5);
  fd_candidates(fs, D, d, cam_used, rank, cand, szv, sred);
  FD_STAMP(6);
}
#undef FR_MFMA

#ifdef SFM_FRONT_STAMPS
// diagnostic build only (scripts/front_ubench.py): the update loop of fr_update on one wave of SIMD 0, alone or beside other
// work on the CU.  mode bits: 1 = a wave of f64 vector FMAs on the same SIMD (the right-hand side's wave), 2 = a wave that
// polls an LDS counter with s_sleep 1 on the same SIMD, 4 = waves folding on the other three SIMDs, 8 = a second folding wave
// on the same SIMD, 16 = two waves polling with s_sleep 8, 32 = a third folding wave on the SIMD.  out[0] = shader clocks of 64 x 8 blocks, out[1] = the same, real time (100 MHz).
__global__ __launch_bounds__(FR_WAVES * 64) void front_ubench(unsigned long long* __restrict__ out, double* __restrict__ sink, int mode) {
  extern __shared__ __attribute__((aligned(16))) double sAll[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* const s_flag = (int*)(sAll + FR_OFF_FLAG);
  for (int i = threadIdx.x; i < FR_OFF_FLAG; i += FR_WAVES * 64) sAll[i] = 1e-3 * ((i * 37) % 101 - 50);
  if (threadIdx.x < 32) s_flag[threadIdx.x] = threadIdx.x < 14 ? 8 : 0;  // (every panel row complete)
  __syncthreads();
  v4d acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
  const bool folder = wave == 8 || ((mode & 4) && wave >= 1 && wave <= 3) || ((mode & 8) && wave == 0) || ((mode & 32) && wave == 4);
  if (folder) {
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    const unsigned subs = 15u;
#pragma unroll 1
    for (int rep = 0; rep < 64; ++rep)
      fr_update(acc, sAll + FR_OFF_PANEL + (size_t)(1 + (rep & 1)) * FR_TILE, s_flag + 1, sAll + FR_OFF_PANEL + (size_t)(3 + (rep & 1)) * FR_TILE, s_flag + 3, 0,
                8, opaque_s((int)subs), lane);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (lane == 0) {
      out[2 * wave] = t1 - t0;
      out[2 * wave + 1] = r1 - r0;
    }
    if (wave == 8) lds_flag_set(s_flag + 20, 1);
    if (wave == 8 && (mode & 64)) {
      // bare MFMAs from registers: 4 per iteration / 8 per iteration on the same four accumulators
      double a0 = sAll[lane], a1 = sAll[64 + lane], b0 = sAll[128 + lane], b1 = sAll[192 + lane];
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int it = 0; it < 512; ++it) {
        FR_MFMA_SUB(acc[0], a0, b0);
        FR_MFMA_SUB(acc[1], a0, b1);
        FR_MFMA_SUB(acc[2], a1, b0);
        FR_MFMA_SUB(acc[3], a1, b1);
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[18] = t1 - t0;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int it = 0; it < 256; ++it) {
        FR_MFMA_SUB(acc[0], a0, b0);
        FR_MFMA_SUB(acc[1], a0, b1);
        FR_MFMA_SUB(acc[2], a1, b0);
        FR_MFMA_SUB(acc[3], a1, b1);
        FR_MFMA_SUB(acc[0], b0, a0);
        FR_MFMA_SUB(acc[1], b0, a1);
        FR_MFMA_SUB(acc[2], b1, a0);
        FR_MFMA_SUB(acc[3], b1, a1);
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[19] = t1 - t0;
      // one accumulator only: the dependent-issue latency
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int it = 0; it < 512; ++it) {
        FR_MFMA_SUB(acc[0], a0, b0);
        FR_MFMA_SUB(acc[0], a1, b1);
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[20] = t1 - t0;
      // LDS reads issued before four MFMAs that do not use them, waited for after: do the reads complete under the MFMAs?
      const double* lp = sAll + FR_OFF_PANEL + (lane & 15) * CBP + (lane >> 4);
      double x0 = 0, x1 = 0, x2 = 0, x3 = 0, sacc = 0;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int it = 0; it < 512; ++it) {
        const double* q = lp + 4 * (it & 7);
        x0 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)q;
        x1 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)(q + 16 * CBP);
        x2 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)(q + FR_TILE);
        x3 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)(q + FR_TILE + 16 * CBP);
        __builtin_amdgcn_sched_barrier(0);
        FR_MFMA_SUB(acc[0], a0, b0);
        FR_MFMA_SUB(acc[1], a0, b1);
        FR_MFMA_SUB(acc[2], a1, b0);
        FR_MFMA_SUB(acc[3], a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        sacc += x0 + x1 + x2 + x3;
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[21] = t1 - t0;
      acc[0][1] += sacc;
      // the same with the reads' results feeding the NEXT iteration's MFMAs (what fr_update does), operands copied after the MFMAs
      double n0 = a0, n1 = a1, m0 = b0, m1 = b1;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int it = 0; it < 512; ++it) {
        const double* q = lp + 4 * (it & 7);
        x0 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)q;
        x1 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)(q + 16 * CBP);
        x2 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)(q + FR_TILE);
        x3 = *(volatile const __attribute__((address_space(3))) double*)(const __attribute__((address_space(3))) double*)(q + FR_TILE + 16 * CBP);
        __builtin_amdgcn_sched_barrier(0);
        FR_MFMA_SUB(acc[0], n0, m0);
        FR_MFMA_SUB(acc[1], n0, m1);
        FR_MFMA_SUB(acc[2], n1, m0);
        FR_MFMA_SUB(acc[3], n1, m1);
        __builtin_amdgcn_sched_barrier(0);
        n0 = x0, n1 = x1, m0 = x2, m1 = x3;
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[22] = t1 - t0;
      // candidate update loops over a panel pair of 8 blocks, 64 repetitions each (clocks per block = total / 512)
      typedef const __attribute__((address_space(3))) double* ldsp;
      const ldsp xr = (ldsp)(sAll + FR_OFF_PANEL + FR_TILE + (lane & 15) * CBP + (lane >> 4));
      const ldsp xc = (ldsp)(sAll + FR_OFF_PANEL + 3 * FR_TILE + (lane & 15) * CBP + (lane >> 4));
      // V2: one block ahead, no branches, reads / MFMAs / copies kept apart
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int rep = 0; rep < 64; ++rep) {
        double R0 = xr[0], R1 = xr[16 * CBP], C0 = xc[0], C1 = xc[16 * CBP];
#pragma unroll 1
        for (int kb = 0; kb < 8; ++kb) {
          const int nk = kb + 1 < 8 ? kb + 1 : kb;
          const double nR0 = xr[4 * nk], nR1 = xr[16 * CBP + 4 * nk], nC0 = xc[4 * nk], nC1 = xc[16 * CBP + 4 * nk];
          __builtin_amdgcn_sched_barrier(0);
          FR_MFMA_SUB(acc[0], C0, R0);
          FR_MFMA_SUB(acc[1], C0, R1);
          FR_MFMA_SUB(acc[2], C1, R0);
          FR_MFMA_SUB(acc[3], C1, R1);
          __builtin_amdgcn_sched_barrier(0);
          R0 = nR0, R1 = nR1, C0 = nC0, C1 = nC1;
        }
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[0] = t1 - t0;
      // V4: four blocks' operands at a time, then their sixteen MFMAs
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int rep = 0; rep < 64; ++rep) {
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
          double R0[4], R1[4], C0[4], C1[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) R0[k] = xr[16 * h + 4 * k], R1[k] = xr[16 * CBP + 16 * h + 4 * k], C0[k] = xc[16 * h + 4 * k], C1[k] = xc[16 * CBP + 16 * h + 4 * k];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            FR_MFMA_SUB(acc[0], C0[k], R0[k]);
            FR_MFMA_SUB(acc[1], C0[k], R1[k]);
            FR_MFMA_SUB(acc[2], C1[k], R0[k]);
            FR_MFMA_SUB(acc[3], C1[k], R1[k]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[1] = t1 - t0;
      // V5: the whole panel pair's operands (8 blocks), then 32 MFMAs
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int rep = 0; rep < 64; ++rep) {
        double R0[8], R1[8], C0[8], C1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) R0[k] = xr[4 * k], R1[k] = xr[16 * CBP + 4 * k], C0[k] = xc[4 * k], C1[k] = xc[16 * CBP + 4 * k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          FR_MFMA_SUB(acc[0], C0[k], R0[k]);
          FR_MFMA_SUB(acc[1], C0[k], R1[k]);
          FR_MFMA_SUB(acc[2], C1[k], R0[k]);
          FR_MFMA_SUB(acc[3], C1[k], R1[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[2] = t1 - t0;
      // V6: four blocks at a time, the next four's reads issued before this four's MFMAs (two register sets)
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
      for (int rep = 0; rep < 64; ++rep) {
        double A0[4], A1[4], B0[4], B1[4], P0[4], P1[4], Q0[4], Q1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) A0[k] = xr[4 * k], A1[k] = xr[16 * CBP + 4 * k], B0[k] = xc[4 * k], B1[k] = xc[16 * CBP + 4 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) P0[k] = xr[16 + 4 * k], P1[k] = xr[16 * CBP + 16 + 4 * k], Q0[k] = xc[16 + 4 * k], Q1[k] = xc[16 * CBP + 16 + 4 * k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          FR_MFMA_SUB(acc[0], B0[k], A0[k]);
          FR_MFMA_SUB(acc[1], B0[k], A1[k]);
          FR_MFMA_SUB(acc[2], B1[k], A0[k]);
          FR_MFMA_SUB(acc[3], B1[k], A1[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          FR_MFMA_SUB(acc[0], Q0[k], P0[k]);
          FR_MFMA_SUB(acc[1], Q0[k], P1[k]);
          FR_MFMA_SUB(acc[2], Q1[k], P0[k]);
          FR_MFMA_SUB(acc[3], Q1[k], P1[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (lane == 0) out[3] = t1 - t0;
    }
  } else if ((mode & 1) && wave == 4) {
    double a = 1.0 + lane, b = 0.5;
    while (*(volatile lds_int*)(s_flag + 20) == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) a = fma(a, 0.999, b);
    }
    acc[0][0] = a;
  } else if (((mode & 2) && wave == 0 && !(mode & 8)) || ((mode & 2) && wave == 4 && !(mode & 1))) {
    lds_wait_ge(s_flag + 20, 1);
  } else if ((mode & 16) && (wave == 0 || wave == 4)) {
    while (__builtin_amdgcn_readfirstlane(*(volatile lds_int*)(s_flag + 20)) == 0) __builtin_amdgcn_s_sleep(8);
  }
  double sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[threadIdx.x] = sum;
}
#endif
