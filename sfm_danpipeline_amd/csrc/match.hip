// match.hip -- brute-force k=2 descriptor matching + ratio test on gfx950 (MI355X).
//
// Replaces cv::BFMatcher(NORM_L2,false).knnMatch(q,t,k=2) + the ratio filter of
// StructFromMotion::getMatching (reference src/Sfm.cpp:590-608), for one pair and for the
// batched all-pairs loop of findBestPair (reference src/Sfm.cpp:511-515).
//
// Design (DESIGN.md section "K1/K2"):
//  * prepare pass: every image becomes an i8 "tile image" (32-row tiles already laid out in the
//    XOR-swizzled order the LDS wants, so staging is a linear copy), plus per-row norms and
//    per-row tie-break key bases.  SIFT rows (integers 0..255 held in f32) are centred to
//    x-128; binary rows are expanded to +-1 so Hamming = (nbits - dot)/2.
//  * knn kernel: 32x32x32 i8 MFMA, trains on the M side (rows -> accumulator registers),
//    queries on the N side (one query per lane).  Exact integer distances; each accumulator
//    element is folded into a 32-bit key (distance << 8 | row-in-chunk) with one v_mad_i32_i24
//    and inserted into a per-lane top-2 with v_med3_u32 + v_min_u32: 3 VALU ops per distance.
//    Keys order by (distance, lower train index) = cv::batchDistance's insertion rule.
//  * exact kernel: f32 rows that are not integer-valued, and the rare queries whose 2nd-best
//    squared distance is >= 2^22 (where sqrtf can merge neighbouring integers), are redone by
//    a VALU kernel that orders by (sqrtf(s), index) like OpenCV does.
//  * compaction kernel: ratio test (float multiply + compare) and order-preserving compaction.
#include "common.h"
#include <vector>
#include <algorithm>
#include <string.h>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

enum { KIND_F32_L2 = 0, KIND_U8_L2 = 1, KIND_U8_HAMMING = 2 };

constexpr int TILE_ROWS = 32;
constexpr int CHUNK_ROWS = 256;  // rows per tie-break chunk: 8 index bits in the key
constexpr int IB = 8;
constexpr unsigned KEY_EMPTY = 0xFFFFFFFFu;
constexpr int FIX_CAP = 1 << 20;

// Pointers read out of the ImgDev records are generic to the compiler (flat_load, which also
// ties up lgkmcnt next to the LDS reads); these casts tell it they are global memory.
#define SFM_GLOBAL __attribute__((address_space(1)))
typedef const SFM_GLOBAL v4i* g_v4i_p;
typedef const SFM_GLOBAL unsigned* g_u32_p;
typedef const SFM_GLOBAL int* g_i32_p;

struct ImgDev {
  const void* raw;  // descriptor rows as handed over (f32 or u8), HBM
  int8_t* tiles;    // [n_pad/32][32][KS*32] swizzled i8
  unsigned* base;   // [n_pad] key base of the row in its train role
  int* nq;          // [n_pad] squared norm of the centred row (query role)
  int n_rows;
  int n_pad;
};

struct WorkItem {
  int pair;
  int qtile0;
};

// physical 16-byte chunk of logical chunk c of row r inside a 32-row tile with NC chunks/row:
// ds_read_b128 then hits 16 distinct 16-byte slots per 16-lane group (MI355X_MICROARCH LDS).
template <int NC>
__host__ __device__ constexpr int chunk_pos(int r, int c) {
  constexpr int SH = (NC == 16) ? 0 : (NC == 8) ? 1 : (NC == 4) ? 2 : 3;
  return r * NC + (c ^ ((r >> SH) & (NC - 1)));
}
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }

// ---------------------------------------------------------------- prepare
template <int KS, int KIND>
__global__ void prepare_kernel(const ImgDev* __restrict__ imgs, const int* __restrict__ tile_img,
                               const int* __restrict__ tile_first, int dim, int* __restrict__ nonintegral,
                               int gen) {
  constexpr int NC = 2 * KS;
  constexpr int RB = 32 * KS;
  const int img = tile_img[blockIdx.x];
  const int tile = blockIdx.x - tile_first[img];
  const ImgDev I = imgs[img];
  const int r = threadIdx.x / NC, c = threadIdx.x % NC;
  const int row = tile * TILE_ROWS + r;
  const bool valid = row < I.n_rows;
  signed char v[16];
  int n2 = 0;
  bool ok = true;
  if (KIND == KIND_U8_HAMMING) {
    const int nbits = dim * 8;
    unsigned bits = 0;
    if (valid) {
      const unsigned char* src = (const unsigned char*)I.raw + (size_t)row * dim;
      const int b0 = 2 * c;
      if (b0 < dim) bits |= src[b0];
      if (b0 + 1 < dim) bits |= (unsigned)src[b0 + 1] << 8;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = c * 16 + e;
      v[e] = (valid && k < nbits) ? (((bits >> e) & 1u) ? 1 : -1) : 0;
    }
  } else {
    // the lane's 16 elements: four 16-byte loads (f32) / one (u8) when the chunk lies inside the
    // row and is 16-byte aligned -- the case of every real descriptor width; element loads otherwise
    float fv[16];
    unsigned char bv[16];
    const bool full = valid && c * 16 + 16 <= dim;
    if (KIND == KIND_F32_L2) {
      const float* src = (const float*)I.raw + (size_t)row * dim + c * 16;
      if (full && (((size_t)src) & 15) == 0) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const SFM_GLOBAL v4f* s4 = (const SFM_GLOBAL v4f*)src;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const v4f x = s4[g4];
          fv[4 * g4] = x[0];
          fv[4 * g4 + 1] = x[1];
          fv[4 * g4 + 2] = x[2];
          fv[4 * g4 + 3] = x[3];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) fv[e] = (valid && c * 16 + e < dim) ? src[e] : 128.f;
      }
    } else {
      const unsigned char* src = (const unsigned char*)I.raw + (size_t)row * dim + c * 16;
      if (full && (((size_t)src) & 15) == 0) {
        const v4i x = *(g_v4i_p)src;
        memcpy(bv, &x, 16);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = (valid && c * 16 + e < dim) ? src[e] : 128;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = c * 16 + e;
      int x = 128;  // centred value 0 for padding
      if (valid && k < dim) {
        if (KIND == KIND_F32_L2) {
          const float f = fv[e];
          ok = ok && (f == rintf(f)) && (f >= 0.f) && (f <= 255.f);
          x = (int)f;
          x = x < 0 ? 0 : (x > 255 ? 255 : x);
        } else {
          x = bv[e];
        }
      }
      const int cv = x - 128;
      v[e] = (signed char)cv;
      n2 += cv * cv;
    }
  }
#pragma unroll
  for (int o = 1; o < NC; o <<= 1) n2 += __shfl_xor(n2, o);
  int4 out;
  memcpy(&out, v, 16);
  int4* dst = (int4*)(I.tiles + (size_t)tile * (TILE_ROWS * RB)) + chunk_pos<NC>(r, c);
  *dst = out;
  if (c == 0) {
    const unsigned jl = (unsigned)(row & (CHUNK_ROWS - 1));
    unsigned b;
    if (KIND == KIND_U8_HAMMING) {
      const int nbits = dim * 8;
      b = valid ? (((unsigned)nbits << (IB - 1)) | jl) : (((unsigned)(nbits + 1) << IB) | jl);
    } else {
      // +RB*16384 (KOFF of the k-NN kernel): keys stay non-negative without the query norm;
      // the largest value, padding rows, is RB*65280 + 3 < 2^24 for RB <= 256
      const unsigned nt_inv = (unsigned)(RB * 48896 + 2);
      b = ((valid ? (unsigned)n2 + 1u : nt_inv + 1u) + (unsigned)(RB * 16384)) << IB | jl;
    }
    I.base[row] = b;
    I.nq[row] = valid ? n2 : 0;
  }
  if (KIND == KIND_F32_L2 && !ok) nonintegral[img] = gen;  // (stamped with the prepare pass: no clearing between passes)
}

// ---------------------------------------------------------------- MFMA k-NN kernel
struct Top2 {
  unsigned s0, s1;  // key >> IB of best / 2nd best (0xFFFFFFFF = empty)
  int j0, j1;       // their train rows
};

__device__ __forceinline__ void chunk_merge(Top2& g, unsigned k0, unsigned k1, int cb) {
  // candidates of a later chunk: on equal distance the earlier (already held) row wins
  const unsigned ns0 = k0 >> IB, ns1 = k1 >> IB;
  const int nj0 = cb + (int)(k0 & (CHUNK_ROWS - 1)), nj1 = cb + (int)(k1 & (CHUNK_ROWS - 1));
  const bool e0 = (k0 == KEY_EMPTY), e1 = (k1 == KEY_EMPTY);
  const unsigned a0 = e0 ? KEY_EMPTY : ns0, a1 = e1 ? KEY_EMPTY : ns1;
  if (a0 < g.s0) {
    if (a1 < g.s0) {
      g.s1 = a1;
      g.j1 = nj1;
    } else {
      g.s1 = g.s0;
      g.j1 = g.j0;
    }
    g.s0 = a0;
    g.j0 = nj0;
  } else if (a0 < g.s1) {
    g.s1 = a0;
    g.j1 = nj0;
  }
}

__device__ __forceinline__ bool lex_less(unsigned sa, int ja, unsigned sb, int jb) {
  return sa < sb || (sa == sb && ja < jb);
}

// MODE 0: L2 on centred i8 rows, MODE 1: Hamming on +-1 rows.  SR = train rows per LDS stage.
//
// Workgroup = 4 waves; wave w owns query tiles qtile0+2w, +2w+1 (64 queries, B fragments and the
// per-query constant accumulator input stay in registers for the whole kernel) and sweeps every
// train tile of the pair's train image.  Train tiles are staged through LDS (two stage buffers,
// register-staged copy issued half a stage ahead), 16 accumulator registers = 16 train rows of
// one query per lane.  "Units" (train tile x query tile) are software-pipelined: the MFMA chain
// of unit n+1 is issued before the VALU top-2 insertion of unit n.
template <int KS, int MODE, int SR>
__global__ __launch_bounds__(256, 2) void knn_mfma_kernel(const ImgDev* __restrict__ imgs,
                                                          const int2* __restrict__ pairs,
                                                          const WorkItem* __restrict__ items,
                                                          const int* __restrict__ nonintegral, int gen,
                                                          int4* __restrict__ knn, int maxq,
                                                          int* __restrict__ fix_count,
                                                          int2* __restrict__ fix_items) {
  constexpr int NC = 2 * KS;
  constexpr int RB = 32 * KS;
  constexpr int KOFF = RB * 16384;  // >= any ||q||^2 of centred i8 rows
  constexpr int STAGE_ROW_BYTES = SR * RB;
  constexpr int STAGE_BYTES = STAGE_ROW_BYTES + SR * 4;
  constexpr int TILES = SR / TILE_ROWS;
  constexpr int PIECES = STAGE_ROW_BYTES / 4096;  // 16-byte pieces per thread per stage
  constexpr int HALF = PIECES > 1 ? PIECES / 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const WorkItem it = items[blockIdx.x];
  const int2 pr = pairs[it.pair];
  if ((nonintegral[pr.x] == gen) | (nonintegral[pr.y] == gen)) return;  // left to the exact kernel
  const ImgDev Q = imgs[pr.x];
  const ImgDev T = imgs[pr.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nqt = (Q.n_rows + TILE_ROWS - 1) / TILE_ROWS;
  const int qt[2] = {it.qtile0 + 2 * wave, it.qtile0 + 2 * wave + 1};

  // key = base - 2^(IB+1) * dot  (L2)   |   base - 2^(IB-1) * dot  (Hamming).  The query's own
  // norm is the same for every train row, so it stays out of the keys (added back when the
  // winners are emitted); base carries +KOFF so that keys stay non-negative without it.
  // The multiplier is made opaque so that hipcc keeps one v_mad_i32_i24 per element instead of
  // strength-reducing it into a shift and a subtract.
  int mul;
  asm volatile("s_mov_b32 %0, %1" : "=s"(mul) : "i"((MODE == 0) ? -(2 << IB) : -(1 << (IB - 1))));

  // query fragments (B operand): lane (r,h) holds bytes [32ks+16h, +16) of query row r
  v4i bq[2][KS];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = qt[u] < nqt ? qt[u] : (nqt > 0 ? nqt - 1 : 0);
    g_v4i_p src = (g_v4i_p)(Q.tiles + (size_t)t * (TILE_ROWS * RB));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[u][ks] = src[chunk_pos<NC>(r, 2 * ks + h)];
  }
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // per-lane LDS byte offsets of the A fragments inside a tile, and of the key bases
  int aoff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) aoff[ks] = chunk_pos<NC>(r, 2 * ks + h) * 16;
  const int boff = STAGE_ROW_BYTES + h * 16;

  Top2 g[2];
  unsigned k0[2], k1[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    g[u].s0 = g[u].s1 = KEY_EMPTY;
    g[u].j0 = g[u].j1 = -1;
    k0[u] = k1[u] = KEY_EMPTY;
  }

  const int nstages = T.n_pad / SR;
  const int tid = threadIdx.x;

  // Stage copy: the tile image is already in LDS order, so a stage is a linear copy; LDS-DMA
  // (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, no ds_write) moves it.
  auto stage_copy = [&](int stage, unsigned char* dstb) {
    const unsigned char* src = (const unsigned char*)T.tiles + (size_t)stage * STAGE_ROW_BYTES;
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      __builtin_amdgcn_global_load_lds((const SFM_GLOBAL void*)(src + (size_t)(i * 256 + tid) * 16),
                                       (__attribute__((address_space(3))) void*)(dstb + i * 4096 + wave * 1024), 16, 0, 0);
    if (wave < SR / 64)
      __builtin_amdgcn_global_load_lds((const SFM_GLOBAL void*)(T.base + (size_t)stage * SR + tid),
                                       (__attribute__((address_space(3))) void*)(dstb + STAGE_ROW_BYTES + wave * 256), 4, 0, 0);
  };
  stage_copy(0, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // one step of the MFMA chain of (fragments fr, query tile u): ks-th K slice
#define SFM_MFMA(ACC, FR, u, ks) \
  ACC = __builtin_amdgcn_mfma_i32_32x32x32_i8(FR[ks], bq[u][ks], (ks) == 0 ? zero16 : ACC, 0, 0, 0)
  // top-2 insertion of accumulator element e of query tile u
#define SFM_INS(ACC, BS, u, e)                                                             \
  do {                                                                                     \
    const unsigned key_ = (unsigned)__mul24(ACC[e], mul) + BS[e];            /* v_mad_i32_i24 */ \
    const unsigned n1_ = umax_(umin_(k0[u], k1[u]), umin_(umax_(k0[u], k1[u]), key_)); /* v_med3_u32 */ \
    k0[u] = umin_(k0[u], key_);                                                            \
    k1[u] = n1_;                                                                           \
    asm volatile("" : "+v"(k0[u]), "+v"(k1[u])); /* no min/max re-association across elements */ \
  } while (0)
  auto ld_afrag = [&](const unsigned char* sb, int tl, int ks) -> v4i {
    const int4 x = *(const int4*)(sb + tl * (TILE_ROWS * RB) + aoff[ks]);
    return v4i{x.x, x.y, x.z, x.w};
  };
  auto ld_bases = [&](const unsigned char* sb, int tl, int gq, unsigned (&bs)[16]) {
    const uint4 x = *(const uint4*)(sb + boff + (tl * TILE_ROWS + 8 * gq) * 4);
    bs[4 * gq] = x.x;
    bs[4 * gq + 1] = x.y;
    bs[4 * gq + 2] = x.z;
    bs[4 * gq + 3] = x.w;
  };

  // MFMA and VALU of a SIMD do not overlap here: the epilogue is VALU-issue-bound (3 ops per
  // distance: scripts/ubench/valu_rate.hip measures time = VALU issue + 8 cycles per MFMA whatever
  // the interleave), and a wave that issues a dependent MFMA chain back to back stalls in order
  // on the matrix pipe.  So per train tile the 2*KS MFMAs of the NEXT tile (both query tiles)
  // are spread one per group through the 96 insertion ops of the CURRENT tile, the insertions of
  // the two query tiles alternate (two independent dependency chains per wave), and the LDS
  // fragment / key-base reads of the tile after ride along.  Groups are pinned by sched_barrier.
  constexpr int MPG = (KS + 3) / 4;  // MFMAs per group and query tile
#define SFM_GROUP(g_, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, do_mfma, do_frag)              \
  do {                                                                                         \
    if (do_mfma) {                                                                             \
      _Pragma("unroll") for (int j_ = 0; j_ < MPG; ++j_) {                                     \
        const int ks_ = ((g_) & 3) * MPG + j_;                                                 \
        if (ks_ < KS) {                                                                        \
          if ((g_) < 4) SFM_MFMA(NA0, FN, 0, ks_);                                             \
          else SFM_MFMA(NA1, FN, 1, ks_);                                                      \
        }                                                                                      \
      }                                                                                        \
      if ((g_) >= 4) ld_bases(SB, (tl) + 1, (g_) - 4, BN);                                     \
      else if (do_frag) {                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < MPG; ++j_)                                     \
          if ((g_) * MPG + j_ < KS) FNN[(g_) * MPG + j_] = ld_afrag(SB, (tl) + 2, (g_) * MPG + j_); \
      }                                                                                        \
    }                                                                                          \
    SFM_INS(CA0, BC, 0, 2 * (g_));                                                             \
    SFM_INS(CA1, BC, 1, 2 * (g_));                                                             \
    SFM_INS(CA0, BC, 0, 2 * (g_) + 1);                                                         \
    SFM_INS(CA1, BC, 1, 2 * (g_) + 1);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
#define SFM_BODY(SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_)                        \
  do {                                                                                         \
    SFM_GROUP(0, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
    SFM_GROUP(1, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
    SFM_GROUP(2, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
    SFM_GROUP(3, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
    SFM_GROUP(4, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
    SFM_GROUP(5, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
    SFM_GROUP(6, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
    SFM_GROUP(7, SB, tl, CA0, CA1, NA0, NA1, FN, FNN, BC, BN, mf_, fr_);                       \
  } while (0)

  // fragments ping-pong between fa / fb, key bases between ba / bb, accumulators between
  // (A0, A1) and (B0, B1): tile tl inserts from one pair while tile tl+1 accumulates in the
  // other.  The pipeline runs across stages: one workgroup barrier per stage, placed before the
  // stage's last tile body, whose MFMAs already read tile 0 of the next stage.
  static_assert(TILES == 4 || TILES == 8, "stage = 4 or 8 train tiles");
  v4i fa[KS], fb[KS];
  unsigned ba[16], bb[16];
  v16i A0, A1, B0, B1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fa[ks] = ld_afrag(lds, 0, ks);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) ld_bases(lds, 0, gq, ba);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) SFM_MFMA(A0, fa, 0, ks);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) SFM_MFMA(A1, fa, 1, ks);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fb[ks] = ld_afrag(lds, 1, ks);
  __builtin_amdgcn_sched_barrier(0);

  for (int s = 0; s < nstages; ++s) {
    const unsigned char* sb = lds + (s & 1) * STAGE_BYTES;
    unsigned char* nb = lds + ((s & 1) ^ 1) * STAGE_BYTES;
    const bool more = s + 1 < nstages;
    // the other buffer was last read before the previous stage's barrier: refill it now
    if (more) stage_copy(s + 1, nb);

    SFM_BODY(sb, 0, A0, A1, B0, B1, fb, fa, ba, bb, true, true);
    SFM_BODY(sb, 1, B0, B1, A0, A1, fa, fb, bb, ba, true, true);
    if (TILES == 8) {
      SFM_BODY(sb, 2, A0, A1, B0, B1, fb, fa, ba, bb, true, true);
      SFM_BODY(sb, 3, B0, B1, A0, A1, fa, fb, bb, ba, true, true);
      SFM_BODY(sb, 4, A0, A1, B0, B1, fb, fa, ba, bb, true, true);
      SFM_BODY(sb, 5, B0, B1, A0, A1, fa, fb, bb, ba, true, true);
    }
    // tile TILES-2: its MFMAs take the last tile of this stage; nothing of this stage left to prefetch
    SFM_BODY(sb, TILES - 2, A0, A1, B0, B1, fb, fa, ba, bb, true, false);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the next stage has landed
    __syncthreads();
    // tile TILES-1: MFMAs on tile 0 of the next stage (fragments fetched now), prefetch of its tile 1
    if (more) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fa[ks] = ld_afrag(nb, 0, ks);
    }
    SFM_BODY(nb, -1, B0, B1, A0, A1, fa, fb, bb, ba, more, more);
    // tie-break chunk boundary: fold the 8-bit-indexed keys into the running (distance, row)
    if (((s + 1) * SR) % CHUNK_ROWS == 0) {
      const int cb = ((s * SR) / CHUNK_ROWS) * CHUNK_ROWS;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        chunk_merge(g[u], k0[u], k1[u], cb);
        k0[u] = k1[u] = KEY_EMPTY;
      }
    }
  }
#undef SFM_BODY
#undef SFM_GROUP
#undef SFM_INS
#undef SFM_MFMA

  // merge the two half-waves (rows 4h.. of each 8-row group) and emit
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    Top2 p;
    p.s0 = __shfl_xor(g[u].s0, 32);
    p.s1 = __shfl_xor(g[u].s1, 32);
    p.j0 = __shfl_xor(g[u].j0, 32);
    p.j1 = __shfl_xor(g[u].j1, 32);
    Top2 m;
    if (lex_less(p.s0, p.j0, g[u].s0, g[u].j0)) {
      m.s0 = p.s0;
      m.j0 = p.j0;
      if (lex_less(g[u].s0, g[u].j0, p.s1, p.j1)) {
        m.s1 = g[u].s0;
        m.j1 = g[u].j0;
      } else {
        m.s1 = p.s1;
        m.j1 = p.j1;
      }
    } else {
      m.s0 = g[u].s0;
      m.j0 = g[u].j0;
      if (lex_less(p.s0, p.j0, g[u].s1, g[u].j1)) {
        m.s1 = p.s0;
        m.j1 = p.j0;
      } else {
        m.s1 = g[u].s1;
        m.j1 = g[u].j1;
      }
    }
    const int q = qt[u] * TILE_ROWS + r;
    if (h == 0 && qt[u] < nqt && q < Q.n_rows) {
      int s0i, s1i;
      if (MODE == 0) {
        const int adj = ((g_i32_p)Q.nq)[q] - 1 - KOFF;  // key >> IB = ||t||^2 + 1 + KOFF - 2 q.t
        s0i = (int)m.s0 + adj;
        s1i = (int)m.s1 + adj;
      } else {
        s0i = (int)m.s0;
        s1i = (int)m.s1;
      }
      const bool v0 = m.j0 >= 0 && m.j0 < T.n_rows, v1 = m.j1 >= 0 && m.j1 < T.n_rows;
      float d0, d1;
      if (MODE == 0) {
        d0 = sqrtf((float)s0i);
        d1 = sqrtf((float)s1i);
        if (v1 && s1i >= (1 << 22)) {  // sqrtf may merge neighbouring integers: redo exactly
          const int k = atomicAdd(fix_count, 1);
          if (k < FIX_CAP) fix_items[k] = make_int2(it.pair, q);
        }
      } else {
        d0 = (float)s0i;
        d1 = (float)s1i;
      }
      int4 o;
      o.x = v0 ? m.j0 : -1;
      o.y = v1 ? m.j1 : -1;
      o.z = __float_as_int(v0 ? d0 : 3.402823466e+38f);
      o.w = __float_as_int(v1 ? d1 : 3.402823466e+38f);
      knn[(size_t)it.pair * maxq + q] = o;
    }
  }
}

// ---------------------------------------------------------------- exact (VALU) k-NN
// One wave per query.  Distances and their ordering restate cv::batchDistance literally:
// float sqrt of the squared distance, insertion by (float distance, lower train index).
template <int KIND>
__device__ __forceinline__ float exact_dist(const void* qrow, const void* trow, int dim) {
  if (KIND == KIND_F32_L2) {
    // f32 accumulation in 8 interleaved partial sums with a fixed combine order (the order
    // the CPU restatement defines; OpenCV leaves its SIMD lane order unspecified)
    const float* a = (const float*)qrow;
    const float* b = (const float*)trow;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= dim; k += 8) {
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const float d = __fsub_rn(a[k + l], b[k + l]);
        acc[l] = __fadd_rn(acc[l], __fmul_rn(d, d));
      }
    }
    for (int l = 0; k < dim; ++k, ++l) {
      const float d = __fsub_rn(a[k], b[k]);
      acc[l] = __fadd_rn(acc[l], __fmul_rn(d, d));
    }
    const float s = __fadd_rn(__fadd_rn(__fadd_rn(acc[0], acc[4]), __fadd_rn(acc[2], acc[6])),
                              __fadd_rn(__fadd_rn(acc[1], acc[5]), __fadd_rn(acc[3], acc[7])));
    return sqrtf(s);
  } else if (KIND == KIND_U8_L2) {
    const unsigned char* a = (const unsigned char*)qrow;
    const unsigned char* b = (const unsigned char*)trow;
    int s = 0;
    for (int k = 0; k < dim; ++k) {
      const int d = (int)a[k] - (int)b[k];
      s += d * d;
    }
    return sqrtf((float)s);
  } else {
    const unsigned char* a = (const unsigned char*)qrow;
    const unsigned char* b = (const unsigned char*)trow;
    int s = 0;
    for (int k = 0; k < dim; ++k) s += __popc((unsigned)(a[k] ^ b[k]));
    return (float)s;
  }
}

template <int KIND>
__device__ void exact_query(const ImgDev& Q, const ImgDev& T, int q, int dim, int4* out) {
  const int lane = threadIdx.x & 63;
  const size_t rowb = (size_t)dim * (KIND == KIND_F32_L2 ? 4 : 1);
  const unsigned char* qrow = (const unsigned char*)Q.raw + (size_t)q * rowb;
  float d0 = 3.402823466e+38f, d1 = 3.402823466e+38f;
  int j0 = -1, j1 = -1;
  for (int j = lane; j < T.n_rows; j += 64) {
    const float d = exact_dist<KIND>(qrow, (const unsigned char*)T.raw + (size_t)j * rowb, dim);
    if (d < d1 || j1 < 0) {
      if (d < d0 || j0 < 0) {
        d1 = d0;
        j1 = j0;
        d0 = d;
        j0 = j;
      } else {
        d1 = d;
        j1 = j;
      }
    }
  }
  // wave merge by (distance, index); empty slots carry j=-1 and sort last
  auto less = [](float da, int ja, float db, int jb) {
    if (ja < 0) return false;
    if (jb < 0) return true;
    return da < db || (da == db && ja < jb);
  };
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float e0 = __shfl_xor(d0, o), e1 = __shfl_xor(d1, o);
    const int i0 = __shfl_xor(j0, o), i1 = __shfl_xor(j1, o);
    float n0, n1;
    int m0, m1;
    if (less(e0, i0, d0, j0)) {
      n0 = e0;
      m0 = i0;
      if (less(d0, j0, e1, i1)) {
        n1 = d0;
        m1 = j0;
      } else {
        n1 = e1;
        m1 = i1;
      }
    } else {
      n0 = d0;
      m0 = j0;
      if (less(e0, i0, d1, j1)) {
        n1 = e0;
        m1 = i0;
      } else {
        n1 = d1;
        m1 = j1;
      }
    }
    d0 = n0;
    d1 = n1;
    j0 = m0;
    j1 = m1;
  }
  if (lane == 0) {
    int4 o;
    o.x = j0;
    o.y = j1;
    o.z = __float_as_int(j0 >= 0 ? d0 : 3.402823466e+38f);
    o.w = __float_as_int(j1 >= 0 ? d1 : 3.402823466e+38f);
    *out = o;
  }
}

// Fixed-size grid.  Phase 1: every pair that touches a non-integer-valued f32 image (or, with
// force_all, every pair: norms/dims the MFMA kernel is not instantiated for).  Phase 2: the
// queries the MFMA kernel flagged.
template <int KIND>
__global__ __launch_bounds__(256) void knn_exact_kernel(const ImgDev* __restrict__ imgs,
                                                        const int2* __restrict__ pairs, int n_pairs,
                                                        const int* __restrict__ nonintegral, int gen,
                                                        int force_all, int dim,
                                                        int4* __restrict__ knn, int maxq,
                                                        const int* __restrict__ fix_count,
                                                        const int2* __restrict__ fix_items) {
  const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
    const int2 pr = pairs[p];
    if (!(force_all | (nonintegral[pr.x] == gen) | (nonintegral[pr.y] == gen))) continue;
    const ImgDev Q = imgs[pr.x], T = imgs[pr.y];
    for (int q = wave; q < Q.n_rows; q += wpb) exact_query<KIND>(Q, T, q, dim, &knn[(size_t)p * maxq + q]);
  }
  if (!force_all) {
    int nfix = *fix_count;
    nfix = nfix < FIX_CAP ? nfix : FIX_CAP;
    for (int i = blockIdx.x * wpb + wave; i < nfix; i += gridDim.x * wpb) {
      const int2 f = fix_items[i];
      const int2 pr = pairs[f.x];
      exact_query<KIND>(imgs[pr.x], imgs[pr.y], f.y, dim, &knn[(size_t)f.x * maxq + f.y]);
    }
  }
}

// ---------------------------------------------------------------- ratio test + compaction
// reference src/Sfm.cpp:603-607: keep knn[i][0] iff d0 <= ratio*d1 (float), ascending queryIdx
__global__ __launch_bounds__(256) void compact_kernel(const ImgDev* __restrict__ imgs,
                                                      const int2* __restrict__ pairs,
                                                      const int4* __restrict__ knn, int maxq,
                                                      float ratio, int* __restrict__ counts,
                                                      int* __restrict__ out_q, int* __restrict__ out_t,
                                                      float* __restrict__ out_d, int* __restrict__ fix_count) {
  __shared__ int wsum[4];
  __shared__ int running;
  const int p = blockIdx.x;
  // the fix-up list of this run has been consumed (knn_exact_kernel ran before this kernel): clear
  // its counter for the plan's next run here instead of with a memset node in front of every run
  if (blockIdx.x == 0 && threadIdx.x == 0) *fix_count = 0;
  const int2 pr = pairs[p];
  const int nq = imgs[pr.x].n_rows, nt = imgs[pr.y].n_rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  if (nt >= 2) {
    for (int q0 = 0; q0 < nq; q0 += 256) {
      const int q = q0 + threadIdx.x;
      int4 e = make_int4(-1, -1, 0, 0);
      bool keep = false;
      if (q < nq) {
        e = knn[(size_t)p * maxq + q];
        keep = e.y >= 0 && __int_as_float(e.z) <= __fmul_rn(ratio, __int_as_float(e.w));
      }
      const unsigned long long b = __ballot(keep);
      const int before = __popcll(b & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[wave] = __popcll(b);
      __syncthreads();
      int off = running;
      for (int w = 0; w < wave; ++w) off += wsum[w];
      if (keep) {
        const size_t o = (size_t)p * maxq + off + before;
        out_q[o] = q;
        out_t[o] = e.x;
        out_d[o] = __int_as_float(e.z);
      }
      __syncthreads();
      if (threadIdx.x == 0) running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) counts[p] = running;
}

}  // namespace

// ================================================================= host side
struct sfmhip_imageset {
  sfmhip_ctx* ctx;
  int n_images, dim, dtype, norm;
  int kind, ks, sr;  // ks==0: no MFMA instantiation -> exact kernel only
  std::vector<int> n_rows, n_pad;
  std::vector<ImgDev> h_imgs;
  std::vector<void*> owned_raw;
  ImgDev* d_imgs = nullptr;
  int8_t* d_tiles = nullptr;
  unsigned* d_base = nullptr;
  int* d_nq = nullptr;
  int* d_tile_img = nullptr;
  int* d_tile_first = nullptr;
  int* d_nonintegral = nullptr;  // per image: the number (gen) of the prepare pass that found non-integer f32 values
  int gen = 0;
  int total_tiles = 0, maxq = 0;
  bool imgs_dirty = true;
  hipEvent_t ev_prep0 = nullptr, ev_prep1 = nullptr;
  bool prep_timed = false;
};

struct sfmhip_matchplan {
  sfmhip_imageset* set;
  int n_pairs, n_items, maxq;
  int cap_pairs = 0;
  size_t cap_items = 0;
  std::vector<int> h_pairs;
  int2* d_pairs = nullptr;
  WorkItem* d_items = nullptr;
  int4* d_knn = nullptr;
  int* d_counts = nullptr;
  int *d_out_q = nullptr, *d_out_t = nullptr;
  float* d_out_d = nullptr;
  int* d_fix_count = nullptr;
  int2* d_fix_items = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  bool timed = false;
};

static int pick_ks(int kind, int dim) {
  if (kind == KIND_U8_HAMMING) {
    const int bits = dim * 8;
    if (bits <= 256) return 8;
    return 0;
  }
  if (dim <= 32) return 1;
  if (dim <= 64) return 2;
  if (dim <= 128) return 4;
  if (dim <= 256) return 8;
  return 0;
}

extern "C" int sfmhip_imageset_create(sfmhip_ctx* ctx, int n_images, const int32_t* n_rows, int dim,
                                      int dtype, int norm, sfmhip_imageset** out) {
  if (!ctx || !out || n_images <= 0 || !n_rows || dim <= 0) return SFMHIP_ERR_ARG;
  if (dtype != SFMHIP_F32 && dtype != SFMHIP_U8) return SFMHIP_ERR_ARG;
  if (norm != SFMHIP_L2 && norm != SFMHIP_HAMMING) return SFMHIP_ERR_ARG;
  if (norm == SFMHIP_HAMMING && dtype != SFMHIP_U8) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  sfmhip_imageset* s = new sfmhip_imageset();
  s->ctx = ctx;
  s->n_images = n_images;
  s->dim = dim;
  s->dtype = dtype;
  s->norm = norm;
  s->kind = norm == SFMHIP_HAMMING ? KIND_U8_HAMMING : (dtype == SFMHIP_F32 ? KIND_F32_L2 : KIND_U8_L2);
  s->ks = pick_ks(s->kind, dim);
  s->sr = s->ks == 8 ? 128 : 256;
  const int rb = 32 * (s->ks ? s->ks : 1);
  size_t tot_pad = 0;
  std::vector<int> tile_img, tile_first(n_images + 1, 0);
  for (int i = 0; i < n_images; ++i) {
    if (n_rows[i] < 0) {
      delete s;
      return SFMHIP_ERR_ARG;
    }
    s->n_rows.push_back(n_rows[i]);
    const int pad = ((n_rows[i] + CHUNK_ROWS - 1) / CHUNK_ROWS) * CHUNK_ROWS;
    s->n_pad.push_back(pad);
    s->maxq = std::max(s->maxq, n_rows[i]);
    tile_first[i] = (int)tile_img.size();
    for (int t = 0; t < pad / TILE_ROWS; ++t) tile_img.push_back(i);
    tot_pad += pad;
  }
  tile_first[n_images] = (int)tile_img.size();
  s->total_tiles = (int)tile_img.size();
  int rc = SFMHIP_OK;
  if ((rc = sfm_dev_alloc(&s->d_imgs, (size_t)n_images)) || (rc = sfm_dev_alloc(&s->d_tiles, tot_pad * rb)) ||
      (rc = sfm_dev_alloc(&s->d_base, tot_pad)) || (rc = sfm_dev_alloc(&s->d_nq, tot_pad)) ||
      (rc = sfm_dev_alloc(&s->d_tile_img, tile_img.size())) ||
      (rc = sfm_dev_alloc(&s->d_tile_first, (size_t)n_images + 1)) ||
      (rc = sfm_dev_alloc(&s->d_nonintegral, (size_t)n_images))) {
    sfmhip_imageset_destroy(s);
    return rc;
  }
  SFM_HIP_TRY(hipMemcpy(s->d_tile_img, tile_img.data(), tile_img.size() * sizeof(int), hipMemcpyHostToDevice));
  SFM_HIP_TRY(hipMemcpy(s->d_tile_first, tile_first.data(), tile_first.size() * sizeof(int), hipMemcpyHostToDevice));
  SFM_HIP_TRY(hipMemset(s->d_nonintegral, 0xFF, n_images * sizeof(int)));  // -1: set by no prepare pass
  s->h_imgs.resize(n_images);
  s->owned_raw.assign(n_images, nullptr);
  size_t off = 0;
  for (int i = 0; i < n_images; ++i) {
    ImgDev& I = s->h_imgs[i];
    I.raw = nullptr;
    I.tiles = s->d_tiles + off * rb;
    I.base = s->d_base + off;
    I.nq = s->d_nq + off;
    I.n_rows = s->n_rows[i];
    I.n_pad = s->n_pad[i];
    off += s->n_pad[i];
  }
  SFM_HIP_TRY(hipEventCreate(&s->ev_prep0));
  SFM_HIP_TRY(hipEventCreate(&s->ev_prep1));
  *out = s;
  return SFMHIP_OK;
}

extern "C" int sfmhip_imageset_upload(sfmhip_imageset* s, int image, const void* host_rows) {
  if (!s || image < 0 || image >= s->n_images || (!host_rows && s->n_rows[image] > 0)) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  const size_t bytes = (size_t)s->n_rows[image] * s->dim * (s->dtype == SFMHIP_F32 ? 4 : 1);
  if (!s->owned_raw[image]) {
    void* p = nullptr;
    SFM_HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    s->owned_raw[image] = p;
  }
  if (bytes) SFM_HIP_TRY(hipMemcpyAsync(s->owned_raw[image], host_rows, bytes, hipMemcpyHostToDevice, s->ctx->stream));
  SFM_HIP_TRY(hipStreamSynchronize(s->ctx->stream));  // host buffer is the caller's: do not outlive the call
  s->h_imgs[image].raw = s->owned_raw[image];
  s->imgs_dirty = true;
  return SFMHIP_OK;
}

extern "C" int sfmhip_imageset_adopt_device(sfmhip_imageset* s, int image, const void* device_rows) {
  if (!s || image < 0 || image >= s->n_images || !device_rows) return SFMHIP_ERR_ARG;
  s->h_imgs[image].raw = device_rows;
  s->imgs_dirty = true;
  return SFMHIP_OK;
}

template <int KS>
static void launch_prepare(sfmhip_imageset* s) {
  const dim3 grid(s->total_tiles), block(32 * 2 * KS);
  hipStream_t st = s->ctx->stream;
  if (s->kind == KIND_F32_L2)
    hipLaunchKernelGGL((prepare_kernel<KS, KIND_F32_L2>), grid, block, 0, st, s->d_imgs, s->d_tile_img, s->d_tile_first, s->dim, s->d_nonintegral, s->gen);
  else if (s->kind == KIND_U8_L2)
    hipLaunchKernelGGL((prepare_kernel<KS, KIND_U8_L2>), grid, block, 0, st, s->d_imgs, s->d_tile_img, s->d_tile_first, s->dim, s->d_nonintegral, s->gen);
  else
    hipLaunchKernelGGL((prepare_kernel<KS, KIND_U8_HAMMING>), grid, block, 0, st, s->d_imgs, s->d_tile_img, s->d_tile_first, s->dim, s->d_nonintegral, s->gen);
}

extern "C" int sfmhip_imageset_prepare_async(sfmhip_imageset* s) {
  if (!s) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  for (int i = 0; i < s->n_images; ++i)
    if (!s->h_imgs[i].raw && s->n_rows[i] > 0) return SFMHIP_ERR_STATE;
  hipStream_t st = s->ctx->stream;
  if (s->imgs_dirty) {
    SFM_HIP_TRY(hipMemcpyAsync(s->d_imgs, s->h_imgs.data(), sizeof(ImgDev) * s->n_images, hipMemcpyHostToDevice, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    s->imgs_dirty = false;
  }
  const bool timing = s->ctx->timing;
  if (timing) SFM_HIP_TRY(hipEventRecord(s->ev_prep0, st));
  ++s->gen;  // the non-integral flags carry the number of the pass that set them
  if (s->total_tiles > 0) {
    switch (s->ks) {
      case 1: launch_prepare<1>(s); break;
      case 2: launch_prepare<2>(s); break;
      case 4: launch_prepare<4>(s); break;
      case 8: launch_prepare<8>(s); break;
      default: break;  // exact kernel reads the raw rows
    }
    SFM_HIP_TRY(hipGetLastError());
  }
  if (timing) SFM_HIP_TRY(hipEventRecord(s->ev_prep1, st));
  s->prep_timed = timing;
  return SFMHIP_OK;
}

extern "C" void sfmhip_imageset_destroy(sfmhip_imageset* s) {
  if (!s) return;
  hipSetDevice(s->ctx->device);
  for (void* p : s->owned_raw)
    if (p) hipFree(p);
  hipFree(s->d_imgs);
  hipFree(s->d_tiles);
  hipFree(s->d_base);
  hipFree(s->d_nq);
  hipFree(s->d_tile_img);
  hipFree(s->d_tile_first);
  hipFree(s->d_nonintegral);
  if (s->ev_prep0) hipEventDestroy(s->ev_prep0);
  if (s->ev_prep1) hipEventDestroy(s->ev_prep1);
  delete s;
}

// work list: one workgroup per (pair, block of 8 query tiles).  Ordered so that the blocks a
// round-robin dispatcher puts on one XCD (equal index mod 8) walk the same train image
// together: train images are dealt to the 8 groups, each group sorted by train image.
static void build_work_items(const sfmhip_imageset* s, const int32_t* pairs, int n_pairs, std::vector<WorkItem>& items) {
  std::vector<WorkItem> lanes[8];
  std::vector<int> order(n_pairs);
  for (int p = 0; p < n_pairs; ++p) order[p] = p;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pairs[2 * a + 1] < pairs[2 * b + 1]; });
  size_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int cur_t = -1, cur_lane = 0;
  for (int p : order) {
    const int qi = pairs[2 * p], ti = pairs[2 * p + 1];
    if (s->n_rows[ti] < 1 || s->n_rows[qi] == 0) continue;  // nt==1 still has a 1-NN list
    if (ti != cur_t) {
      cur_t = ti;
      cur_lane = (int)(std::min_element(load, load + 8) - load);
    }
    const int nqt = (s->n_rows[qi] + TILE_ROWS - 1) / TILE_ROWS;
    for (int t0 = 0; t0 < nqt; t0 += 8) {
      lanes[cur_lane].push_back(WorkItem{p, t0});
      load[cur_lane] += (size_t)s->n_pad[ti];
    }
  }
  items.clear();
  size_t mx = 0;
  for (auto& l : lanes) mx = std::max(mx, l.size());
  // interleave; shorter lanes are padded by stealing from the longest so the list stays dense
  std::vector<size_t> pos(8, 0);
  for (size_t i = 0; i < mx; ++i)
    for (int x = 0; x < 8; ++x)
      if (pos[x] < lanes[x].size()) items.push_back(lanes[x][pos[x]++]);
}

static int plan_check_pairs(const sfmhip_imageset* s, const int32_t* pairs, int n_pairs) {
  for (int p = 0; p < n_pairs; ++p)
    if (pairs[2 * p] < 0 || pairs[2 * p] >= s->n_images || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= s->n_images)
      return SFMHIP_ERR_ARG;
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_create(sfmhip_imageset* s, const int32_t* pairs, int n_pairs, sfmhip_matchplan** out) {
  if (!s || !out || n_pairs < 0 || (n_pairs && !pairs)) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  SFM_TRY(plan_check_pairs(s, pairs, n_pairs));
  sfmhip_matchplan* pl = new sfmhip_matchplan();
  pl->set = s;
  pl->n_pairs = n_pairs;
  pl->cap_pairs = std::max(n_pairs, 1);
  pl->maxq = std::max(s->maxq, 1);
  pl->h_pairs.assign(pairs, pairs + 2 * (size_t)n_pairs);
  std::vector<WorkItem> items;
  build_work_items(s, pairs, n_pairs, items);
  pl->n_items = (int)items.size();
  // room for any pair list of up to cap_pairs pairs (sfmhip_matchplan_set_pairs)
  pl->cap_items = (size_t)pl->cap_pairs * (size_t)(((pl->maxq + TILE_ROWS - 1) / TILE_ROWS + 7) / 8);
  int rc = SFMHIP_OK;
  const size_t slots = (size_t)pl->cap_pairs * pl->maxq;
  if ((rc = sfm_dev_alloc(&pl->d_pairs, (size_t)pl->cap_pairs)) || (rc = sfm_dev_alloc(&pl->d_items, pl->cap_items)) ||
      (rc = sfm_dev_alloc(&pl->d_knn, slots)) || (rc = sfm_dev_alloc(&pl->d_counts, (size_t)pl->cap_pairs)) ||
      (rc = sfm_dev_alloc(&pl->d_out_q, slots)) || (rc = sfm_dev_alloc(&pl->d_out_t, slots)) ||
      (rc = sfm_dev_alloc(&pl->d_out_d, slots)) || (rc = sfm_dev_alloc(&pl->d_fix_count, (size_t)1)) ||
      (rc = sfm_dev_alloc(&pl->d_fix_items, (size_t)FIX_CAP))) {
    sfmhip_matchplan_destroy(pl);
    return rc;
  }
  if (n_pairs) SFM_HIP_TRY(hipMemcpy(pl->d_pairs, pairs, sizeof(int2) * n_pairs, hipMemcpyHostToDevice));
  if (!items.empty()) SFM_HIP_TRY(hipMemcpy(pl->d_items, items.data(), sizeof(WorkItem) * items.size(), hipMemcpyHostToDevice));
  SFM_HIP_TRY(hipMemset(pl->d_counts, 0, sizeof(int) * pl->cap_pairs));
  SFM_HIP_TRY(hipMemset(pl->d_fix_count, 0, sizeof(int)));  // (every run leaves it cleared: compact_kernel)
  for (auto& e : pl->ev) SFM_HIP_TRY(hipEventCreate(&e));
  *out = pl;
  return SFMHIP_OK;
}

// Re-target a plan at another pair list (at most as many pairs as it was created with): the
// device buffers are reused, so a one-pair plan serves every getMatching call of a session.
extern "C" int sfmhip_matchplan_set_pairs(sfmhip_matchplan* pl, const int32_t* pairs, int n_pairs) {
  if (!pl || n_pairs < 0 || (n_pairs && !pairs) || n_pairs > pl->cap_pairs) return SFMHIP_ERR_ARG;
  sfmhip_imageset* s = pl->set;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  SFM_TRY(plan_check_pairs(s, pairs, n_pairs));
  std::vector<WorkItem> items;
  build_work_items(s, pairs, n_pairs, items);
  if (items.size() > pl->cap_items) return SFMHIP_ERR_STATE;
  hipStream_t st = s->ctx->stream;
  SFM_HIP_TRY(hipStreamSynchronize(st));  // a previous run may still read the old lists
  pl->n_pairs = n_pairs;
  pl->n_items = (int)items.size();
  pl->h_pairs.assign(pairs, pairs + 2 * (size_t)n_pairs);
  if (n_pairs) SFM_HIP_TRY(hipMemcpyAsync(pl->d_pairs, pairs, sizeof(int2) * n_pairs, hipMemcpyHostToDevice, st));
  if (!items.empty())
    SFM_HIP_TRY(hipMemcpyAsync(pl->d_items, items.data(), sizeof(WorkItem) * items.size(), hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemsetAsync(pl->d_counts, 0, sizeof(int) * pl->cap_pairs, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));  // items / pairs are host temporaries
  pl->timed = false;
  return SFMHIP_OK;
}

template <int KS, int MODE, int SR>
static int launch_knn(sfmhip_matchplan* pl) {
  sfmhip_imageset* s = pl->set;
  constexpr int LDS = 2 * (SR * 32 * KS + SR * 4);
  static bool attr_set = false;
  if (!attr_set) {
    SFM_HIP_TRY(hipFuncSetAttribute((const void*)knn_mfma_kernel<KS, MODE, SR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL((knn_mfma_kernel<KS, MODE, SR>), dim3(pl->n_items), dim3(256), LDS, s->ctx->stream, s->d_imgs,
                     pl->d_pairs, pl->d_items, s->d_nonintegral, s->gen, pl->d_knn, pl->maxq, pl->d_fix_count, pl->d_fix_items);
  SFM_HIP_TRY(hipGetLastError());
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_run_async(sfmhip_matchplan* pl, float ratio) {
  if (!pl) return SFMHIP_ERR_ARG;
  sfmhip_imageset* s = pl->set;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  const bool timing = s->ctx->timing;
  if (timing) SFM_HIP_TRY(hipEventRecord(pl->ev[0], st));
  if (pl->n_pairs > 0) {
    const bool mfma = s->ks != 0;
    if (mfma && pl->n_items > 0) {
      if (s->kind == KIND_U8_HAMMING) {
        SFM_TRY((launch_knn<8, 1, 128>(pl)));
      } else {
        switch (s->ks) {
          case 1: SFM_TRY((launch_knn<1, 0, 256>(pl))); break;
          case 2: SFM_TRY((launch_knn<2, 0, 256>(pl))); break;
          case 4: SFM_TRY((launch_knn<4, 0, 256>(pl))); break;
          case 8: SFM_TRY((launch_knn<8, 0, 128>(pl))); break;
        }
      }
    }
    const int force_all = mfma ? 0 : 1;
    const int grid = std::min(std::max(pl->n_pairs, 256), 2048);
    if (s->kind == KIND_F32_L2)
      hipLaunchKernelGGL((knn_exact_kernel<KIND_F32_L2>), dim3(grid), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->n_pairs,
                         s->d_nonintegral, s->gen, force_all, s->dim, pl->d_knn, pl->maxq, pl->d_fix_count, pl->d_fix_items);
    else if (s->kind == KIND_U8_L2)
      hipLaunchKernelGGL((knn_exact_kernel<KIND_U8_L2>), dim3(grid), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->n_pairs,
                         s->d_nonintegral, s->gen, force_all, s->dim, pl->d_knn, pl->maxq, pl->d_fix_count, pl->d_fix_items);
    else
      hipLaunchKernelGGL((knn_exact_kernel<KIND_U8_HAMMING>), dim3(grid), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->n_pairs,
                         s->d_nonintegral, s->gen, force_all, s->dim, pl->d_knn, pl->maxq, pl->d_fix_count, pl->d_fix_items);
    SFM_HIP_TRY(hipGetLastError());
  }
  if (timing) SFM_HIP_TRY(hipEventRecord(pl->ev[1], st));
  if (pl->n_pairs > 0) {
    hipLaunchKernelGGL(compact_kernel, dim3(pl->n_pairs), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->d_knn, pl->maxq,
                       ratio, pl->d_counts, pl->d_out_q, pl->d_out_t, pl->d_out_d, pl->d_fix_count);
    SFM_HIP_TRY(hipGetLastError());
  }
  if (timing) SFM_HIP_TRY(hipEventRecord(pl->ev[2], st));
  pl->timed = timing;
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_fetch(sfmhip_matchplan* pl, int32_t* counts, int32_t* out_q, int32_t* out_t,
                                      float* out_dist, int64_t capacity, int64_t* total) {
  if (!pl || !counts) return SFMHIP_ERR_ARG;
  sfmhip_imageset* s = pl->set;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  if (pl->n_pairs) SFM_HIP_TRY(hipMemcpyAsync(counts, pl->d_counts, sizeof(int) * pl->n_pairs, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  int64_t tot = 0;
  for (int p = 0; p < pl->n_pairs; ++p) tot += counts[p];
  if (total) *total = tot;
  if (!out_q && !out_t && !out_dist) return SFMHIP_OK;
  if (tot > capacity) return SFMHIP_ERR_ARG;
  int64_t off = 0;
  for (int p = 0; p < pl->n_pairs; ++p) {
    const size_t n = (size_t)counts[p], src = (size_t)p * pl->maxq;
    if (n) {
      if (out_q) SFM_HIP_TRY(hipMemcpyAsync(out_q + off, pl->d_out_q + src, n * 4, hipMemcpyDeviceToHost, st));
      if (out_t) SFM_HIP_TRY(hipMemcpyAsync(out_t + off, pl->d_out_t + src, n * 4, hipMemcpyDeviceToHost, st));
      if (out_dist) SFM_HIP_TRY(hipMemcpyAsync(out_dist + off, pl->d_out_d + src, n * 4, hipMemcpyDeviceToHost, st));
    }
    off += (int64_t)n;
  }
  SFM_HIP_TRY(hipStreamSynchronize(st));
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_fetch_knn(sfmhip_matchplan* pl, int pair, int32_t* idx, float* dist) {
  if (!pl || pair < 0 || pair >= pl->n_pairs || !idx || !dist) return SFMHIP_ERR_ARG;
  sfmhip_imageset* s = pl->set;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  const int nq = s->n_rows[pl->h_pairs[2 * pair]];
  const int nt = s->n_rows[pl->h_pairs[2 * pair + 1]];
  std::vector<int4> tmp((size_t)std::max(nq, 1));
  if (nq) {
    SFM_HIP_TRY(hipMemcpyAsync(tmp.data(), pl->d_knn + (size_t)pair * pl->maxq, sizeof(int4) * nq, hipMemcpyDeviceToHost, s->ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
  }
  for (int q = 0; q < nq; ++q) {
    int4 e = tmp[q];
    if (nt < 1) {  // the knn kernels are not run for such pairs
      e.x = e.y = -1;
      const float fm = 3.402823466e+38f;
      memcpy(&e.z, &fm, 4);
      memcpy(&e.w, &fm, 4);
    }
    idx[2 * q] = e.x;
    idx[2 * q + 1] = e.y;
    memcpy(&dist[2 * q], &e.z, 4);
    memcpy(&dist[2 * q + 1], &e.w, 4);
  }
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_last_timing(sfmhip_matchplan* pl, double seconds[3]) {
  if (!pl || !seconds) return SFMHIP_ERR_ARG;
  seconds[0] = seconds[1] = seconds[2] = 0;
  float ms = 0;
  if (pl->set->prep_timed) {
    SFM_HIP_TRY(hipEventSynchronize(pl->set->ev_prep1));
    SFM_HIP_TRY(hipEventElapsedTime(&ms, pl->set->ev_prep0, pl->set->ev_prep1));
    seconds[0] = ms * 1e-3;
  }
  if (pl->timed) {
    SFM_HIP_TRY(hipEventSynchronize(pl->ev[2]));
    SFM_HIP_TRY(hipEventElapsedTime(&ms, pl->ev[0], pl->ev[1]));
    seconds[1] = ms * 1e-3;
    SFM_HIP_TRY(hipEventElapsedTime(&ms, pl->ev[1], pl->ev[2]));
    seconds[2] = ms * 1e-3;
  }
  return SFMHIP_OK;
}

extern "C" void sfmhip_matchplan_destroy(sfmhip_matchplan* pl) {
  if (!pl) return;
  hipSetDevice(pl->set->ctx->device);
  hipFree(pl->d_pairs);
  hipFree(pl->d_items);
  hipFree(pl->d_knn);
  hipFree(pl->d_counts);
  hipFree(pl->d_out_q);
  hipFree(pl->d_out_t);
  hipFree(pl->d_out_d);
  hipFree(pl->d_fix_count);
  hipFree(pl->d_fix_items);
  for (auto& e : pl->ev)
    if (e) hipEventDestroy(e);
  delete pl;
}

// getMatching, one pair, host buffers (reference src/Sfm.cpp:590-608)
extern "C" int sfmhip_match_knn2(sfmhip_ctx* ctx, const void* q, int nq, const void* t, int nt, int dim, int dtype,
                                 int norm, float ratio, int32_t* out_q, int32_t* out_t, float* out_dist, int32_t* out_n) {
  if (!ctx || nq < 0 || nt < 0 || dim <= 0 || !out_n) return SFMHIP_ERR_ARG;
  *out_n = 0;
  if (nq == 0 || nt < 2) return SFMHIP_OK;  // knn[i][1] does not exist: nothing can pass
  if (!q || !t) return SFMHIP_ERR_ARG;
  const int32_t rows[2] = {nq, nt};
  sfmhip_imageset* s = nullptr;
  sfmhip_matchplan* pl = nullptr;
  int rc = sfmhip_imageset_create(ctx, 2, rows, dim, dtype, norm, &s);
  if (rc == SFMHIP_OK) rc = sfmhip_imageset_upload(s, 0, q);
  if (rc == SFMHIP_OK) rc = sfmhip_imageset_upload(s, 1, t);
  if (rc == SFMHIP_OK) rc = sfmhip_imageset_prepare_async(s);
  const int32_t pr[2] = {0, 1};
  if (rc == SFMHIP_OK) rc = sfmhip_matchplan_create(s, pr, 1, &pl);
  if (rc == SFMHIP_OK) rc = sfmhip_matchplan_run_async(pl, ratio);
  int32_t cnt = 0;
  int64_t tot = 0;
  if (rc == SFMHIP_OK) rc = sfmhip_matchplan_fetch(pl, &cnt, out_q, out_t, out_dist, nq, &tot);
  if (rc == SFMHIP_OK) *out_n = cnt;
  sfmhip_matchplan_destroy(pl);
  sfmhip_imageset_destroy(s);
  return rc;
}
