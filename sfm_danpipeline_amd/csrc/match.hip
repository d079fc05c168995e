// match.hip -- brute-force k=2 descriptor matching + ratio test on gfx950 (MI355X).
//
// Replaces cv::BFMatcher(NORM_L2,false).knnMatch(q,t,k=2) + the ratio filter of
// StructFromMotion::getMatching (reference src/Sfm.cpp:590-608), for one pair and for the
// batched all-pairs loop of findBestPair (reference src/Sfm.cpp:511-515).
//
// Design (DESIGN.md section "K1/K2"):
//  * prepare pass: every image becomes an i8 "tile image" (32-row tiles already laid out in the
//    XOR-swizzled order the LDS wants, so staging is a linear copy).  SIFT rows (integers 0..255
//    held in f32) are centred to x-128; binary rows are expanded to +-1.  L2 images are stored
//    in PARITY ORDER: rows whose centred squared norm is odd first, then (from a tile boundary)
//    the even ones, original order inside each class; perm[] maps a position back to the row.
//  * k-NN kernel: 32x32x32 i8 MFMA, trains on the M side (rows -> accumulator registers),
//    queries on the N side (one query per lane).  With c = ceil(||t||^2/2) fed through the MFMA's
//    C operand the accumulator IS h = q.t - c, and d = ||q||^2 - 2h - (||t||^2 odd): larger h is a
//    smaller distance, equal h means the odd-norm row is closer by one -- which is why odd rows
//    come first: the order cv::batchDistance inserts by, (distance, lower train index), becomes
//    (h descending, position ascending) and no per-distance key has to be built.
//    Per lane the epilogue keeps VALUES only, in two structures that between them separate any
//    two rows: the maximum of every accumulator slot over all train tiles (one v_max3_i32 per two
//    distances) and the top-2 of the per-tile maxima, tagged with their tile (a v_max3_i32 tree):
//    19 VALU ops per 32x32 tile instead of 48.  The best two rows of a lane lie in
//    (top-2 tiles) x (slots whose maximum reaches the second-best value); that handful of
//    candidates is recomputed exactly at the end of the sweep (v_dot4_i32_i8) and ranked.
//  * exact kernel: f32 rows that are not integer-valued are redone by a VALU kernel that orders
//    by (sqrtf(s), index) like OpenCV does; so are, inside the compaction kernel, the rare
//    queries whose 2nd-best squared distance is >= 2^22 (where sqrtf can merge neighbouring
//    integers) -- they are flagged in-band, so there is no list that could overflow.
//  * compaction kernel: ratio test (float multiply + compare) and order-preserving compaction.
#include "common.h"
#include <type_traits>
#include <vector>
#include <algorithm>
#include <string.h>
#include <stdlib.h>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

enum { KIND_F32_L2 = 0, KIND_U8_L2 = 1, KIND_U8_HAMMING = 2 };

constexpr int TILE_ROWS = 32;
constexpr int HPAD = -(1 << 23);     // h of padding rows and of empty structures; real rows: h > -2^23
constexpr int TAG_BITS = 8;          // tile tag in the low bits of a tile-maximum key
constexpr int EPOCH_TILES = 1 << TAG_BITS;  // train tiles per epoch (8192 rows): structures are resolved per epoch
constexpr int TAG_MASK = EPOCH_TILES - 1;
constexpr int FIX_FLAG = 1 << 30;    // in knn[].y: redo this query exactly (sqrtf merge range)
constexpr int FIX_GRID = 1024;       // workgroups of knn_fixup_kernel
constexpr int FIX_CAP = 1 << 20;     // flagged queries that are also LISTED, for a one-wave-per-query fix-up kernel;
                                     // beyond it they stay flagged in-band and the compaction kernel redoes them
constexpr int DIST_EMPTY = 0x7FFFFFFF;
constexpr int HCHUNK = 1024;  // Hamming kernel: rows per tie-break chunk (10 index bits in the key: the query's bytes are scaled to +-64)
#ifndef SFM_RESOLVE_V2
#define SFM_RESOLVE_V2 1  // 0: the lanes with two slots at the threshold recompute all four rows (the form of rounds 2-5; A/B builds)
#endif
#ifndef SFM_DBG
#define SFM_DBG 0  // diagnostic builds (scripts/build_match_variants.py): 1 no candidate loop, 2 no epilogue, 3 no MFMA, 4 stamps, 5 no exact redo
#endif
#if SFM_DBG == 4
// per workgroup (first 4096) and wave: s_memtime at 6 points + candidate-loop trips (sfmhip_dbg_read_stamps)
__device__ unsigned long long g_stamps[4096 * 8 * 16];
#define SFM_STAMP(i) do { if (blockIdx.x < 4096 && lane == 0) g_stamps[(blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SFM_STAMP(i) do { } while (0)
#endif

// Pointers read out of the ImgDev records are generic to the compiler (flat_load, which also
// ties up lgkmcnt next to the LDS reads); these casts tell it they are global memory.
#define SFM_GLOBAL __attribute__((address_space(1)))
typedef const SFM_GLOBAL v4i* g_v4i_p;
typedef const SFM_GLOBAL unsigned* g_u32_p;
typedef const SFM_GLOBAL int* g_i32_p;

struct ImgDev {
  const void* raw;  // descriptor rows as handed over (f32 or u8), HBM
  int8_t* tiles;    // [n_pad/32][32][KS*32] swizzled i8, rows in position order
  int* cin;         // [n_pad] MFMA C input of the row in its train role: -ceil(||t||^2/2), HPAD for padding
  int* nq;          // [n_pad] squared norm of the centred row
  int* perm;        // [n_pad] position -> original row (-1: padding)
  int* par;         // [n_pad] scratch of the prepare pass: parity of the row's norm, by original row
  int* nodd;        // [1] positions of the odd class (a multiple of 32), written by the order pass
  int n_rows;
  int n_pad;
};

struct WorkItem {
  int pair;
  int qtile0;
  int qimg, timg;  // the pair's images (saves the k-NN kernel one dependent load)
};

// physical 16-byte chunk of logical chunk c of row r inside a 32-row tile with NC chunks/row:
// ds_read_b128 then hits 16 distinct 16-byte slots per 16-lane group (MI355X_MICROARCH LDS).
template <int NC>
__host__ __device__ constexpr int chunk_pos(int r, int c) {
  constexpr int SH = (NC == 16) ? 0 : (NC == 8) ? 1 : (NC == 4) ? 2 : 3;
  return r * NC + (c ^ ((r >> SH) & (NC - 1)));
}
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }       // v_max3_i32
__device__ __forceinline__ int imed3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }  // v_med3_i32

// ---------------------------------------------------------------- prepare
// Pass 1 (L2 kinds): parity of every row's centred squared norm = parity of the sum of its
// elements, by original row.  32 lanes per row.
template <int KIND>
__global__ __launch_bounds__(256) void parity_kernel(const ImgDev* __restrict__ imgs, const int* __restrict__ tile_img,
                                                     const int* __restrict__ tile_first, int dim) {
  const int img = tile_img[blockIdx.x];
  const int tile = blockIdx.x - tile_first[img];
  const ImgDev I = imgs[img];
  const int l = threadIdx.x & 31;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = tile * TILE_ROWS + rr * 8 + (threadIdx.x >> 5);
    int s = 0;
    if (row < I.n_rows) {
      if (KIND == KIND_F32_L2) {
        const float* src = (const float*)I.raw + (size_t)row * dim;
        for (int k = l; k < dim; k += 32) {
          const float f = src[k];
          int x = (int)f;
          x = x < 0 ? 0 : (x > 255 ? 255 : x);
          s ^= x;
        }
      } else {
        const unsigned char* src = (const unsigned char*)I.raw + (size_t)row * dim;
        for (int k = l; k < dim; k += 32) s ^= src[k];
      }
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) s ^= __shfl_xor(s, o);
    if (l == 0 && row < I.n_pad) I.par[row] = (row < I.n_rows) ? (s & 1) : 0;
  }
}

// Pass 2: positions.  One workgroup per image: odd rows first (original order), padded to a
// tile boundary, then the even rows.  by_parity == 0 (Hamming): identity.
__global__ __launch_bounds__(256) void order_kernel(const ImgDev* __restrict__ imgs, int by_parity) {
  __shared__ int wsum[4];
  __shared__ int run_odd, run_even, n_odd_pad_s;
  const ImgDev I = imgs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = threadIdx.x; p < I.n_pad; p += 256) I.perm[p] = (!by_parity && p < I.n_rows) ? p : -1;
  if (!by_parity) {
    if (threadIdx.x == 0) *I.nodd = 0;
    return;
  }
  // count the odd rows
  int c = 0;
  for (int r = threadIdx.x; r < I.n_rows; r += 256) c += I.par[r];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) c += __shfl_xor(c, o);
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n_odd = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    n_odd_pad_s = (n_odd + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    *I.nodd = n_odd_pad_s;
    run_odd = 0;
    run_even = 0;
  }
  __syncthreads();
  const int n_odd_pad = n_odd_pad_s;
  for (int r0 = 0; r0 < I.n_rows; r0 += 256) {
    const int r = r0 + threadIdx.x;
    const bool valid = r < I.n_rows;
    const bool odd = valid && I.par[r];
    const unsigned long long bo = __ballot(odd), bv = __ballot(valid);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) wsum[wave] = __popcll(bo) | (__popcll(bv) << 16);
    __syncthreads();
    int odd_before = run_odd, valid_before = 0;
    for (int w = 0; w < wave; ++w) {
      odd_before += wsum[w] & 0xFFFF;
      valid_before += wsum[w] >> 16;
    }
    const int my_odd = odd_before + __popcll(bo & below);
    const int my_valid = valid_before + __popcll(bv & below);
    if (valid) {
      const int even_before = run_even + (my_valid - (my_odd - run_odd));
      I.perm[odd ? my_odd : n_odd_pad + even_before] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int so = 0, sv = 0;
      for (int w = 0; w < 4; ++w) {
        so += wsum[w] & 0xFFFF;
        sv += wsum[w] >> 16;
      }
      run_odd += so;
      run_even += sv - so;
    }
    __syncthreads();
  }
}

// Pass 3: tile image, C inputs and norms, in position order.
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
  const int pos = tile * TILE_ROWS + r;
  const int row = I.perm[pos];
  const bool valid = row >= 0;
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
      v[e] = (valid && k < nbits) ? (((bits >> e) & 1u) ? 8 : -8) : 0;  // (+-8: the products of the keyed kernel are +-64)
    }
  } else {
    // the lane's 16 elements: four 16-byte loads (f32) / one (u8) when the chunk lies inside the
    // row and is 16-byte aligned -- the case of every real descriptor width; element loads otherwise
    float fv[16];
    unsigned char bv[16];
    const bool full = valid && c * 16 + 16 <= dim;
    if (KIND == KIND_F32_L2) {
      const float* src = (const float*)I.raw + (size_t)(valid ? row : 0) * dim + c * 16;
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
      const unsigned char* src = (const unsigned char*)I.raw + (size_t)(valid ? row : 0) * dim + c * 16;
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
    int ci;
    if (KIND == KIND_U8_HAMMING)  // key base of knn_keyed_kernel: key = base - 512 q.t = 1024 hamming + (row mod 1024)
      ci = ((valid ? dim * 8 * (HCHUNK / 2) : (dim * 8 + 1) * HCHUNK)) | (pos & (HCHUNK - 1));
    else ci = valid ? -((n2 + 1) >> 1) : HPAD;              // h = q.t - ceil(||t||^2 / 2)
    I.cin[pos] = ci;
    I.nq[pos] = valid ? n2 : 0;
  }
  if (KIND == KIND_F32_L2 && !ok) nonintegral[img] = gen;  // (stamped with the prepare pass: no clearing between passes)
}

// sum over groups of NC consecutive lanes (NC = 2, 4, 8, 16), left in every lane of the group: DPP only
template <int NC>
__device__ __forceinline__ int group_sum(int x) {
  if (NC >= 2) x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
  if (NC >= 4) x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
  if (NC >= 8) x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);     // row_half_mirror
  if (NC >= 16) x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);    // row_mirror
  return x;
}

// ---------------------------------------------------------------- MFMA k-NN kernel
struct Best2 {  // exact (squared distance, position; original train row) of the best two; DIST_EMPTY = none
  int d0, i0, x0, d1, i1, x1;
};
__device__ __forceinline__ bool lex_less(int da, int ia, int db, int ib) { return da < db || (da == db && ia < ib); }
__device__ __forceinline__ void best2_insert(Best2& b, int d, int i, int x) {
  if (lex_less(d, i, b.d0, b.i0)) {
    b.d1 = b.d0;
    b.i1 = b.i0;
    b.x1 = b.x0;
    b.d0 = d;
    b.i0 = i;
    b.x0 = x;
  } else if (lex_less(d, i, b.d1, b.i1)) {
    b.d1 = d;
    b.i1 = i;
    b.x1 = x;
  }
}

// epilogue micro-ops of one drained tile pair (accumulators D0 = tile 2p, D1 = tile 2p+1 of one
// query tile): 16 slot maxima, then per tile a 7-op max3 tree, the tagged key and its insertion
// into the lane's top-2 of tile maxima.  38 ops, executed in index order.
template <int I>
__device__ __forceinline__ void epi_op(const v16i& D0, const v16i& D1, int (&sl)[16], int& k0, int& k1, int (&T)[16],
                                       int tag0, int tag1) {
  if constexpr (I < 16) {
    sl[I] = imax3(sl[I], D0[I], D1[I]);
  } else {
    constexpr int J = (I - 16) % 11;
    constexpr int B = (I - 16) < 11 ? 0 : 8;  // each tree has its own temporaries
    const v16i& D = (I - 16) < 11 ? D0 : D1;
    const int tag = (I - 16) < 11 ? tag0 : tag1;
    if constexpr (J == 0) T[B + 0] = imax3(D[0], D[1], D[2]);
    if constexpr (J == 1) T[B + 1] = imax3(D[3], D[4], D[5]);
    if constexpr (J == 2) T[B + 2] = imax3(D[6], D[7], D[8]);
    if constexpr (J == 3) T[B + 3] = imax3(D[9], D[10], D[11]);
    if constexpr (J == 4) T[B + 4] = imax3(D[12], D[13], D[14]);
    if constexpr (J == 5) T[B + 5] = imax3(T[B + 0], T[B + 1], T[B + 2]);
    if constexpr (J == 6) T[B + 6] = imax3(T[B + 3], T[B + 4], D[15]);
    if constexpr (J == 7) T[B + 7] = max(T[B + 5], T[B + 6]);
    if constexpr (J == 8) T[B + 7] = (T[B + 7] << TAG_BITS) | tag;
    if constexpr (J == 9) k1 = imed3(k0, k1, T[B + 7]);
    if constexpr (J == 10) k0 = max(k0, T[B + 7]);
  }
}
// Issue order of the 38 ops: the tree of D0 first (D0 was finished four MFMAs before D1: nothing waits for
// the matrix pipe), the dependent tails of both trees spread between the independent slot maxima.
// ids: 0..15 slot maxima, 16..26 tree/key/insert of D0, 27..37 of D1.
constexpr int EPI_ORDER[38] = {16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 0, 1,  2,
                                          32, 33, 3,  4,  5,  34, 6,  7,  8,  9,  35, 10, 11, 12, 13, 36, 37, 14, 15};
template <int LO, int HI>
__device__ __forceinline__ void epi_range(const v16i& D0, const v16i& D1, int (&sl)[16], int& k0, int& k1, int (&T)[16],
                                          int tag0, int tag1) {
  if constexpr (LO < HI && LO < 38) {
    epi_op<EPI_ORDER[LO]>(D0, D1, sl, k0, k1, T, tag0, tag1);
    epi_range<LO + 1, HI>(D0, D1, sl, k0, k1, T, tag0, tag1);
  }
}

// MODE 0: L2 on centred i8 rows, MODE 1: Hamming on +-1 rows.  SR = train rows per LDS stage,
// NU = query tiles per wave.
//
// Workgroup = 8 waves = two per SIMD; wave w owns query tiles qtile0 + NU*w .. (B fragments stay in
// registers for the whole kernel) and sweeps every train tile of the pair's train image.  Train
// tiles are staged through LDS by LDS-DMA (two stage buffers, one barrier per stage).  The sweep
// goes by CHAINS: a chain is the MFMAs of one query tile against one train tile PAIR (2*KS
// MFMAs into two accumulator blocks); between its MFMAs ride the 38 epilogue ops of the
// previously filled accumulator blocks, so a SIMD's matrix and vector pipes run side by side
// (4.75 VALU ops per MFMA: scripts/ubench/epi_mix.hip, the gap stays MFMA-paced).
template <int KS, int MODE, int NU, int SR, int NW>
__global__ __launch_bounds__(NW * 64, 2) void knn_kernel(const ImgDev* __restrict__ imgs,
                                                         const WorkItem* __restrict__ items,
                                                         const int* __restrict__ nonintegral, int gen, int dim,
                                                         int4* __restrict__ knn, int maxq, int late_start,
                                                         int* __restrict__ fix_count, int2* __restrict__ fix_items) {
  constexpr int NC = 2 * KS;
  constexpr int RB = 32 * KS;
  constexpr int TILE_BYTES = TILE_ROWS * RB;
  constexpr int STAGE_ROW_BYTES = SR * RB;
  constexpr int NT = NW * 64;                            // threads
  constexpr int CIN_COPIES = (SR + NT - 1) / NT;           // every thread copies CIN_COPIES C inputs: no divergent copy code
  constexpr int STAGE_BYTES = STAGE_ROW_BYTES + CIN_COPIES * NT * 4;
  constexpr int TILES = SR / TILE_ROWS;
  constexpr int TP = TILES / 2;                       // tile pairs per stage
  constexpr int PIECES = STAGE_ROW_BYTES / (NT * 16);   // 16-byte pieces per thread per stage
  constexpr int BATCH = KS <= 4 ? 4 : 2;                // candidate rows per lane resolved per round
  constexpr int NG = 2 * KS;                          // MFMAs (= op groups) per chain
  constexpr int STAGES_PER_EPOCH = EPOCH_TILES / TILES;
  static_assert(PIECES >= 1 && TP >= 2 && TP % 2 == 0 && STAGES_PER_EPOCH % 2 == 0, "stage shape");
  // Two separate LDS objects, not one array cut in two: only then can the compiler tell that the
  // LDS-DMA filling one buffer does not alias the ds_reads of the other, and leave out the
  // s_waitcnt vmcnt(0) it otherwise puts in front of every ds_read that follows an LDS-DMA
  // (which would make the stage copy synchronous).  The stage loop is unrolled by two for it.
  // (resolve scratch per wave and query tile: 2 KB of lists + a copy of the query tile; tile u of a wave in buffer u)
  // (... whose 16 x 64 slot maxima pass through the same bytes first: 4 KB at least)
  constexpr int SCRATCH_W = 2048 + (TILE_ROWS * RB > 4096 ? TILE_ROWS * RB : 4096);
  constexpr int BUFA_BYTES = STAGE_BYTES > NW * SCRATCH_W ? STAGE_BYTES : NW * SCRATCH_W;
  constexpr int BUFB_BYTES = (NU < 2 || STAGE_BYTES > NW * SCRATCH_W) ? STAGE_BYTES : NW * SCRATCH_W;
  __shared__ __attribute__((aligned(16))) unsigned char ldsA[BUFA_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char ldsB[BUFB_BYTES];

  const WorkItem it = items[blockIdx.x];
  if ((nonintegral[it.qimg] == gen) | (nonintegral[it.timg] == gen)) return;  // left to the exact kernel
  const ImgDev Q = imgs[it.qimg];
  const ImgDev T = imgs[it.timg];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: LDS-DMA targets and query tiles stay in SGPRs
  const int r = lane & 31, h = lane >> 5;
  const int nqt = Q.n_pad / TILE_ROWS;
  int qt[NU], qtc[NU];  // query tiles of the wave; clamped to the image for the loads
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    qt[u] = it.qtile0 + NU * wave + u;
    qtc[u] = qt[u] < Q.n_pad / TILE_ROWS ? qt[u] : (Q.n_pad >= TILE_ROWS ? Q.n_pad / TILE_ROWS - 1 : 0);
  }
  SFM_STAMP(0);
  int dbg_trips = 0;
  // Two workgroups share a CU (NW = 4).  Launched together they would stay in step -- both in the
  // MFMA-bound sweep, then both in the memory-bound resolve.  The second set of resident workgroups
  // starts late_start * 8 k cycles late, and every later workgroup inherits the phase of the one it replaces.
  if (late_start > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < late_start; ++i) __builtin_amdgcn_s_sleep(127);

  // query fragments (B operand): lane (r,h) holds bytes [32ks+16h, +16) of query row r
  v4i bq[NU][KS];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    g_v4i_p src = (g_v4i_p)(Q.tiles + (size_t)qtc[u] * TILE_BYTES);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[u][ks] = src[chunk_pos<NC>(r, 2 * ks + h)];
  }

  // per-lane LDS byte offsets of the A fragments inside a tile, and of the C inputs
  int aoff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) aoff[ks] = chunk_pos<NC>(r, 2 * ks + h) * 16;
  const int coff = STAGE_ROW_BYTES + h * 16;

  const int nstages = T.n_pad / SR;
  const int tid = threadIdx.x;

  // Stage copy: the tile image is already in LDS order, so a stage is a linear copy; LDS-DMA
  // (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, no ds_write) moves it.
  // (buffer_load ... lds, not global_load_lds: the compiler books the latter as a FLAT access that may touch
  // LDS, and while one is pending it turns every counted LDS wait into s_waitcnt lgkmcnt(0) -- each MFMA
  // that needs a fragment would then wait for the youngest ds_read in flight, not for its own)
  const __amdgpu_buffer_rsrc_t rs_tiles = __builtin_amdgcn_make_buffer_rsrc((void*)T.tiles, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cin = __builtin_amdgcn_make_buffer_rsrc((void*)T.cin, 0, 0x7FFFFFFF, 0x00020000);
  auto stage_copy = [&](int stage, unsigned char* dstb) {
    const int off = stage * STAGE_ROW_BYTES;  // (a tile image stays below 2 GB: n_pad * RB)
#if SFM_DBG == 6
    const unsigned char* src = (const unsigned char*)T.tiles + (size_t)stage * STAGE_ROW_BYTES;
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      __builtin_amdgcn_global_load_lds((const SFM_GLOBAL void*)(src + (size_t)(i * NT + tid) * 16),
                                       (__attribute__((address_space(3))) void*)(dstb + i * (NT * 16) + wave * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < CIN_COPIES; ++i)
      __builtin_amdgcn_global_load_lds((const SFM_GLOBAL void*)(T.cin + (size_t)stage * SR + i * NT + tid),
                                       (__attribute__((address_space(3))) void*)(dstb + STAGE_ROW_BYTES + i * (NT * 4) + wave * 256), 4, 0, 0);
    (void)off;
#else
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_tiles, (__attribute__((address_space(3))) void*)(dstb + i * (NT * 16) + wave * 1024),
                                               16, (i * NT + tid) * 16, off, 0, 0);
    // SR C inputs; threads beyond SR copy those of the next stage (the array is padded by 512)
#pragma unroll
    for (int i = 0; i < CIN_COPIES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cin, (__attribute__((address_space(3))) void*)(dstb + STAGE_ROW_BYTES + i * (NT * 4) + wave * 256),
                                               4, (i * NT + tid) * 4, stage * SR * 4, 0, 0);
#endif
  };
  auto ld_afrag = [&](const unsigned char* tb, int ks) -> v4i {  // tb: the tile inside a stage buffer
    const int4 x = *(const int4*)(tb + aoff[ks]);
    return v4i{x.x, x.y, x.z, x.w};
  };
  // C input of the lane's 16 accumulator rows of tile tl: rows 8g + 4h + j
  auto ld_cin = [&](const unsigned char* sb, int tl, v16i& C) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int4 x = *(const int4*)(sb + coff + (tl * TILE_ROWS + 8 * g) * 4);
      C[4 * g] = x.x;
      C[4 * g + 1] = x.y;
      C[4 * g + 2] = x.z;
      C[4 * g + 3] = x.w;
    }
  };

  // (No per-lane state is kept across the sweep -- it would be spilled: the sweep uses the whole register
  // budget.  An epoch's result goes to the k-NN buffer; a later epoch merges with what is there.)
  const int nodd_t = MODE == 0 ? *(g_i32_p)T.nodd : 0;
  const int nbits = dim * 8;

  for (int ep0 = 0; ep0 < nstages; ep0 += STAGES_PER_EPOCH) {
    const int ep1 = (ep0 + STAGES_PER_EPOCH < nstages) ? ep0 + STAGES_PER_EPOCH : nstages;
    const int ep_tile0 = ep0 * TILES;
    int sl[NU][16], k0[NU], k1[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      k0[u] = k1[u] = (int)0x80000000;
#pragma unroll
      for (int e = 0; e < 16; ++e) sl[u][e] = HPAD;
    }
    if (ep0 > 0) __syncthreads();  // every wave is done with the stage buffers of the previous epoch
    stage_copy(ep0, ldsA);         // (epochs start at even stages: STAGES_PER_EPOCH is even)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    v4i F0[KS], F1[KS];
    v16i C0, C1, X0, X1, Y0, Y1;
    int TT[16];
    {
      const unsigned char* sb = ldsA;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) F0[ks] = ld_afrag(sb, ks);
      ld_cin(sb, 0, C0);  // (F1 / C1 of a pair are fetched by the pair's first chain)
#pragma unroll
      for (int e = 0; e < 16; ++e) X0[e] = X1[e] = Y0[e] = Y1[e] = HPAD;  // draining these changes nothing
    }
    // tag of the first tile of the pair drained next; the first drain is the dummy one: its blocks hold HPAD,
    // so whatever (masked) tag it gets, its keys carry the value HPAD and never become candidates
    int tag = EPOCH_TILES - 1 + 2;
    SFM_STAMP(1);

    // One chain: MFMAs of query tile UF against (F0, F1) into (FA, FB); the epilogue of (DA, DB)
    // (query tile UD, tile tags tag, tag-1) spread between them.  The LDS reads come in two batches, each
    // KS MFMAs ahead of its first use (the compiler waits for ALL outstanding LDS reads before an MFMA that
    // needs any of them, so a read issued just before such a wait would cost its whole latency):
    //   LD1: at the chain's first group, F1 / C1 of THIS pair (tile T1 of buffer SB1), used from group KS on;
    //   LD0: in groups 0..KS-1, each F0 fragment right after its last use (and C0 after the first MFMA), of the
    //        NEXT pair (tile T0 of SB0): used by the next chain, KS groups later at the earliest.
#define SFM_CHAIN(FA, FB, UF, DA, DB, UD, LD1, SB1, T1, LD0, SB0, T0)                                       \
  do {                                                                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                     \
      const int ks_ = g_ % KS;                                                                              \
      if (SFM_DBG == 3) {                                                                                   \
        if (g_ < KS) FA[ks_] += F0[ks_][0] + C0[ks_];                                                       \
        else FB[ks_] += F1[ks_][0] + C1[ks_];                                                               \
      } else if (g_ < KS) FA = __builtin_amdgcn_mfma_i32_32x32x32_i8(F0[ks_], bq[UF][ks_], ks_ == 0 ? C0 : FA, 0, 0, 0); \
      else FB = __builtin_amdgcn_mfma_i32_32x32x32_i8(F1[ks_], bq[UF][ks_], ks_ == 0 ? C1 : FB, 0, 0, 0);  \
      if ((LD1) && g_ == 0) {                                                                               \
        _Pragma("unroll") for (int k2_ = 0; k2_ < KS; ++k2_) F1[k2_] = ld_afrag((SB1) + (T1) * TILE_BYTES, k2_); \
        ld_cin(SB1, T1, C1);                                                                                \
      }                                                                                                     \
      if ((LD0) && g_ < KS) {                                                                               \
        F0[ks_] = ld_afrag((SB0) + (T0) * TILE_BYTES, ks_);                                                 \
        if (g_ == 0) ld_cin(SB0, T0, C0);                                                                   \
      }                                                                                                     \
      if (SFM_DBG != 2) switch (g_) {                                                                       \
        SFM_EPI_CASES(DA, DB, UD)                                                                           \
      }                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
  } while (0)
#define SFM_EPI_CASE(G, DA, DB, UD) \
  case G: epi_range<(G) * 38 / NG, ((G) + 1) * 38 / NG>(DA, DB, sl[UD], k0[UD], k1[UD], TT, tag & TAG_MASK, (tag - 1) & TAG_MASK); break;
#define SFM_EPI_CASES(DA, DB, UD)                                                                          \
  SFM_EPI_CASE(0, DA, DB, UD) SFM_EPI_CASE(1, DA, DB, UD) SFM_EPI_CASE(2, DA, DB, UD) SFM_EPI_CASE(3, DA, DB, UD)     \
  SFM_EPI_CASE(4, DA, DB, UD) SFM_EPI_CASE(5, DA, DB, UD) SFM_EPI_CASE(6, DA, DB, UD) SFM_EPI_CASE(7, DA, DB, UD)     \
  SFM_EPI_CASE(8, DA, DB, UD) SFM_EPI_CASE(9, DA, DB, UD) SFM_EPI_CASE(10, DA, DB, UD) SFM_EPI_CASE(11, DA, DB, UD)   \
  SFM_EPI_CASE(12, DA, DB, UD) SFM_EPI_CASE(13, DA, DB, UD) SFM_EPI_CASE(14, DA, DB, UD) SFM_EPI_CASE(15, DA, DB, UD)

    auto stage = [&](int s, const unsigned char* sb, unsigned char* nb) __attribute__((always_inline)) {
      const bool more = s + 1 < ep1;
      // the other buffer was last read before the previous stage's barrier: refill it now
      if (more) stage_copy(s + 1, nb);
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const bool last = tp == TP - 1;
        if (NU == 2) {
          // query tile 0 against the pair (drains the previous pair of query tile 1); fetches the pair's second tile
          SFM_CHAIN(X0, X1, 0, Y0, Y1, 1, true, sb, 2 * tp + 1, false, sb, 0);
          tag -= 2;
          if (last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the next stage has landed
            __syncthreads();
          }
          // query tile 1 against the pair (drains query tile 0 of it); fetches the next pair's first tile
          SFM_CHAIN(Y0, Y1, NU - 1, X0, X1, 0, false, sb, 0, true, last ? nb : sb, last ? 0 : 2 * tp + 2);
        } else {
          if (last) {
            // (the pair's second tile before the barrier: past it nothing may read this stage's buffer)
#pragma unroll
            for (int k2 = 0; k2 < KS; ++k2) F1[k2] = ld_afrag(sb + (2 * tp + 1) * TILE_BYTES, k2);
            ld_cin(sb, 2 * tp + 1, C1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
          }
          if ((tp & 1) == 0) SFM_CHAIN(X0, X1, 0, Y0, Y1, 0, !last, sb, 2 * tp + 1, true, last ? nb : sb, last ? 0 : 2 * tp + 2);
          else SFM_CHAIN(Y0, Y1, 0, X0, X1, 0, !last, sb, 2 * tp + 1, true, last ? nb : sb, last ? 0 : 2 * tp + 2);
          tag -= 2;
        }
      }
    
    };
    for (int s = ep0; s < ep1; s += 2) {
      stage(s, ldsA, ldsB);
      if (s + 1 < ep1) stage(s + 1, ldsB, ldsA);
    }
    SFM_STAMP(2);
    // the blocks filled last are still to be drained (Y for both shapes: TP is even)
    epi_range<0, 38>(Y0, Y1, sl[NU - 1], k0[NU - 1], k1[NU - 1], TT, tag & TAG_MASK, (tag - 1) & TAG_MASK);
#undef SFM_CHAIN
#undef SFM_EPI_CASES
#undef SFM_EPI_CASE

    // ---- resolve the epoch: exact h of the lane's candidate rows, then (d, row) per query.
    // Written as phases over the wave's NU query tiles, branch-free where it can be, so that their
    // shuffles, LDS round trips and global loads are in flight together: this part runs once per
    // sweep on two waves per SIMD and is bound by instruction count and latency, not by throughput.
    // Per-wave scratch in the stage buffers (query tile u in buffer u): past the last stage's barrier
    // nothing reads them any more (the reloads of the last chain fetch fragments that are never used).
    int lane_p = lane;
    asm volatile("" : "+v"(lane_p));  // (what follows is computed after the sweep, not kept in registers or spilled across it)
    const int r_p = lane_p & 31, h_p = lane_p >> 5;
    typedef __attribute__((address_space(3))) int* lds_vi_p;
    // (one wave's LDS operations execute in order; the fences keep the compiler from moving them across a phase boundary)
#define SFM_WAVE_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    // ONE packed list of the rows to recompute for the wave's NU query tiles (NU * 256 entries) and their values: the list in
    // buffer 0's scratch, the values in buffer NU - 1's (NU == 1: behind the list)
    __attribute__((address_space(3))) unsigned char* wq[NU];  // the wave's query tile, in the tile image's layout
#pragma unroll
    for (int u = 0; u < NU; ++u)
      wq[u] = (__attribute__((address_space(3))) unsigned char*)(u == 0 ? ldsA : ldsB) + wave * SCRATCH_W + 2048;
    const lds_vi_p wl = (lds_vi_p)(wq[0] - 2048);
    const lds_vi_p wr = NU == 1 ? wl + 256 : (lds_vi_p)(wq[NU - 1] - 2048);
    // (the query rows are in registers, half a row per lane: through LDS every lane can reach any of them,
    // and the recompute below reads half as many bytes from the vector cache)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        *(__attribute__((address_space(3))) v4i*)(wq[u] + chunk_pos<NC>(r_p, 2 * ks + h_p) * 16) = bq[u][ks];
    constexpr int SH = (NC == 16) ? 0 : (NC == 8) ? 1 : (NC == 4) ? 2 : 3;  // (the swizzle of chunk_pos)
    int cpos[NU][4], chv[NU][4], slot_[NU][4], count[NU];
    bool cval[NU][4], cload[NU][4];
    int qrow[NU], nqq[NU];  // original row and norm of the lane's query (in flight during the phases)
    int ovf[NU];            // the query goes to the exact kernel: too many equal candidates
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = qtc[u] * TILE_ROWS + r_p;
      qrow[u] = ((g_i32_p)Q.perm)[q];
      nqq[u] = ((g_i32_p)Q.nq)[q];
      if (qt[u] >= nqt) qrow[u] = -1;  // (a query tile beyond the image: the wave swept a copy of the last one)
    }
    // phase A: thresholds, the slots that reach them, the candidate rows
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int bv0 = k0[u] >> TAG_BITS, bv1 = k1[u] >> TAG_BITS;  // the two largest tile maxima of the lane
      const int tl0 = ep_tile0 + (EPOCH_TILES - 1) - (k0[u] & TAG_MASK);
      const int tl1 = ep_tile0 + (EPOCH_TILES - 1) - (k1[u] & TAG_MASK);
      const int pb0 = __shfl_xor(bv0, 32), pb1 = __shfl_xor(bv1, 32);
      // The three largest slot maxima of the lane WITH their slots: keys (maximum << 4 | slot), a running top-3 (3 ops per
      // slot).  They give, without a mask over the sixteen slots and without indexing registers per lane, how many slots reach
      // the threshold (more than two only matters as "more than two"), which slots those are, and their maxima S0 >= S1.
      int a0 = (HPAD - 1) * 16, a1 = a0, a2 = a0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (int)((unsigned)sl[u][e] << 4) | e;  // (v_lshl_or_b32; |h| < 2^23)
        a2 = imed3(a1, a2, key);
        a1 = imed3(a0, a1, key);
        a0 = max(a0, key);
      }
      const int S0 = a0 >> 4, S1 = a1 >> 4, S2 = a2 >> 4;
      const int e0 = a0 & 15, e1 = a1 & 15;
      // thr: the second largest of the four tile maxima of the query's two lanes (four different rows: a lower bound of the
      // query's second-best h) and the lane's second largest slot maximum (another row's value as well: when the best two rows
      // share a tile, the tile maxima hide the second).  Rows below it are out.
      const int thr = max(max(min(bv0, pb0), max(bv1, pb1)), S1);
      const int ns = (S0 >= thr ? 1 : 0) + (S1 >= thr ? 1 : 0) + (S2 >= thr ? 1 : 0);  // slots that reach thr (3: three or more)
      // (a tile maximum of HPAD is a padding tile, the dummy drain or an empty slot: no candidates there)
      const bool t0in = bv0 >= thr && bv0 > HPAD, t1in = bv1 >= thr && bv1 > HPAD;
      // The rows >= thr of the lane lie in (tiles tl0, tl1) x (slots that reach thr).  Two slots are
      // resolved; more (only equal values do that) sends the query to the exact kernel.
      const int rho0 = 8 * (e0 >> 2) + 4 * h_p + (e0 & 3), rho1 = ns >= 2 ? 8 * (e1 >> 2) + 4 * h_p + (e1 & 3) : rho0;
      int over = (ns > 2 && (t0in || t1in)) ? 1 : 0;
      over |= __shfl_xor(over, 32);
      ovf[u] = over;
      // One slot reaches thr: every row >= thr of the lane sits in it, so a tile whose maximum reaches
      // thr has that maximum in this slot -- row and value are known.  (tl0 is the first tile that
      // reaches the lane's maximum, tl1 the first other tile that reaches bv1.)
      const bool known = ns == 1;
      cpos[u][0] = tl0 * TILE_ROWS + rho0;
      cpos[u][1] = tl0 * TILE_ROWS + rho1;
      cpos[u][2] = tl1 * TILE_ROWS + rho0;
      cpos[u][3] = tl1 * TILE_ROWS + rho1;
      cval[u][0] = t0in && ns >= 1;
      cval[u][1] = t0in && ns >= 2;
      cval[u][2] = t1in && ns >= 1;
      cval[u][3] = t1in && ns >= 2;
      chv[u][0] = known ? bv0 : HPAD - 1;
      chv[u][1] = HPAD - 1;
      chv[u][2] = known ? bv1 : HPAD - 1;
      chv[u][3] = HPAD - 1;
      bool need[4] = {!known, !known, !known, !known};
      // Two slots reach thr, with DIFFERENT maxima S0 > S1 (slots e0, e1).  The lane's maximum bv0 = S0 is row (tl0, e0):
      // tile tl0 holds bv0 in a slot whose maximum reaches it, and only e0's does.  The lane's second row then has the value
      // max(S1, bv1), and where it sits follows from the four maxima, except when S1 == bv1 =: v (the usual case -- the second
      // row is its slot's AND its tile's maximum):
      //   S1 > bv1: no other tile reaches S1, so e1's maximum sits in tile tl0: row (tl0, e1), value S1;
      //   S1 < bv1: tile tl1's maximum exceeds e1's, so it sits in slot e0: row (tl1, e0), value bv1;
      //   S1 == v:  the rows that can hold v before any other row does are (tl0, e1) -- when tile tl0 comes before tl1 -- and, in
      //             tile tl1, slots e0 and e1, of which at least one holds v (the tile's maximum sits in a slot that reaches
      //             it): the EARLIER of the two rows is recomputed, and the later one is entered WITH the value v -- if the
      //             earlier holds v it precedes the later in the ranking whatever that really holds, and if it does not, the
      //             later does.  (tl0, e1) is recomputed when tl0 < tl1, and cannot be the second row otherwise.
      // One or two rows are read where the four of (tl0, tl1) x (e0, e1) used to be (22 rows per query tile at cfg2 instead of
      // 60); equal slot maxima keep the four.
#if SFM_RESOLVE_V2
      if (ns == 2 && S0 != S1 && t0in) {
        const bool gt = S1 > bv1, eq = S1 == bv1, lt = S1 < bv1;
        const bool e1_in = gt || (eq && tl0 < tl1);  // row (tl0, e1) can be the lane's second row
        const bool first0 = rho0 < rho1;             // of tile tl1's two rows, slot e0's comes first
        cval[u][0] = true, chv[u][0] = bv0, need[0] = false;
        cval[u][1] = e1_in, chv[u][1] = S1, need[1] = e1_in && !gt;
        cval[u][2] = eq || lt, chv[u][2] = bv1, need[2] = eq && first0;
        cval[u][3] = eq, chv[u][3] = bv1, need[3] = eq && !first0;
      }
#endif
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cval[u][k] = cval[u][k] && cpos[u][k] < T.n_pad;
        cload[u][k] = cval[u][k] && need[k] && SFM_DBG != 1;
      }
    }
    SFM_STAMP(8);
    // phase B: pack the rows to recompute into the wave's list
    int total = 0;  // wave-uniform
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned long long bm = __ballot(cload[u][k]);
        slot_[u][k] = total + __popcll(bm & ((1ull << lane_p) - 1ull));
        // entry: position (26 bits: a tile image stays below 2 GB) | the query's row in its tile << 26 | query tile << 31
        if (cload[u][k]) wl[slot_[u][k]] = cpos[u][k] | (r_p << 26) | (u << 31);
        total += __popcll(bm);
      }
      count[u] = total;
    }
    SFM_WAVE_LDS_FENCE();
    SFM_STAMP(9);
#if SFM_DBG == 4
    if (blockIdx.x < 4096 && lane == 0) {
      g_stamps[(blockIdx.x * 8 + wave) * 16 + 13] = count[0];
      g_stamps[(blockIdx.x * 8 + wave) * 16 + 7] = total - count[0];
    }
#endif
    // phase C: the listed rows are recomputed LPR lanes to a row, two 16-byte chunks per lane (a row's lanes
    // read one cache line together: the texture addresser, not the arithmetic, bounds this phase).  Train
    // row chunk p (physical order) holds logical chunk p ^ key(train row), which the query's row keeps
    // at physical chunk p ^ key(train) ^ key(query).
    {
      constexpr int LPR = NC >= 2 ? NC / 2 : 1;  // lanes per row
      constexpr int CPL = NC / LPR;              // chunks per lane
      constexpr int RPS = 64 / LPR;              // rows per step
      constexpr int STEPS = 64 / RPS;            // steps that cover 64 rows
      const int grp = lane_p / LPR, c = lane_p % LPR;
      // a trip of NS steps (16 rows of the list per step at KS = 4): every load of the trip is issued before its first use
      auto trip = [&](int j0, auto ns_) __attribute__((always_inline)) {
        constexpr int NS = decltype(ns_)::value;
        v4i xt[NS][CPL], xq[NS][CPL];
        int ci[NS];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int j = j0 + st * RPS + grp;
          unsigned ent = (unsigned)wl[j & (NU * 256 - 1)];  // position | query row << 26 | query tile << 31
          if (j >= total) ent = 0u;                          // (row 0 of the image: fetched, not used)
          const unsigned ps = ent & ((1u << 26) - 1u), qr = (ent >> 26) & 31u;
          const unsigned x = ((ps >> SH) ^ (qr >> SH)) & (unsigned)(NC - 1);
          const unsigned char* trow = (const unsigned char*)T.tiles + ps * (unsigned)RB;
          const __attribute__((address_space(3))) unsigned char* qrw = ((NU > 1 && (ent >> 31)) ? wq[NU - 1] : wq[0]) + qr * (unsigned)RB;
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            const unsigned p_ = (unsigned)(c + i * LPR);
            xt[st][i] = *(g_v4i_p)(trow + p_ * 16);
            xq[st][i] = *(const __attribute__((address_space(3))) v4i*)(qrw + ((p_ ^ x) << 4));
          }
          ci[st] = ((g_i32_p)T.cin)[ps];
        }
        SFM_STAMP(11);
#if SFM_DBG == 4
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SFM_STAMP(12);
#endif
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          int dot = 0;
#pragma unroll
          for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) dot = __builtin_amdgcn_sdot4(xt[st][i][w], xq[st][i][w], dot, false);
          dot = group_sum<LPR>(dot) + ci[st];
          wr[(j0 + st * RPS + grp) & (NU * 256 - 1)] = dot;  // (every lane of the group writes the same value)
        }
      };
      // (the last trip takes the steps the list still needs -- a wave-uniform choice between unrolled bodies; a guard per step
      // inside one body turns the fragments into loop-carried values the compiler keeps alive across the SWEEP: scratch)
      for (int j0 = 0; j0 < total; j0 += RPS * STEPS) {
        ++dbg_trips;
        const int left = total - j0;
        if (STEPS >= 4 && left <= RPS * (STEPS / 4)) trip(j0, std::integral_constant<int, (STEPS >= 4 ? STEPS / 4 : 1)>{});
        else if (STEPS >= 2 && left <= RPS * (STEPS / 2)) trip(j0, std::integral_constant<int, (STEPS >= 2 ? STEPS / 2 : 1)>{});
        else if (STEPS >= 4 && left <= RPS * (3 * STEPS / 4)) trip(j0, std::integral_constant<int, (STEPS >= 4 ? 3 * STEPS / 4 : 1)>{});
        else trip(j0, std::integral_constant<int, STEPS>{});
      }
    }
    SFM_WAVE_LDS_FENCE();
    SFM_STAMP(10);
    // phase D: lane top-2 by (h descending, position ascending) as 64-bit keys; phase E: distances, merge of
    // the query's two lanes by (d, position) -- the same order as (d, original row): equal d means equal
    // parity class, and positions keep the original order inside a class -- then the rows' original indices
    // (NU == 2: the half-wave h_p finalises query tile h_p -- every lane ends up with ONE query to merge, look up and
    // emit, so that part of the code runs once per wave instead of once per query tile; the resolve's instructions compete
    // with the SIMD partner's sweep for issue slots, and this half was a fifth of them.  NU == 1: lanes h_p == 0 emit.)
    long long lane_dk[NU][2];  // the lane's best two of query tile u: (d << 32 | position), ascending; 0x7FFF...: none
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      long long key[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (cload[u][k]) chv[u][k] = wr[slot_[u][k]];
        const int hv = cval[u][k] ? chv[u][k] : HPAD - 1;
        key[k] = ((long long)hv << 32) | (unsigned)(0x7FFFFFFF - cpos[u][k]);  // larger = better
      }
      const long long a = max(key[0], key[1]), b = min(key[0], key[1]);
      const long long c = max(key[2], key[3]), dd = min(key[2], key[3]);
      const long long lk[2] = {max(a, c), max(min(a, c), max(b, dd))};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int hv = (int)(lk[k] >> 32), ps = 0x7FFFFFFF - (int)(unsigned)lk[k];
        const int d = MODE == 0 ? nqq[u] - 2 * hv - (ps < nodd_t ? 1 : 0) : (nbits - hv) >> 1;
        lane_dk[u][k] = hv > HPAD ? (((long long)d << 32) | (unsigned)ps) : 0x7FFFFFFFFFFFFFFFll;  // smaller = better
      }
    }
    // the query this lane finalises: tile UM = h_p (NU == 2) or tile 0; its other lane's pair comes by one exchange
    constexpr bool SPLIT = NU == 2;
    const bool um = SPLIT && h_p != 0;
    long long mine[2], give[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      mine[k] = um ? lane_dk[NU - 1][k] : lane_dk[0][k];
      give[k] = SPLIT ? (um ? lane_dk[0][k] : lane_dk[NU - 1][k]) : lane_dk[0][k];  // what the partner lane merges with ITS pair
    }
    const long long o0 = __shfl_xor(give[0], 32), o1 = __shfl_xor(give[1], 32);
    const long long best0 = min(mine[0], o0), best1 = min(max(mine[0], o0), min(mine[1], o1));
    const int my_qrow = um ? qrow[NU - 1] : qrow[0];
    int my_ovf = um ? ovf[NU - 1] : ovf[0];
    const unsigned p0 = (unsigned)best0, p1 = (unsigned)best1;
    const int px0 = ((g_i32_p)T.perm)[p0 < (unsigned)T.n_pad ? p0 : 0u];
    const int px1 = ((g_i32_p)T.perm)[p1 < (unsigned)T.n_pad ? p1 : 0u];
    {
      Best2 m;
      m.d0 = (int)(best0 >> 32), m.i0 = (int)(unsigned)best0, m.x0 = px0;
      m.d1 = (int)(best1 >> 32), m.i1 = (int)(unsigned)best1, m.x1 = px1;
      if (best0 == 0x7FFFFFFFFFFFFFFFll || m.x0 < 0) m.d0 = DIST_EMPTY, m.i0 = 0x7FFFFFFF, m.x0 = -1;  // (a padding position is no row)
      if (best1 == 0x7FFFFFFFFFFFFFFFll || m.x1 < 0) m.d1 = DIST_EMPTY, m.i1 = 0x7FFFFFFF, m.x1 = -1;
      // emit.  Between epochs the entry holds {row0, row1, d0, d1} as integers (an earlier epoch's rows have the
      // lower indices: on equal d they stay); the last epoch turns it into the final {row0, row1|flag, dist bits}.
      if ((SPLIT || h_p == 0) && my_qrow >= 0) {
        int4* dst = &knn[(size_t)it.pair * maxq + my_qrow];
        if (ep0 > 0) {
          const int4 pv = *dst;
          Best2 a;
          a.d0 = pv.z, a.x0 = pv.x, a.i0 = -2, a.d1 = pv.w, a.x1 = pv.y, a.i1 = -1;  // (positions below any of this epoch)
          if (pv.y == FIX_FLAG) my_ovf = 1;
          if (m.d0 != DIST_EMPTY) best2_insert(a, m.d0, m.i0, m.x0);
          if (m.d1 != DIST_EMPTY) best2_insert(a, m.d1, m.i1, m.x1);
          m = a;
        }
        int4 o;
        if (ep1 < nstages) {
          o.x = m.x0;
          o.y = my_ovf ? FIX_FLAG : m.x1;
          o.z = m.d0;
          o.w = m.d1;
        } else {
          const bool v0 = m.d0 != DIST_EMPTY, v1 = m.d1 != DIST_EMPTY;
          float d0, d1;
          int fix = 0;
          if (MODE == 0) {
            d0 = sqrtf((float)m.d0);
            d1 = sqrtf((float)m.d1);
            if (v1 && m.d1 >= (1 << 22)) fix = FIX_FLAG;  // sqrtf may merge neighbouring integers: redone exactly
          } else {
            d0 = (float)m.d0;
            d1 = (float)m.d1;
          }
          o.x = v0 ? m.x0 : -1;
          o.y = v1 ? (m.x1 | fix) : -1;
          if (my_ovf) o.y = FIX_FLAG;  // (any non-negative flagged value: the compaction kernel recomputes the query)
          o.z = __float_as_int(v0 ? d0 : 3.402823466e+38f);
          o.w = __float_as_int(v1 ? d1 : 3.402823466e+38f);
          if (o.y >= 0 && (o.y & FIX_FLAG)) {
            const int k = atomicAdd(fix_count, 1);
            if (k < FIX_CAP) fix_items[k] = make_int2(it.pair, my_qrow);
          }
        }
        *dst = o;
      }
    }
    SFM_STAMP(3);
    SFM_STAMP(4);
  }

  SFM_STAMP(5);
#if SFM_DBG == 4
  if (blockIdx.x < 4096 && lane == 0) {
    g_stamps[(blockIdx.x * 8 + wave) * 16 + 14] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    g_stamps[(blockIdx.x * 8 + wave) * 16 + 15] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  }
  if (blockIdx.x < 4096 && lane == 0) g_stamps[(blockIdx.x * 8 + wave) * 16 + 6] = dbg_trips;
#endif
}

// ---------------------------------------------------------------- MFMA k-NN kernel, Hamming
// Binary rows are stored as +-8 bytes.  With the query fragments negated AND scaled to -+64 the i8 MFMA
// yields C - 512 q.t (q, t in units of +-1), and with C = 512 nbits + (row mod 1024) that IS the key
// (1024 hamming + row-in-chunk): minima of unsigned keys order by (distance, lower train index) =
// cv::batchDistance's insertion rule, at two VALU ops per distance (v_med3_u32 + v_min_u32) and nothing
// to build.  Integer distances tie all the time, so the value-only structures of the L2 kernel are no
// use here.  (Until round 6 the query kept its +-8: seven index bits, a chunk merge every 128 rows --
// a sixth of the sweep's vector instructions; ten bits make it every 1024.)
constexpr int HIB = 10;
static_assert((1 << HIB) == HCHUNK, "the key's index bits");
constexpr unsigned KEY_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
struct Top2 {
  unsigned s0, s1;  // key >> HIB of best / 2nd best (0xFFFFFFFF = empty)
  int j0, j1;       // their train rows
};

__device__ __forceinline__ void chunk_merge(Top2& g, unsigned k0, unsigned k1, int cb) {
  // candidates of a later chunk: on equal distance the earlier (already held) row wins
  const unsigned ns0 = k0 >> HIB, ns1 = k1 >> HIB;
  const int nj0 = cb + (int)(k0 & (HCHUNK - 1)), nj1 = cb + (int)(k1 & (HCHUNK - 1));
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

__device__ __forceinline__ bool lex_less_u(unsigned sa, int ja, unsigned sb, int jb) {
  return sa < sb || (sa == sb && ja < jb);
}

// Workgroup = 4 waves; wave w owns query tiles qtile0+2w, +2w+1 (64 queries, B fragments stay in
// registers for the whole kernel) and sweeps every train tile of the pair's train image.  Train
// tiles are staged through LDS (two stage buffers: two LDS objects, so that the LDS-DMA of one does
// not make the compiler wait before the ds_reads of the other), 16 accumulator registers = 16
// train rows of one query per lane.  "Units" (train tile x query tile) are software-pipelined: the
// MFMA chain of unit n+1 is issued between the top-2 insertions of unit n.
template <int KS, int SR>
__global__ __launch_bounds__(256, 2) void knn_keyed_kernel(const ImgDev* __restrict__ imgs,
                                                           const WorkItem* __restrict__ items, int dim,
                                                           int4* __restrict__ knn, int maxq) {
  constexpr int MODE = 1;
  constexpr int NC = 2 * KS;
  constexpr int RB = 32 * KS;
  constexpr int STAGE_ROW_BYTES = SR * RB;
  constexpr int STAGE_BYTES = STAGE_ROW_BYTES + SR * 4;
  constexpr int TILES = SR / TILE_ROWS;
  constexpr int PIECES = STAGE_ROW_BYTES / 4096;  // 16-byte pieces per thread per stage
  constexpr int HALF = PIECES > 1 ? PIECES / 2 : 1;
  __shared__ __attribute__((aligned(16))) unsigned char ldsA[STAGE_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char ldsB[STAGE_BYTES];

  const WorkItem it = items[blockIdx.x];
  const ImgDev Q = imgs[it.qimg];
  const ImgDev T = imgs[it.timg];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nqt = (Q.n_rows + TILE_ROWS - 1) / TILE_ROWS;
  const int qt[2] = {it.qtile0 + 2 * wave, it.qtile0 + 2 * wave + 1};

  // query fragments (B operand): lane (r,h) holds bytes [32ks+16h, +16) of query row r
  v4i bq[2][KS];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = qt[u] < nqt ? qt[u] : (nqt > 0 ? nqt - 1 : 0);
    g_v4i_p src = (g_v4i_p)(Q.tiles + (size_t)t * (TILE_ROWS * RB));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      v4i x = src[chunk_pos<NC>(r, 2 * ks + h)];
      // bytes are +8, -8 or 0 (padding): negate them in place (0x08 * 30 = 0xF0 flips 0x08 <-> 0xF8), then times 8 -- 0x08 ->
      // 0x40, 0xF8 -> 0xC0; what a byte shifts into its neighbour is masked off
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned neg = (unsigned)x[w] ^ (((unsigned)x[w] & 0x08080808u) * 30u);
        x[w] = (int)((neg << 3) & 0xC0C0C0C0u);
      }
      bq[u][ks] = x;
    }
  }

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
  // (buffer_load ... lds: see knn_kernel)
  const __amdgpu_buffer_rsrc_t rs_tiles = __builtin_amdgcn_make_buffer_rsrc((void*)T.tiles, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cin = __builtin_amdgcn_make_buffer_rsrc((void*)T.cin, 0, 0x7FFFFFFF, 0x00020000);
  auto stage_copy = [&](int stage, unsigned char* dstb) {
    const int off = stage * STAGE_ROW_BYTES;
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_tiles, (__attribute__((address_space(3))) void*)(dstb + i * 4096 + wave * 1024), 16,
                                               (i * 256 + tid) * 16, off, 0, 0);
    if (wave < SR / 64)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cin, (__attribute__((address_space(3))) void*)(dstb + STAGE_ROW_BYTES + wave * 256), 4,
                                               tid * 4, stage * SR * 4, 0, 0);
  };

  stage_copy(0, ldsA);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // one step of the MFMA chain of (fragments fr, query tile u): ks-th K slice
#define SFM_MFMA(ACC, FR, u, ks, BS) \
  ACC = __builtin_amdgcn_mfma_i32_32x32x32_i8(FR[ks], bq[u][ks], (ks) == 0 ? BS : ACC, 0, 0, 0)
  // top-2 insertion of accumulator element e of query tile u
#define SFM_INS(ACC, u, e)                                                             \
  do {                                                                                     \
    const unsigned key_ = (unsigned)ACC[e];                 /* the accumulator is the key */ \
    const unsigned n1_ = umax_(umin_(k0[u], k1[u]), umin_(umax_(k0[u], k1[u]), key_)); /* v_med3_u32 */ \
    k0[u] = umin_(k0[u], key_);                                                            \
    k1[u] = n1_;                                                                           \
    asm volatile("" : "+v"(k0[u]), "+v"(k1[u])); /* no min/max re-association across elements */ \
  } while (0)
  auto ld_afrag = [&](const unsigned char* sb, int tl, int ks) -> v4i {
    const int4 x = *(const int4*)(sb + tl * (TILE_ROWS * RB) + aoff[ks]);
    return v4i{x.x, x.y, x.z, x.w};
  };
  auto ld_bases = [&](const unsigned char* sb, int tl, int gq, v16i& bs) {
    const int4 x = *(const int4*)(sb + boff + (tl * TILE_ROWS + 8 * gq) * 4);
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
          if ((g_) < 4) SFM_MFMA(NA0, FN, 0, ks_, BN);                                         \
          else SFM_MFMA(NA1, FN, 1, ks_, BN);                                                  \
        }                                                                                      \
      }                                                                                        \
      if ((g_) >= 4) {                                                                         \
        if (do_frag) ld_bases(SB, (tl) + 2, (g_) - 4, BC); /* C inputs of the chain after next */ \
      }                                                                                        \
      else if (do_frag) {                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < MPG; ++j_)                                     \
          if ((g_) * MPG + j_ < KS) FNN[(g_) * MPG + j_] = ld_afrag(SB, (tl) + 2, (g_) * MPG + j_); \
      }                                                                                        \
    }                                                                                          \
    SFM_INS(CA0, 0, 2 * (g_));                                                                 \
    SFM_INS(CA1, 1, 2 * (g_));                                                                 \
    SFM_INS(CA0, 0, 2 * (g_) + 1);                                                             \
    SFM_INS(CA1, 1, 2 * (g_) + 1);                                                             \
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
  v16i ba, bb;  // key bases = C inputs of the chains (ba: the tile being inserted was started from it; bb: the next)
  v16i A0, A1, B0, B1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fa[ks] = ld_afrag(ldsA, 0, ks);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) ld_bases(ldsA, 0, gq, ba);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) ld_bases(ldsA, 1, gq, bb);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) SFM_MFMA(A0, fa, 0, ks, ba);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) SFM_MFMA(A1, fa, 1, ks, ba);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fb[ks] = ld_afrag(ldsA, 1, ks);
  __builtin_amdgcn_sched_barrier(0);

  auto stage = [&](int s, const unsigned char* sb, unsigned char* nb) __attribute__((always_inline)) {
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
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) ld_bases(nb, 0, gq, ba);  // C inputs of the next stage's first chain
    }
    SFM_BODY(nb, -1, B0, B1, A0, A1, fa, fb, bb, ba, more, more);
    // tie-break chunk boundary: fold the keys (distance, row in chunk) into the running (distance, row)
    if (((s + 1) * SR) % HCHUNK == 0) {
      const int cb = ((s * SR) / HCHUNK) * HCHUNK;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        chunk_merge(g[u], k0[u], k1[u], cb);
        k0[u] = k1[u] = KEY_EMPTY;
      }
    }
  };
  for (int s = 0; s < nstages; s += 2) {
    stage(s, ldsA, ldsB);
    if (s + 1 < nstages) stage(s + 1, ldsB, ldsA);
  }
  if ((nstages * SR) % HCHUNK != 0) {  // the last, partial chunk
    const int cb = ((nstages * SR) / HCHUNK) * HCHUNK;
#pragma unroll
    for (int u = 0; u < 2; ++u) chunk_merge(g[u], k0[u], k1[u], cb);
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
    if (lex_less_u(p.s0, p.j0, g[u].s0, g[u].j0)) {
      m.s0 = p.s0;
      m.j0 = p.j0;
      if (lex_less_u(g[u].s0, g[u].j0, p.s1, p.j1)) {
        m.s1 = g[u].s0;
        m.j1 = g[u].j0;
      } else {
        m.s1 = p.s1;
        m.j1 = p.j1;
      }
    } else {
      m.s0 = g[u].s0;
      m.j0 = g[u].j0;
      if (lex_less_u(p.s0, p.j0, g[u].s1, g[u].j1)) {
        m.s1 = p.s0;
        m.j1 = p.j0;
      } else {
        m.s1 = g[u].s1;
        m.j1 = g[u].j1;
      }
    }
    const int q = qt[u] * TILE_ROWS + r;
    if (h == 0 && qt[u] < nqt && q < Q.n_rows) {
      const int s0i = (int)m.s0, s1i = (int)m.s1;  // hamming distances
      const bool v0 = m.j0 >= 0 && m.j0 < T.n_rows, v1 = m.j1 >= 0 && m.j1 < T.n_rows;
      const float d0 = (float)s0i, d1 = (float)s1i;
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
__device__ int4 exact_query(const ImgDev& Q, const ImgDev& T, int q, int dim) {
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
  int4 o;  // the butterfly leaves the same result in every lane
  o.x = j0;
  o.y = j1;
  o.z = __float_as_int(j0 >= 0 ? d0 : 3.402823466e+38f);
  o.w = __float_as_int(j1 >= 0 ? d1 : 3.402823466e+38f);
  return o;
}

// Every pair that touches a non-integer-valued f32 image (or, with force_all, every pair:
// norms/dims the MFMA kernel is not instantiated for).  One wave per query.
template <int KIND>
__global__ __launch_bounds__(256) void knn_exact_kernel(const ImgDev* __restrict__ imgs,
                                                        const int2* __restrict__ pairs, int n_pairs,
                                                        const int* __restrict__ nonintegral, int gen,
                                                        int force_all, int dim,
                                                        int4* __restrict__ knn, int maxq) {
  const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, lane = threadIdx.x & 63;
  for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
    const int2 pr = pairs[p];
    if (!(force_all | (nonintegral[pr.x] == gen) | (nonintegral[pr.y] == gen))) continue;
    const ImgDev Q = imgs[pr.x], T = imgs[pr.y];
    for (int q = wave; q < Q.n_rows; q += wpb) {
      const int4 o = exact_query<KIND>(Q, T, q, dim);
      if (lane == 0) knn[(size_t)p * maxq + q] = o;
    }
  }
}

// ---------------------------------------------------------------- ratio test + compaction
// One query by a whole 256-thread block, for the L2 kinds: eight lanes share a train row -- lane l takes
// the elements k = l (mod 8), in ascending k: exactly the eight interleaved partial sums of exact_dist, then
// combined in its fixed order -- so a wave does eight rows per round from coalesced 32-byte segments, and the
// four waves split the rows.  Same arithmetic, same bits as exact_query; ~8x less time per query.
template <int KIND>
__device__ int4 exact_query_block(const ImgDev& Q, const ImgDev& T, int q, int dim, int row0, int row1, int4* sh /* [4] */) {
  static_assert(KIND == KIND_F32_L2 || KIND == KIND_U8_L2, "L2 kinds");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = lane & 7, grp = lane >> 3;
  const size_t rowb = (size_t)dim * (KIND == KIND_F32_L2 ? 4 : 1);
  const unsigned char* qrow = (const unsigned char*)Q.raw + (size_t)q * rowb;
  float d0 = 3.402823466e+38f, d1 = 3.402823466e+38f;
  int j0 = -1, j1 = -1;
  constexpr int RU = 4;  // rows per group and round: four independent chains, their loads in flight together
  for (int jb = row0 + wave * 8 * RU; jb < row1; jb += 32 * RU) {
    int jr[RU];
    const unsigned char* trow[RU];
#pragma unroll
    for (int a = 0; a < RU; ++a) {
      jr[a] = jb + a * 8 + grp;  // (ascending within the group: round by round, a by a)
      trow[a] = (const unsigned char*)T.raw + (size_t)(jr[a] < row1 ? jr[a] : 0) * rowb;
    }
    float d[RU];
    if (KIND == KIND_F32_L2) {
      float acc[RU];
#pragma unroll
      for (int a = 0; a < RU; ++a) acc[a] = 0.f;
      for (int k = l; k < dim; k += 8) {
        const float qv = ((const float*)qrow)[k];
#pragma unroll
        for (int a = 0; a < RU; ++a) {
          const float x = __fsub_rn(qv, ((const float*)trow[a])[k]);
          acc[a] = __fadd_rn(acc[a], __fmul_rn(x, x));
        }
      }
      // ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)): float addition commutes, so both lanes of a pair get the same bits
#pragma unroll
      for (int a = 0; a < RU; ++a) {
        acc[a] = __fadd_rn(acc[a], __shfl_xor(acc[a], 4));
        acc[a] = __fadd_rn(acc[a], __shfl_xor(acc[a], 2));
        acc[a] = __fadd_rn(acc[a], __shfl_xor(acc[a], 1));
        d[a] = sqrtf(acc[a]);
      }
    } else {
      int si[RU];
#pragma unroll
      for (int a = 0; a < RU; ++a) si[a] = 0;
      for (int k = l; k < dim; k += 8) {
        const int qv = (int)qrow[k];
#pragma unroll
        for (int a = 0; a < RU; ++a) {
          const int x = qv - (int)trow[a][k];
          si[a] += x * x;
        }
      }
#pragma unroll
      for (int a = 0; a < RU; ++a) {
        si[a] += __shfl_xor(si[a], 4);
        si[a] += __shfl_xor(si[a], 2);
        si[a] += __shfl_xor(si[a], 1);
        d[a] = sqrtf((float)si[a]);
      }
    }
#pragma unroll
    for (int a = 0; a < RU; ++a) {
      if (jr[a] < row1 && (d[a] < d1 || j1 < 0)) {  // (a group meets its rows in ascending order)
        if (d[a] < d0 || j0 < 0) {
          d1 = d0;
          j1 = j0;
          d0 = d[a];
          j0 = jr[a];
        } else {
          d1 = d[a];
          j1 = jr[a];
        }
      }
    }
  }
  auto less = [](float da, int ja, float db, int jb) {
    if (ja < 0) return false;
    if (jb < 0) return true;
    return da < db || (da == db && ja < jb);
  };
  auto merge = [&](float e0, int i0, float e1, int i1) {
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
  };
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) merge(__shfl_xor(d0, o), __shfl_xor(j0, o), __shfl_xor(d1, o), __shfl_xor(j1, o));
  __syncthreads();  // (sh may still be read from the previous query)
  if (lane == 0) sh[wave] = make_int4(j0, j1, __float_as_int(d0), __float_as_int(d1));
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w != wave) merge(__int_as_float(sh[w].z), sh[w].x, __int_as_float(sh[w].w), sh[w].y);
  int4 o;
  o.x = j0;
  o.y = j1;
  o.z = __float_as_int(j0 >= 0 ? d0 : 3.402823466e+38f);
  o.w = __float_as_int(j1 >= 0 ? d1 : 3.402823466e+38f);
  return o;
}

// The listed flagged queries.  With few of them (the usual case) up to eight workgroups share a query --
// slices of the train rows, partial results in `part`, the workgroup that arrives last merges -- so that a
// handful of flagged queries does not leave the chip waiting for a handful of workgroups.
template <int KIND>
__global__ __launch_bounds__(256) void knn_fixup_kernel(const ImgDev* __restrict__ imgs, const int2* __restrict__ pairs,
                                                        int dim, int4* __restrict__ knn, int maxq,
                                                        const int* __restrict__ fix_count,
                                                        const int2* __restrict__ fix_items, int4* __restrict__ part,
                                                        int* __restrict__ arrived) {
  __shared__ int4 sh[4];
  __shared__ int last_s;
  int nfix = *fix_count;
  nfix = nfix < FIX_CAP ? nfix : FIX_CAP;
  if (nfix == 0) return;
  int S = (int)gridDim.x / nfix;  // workgroups per query (nfix * S <= gridDim.x: the size of part / arrived)
  S = S < 1 ? 1 : (S > 8 ? 8 : S);
  for (int unit = blockIdx.x; unit < nfix * S; unit += gridDim.x) {
    const int i = unit / S, c = unit % S;
    const int2 f = fix_items[i];
    const int2 pr = pairs[f.x];
    const ImgDev T = imgs[pr.y];
    const int chunk = ((T.n_rows + S - 1) / S + 127) / 128 * 128;
    const int row0 = c * chunk, row1 = (c + 1) * chunk < T.n_rows ? (c + 1) * chunk : T.n_rows;
    int4 o = exact_query_block<KIND>(imgs[pr.x], T, f.y, dim, row0, row1, sh);
    if (S > 1) {
      if (threadIdx.x == 0) {
        part[i * S + c] = o;
        __threadfence();
        last_s = atomicAdd(&arrived[i], 1) == S - 1;
      }
      __syncthreads();
      if (!last_s) continue;
      if (threadIdx.x == 0) {
        __threadfence();
        arrived[i] = 0;  // (for the next run)
        float d0 = 3.402823466e+38f, d1 = 3.402823466e+38f;
        int j0 = -1, j1 = -1;
        for (int cc = 0; cc < S; ++cc) {  // slices in row order: on equal distance the earlier row stays in front
          const int4 e = part[i * S + cc];  // (behind the acquire side of the fence above)
          const int js[2] = {e.x, e.y};
          const float ds[2] = {__int_as_float(e.z), __int_as_float(e.w)};
          for (int k = 0; k < 2; ++k) {
            if (js[k] < 0) continue;
            if (j0 < 0 || ds[k] < d0) {
              d1 = d0;
              j1 = j0;
              d0 = ds[k];
              j0 = js[k];
            } else if (j1 < 0 || ds[k] < d1) {
              d1 = ds[k];
              j1 = js[k];
            }
          }
        }
        o = make_int4(j0, j1, __float_as_int(j0 >= 0 ? d0 : 3.402823466e+38f), __float_as_int(j1 >= 0 ? d1 : 3.402823466e+38f));
      }
    }
    if (threadIdx.x == 0) knn[(size_t)f.x * maxq + f.y] = o;
  }
}

// reference src/Sfm.cpp:603-607: keep knn[i][0] iff d0 <= ratio*d1 (float), ascending queryIdx.
// Queries the MFMA kernel flagged (2nd-best squared distance >= 2^22, where sqrtf can merge
// neighbouring integers and the order by (sqrtf(s), index) can differ from the order by (s, index))
// are redone here with OpenCV's literal float arithmetic, one wave per flagged query.
template <int KIND>
__global__ __launch_bounds__(256) void compact_kernel(const ImgDev* __restrict__ imgs,
                                                      const int2* __restrict__ pairs,
                                                      int4* __restrict__ knn, int maxq, int dim,
                                                      float ratio, int* __restrict__ counts,
                                                      int* __restrict__ out_q, int* __restrict__ out_t,
                                                      float* __restrict__ out_d, int* __restrict__ fix_count) {
  __shared__ int wsum[4];
  __shared__ int running;
  const int p = blockIdx.x;
  // the fix-up list of this run has been consumed (knn_fixup_kernel ran before this kernel): clear its
  // counter for the plan's next run here instead of with a memset node in front of every run
  if (blockIdx.x == 0 && threadIdx.x == 0) *fix_count = 0;
  const int2 pr = pairs[p];
  const ImgDev Q = imgs[pr.x], T = imgs[pr.y];
  const int nq = Q.n_rows, nt = T.n_rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  if (nt >= 1) {
    for (int q0 = 0; q0 < nq; q0 += 256) {
      const int q = q0 + threadIdx.x;
      int4 e = make_int4(-1, -1, 0, 0);
      if (q < nq) e = knn[(size_t)p * maxq + q];
      unsigned long long fb = __ballot(q < nq && e.y >= 0 && (e.y & FIX_FLAG) != 0);
#if SFM_DBG == 4
      if (lane == 0 && fb) atomicAdd(&g_stamps[4096 * 8 * 16 - 1], (unsigned long long)__popcll(fb));
      if (lane == 0) atomicAdd(&g_stamps[4096 * 8 * 16 - 2], (unsigned long long)__popcll(__ballot(q < nq && e.y == FIX_FLAG)));
#endif
      while (fb && SFM_DBG != 5) {  // wave-uniform
        const int l = __ffsll((long long)fb) - 1;
        fb &= fb - 1;
        const int4 o = exact_query<KIND>(Q, T, q0 + wave * 64 + l, dim);
        if (lane == l) {
          e = o;
          knn[(size_t)p * maxq + q] = o;
        }
      }
      const bool keep = q < nq && nt >= 2 && e.y >= 0 && __int_as_float(e.z) <= __fmul_rn(ratio, __int_as_float(e.w));
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

// ---------------------------------------------------------------- packing for the host
// exclusive scan of the pair counts (one workgroup) ...
// h_counts / h_total (may be null): the same counts and the grand total written into mapped host memory (the
// pipelined fetch: the host reads them after the copy stream's event, no hipMemcpy in between)
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const int* __restrict__ counts, int n_pairs,
                                                            long long* __restrict__ offsets, int* __restrict__ h_counts = nullptr,
                                                            long long* __restrict__ h_total = nullptr) {
  __shared__ long long wsum[16];
  __shared__ long long running;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (int p0 = 0; p0 < n_pairs; p0 += 1024) {
    const int p = p0 + threadIdx.x;
    const long long c = p < n_pairs ? counts[p] : 0;
    long long x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    long long before = running;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (p < n_pairs) offsets[p] = before + x - c;
    if (p < n_pairs && h_counts) h_counts[p] = (int)c;
    __syncthreads();
    if (threadIdx.x == 0) {
      long long t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      running += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    offsets[n_pairs] = running;
    if (h_total) *h_total = running;
  }
}
// ... and the pairs' lists, gathered into three arrays back to back: queryIdx | trainIdx | distance bits, each of
// offsets[n_pairs] entries (the host copies them out with three memcpy, not record by record)
__global__ __launch_bounds__(256) void pack_lists_kernel(const int* __restrict__ counts, const long long* __restrict__ offsets,
                                                         int n_pairs, int maxq, const int* __restrict__ out_q,
                                                         const int* __restrict__ out_t, const float* __restrict__ out_d,
                                                         int* __restrict__ packed, long long capacity = -1) {
  const long long tot = offsets[n_pairs];
  if (capacity >= 0 && tot > capacity) return;  // (the host sees total > capacity and reports it)
  for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {  // (the pipelined fetch launches few workgroups: they share the device with a sweep)
    const int n = counts[p];
    const long long o = offsets[p];
    for (int i = threadIdx.x; i < n; i += 256) {
      const size_t src = (size_t)p * maxq + i;
      packed[o + i] = out_q[src];
      packed[tot + o + i] = out_t[src];
      packed[2 * tot + o + i] = __float_as_int(out_d[src]);
    }
  }
}

}  // namespace

// ================================================================= host side
struct sfmhip_imageset {
  sfmhip_ctx* ctx;
  int n_images, dim, dtype, norm;
  int kind, ks, sr, nu;  // ks==0: no MFMA instantiation -> exact kernel only; nu = query tiles per wave
  int nw = 4;            // waves per k-NN workgroup: 4 (two workgroups per CU, which drift half a sweep apart) or 8
  int late_start = 0;    // nw == 4: start delay of the second set of resident workgroups, in 8 k cycles
  std::vector<int> n_rows, n_pad;
  std::vector<ImgDev> h_imgs;
  std::vector<void*> owned_raw;
  ImgDev* d_imgs = nullptr;
  int8_t* d_tiles = nullptr;
  int* d_cin = nullptr;
  int* d_nq = nullptr;
  int* d_perm = nullptr;
  int* d_par = nullptr;
  int* d_nodd = nullptr;
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
  long long* d_offsets = nullptr;  // fetch: exclusive scan of the counts, packed {q, t, dist} records
  int* d_packed = nullptr;
  int4* d_fix_part = nullptr;   // knn_fixup_kernel: partial results of the workgroups sharing a query
  int* d_fix_arrived = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_k = nullptr;  // end of the k-NN kernel proper (ev[0] .. ev_k = that one launch)
  bool timed = false;
  // pipelined fetch (sfmhip_matchplan_pipeline): after every run a second stream packs the lists straight into one of
  // two pinned host buffers while the first stream goes on with the next sweep
  bool pipe_on = false;
  hipStream_t pipe_st = nullptr;
  hipEvent_t ev_ready = nullptr, ev_host[2] = {nullptr, nullptr};
  void* h_pipe[2] = {nullptr, nullptr};   // [int64 total | int64 capacity | int32 counts[cap_pairs] | int32 q[], t[], d[]]
  int* dh_pipe[2] = {nullptr, nullptr};   // the same buffers as the device sees them
  // round 5: the lists are packed into a DEVICE buffer of the same layout by a short many-workgroup launch and leave over the
  // DMA engine (hipMemcpyAsync on the copy stream): eight workgroups writing 3 MB through the PCIe link held their compute units
  // for a sixth of a sweep (host-visible 0.86 of the device-only rate).  The copy carries the matches the last runs had, plus a
  // quarter (the count is only known on the device); a run that has more gets the rest in fetch_wait.
  int* d_pipe[2] = {nullptr, nullptr};
  long long pipe_copied[2] = {0, 0};      // matches the slot's copy carried
  long long pipe_est = -1;                // matches to copy (< 0: not known yet, the whole capacity)
  bool pipe_zero_copy = false;            // SFMHIP_PIPE_ZEROCOPY=1: the round-3 path (measurement)
  long long pipe_capacity = 0;            // matches per slot
  long long runs = 0;                     // runs enqueued since the pipeline was switched on
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
  s->nu = s->ks == 8 ? 1 : 2;
  if (s->kind == KIND_U8_HAMMING) s->nu = 2;  // (knn_keyed_kernel: 4 waves x 2 query tiles)
  if (const char* e = getenv("SFMHIP_KNN_NW")) s->nw = atoi(e) == 8 ? 8 : 4;          // (tuning knobs, not API)
  if (const char* e = getenv("SFMHIP_KNN_LATE")) s->late_start = std::max(0, atoi(e));
  if (s->kind == KIND_U8_HAMMING) s->nw = 4;
  const int rb = 32 * (s->ks ? s->ks : 1);
  size_t tot_pad = 0;
  std::vector<int> tile_img, tile_first(n_images + 1, 0);
  for (int i = 0; i < n_images; ++i) {
    if (n_rows[i] < 0) {
      delete s;
      return SFMHIP_ERR_ARG;
    }
    s->n_rows.push_back(n_rows[i]);
    // L2 kinds store the rows in parity order, the odd class padded to a tile boundary: up to 31 more positions
    const int need = n_rows[i] ? n_rows[i] + (s->kind == KIND_U8_HAMMING ? 0 : TILE_ROWS - 1) : 0;
    const int pad = (need + s->sr - 1) / s->sr * s->sr;
    if ((size_t)pad * rb >= (1ull << 31)) {  // (one image's tile image is addressed by 31-bit offsets, its positions fit 26 bits)
      delete s;
      return SFMHIP_ERR_ARG;
    }
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
      (rc = sfm_dev_alloc(&s->d_cin, tot_pad + 512)) || (rc = sfm_dev_alloc(&s->d_nq, tot_pad)) ||
      (rc = sfm_dev_alloc(&s->d_perm, tot_pad)) || (rc = sfm_dev_alloc(&s->d_par, tot_pad)) ||
      (rc = sfm_dev_alloc(&s->d_nodd, (size_t)n_images)) ||
      (rc = sfm_dev_alloc(&s->d_tile_img, tile_img.size())) ||
      (rc = sfm_dev_alloc(&s->d_tile_first, (size_t)n_images + 1)) ||
      (rc = sfm_dev_alloc(&s->d_nonintegral, (size_t)n_images))) {
    sfmhip_imageset_destroy(s);
    return rc;
  }
  // (from here on a HIP failure must not leak the object: destroy, then report)
#define SFM_HIP_TRY_OR_DESTROY(expr, obj, destroy)                                        \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) {                                                              \
      g_sfmhip_last_hip_error = (int)e__;                                                 \
      fprintf(stderr, "[sfmhip] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
      destroy(obj);                                                                       \
      return SFMHIP_ERR_HIP;                                                              \
    }                                                                                     \
  } while (0)
  SFM_HIP_TRY_OR_DESTROY(hipMemcpy(s->d_tile_img, tile_img.data(), tile_img.size() * sizeof(int), hipMemcpyHostToDevice), s,
                         sfmhip_imageset_destroy);
  SFM_HIP_TRY_OR_DESTROY(hipMemcpy(s->d_tile_first, tile_first.data(), tile_first.size() * sizeof(int), hipMemcpyHostToDevice), s,
                         sfmhip_imageset_destroy);
  SFM_HIP_TRY_OR_DESTROY(hipMemset(s->d_nonintegral, 0xFF, n_images * sizeof(int)), s, sfmhip_imageset_destroy);  // -1: set by no prepare pass
  s->h_imgs.resize(n_images);
  s->owned_raw.assign(n_images, nullptr);
  size_t off = 0;
  for (int i = 0; i < n_images; ++i) {
    ImgDev& I = s->h_imgs[i];
    I.raw = nullptr;
    I.tiles = s->d_tiles + off * rb;
    I.cin = s->d_cin + off;
    I.nq = s->d_nq + off;
    I.perm = s->d_perm + off;
    I.par = s->d_par + off;
    I.nodd = s->d_nodd + i;
    I.n_rows = s->n_rows[i];
    I.n_pad = s->n_pad[i];
    off += s->n_pad[i];
  }
  SFM_HIP_TRY_OR_DESTROY(hipEventCreate(&s->ev_prep0), s, sfmhip_imageset_destroy);
  SFM_HIP_TRY_OR_DESTROY(hipEventCreate(&s->ev_prep1), s, sfmhip_imageset_destroy);
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
  // positions first: parity of the norms (L2 kinds), then one workgroup per image orders the rows
  if (s->kind == KIND_F32_L2)
    hipLaunchKernelGGL((parity_kernel<KIND_F32_L2>), grid, dim3(256), 0, st, s->d_imgs, s->d_tile_img, s->d_tile_first, s->dim);
  else if (s->kind == KIND_U8_L2)
    hipLaunchKernelGGL((parity_kernel<KIND_U8_L2>), grid, dim3(256), 0, st, s->d_imgs, s->d_tile_img, s->d_tile_first, s->dim);
  hipLaunchKernelGGL(order_kernel, dim3(s->n_images), dim3(256), 0, st, s->d_imgs, s->kind == KIND_U8_HAMMING ? 0 : 1);
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
  hipFree(s->d_cin);
  hipFree(s->d_nq);
  hipFree(s->d_perm);
  hipFree(s->d_par);
  hipFree(s->d_nodd);
  hipFree(s->d_tile_img);
  hipFree(s->d_tile_first);
  hipFree(s->d_nonintegral);
  if (s->ev_prep0) hipEventDestroy(s->ev_prep0);
  if (s->ev_prep1) hipEventDestroy(s->ev_prep1);
  delete s;
}

// work list: one workgroup per (pair, block of nw*nu query tiles: nu per wave).  Ordered so that the blocks a
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
    // query tiles cover the positions: the rows plus the padding of the odd class (L2 kinds)
    const int npos = s->n_rows[qi] + (s->kind == KIND_U8_HAMMING ? 0 : TILE_ROWS - 1);
    const int nqt = std::min((npos + TILE_ROWS - 1) / TILE_ROWS, s->n_pad[qi] / TILE_ROWS);
    for (int t0 = 0; t0 < nqt; t0 += s->nw * s->nu) {
      lanes[cur_lane].push_back(WorkItem{p, t0, qi, ti});
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
  pl->cap_items = (size_t)pl->cap_pairs * (size_t)(((pl->maxq + 2 * TILE_ROWS - 2) / TILE_ROWS + s->nw * s->nu - 1) / (s->nw * s->nu));
  int rc = SFMHIP_OK;
  const size_t slots = (size_t)pl->cap_pairs * pl->maxq;
  if ((rc = sfm_dev_alloc(&pl->d_pairs, (size_t)pl->cap_pairs)) || (rc = sfm_dev_alloc(&pl->d_items, pl->cap_items)) ||
      (rc = sfm_dev_alloc(&pl->d_knn, slots)) || (rc = sfm_dev_alloc(&pl->d_counts, (size_t)pl->cap_pairs)) ||
      (rc = sfm_dev_alloc(&pl->d_out_q, slots)) || (rc = sfm_dev_alloc(&pl->d_out_t, slots)) ||
      (rc = sfm_dev_alloc(&pl->d_out_d, slots)) || (rc = sfm_dev_alloc(&pl->d_fix_count, (size_t)1)) ||
      (rc = sfm_dev_alloc(&pl->d_fix_items, (size_t)FIX_CAP)) || (rc = sfm_dev_alloc(&pl->d_fix_part, (size_t)FIX_GRID)) ||
      (rc = sfm_dev_alloc(&pl->d_fix_arrived, (size_t)FIX_GRID))) {
    sfmhip_matchplan_destroy(pl);
    return rc;
  }
  if (n_pairs) SFM_HIP_TRY_OR_DESTROY(hipMemcpy(pl->d_pairs, pairs, sizeof(int2) * n_pairs, hipMemcpyHostToDevice), pl, sfmhip_matchplan_destroy);
  if (!items.empty())
    SFM_HIP_TRY_OR_DESTROY(hipMemcpy(pl->d_items, items.data(), sizeof(WorkItem) * items.size(), hipMemcpyHostToDevice), pl,
                           sfmhip_matchplan_destroy);
  SFM_HIP_TRY_OR_DESTROY(hipMemset(pl->d_counts, 0, sizeof(int) * pl->cap_pairs), pl, sfmhip_matchplan_destroy);
  SFM_HIP_TRY_OR_DESTROY(hipMemset(pl->d_fix_count, 0, sizeof(int)), pl, sfmhip_matchplan_destroy);  // (every run leaves it cleared: compact_kernel)
  SFM_HIP_TRY_OR_DESTROY(hipMemset(pl->d_fix_arrived, 0, sizeof(int) * FIX_GRID), pl, sfmhip_matchplan_destroy);  // (and these: the merging workgroup)
  for (auto& e : pl->ev) SFM_HIP_TRY_OR_DESTROY(hipEventCreate(&e), pl, sfmhip_matchplan_destroy);
  SFM_HIP_TRY_OR_DESTROY(hipEventCreate(&pl->ev_k), pl, sfmhip_matchplan_destroy);
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
  if (pl->pipe_st) SFM_HIP_TRY(hipStreamSynchronize(pl->pipe_st));
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

template <int KS, int MODE, int NU, int SR>
static int launch_knn(sfmhip_matchplan* pl) {
  sfmhip_imageset* s = pl->set;
  if (s->nw == 4)
    hipLaunchKernelGGL((knn_kernel<KS, MODE, NU, SR, 4>), dim3(pl->n_items), dim3(256), 0, s->ctx->stream, s->d_imgs,
                       pl->d_items, s->d_nonintegral, s->gen, s->dim, pl->d_knn, pl->maxq, s->late_start, pl->d_fix_count,
                       pl->d_fix_items);
  else
    hipLaunchKernelGGL((knn_kernel<KS, MODE, NU, SR, 8>), dim3(pl->n_items), dim3(512), 0, s->ctx->stream, s->d_imgs,
                       pl->d_items, s->d_nonintegral, s->gen, s->dim, pl->d_knn, pl->maxq, 0, pl->d_fix_count, pl->d_fix_items);
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
        hipLaunchKernelGGL((knn_keyed_kernel<8, 128>), dim3(pl->n_items), dim3(256), 0, st, s->d_imgs, pl->d_items, s->dim,
                           pl->d_knn, pl->maxq);
        SFM_HIP_TRY(hipGetLastError());
      } else {
        switch (s->ks) {
          case 1: SFM_TRY((launch_knn<1, 0, 2, 256>(pl))); break;
          case 2: SFM_TRY((launch_knn<2, 0, 2, 256>(pl))); break;
          case 4: SFM_TRY((launch_knn<4, 0, 2, 256>(pl))); break;
          case 8: SFM_TRY((launch_knn<8, 0, 1, 128>(pl))); break;
        }
      }
    }
    if (timing) SFM_HIP_TRY(hipEventRecord(pl->ev_k, st));
    // pairs the MFMA kernel leaves alone: a non-integer-valued f32 image, or no MFMA instantiation
    const int force_all = mfma ? 0 : 1;
    if (force_all || s->kind == KIND_F32_L2) {
      const int grid = std::min(std::max(pl->n_pairs, 256), 2048);
      if (s->kind == KIND_F32_L2)
        hipLaunchKernelGGL((knn_exact_kernel<KIND_F32_L2>), dim3(grid), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->n_pairs,
                           s->d_nonintegral, s->gen, force_all, s->dim, pl->d_knn, pl->maxq);
      else if (s->kind == KIND_U8_L2)
        hipLaunchKernelGGL((knn_exact_kernel<KIND_U8_L2>), dim3(grid), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->n_pairs,
                           s->d_nonintegral, s->gen, force_all, s->dim, pl->d_knn, pl->maxq);
      else
        hipLaunchKernelGGL((knn_exact_kernel<KIND_U8_HAMMING>), dim3(grid), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->n_pairs,
                           s->d_nonintegral, s->gen, force_all, s->dim, pl->d_knn, pl->maxq);
      SFM_HIP_TRY(hipGetLastError());
    }
    if (mfma && pl->n_items > 0 && s->kind != KIND_U8_HAMMING) {  // the listed sqrtf-merge candidates, one wave each
      if (s->kind == KIND_F32_L2)
        hipLaunchKernelGGL((knn_fixup_kernel<KIND_F32_L2>), dim3(FIX_GRID), dim3(256), 0, st, s->d_imgs, pl->d_pairs, s->dim, pl->d_knn,
                           pl->maxq, pl->d_fix_count, pl->d_fix_items, pl->d_fix_part, pl->d_fix_arrived);
      else
        hipLaunchKernelGGL((knn_fixup_kernel<KIND_U8_L2>), dim3(FIX_GRID), dim3(256), 0, st, s->d_imgs, pl->d_pairs, s->dim, pl->d_knn,
                           pl->maxq, pl->d_fix_count, pl->d_fix_items, pl->d_fix_part, pl->d_fix_arrived);
      SFM_HIP_TRY(hipGetLastError());
    }
  }
  if (timing) SFM_HIP_TRY(hipEventRecord(pl->ev[1], st));
  // (pipelined fetch: the previous run's lists are still being packed off d_counts / d_out_* by the copy stream --
  // in practice long done: that takes tens of microseconds and a sweep hundreds)
  if (pl->pipe_on && pl->runs > 0) SFM_HIP_TRY(hipStreamWaitEvent(st, pl->ev_host[(pl->runs - 1) & 1], 0));
  if (pl->n_pairs > 0) {
    if (s->kind == KIND_F32_L2)
      hipLaunchKernelGGL((compact_kernel<KIND_F32_L2>), dim3(pl->n_pairs), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->d_knn,
                         pl->maxq, s->dim, ratio, pl->d_counts, pl->d_out_q, pl->d_out_t, pl->d_out_d, pl->d_fix_count);
    else if (s->kind == KIND_U8_L2)
      hipLaunchKernelGGL((compact_kernel<KIND_U8_L2>), dim3(pl->n_pairs), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->d_knn,
                         pl->maxq, s->dim, ratio, pl->d_counts, pl->d_out_q, pl->d_out_t, pl->d_out_d, pl->d_fix_count);
    else
      hipLaunchKernelGGL((compact_kernel<KIND_U8_HAMMING>), dim3(pl->n_pairs), dim3(256), 0, st, s->d_imgs, pl->d_pairs, pl->d_knn,
                         pl->maxq, s->dim, ratio, pl->d_counts, pl->d_out_q, pl->d_out_t, pl->d_out_d, pl->d_fix_count);
    SFM_HIP_TRY(hipGetLastError());
  }
  if (timing) SFM_HIP_TRY(hipEventRecord(pl->ev[2], st));
  pl->timed = timing;
  if (pl->pipe_on) {
    const int slot = (int)(pl->runs & 1);
    SFM_HIP_TRY(hipEventRecord(pl->ev_ready, st));
    SFM_HIP_TRY(hipStreamWaitEvent(pl->pipe_st, pl->ev_ready, 0));
    const bool zc = pl->pipe_zero_copy;
    int* base = zc ? pl->dh_pipe[slot] : pl->d_pipe[slot];
    long long* h_hdr = (long long*)base;
    int* h_counts = base + 4;
    int* h_rec = h_counts + pl->cap_pairs;
    if (pl->n_pairs > 0) {
      hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, pl->pipe_st, pl->d_counts, pl->n_pairs, pl->d_offsets, h_counts, h_hdr);
      static const int pipe_wgs_env = getenv("SFMHIP_PIPE_WGS") ? atoi(getenv("SFMHIP_PIPE_WGS")) : 0;
      // (zero copy: 8 workgroups, 0.977 of the device-only sweep rate on one stream, 64 or more 0.91 -- scripts/gpu_hostvisible_ab.py;
      // into device memory the launch is over in microseconds whatever its size)
      const int pipe_wgs = pipe_wgs_env > 0 ? pipe_wgs_env : zc ? 8 : 128;
      hipLaunchKernelGGL(pack_lists_kernel, dim3(std::max(1, std::min(pl->n_pairs, pipe_wgs))), dim3(256), 0, pl->pipe_st, pl->d_counts,
                         pl->d_offsets, pl->n_pairs, pl->maxq, pl->d_out_q, pl->d_out_t, pl->d_out_d, h_rec, pl->pipe_capacity);
      SFM_HIP_TRY(hipGetLastError());
      if (!zc) {
        // how much to copy: what the finished runs had, plus a quarter (their totals are in the host buffers already)
        for (int k = 0; k < 2; ++k)
          if (pl->runs > k && hipEventQuery(pl->ev_host[(pl->runs - 1 - k) & 1]) == hipSuccess) {
            const long long seen = ((const long long*)pl->h_pipe[(pl->runs - 1 - k) & 1])[0];
            if (seen <= pl->pipe_capacity) pl->pipe_est = std::max(pl->pipe_est, seen + seen / 4 + 1024);
          }
        const long long n_copy = pl->pipe_est < 0 ? pl->pipe_capacity : std::min(pl->pipe_capacity, pl->pipe_est);
        pl->pipe_copied[slot] = n_copy;
        SFM_HIP_TRY(hipMemcpyAsync(pl->h_pipe[slot], pl->d_pipe[slot], 16 + sizeof(int) * ((size_t)pl->cap_pairs + 3 * (size_t)n_copy),
                                   hipMemcpyDeviceToHost, pl->pipe_st));
      }
    } else {
      ((long long*)pl->h_pipe[slot])[0] = 0;
    }
    SFM_HIP_TRY(hipEventRecord(pl->ev_host[slot], pl->pipe_st));
    ++pl->runs;
  }
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_pipeline(sfmhip_matchplan* pl, int64_t capacity) {
  if (!pl) return SFMHIP_ERR_ARG;
  sfmhip_imageset* s = pl->set;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  SFM_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
  if (pl->pipe_st) SFM_HIP_TRY(hipStreamSynchronize(pl->pipe_st));
  if (capacity < 0) {  // off: runs are no longer followed by the packing pass
    pl->pipe_on = false;
    pl->runs = 0;
    return SFMHIP_OK;
  }
  if (capacity == 0) capacity = std::max<long long>(4096, (long long)pl->cap_pairs * pl->maxq / 4);
  if (!pl->pipe_st) {
    // (highest priority: the packing launch gets the first compute units a sweep's workgroups give back, not the last)
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    if (getenv("SFMHIP_PIPE_PRIORITY") && atoi(getenv("SFMHIP_PIPE_PRIORITY")) == 0) prio_hi = prio_lo;  // (measurement)
    SFM_HIP_TRY(hipStreamCreateWithPriority(&pl->pipe_st, hipStreamNonBlocking, prio_hi));
    SFM_HIP_TRY(hipEventCreateWithFlags(&pl->ev_ready, hipEventDisableTiming));
    for (auto& e : pl->ev_host) SFM_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (!pl->d_offsets) SFM_TRY(sfm_dev_alloc(&pl->d_offsets, (size_t)pl->cap_pairs + 1));
  if (capacity != pl->pipe_capacity || !pl->h_pipe[0] || !pl->h_pipe[1]) {
    pl->pipe_on = false;  // (a refused allocation leaves the plan usable without the pipeline)
    pl->pipe_capacity = 0;
    for (int k = 0; k < 2; ++k) {
      if (pl->h_pipe[k]) hipHostFree(pl->h_pipe[k]);
      pl->h_pipe[k] = nullptr;
      const size_t bytes = 16 + sizeof(int) * ((size_t)pl->cap_pairs + 3 * (size_t)capacity);
      SFM_HIP_TRY(hipHostMalloc(&pl->h_pipe[k], bytes, hipHostMallocMapped));
      void* dp = nullptr;
      SFM_HIP_TRY(hipHostGetDevicePointer(&dp, pl->h_pipe[k], 0));
      pl->dh_pipe[k] = (int*)dp;
      memset(pl->h_pipe[k], 0, 16);
      if (pl->d_pipe[k]) hipFree(pl->d_pipe[k]);
      pl->d_pipe[k] = nullptr;
      SFM_HIP_TRY(hipMalloc((void**)&pl->d_pipe[k], bytes));
    }
    pl->pipe_est = -1;
    pl->pipe_capacity = capacity;
  }
  pl->pipe_on = true;
  pl->runs = 0;
  pl->pipe_zero_copy = getenv("SFMHIP_PIPE_ZEROCOPY") && atoi(getenv("SFMHIP_PIPE_ZEROCOPY")) != 0;
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_fetch_wait(sfmhip_matchplan* pl, int back, const int32_t** counts, const int32_t** out_q,
                                           const int32_t** out_t, const float** out_dist, int64_t* total) {
  if (!pl || !pl->pipe_on || back < 0 || back > 1 || pl->runs <= back) return pl && pl->pipe_on ? SFMHIP_ERR_STATE : SFMHIP_ERR_ARG;
  const int slot = (int)((pl->runs - 1 - back) & 1);
  SFM_HIP_TRY(hipEventSynchronize(pl->ev_host[slot]));
  const long long tot = ((const long long*)pl->h_pipe[slot])[0];
  const int32_t* c = (const int32_t*)pl->h_pipe[slot] + 4;
  const int32_t* rec = c + pl->cap_pairs;
  if (total) *total = tot;
  if (counts) *counts = c;
  if (tot > pl->pipe_capacity) return SFMHIP_ERR_ALLOC;  // (total is set: switch the pipeline on again with that much room)
  if (!pl->pipe_zero_copy && tot > pl->pipe_copied[slot]) {
    // the run found more matches than the copy was sized for: the rest of the three arrays, now (the device buffer of the slot
    // is untouched until the slot's next run)
    const size_t lo = (size_t)pl->cap_pairs + 3 * (size_t)pl->pipe_copied[slot], hi = (size_t)pl->cap_pairs + 3 * (size_t)tot;
    SFM_HIP_TRY(hipSetDevice(pl->set->ctx->device));
    SFM_HIP_TRY(hipMemcpy((int*)pl->h_pipe[slot] + 4 + lo, pl->d_pipe[slot] + 4 + lo, sizeof(int) * (hi - lo), hipMemcpyDeviceToHost));
    pl->pipe_copied[slot] = tot;
    pl->pipe_est = std::max(pl->pipe_est, tot + tot / 4 + 1024);
  }
  if (out_q) *out_q = rec;
  if (out_t) *out_t = rec + tot;
  if (out_dist) *out_dist = (const float*)(rec + 2 * tot);
  return SFMHIP_OK;
}

extern "C" int sfmhip_matchplan_fetch(sfmhip_matchplan* pl, int32_t* counts, int32_t* out_q, int32_t* out_t,
                                      float* out_dist, int64_t capacity, int64_t* total) {
  if (!pl || !counts) return SFMHIP_ERR_ARG;
  sfmhip_imageset* s = pl->set;
  SFM_HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  const bool lists = out_q || out_t || out_dist;
  if (total) *total = 0;
  if (pl->n_pairs == 0) return SFMHIP_OK;
  if (pl->pipe_st) SFM_HIP_TRY(hipStreamSynchronize(pl->pipe_st));  // (the pipelined packing shares d_offsets)
  // The lists leave the device packed: a scan of the counts and a gather into {q, t, dist} records on the
  // device, then two copies (counts + offsets, records) instead of three per pair.
  if (lists) {
    if (!pl->d_offsets) SFM_TRY(sfm_dev_alloc(&pl->d_offsets, (size_t)pl->cap_pairs + 1));
    if (!pl->d_packed) SFM_TRY(sfm_dev_alloc(&pl->d_packed, 3 * (size_t)pl->cap_pairs * pl->maxq));
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, st, pl->d_counts, pl->n_pairs, pl->d_offsets);
    hipLaunchKernelGGL(pack_lists_kernel, dim3(pl->n_pairs), dim3(256), 0, st, pl->d_counts, pl->d_offsets, pl->n_pairs,
                       pl->maxq, pl->d_out_q, pl->d_out_t, pl->d_out_d, pl->d_packed);
    SFM_HIP_TRY(hipGetLastError());
  }
  SFM_HIP_TRY(hipMemcpyAsync(counts, pl->d_counts, sizeof(int) * pl->n_pairs, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  int64_t tot = 0;
  for (int p = 0; p < pl->n_pairs; ++p) tot += counts[p];
  if (total) *total = tot;
  if (!lists) return SFMHIP_OK;
  if (tot > capacity) return SFMHIP_ERR_ARG;
  if (tot == 0) return SFMHIP_OK;
  void* stage = nullptr;
  SFM_TRY(sfm_ctx_pinned(s->ctx, (size_t)tot * 12, &stage));
  SFM_HIP_TRY(hipMemcpyAsync(stage, pl->d_packed, (size_t)tot * 12, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  const int32_t* rec = (const int32_t*)stage;
  if (out_q) memcpy(out_q, rec, (size_t)tot * 4);
  if (out_t) memcpy(out_t, rec + tot, (size_t)tot * 4);
  if (out_dist) memcpy(out_dist, rec + 2 * tot, (size_t)tot * 4);
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

extern "C" int sfmhip_matchplan_last_knn_kernel_time(sfmhip_matchplan* pl, double* seconds) {
  if (!pl || !seconds) return SFMHIP_ERR_ARG;
  *seconds = 0;
  if (pl->timed && pl->n_pairs > 0) {
    float ms = 0;
    SFM_HIP_TRY(hipEventSynchronize(pl->ev_k));
    SFM_HIP_TRY(hipEventElapsedTime(&ms, pl->ev[0], pl->ev_k));
    *seconds = ms * 1e-3;
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
  hipFree(pl->d_fix_part);
  hipFree(pl->d_fix_arrived);
  hipFree(pl->d_offsets);
  hipFree(pl->d_packed);
  if (pl->pipe_st) {
    hipStreamSynchronize(pl->pipe_st);
    hipStreamDestroy(pl->pipe_st);
  }
  if (pl->ev_ready) hipEventDestroy(pl->ev_ready);
  for (auto& e : pl->ev_host)
    if (e) hipEventDestroy(e);
  for (int* dp : pl->d_pipe)
    if (dp) hipFree(dp);
  for (void* h : pl->h_pipe)
    if (h) hipHostFree(h);
  for (auto& e : pl->ev)
    if (e) hipEventDestroy(e);
  if (pl->ev_k) hipEventDestroy(pl->ev_k);
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

#if SFM_DBG == 4
extern "C" int sfmhip_dbg_read_stamps(void* out, size_t bytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), bytes) == hipSuccess ? 0 : -1;
}
#endif
