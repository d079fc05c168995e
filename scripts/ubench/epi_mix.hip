// Micro-benchmark (diagnostic, not product): cost of candidate k-NN epilogues beside the i8 MFMA
// chain on gfx950, at 1/2/3 waves per SIMD.  One "unit" = one 32x32 (train x query) tile of
// SIFT-128 = 4 x v_mfma_i32_32x32x32_i8 + the epilogue of its 16 accumulator registers.
//   mix 0: round-1 epilogue: per element v_mad_i32_i24 + v_med3_u32 + v_min_u32   (48 ops/unit)
//   mix 1: value-only top-2 per element: v_max_i32 + v_med3_i32                    (32 ops/unit)
//   mix 2: slot maxima over two tiles (v_max3_i32) + per-tile max3 tree + keyed top-2 of the
//          tile maxima                                                            (19 ops/unit)
//   mix 3: MFMA only
//   mix 4: mix 2's VALU only
// Build: hipcc -O3 --offload-arch=gfx950 epi_mix.hip -o epi_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int max3i(int a, int b, int c) {
  int d;
  asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int med3i(int a, int b, int c) {
  int d;
  asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ int maxi(int a, int b) {
  int d;
  asm volatile("v_max_i32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

template <int MIX>
__global__ __launch_bounds__(256, 2) void k(int* out, int iters, int mul, int tile0) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  v16i cin0, cin1, accA, accB, nxtA, nxtB;
  for (int e = 0; e < 16; ++e) {
    cin0[e] = threadIdx.x + e;
    cin1[e] = threadIdx.x * 3 + e;
    accA[e] = e * 77 + threadIdx.x;
    accB[e] = e * 31 + threadIdx.x;
  }
  nxtA = accA;
  nxtB = accB;
  unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
  unsigned bs[16];
  for (int e = 0; e < 16; ++e) bs[e] = 1000 * e + threadIdx.x;
  int v0 = -0x7fffffff, v1 = -0x7fffffff;
  int slot[16];
  for (int e = 0; e < 16; ++e) slot[e] = -0x7fffffff;
  int b0 = -0x7fffffff, b1 = -0x7fffffff;
  int tile = tile0;
#define KBODY(accA, accB, nxtA, nxtB)                                                           \
  do {                                                                                          \
    asm volatile("" : "+v"(a), "+v"(b));                                                        \
    if (MIX != 4) {                                                                             \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                          \
          nxtA = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, ks == 0 ? cin0 : nxtA, 0, 0, 0);   \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                          \
          nxtB = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, ks == 0 ? cin1 : nxtB, 0, 0, 0);   \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (MIX == 0) {                                                                             \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                          \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                         \
          const int x = u ? accB[e] : accA[e];                                                  \
          const unsigned key = (unsigned)__mul24(x, mul) + bs[e];                               \
          const unsigned lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;                        \
          const unsigned m = hi < key ? hi : key;                                               \
          k1 = lo > m ? lo : m;                                                                 \
          k0 = k0 < key ? k0 : key;                                                             \
          asm volatile("" : "+v"(k0), "+v"(k1));                                                \
        }                                                                                       \
      }                                                                                         \
    } else if (MIX == 1) {                                                                      \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                          \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                         \
          const int x = u ? accB[e] : accA[e];                                                  \
          const int lo = min(v0, x);                                                            \
          v0 = max(v0, x);                                                                      \
          v1 = max(v1, lo);                                                                     \
          asm volatile("" : "+v"(v0), "+v"(v1));                                                \
        }                                                                                       \
      }                                                                                         \
    } else if (MIX == 2 || MIX == 4) {                                                          \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) slot[e] = max(max(slot[e], accA[e]), accB[e]); \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                           \
        const v16i& x = u ? accB : accA;                                                        \
        const int t0 = max(max(x[0], x[1]), x[2]), t1 = max(max(x[3], x[4]), x[5]);             \
        const int t2 = max(max(x[6], x[7]), x[8]), t3 = max(max(x[9], x[10]), x[11]);           \
        const int t4 = max(max(x[12], x[13]), x[14]);                                           \
        const int m = max(max(max(t0, t1), t2), max(max(t3, t4), x[15]));                       \
        const int key = (m << 8) | tile;                                                        \
        const int lo = min(b0, key);                                                            \
        b0 = max(b0, key);                                                                      \
        b1 = max(b1, lo);                                                                       \
        tile -= 1;                                                                              \
      }                                                                                         \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (MIX == 4) {                                                                             \
      nxtA[0] += it;                                                                            \
      nxtB[3] += it;                                                                            \
    }                                                                                           \
  } while (0)
  for (int it = 0; it < iters; it += 2) {
    KBODY(accA, accB, nxtA, nxtB);
    KBODY(nxtA, nxtB, accA, accB);
  }
#undef KBODY
  int s = (int)(k0 ^ k1) ^ v0 ^ v1 ^ b0 ^ b1;
  for (int e = 0; e < 16; ++e) s ^= accA[e] ^ accB[e] ^ slot[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// mix 5: the 19-op epilogue of the PREVIOUS tile pair spread between the 8 MFMAs of the next one
// (plain C++ max chains -> v_max3_i32; one MFMA + 5 VALU per group by sched_group_barrier);
// STAG: waves 4..7 of a 512-thread workgroup start half a period late.
__device__ __forceinline__ int mx3(int a, int b, int c) { return max(max(a, b), c); }
template <int THREADS, int STAG>
__global__ __launch_bounds__(THREADS, 2) void k5(int* out, int iters, int tile0) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  v16i cin0, cin1, accA, accB, nxtA, nxtB;
  for (int e = 0; e < 16; ++e) {
    cin0[e] = threadIdx.x + e;
    cin1[e] = threadIdx.x * 3 + e;
    accA[e] = e * 77 + threadIdx.x;
    accB[e] = e * 31 + threadIdx.x;
  }
  nxtA = accA;
  nxtB = accB;
  int slot[16];
  for (int e = 0; e < 16; ++e) slot[e] = -0x7fffffff;
  int b0 = -0x7fffffff, b1 = -0x7fffffff;
  int tile = tile0;
  if (STAG && threadIdx.x >= 256) {
    for (int i = 0; i < 40; ++i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(b0) : "v"(b1));
  }
#define BODY(accA, accB, nxtA, nxtB)                                                            \
  do {                                                                                          \
    asm volatile("" : "+v"(a), "+v"(b));                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                            \
        nxtA = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, ks == 0 ? cin0 : nxtA, 0, 0, 0);     \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                            \
        nxtB = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, ks == 0 ? cin1 : nxtB, 0, 0, 0);     \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) slot[e] = mx3(slot[e], accA[e], accB[e]);    \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                             \
      const v16i& x = u ? accB : accA;                                                          \
      const int t0 = mx3(x[0], x[1], x[2]), t1 = mx3(x[3], x[4], x[5]), t2 = mx3(x[6], x[7], x[8]); \
      const int t3 = mx3(x[9], x[10], x[11]), t4 = mx3(x[12], x[13], x[14]);                    \
      const int m = max(mx3(t0, t1, t2), mx3(t3, t4, x[15]));                                   \
      const int key = (m << 8) | tile;                                                          \
      const int lo = min(b0, key);                                                              \
      b0 = max(b0, key);                                                                        \
      b1 = max(b1, lo);                                                                         \
      tile -= 1;                                                                                \
    }                                                                                           \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                        \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
  for (int it = 0; it < iters; it += 2) {
    BODY(accA, accB, nxtA, nxtB);
    BODY(nxtA, nxtB, accA, accB);
  }
#undef BODY
  int s = b0 ^ b1;
  for (int e = 0; e < 16; ++e) s ^= accA[e] ^ accB[e] ^ slot[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int THREADS, int STAG>
static void run5(const char* name) {
  int* out;
  (void)hipMalloc(&out, sizeof(int) * 256 * 4 * 512);
  for (int w = 1; w <= 3; ++w) {
    const int wpw = THREADS / 256;  // waves per SIMD per workgroup
    const int grid = 256 * w, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k5<THREADS, STAG><<<grid, THREADS>>>(out, 200, 255);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k5<THREADS, STAG><<<grid, THREADS>>>(out, iters, 255);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d: %8.3f ms -> %6.1f ns per unit per SIMD\n", name, w * wpw, ms, ms * 1e6 / iters / 2 / (w * wpw));
  }
  (void)hipFree(out);
}

template <int MIX>
static void run(const char* name) {
  int* out;
  (void)hipMalloc(&out, sizeof(int) * 256 * 4 * 256);
  for (int w = 1; w <= 3; ++w) {
    const int grid = 256 * w, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<MIX><<<grid, 256>>>(out, 200, -512, 255);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MIX><<<grid, 256>>>(out, iters, -512, 255);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d: %8.3f ms -> %6.1f ns per unit per SIMD\n", name, w, ms, ms * 1e6 / iters / 2 / w);
  }
  (void)hipFree(out);
}

// ---- op rates: 8 independent chains per wave, 2 and 4 waves per SIMD
template <int OP>
__global__ __launch_bounds__(256) void kop(unsigned* out, int iters) {
  unsigned x[8];
  for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 977 + e * 131;
  unsigned y = threadIdx.x * 31 + 7, z = threadIdx.x ^ 0x5555;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 2) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 3) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 4) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 7) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 8) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 9) asm volatile("v_max_i16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 9, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 12) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 13) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[e]) : "v"(y) : "vcc");
        if (OP == 14) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 15) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 17) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 18) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 19) asm volatile("v_mov_b32 %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 20) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 21) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 22) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 23) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 24) asm volatile("v_max3_i16 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 25) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 26) asm volatile("v_accvgpr_write_b32 a0, %0" : : "v"(x[e]) : "a0");
        if (OP == 27) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
      }
    }
  }
  unsigned s = 0;
  for (int e = 0; e < 8; ++e) s ^= x[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
static void runop(const char* name) {
  unsigned* out;
  (void)hipMalloc(&out, sizeof(unsigned) * 1024 * 256);
  printf("%-22s", name);
  for (int w = 1; w <= 4; w *= 2) {
    const int grid = 256 * w, iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kop<OP><<<grid, 256>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kop<OP><<<grid, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  w%d: %.3f ns/inst/SIMD", w, ms * 1e6 / iters / 48 / w);
  }
  printf("\n");
  (void)hipFree(out);
}

int main() {
  run<3>("mfma only (4/unit)");
  run<0>("r1 mad24+med3+min (48)");
  run<1>("value top2 max+med3 (32)");
  run<2>("slot max3 + tile tree (19)");
  run<4>("mix2 VALU only");
  run5<256, 0>("interleaved 19, 256thr");
  run5<512, 0>("interleaved 19, 512thr");
  run5<512, 1>("interleaved 19, 512thr stag");
  runop<0>("v_add_u32");
  runop<17>("v_sub_u32");
  runop<18>("v_xor_b32");
  runop<19>("v_mov_b32");
  runop<1>("v_max_i32");
  runop<2>("v_max_u32");
  runop<3>("v_max3_i32");
  runop<23>("v_min3_u32");
  runop<4>("v_med3_i32");
  runop<5>("v_max_f32");
  runop<6>("v_max3_f32");
  runop<27>("v_fma_f32");
  runop<7>("v_pk_max_i16");
  runop<8>("v_pk_min_u16");
  runop<20>("v_pk_max_f16");
  runop<25>("v_pk_add_i16");
  runop<9>("v_max_i16");
  runop<24>("v_max3_i16");
  runop<10>("v_lshl_add_u32");
  runop<11>("v_lshl_or_b32");
  runop<12>("v_mad_i32_i24");
  runop<21>("v_add3_u32");
  runop<22>("v_and_or_b32");
  runop<13>("v_cmp+v_cndmask (2)");
  runop<14>("v_dot4_i32_i8");
  runop<15>("v_cvt_pk_i16_i32");
  runop<16>("v_perm_b32");
  runop<26>("v_accvgpr_write");
  return 0;
}
