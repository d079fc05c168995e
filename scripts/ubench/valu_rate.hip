// Micro-benchmark (diagnostic, not product): VALU issue rate of the matcher's epilogue ops on
// gfx950, alone and beside i8 MFMA, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: VALU only (mad+med3+min x16 per iter)  1: MFMA only (4 per iter)  2: both
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, int mul) {
  unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
  v16i acc, cin;
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  for (int e = 0; e < 16; ++e) { cin[e] = threadIdx.x + e; acc[e] = e * 77 + threadIdx.x; }
  unsigned bs[16];
  for (int e = 0; e < 16; ++e) bs[e] = 1000 * e + threadIdx.x;
  v16i nxt = acc;
  for (int it = 0; it < iters; ++it) {
    if (MODE != 0) {
      nxt = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, cin, 0, 0, 0);
      nxt = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, nxt, 0, 0, 0);
      nxt = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, nxt, 0, 0, 0);
      nxt = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, nxt, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned key = (unsigned)__mul24(acc[e], mul) + bs[e];
        const unsigned lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
        const unsigned m = hi < key ? hi : key;
        k1 = lo > m ? lo : m;
        k0 = k0 < key ? k0 : key;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 0) acc = nxt; else { acc[0] += it; }
  }
  unsigned s = k0 ^ k1;
  for (int e = 0; e < 16; ++e) s ^= (unsigned)acc[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE 3: 1 MFMA : 12 VALU interleave (code order pinned by sched_barrier), one insertion chain
// MODE 4: same, two independent insertion chains (two units per iteration)
template <int MODE>
__global__ __launch_bounds__(256) void k2(unsigned* out, int iters, int mul) {
  unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, j0 = 0xFFFFFFFFu, j1 = 0xFFFFFFFFu;
  v16i acc, acc2, cin, nxt, nxt2;
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  for (int e = 0; e < 16; ++e) { cin[e] = threadIdx.x + e; acc[e] = e * 77 + threadIdx.x; acc2[e] = e * 31 + threadIdx.x; }
  unsigned bs[16];
  for (int e = 0; e < 16; ++e) bs[e] = 1000 * e + threadIdx.x;
  nxt = acc; nxt2 = acc2;
#define INS(K0, K1, ACC, e)                                            \
  {                                                                     \
    const unsigned key = (unsigned)__mul24(ACC[e], mul) + bs[e];        \
    const unsigned lo = K0 < K1 ? K0 : K1, hi = K0 < K1 ? K1 : K0;      \
    const unsigned m = hi < key ? hi : key;                             \
    K1 = lo > m ? lo : m;                                               \
    K0 = K0 < key ? K0 : key;                                           \
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        nxt = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, g == 0 ? cin : nxt, 0, 0, 0);
#pragma unroll
        for (int e = 4 * g; e < 4 * g + 4; ++e) INS(k0, k1, acc, e)
        __builtin_amdgcn_sched_barrier(0);
      }
      acc = nxt;
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g < 4) nxt = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, g == 0 ? cin : nxt, 0, 0, 0);
        else nxt2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, g == 4 ? cin : nxt2, 0, 0, 0);
#pragma unroll
        for (int e = 2 * g; e < 2 * g + 2; ++e) { INS(k0, k1, acc, e) INS(j0, j1, acc2, e) }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc = nxt; acc2 = nxt2;
    }
  }
  unsigned s = k0 ^ k1 ^ j0 ^ j1;
  for (int e = 0; e < 16; ++e) s ^= (unsigned)acc[e] ^ (unsigned)acc2[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run2(const char* name, int wgs_per_cu) {
  unsigned* out;
  const int grid = 256 * wgs_per_cu;
  hipMalloc(&out, sizeof(unsigned) * grid * 256);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k2<MODE><<<grid, 256>>>(out, 100, -512);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k2<MODE><<<grid, 256>>>(out, iters, -512);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int units = MODE == 3 ? 1 : 2;
  printf("%-10s waves/SIMD %d: %.3f ms -> %.1f ns per unit per SIMD\n", name, wgs_per_cu, ms, ms * 1e6 / iters / wgs_per_cu / units);
  hipFree(out);
}

template <int MODE>
static void run(const char* name, int wgs_per_cu) {
  unsigned* out;
  const int grid = 256 * wgs_per_cu;
  hipMalloc(&out, sizeof(unsigned) * grid * 256);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(out, 100, -512);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, iters, -512);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: wgs_per_cu waves, each iters iterations
  const double ns_per_iter_per_simd = ms * 1e6 / iters;  // wall per iteration (all waves of a SIMD progress together)
  printf("%-10s waves/SIMD %d: %.3f ms  -> %.1f ns per iteration-round (%d waves) = %.1f ns per wave-iteration-equivalent\n", name,
         wgs_per_cu, ms, ns_per_iter_per_simd, wgs_per_cu, ns_per_iter_per_simd / wgs_per_cu);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) run<0>("valu48", w);
  for (int w = 1; w <= 4; w *= 2) run<1>("mfma4", w);
  for (int w = 1; w <= 4; w *= 2) run<2>("both", w);
  for (int w = 1; w <= 4; w *= 2) run2<3>("fine1", w);
  for (int w = 1; w <= 2; w *= 2) run2<4>("fine2", w);
  return 0;
}
