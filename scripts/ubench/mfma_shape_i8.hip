// Micro-benchmark (diagnostic, not product): the k-NN sweep's instruction mix -- i8 MFMAs fed by ds_read_b128 from an
// LDS image, the 19-op value epilogue per 32x32 block of distances -- with v_mfma_i32_32x32x32_i8 and with
// v_mfma_i32_16x16x64_i8, on RANDOM operands and long enough for the clock to settle: the chip lowers its clock under
// load, and the clock it holds can depend on the MFMA shape (MI355X_MICROARCH.md, 'DVFS give-back' item 7), so cycles per
// operation do not decide which shape delivers more.  Per wave and iteration both shapes do the same work: 32 train rows
// x 64 queries x 128 dimensions (8 MFMAs of 32 cycles or 16 of 16), 8 resp. 6 ds_read_b128, 38 epilogue ops.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_shape_i8.hip -o mfma_shape_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mx3(int a, int b, int c) { return max(max(a, b), c); }

// the epilogue of 32 accumulator values (two blocks of 16): slot maxima + two 7-op trees + keyed top-2
#define EPILOGUE(P0, P1)                                                                          \
  do {                                                                                            \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) slot[e] = mx3(slot[e], P0[e], P1[e]);          \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                               \
      const int* x = u ? P1 : P0;                                                                 \
      const int t0 = mx3(x[0], x[1], x[2]), t1 = mx3(x[3], x[4], x[5]), t2 = mx3(x[6], x[7], x[8]); \
      const int t3 = mx3(x[9], x[10], x[11]), t4 = mx3(x[12], x[13], x[14]);                      \
      const int m = max(mx3(t0, t1, t2), mx3(t3, t4, x[15]));                                     \
      const int key = (m << 8) | (tile & 255);                                                    \
      const int lo = min(b0, key);                                                                \
      b0 = max(b0, key);                                                                          \
      b1 = max(b1, lo);                                                                           \
      tile -= 1;                                                                                  \
    }                                                                                             \
  } while (0)

template <int SHAPE, int SCHED, int LDSRD = 1, int EPI = 1>
__global__ __launch_bounds__(256, 2) void sweep(const int* __restrict__ rnd, int* __restrict__ out, unsigned long long* __restrict__ clk,
                                                int iters) {
  __shared__ __attribute__((aligned(16))) int lds[16384];  // 64 KB image
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = rnd[(blockIdx.x * 977 + i) & 0xFFFFF];
  v4i bq[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    bq[k] = v4i{rnd[(threadIdx.x * 32 + k * 4 + blockIdx.x * 131) & 0xFFFFF], rnd[(threadIdx.x * 32 + k * 4 + 1 + blockIdx.x * 131) & 0xFFFFF],
                rnd[(threadIdx.x * 32 + k * 4 + 2 + blockIdx.x * 131) & 0xFFFFF], rnd[(threadIdx.x * 32 + k * 4 + 3 + blockIdx.x * 131) & 0xFFFFF]};
  __syncthreads();
  int slot[16];
  for (int e = 0; e < 16; ++e) slot[e] = -0x7fffffff;
  int b0 = -0x7fffffff, b1 = -0x7fffffff, tile = 255;
  int P0[16], P1[16], Q0[16], Q1[16];
  for (int e = 0; e < 16; ++e) P0[e] = P1[e] = Q0[e] = Q1[e] = e - 100000;
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  const v4i* L = (const v4i*)lds;
  v4i Afix[4], Cfix[4];
  for (int k = 0; k < 4; ++k) {
    Afix[k] = L[lane + 64 * k];
    Cfix[k] = L[lane + 256 + 64 * k];
  }
#define BODY(X0, X1, D0, D1, IT)                                                                                 \
  do {                                                                                                           \
    const int base = (LDSRD ? (((IT) * 37) & 7) * 512 : 0) + lane; /* a different 8 KB tile every iteration */            \
    if (SHAPE == 0) {                                                                                            \
      v4i A[4], C[4];                                                                                            \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) A[k] = LDSRD ? L[base + 64 * k] : Afix[k];                   \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) C[k] = LDSRD ? L[base + 256 + 64 * k] : Cfix[k];             \
      if (!LDSRD) asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]));  /* (opaque: not one body for two) */ \
      v16i c, a0, a1;                                                                                            \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) c[e] = C[e >> 2][e & 3];                                    \
      a0 = c;                                                                                                    \
      a1 = c;                                                                                                    \
      if (!EPI) { _Pragma("unroll") for (int e = 0; e < 16; ++e) { a0[e] = D0[e]; a1[e] = D1[e]; } }             \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[k], bq[k], a0, 0, 0, 0);     \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[k], bq[4 + k], a1, 0, 0, 0); \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                           \
        X0[e] = a0[e];                                                                                           \
        X1[e] = a1[e];                                                                                           \
      }                                                                                                          \
    } else {                                                                                                     \
      v4i A[4], C[2];                                                                                            \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) A[k] = LDSRD ? L[base + 64 * k] : Afix[k];                   \
      _Pragma("unroll") for (int k = 0; k < 2; ++k) C[k] = LDSRD ? L[base + 256 + 64 * k] : Cfix[k];             \
      if (!LDSRD) asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]));                             \
      _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) /* train tile of 16 rows */                                \
        _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) { /* query tile of 16 */                                 \
          v4i a = C[tt];                                                                                         \
          if (!EPI) { _Pragma("unroll") for (int j = 0; j < 4; ++j) a[j] = qq < 2 ? D0[(tt * 2 + qq) * 4 + j] : D1[(tt * 2 + qq - 2) * 4 + j]; } \
          a = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * tt], bq[2 * qq], a, 0, 0, 0);                          \
          a = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * tt + 1], bq[2 * qq + 1], a, 0, 0, 0);                  \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
            if (qq < 2) X0[(tt * 2 + qq) * 4 + j] = a[j];                                                        \
            else X1[(tt * 2 + qq - 2) * 4 + j] = a[j];                                                           \
          }                                                                                                      \
        }                                                                                                        \
    }                                                                                                            \
    if (EPI) EPILOGUE(D0, D1);                                                                                   \
    if (SCHED) {                                                                                                 \
      /* the epilogue of the previous blocks spread between this body's MFMAs: 1 MFMA + 1 DS read + VALU ops */  \
      if (SHAPE == 0) {                                                                                          \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                          \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                     \
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                                     \
        }                                                                                                        \
      } else {                                                                                                   \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                          \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                     \
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                     \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                     \
        }                                                                                                        \
      }                                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
    }                                                                                                            \
  } while (0)
  for (int it = 0; it < iters; it += 2) {
    BODY(P0, P1, Q0, Q1, it);
    BODY(Q0, Q1, P0, P1, it + 1);
  }
#undef BODY
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  int s = b0 ^ b1;
  for (int e = 0; e < 16; ++e) s ^= P0[e] ^ P1[e] ^ Q0[e] ^ Q1[e] ^ slot[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int SCHED, int LDSRD = 1, int EPI = 1>
static void run(const char* name, const int* rnd, int* out, unsigned long long* clk, int grid, double seconds) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  sweep<SHAPE, SCHED, LDSRD, EPI><<<grid, 256>>>(rnd, out, clk, 2000);
  (void)hipDeviceSynchronize();
  float ms = 0, total = 0;
  int n = 0;
  while (total < seconds * 1e3) {  // back to back until the clock has settled; the last launch is the one reported
    (void)hipEventRecord(e0);
    sweep<SHAPE, SCHED, LDSRD, EPI><<<grid, 256>>>(rnd, out, clk, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    total += ms;
    ++n;
  }
  std::vector<unsigned long long> h(2 * grid);
  (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < grid; ++b) ghz.push_back((double)h[2 * b] / ((double)h[2 * b + 1] * 10.0));  // s_memrealtime: 100 MHz
  std::sort(ghz.begin(), ghz.end());
  const double ops = (double)grid * 4 * iters * 2.0 * 32 * 64 * 128;
  printf("%-34s grid %4d: last of %2d launches %8.2f ms  %6.3f POP/s  in-kernel clock %.3f GHz  cycles/iteration/wave %.1f\n", name, grid, n, ms,
         ops / (ms * 1e-3) / 1e15, ghz[grid / 2], (double)h[2 * (grid / 2)] / iters);
}

int main() {
  int *rnd, *out;
  unsigned long long* clk;
  std::vector<int> h(1 << 20);
  srand(12345);
  for (auto& v : h) v = (rand() << 16) ^ rand();
  (void)hipMalloc(&rnd, sizeof(int) << 20);
  (void)hipMalloc(&out, sizeof(int) * 256 * 1024);
  (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * 1024);
  (void)hipMemcpy(rnd, h.data(), sizeof(int) << 20, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>("32x32x32, two waves per SIMD", rnd, out, clk, 512, 2.0);
    run<1, 0>("16x16x64, two waves per SIMD", rnd, out, clk, 512, 2.0);
    run<0, 1>("32x32x32 interleaved, two waves", rnd, out, clk, 512, 2.0);
    run<1, 1>("16x16x64 interleaved, two waves", rnd, out, clk, 512, 2.0);
  }
  // where the energy goes: the MFMAs alone (operands in registers), + the LDS reads, + the epilogue
  run<0, 0, 0, 0>("32x32x32 MFMA only", rnd, out, clk, 512, 2.0);
  run<1, 0, 0, 0>("16x16x64 MFMA only", rnd, out, clk, 512, 2.0);
  run<0, 0, 1, 0>("32x32x32 MFMA + LDS reads", rnd, out, clk, 512, 2.0);
  run<1, 0, 1, 0>("16x16x64 MFMA + LDS reads", rnd, out, clk, 512, 2.0);
  run<0, 0, 0, 1>("32x32x32 MFMA + epilogue", rnd, out, clk, 512, 2.0);
  run<1, 0, 0, 1>("16x16x64 MFMA + epilogue", rnd, out, clk, 512, 2.0);
  run<0, 1>("32x32x32 interleaved, one wave", rnd, out, clk, 256, 1.5);
  run<1, 1>("16x16x64 interleaved, one wave", rnd, out, clk, 256, 1.5);
  return 0;
}
