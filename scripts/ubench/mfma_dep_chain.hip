// How fast does ONE wave issue v_mfma_i32_32x32x32_i8 when consecutive MFMAs accumulate into the same registers (K1's chain: four
// k-steps of tile A, then four of tile B) and when they alternate between two accumulators (A B A B ...)?  One or two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o mfma_dep_chain mfma_dep_chain.hip && ./mfma_dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int* out, unsigned long long* cyc, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  v16i A, B;
  for (int e = 0; e < 16; ++e) A[e] = e, B[e] = -e;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // A A A A B B B B
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) A = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, A, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) B = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, B, 0, 0, 0);
    } else {  // A B A B A B A B
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        A = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, A, 0, 0, 0);
        B = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, B, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int e = 0; e < 16; ++e) s += A[e] + B[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  int* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int waves = 4; waves <= 8; waves += 4)
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
      }
      unsigned long long c;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%d wave(s) per SIMD, %s: %.1f shader-clock ticks per MFMA per wave\n", waves / 4, mode ? "A B A B A B A B" : "A A A A B B B B",
             (double)c / (8.0 * iters));
    }
  return 0;
}
