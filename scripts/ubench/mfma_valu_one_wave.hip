// Does ONE wave's vector work run under its OWN matrix instructions?  Per iteration: 8 v_mfma_i32_32x32x32_i8 (two accumulator blocks)
// and 0 / 19 / 38 v_max3_i32 on registers the MFMAs do not touch, interleaved; one or two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_one_wave mfma_valu_one_wave.hip && ./mfma_valu_one_wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

template <int NV>
__global__ __launch_bounds__(512) void k(int* out, unsigned long long* cyc, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  v16i A, B;
  int s[20];
  for (int e = 0; e < 16; ++e) A[e] = e, B[e] = -e;
  for (int e = 0; e < 20; ++e) s[e] = (int)threadIdx.x * e;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g & 1) B = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, B, 0, 0, 0);
      else A = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, A, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = (g * NV + v) % 19;
        s[e] = imax3(s[e], s[e + 1], i + v);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int r = 0;
  for (int e = 0; e < 16; ++e) r += A[e] + B[e];
  for (int e = 0; e < 20; ++e) r += s[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// the same with the vector work READING the accumulator block that the previous chain of eight MFMAs filled (K1's ping-pong of X and Y)
template <int NV>
__global__ __launch_bounds__(512) void k2(int* out, unsigned long long* cyc, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  v16i X, Y;
  int s[16];
  for (int e = 0; e < 16; ++e) X[e] = e, Y[e] = -e, s[e] = (int)threadIdx.x * e;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i += 2) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {  // chain into X, epilogue on Y
      X = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, X, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = (g * NV + v) % 15;
        s[e] = imax3(s[e], Y[e], Y[e + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {  // chain into Y, epilogue on X
      Y = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, Y, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int e = (g * NV + v) % 15;
        s[e] = imax3(s[e], X[e], X[e + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int r = 0;
  for (int e = 0; e < 16; ++e) r += X[e] + Y[e] + s[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV>
void run2(int* out, unsigned long long* cyc) {
  const int iters = 20000;
  for (int waves = 4; waves <= 8; waves += 4) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k2<NV>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
      (void)hipDeviceSynchronize();
    }
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d wave(s) per SIMD, %d v_max3 per MFMA on the OTHER accumulator block: %.1f ticks per MFMA per wave\n", waves / 4, NV, (double)c / (8.0 * iters));
  }
}

template <int NV>
void run(int* out, unsigned long long* cyc) {
  const int iters = 20000;
  for (int waves = 4; waves <= 8; waves += 4) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<NV>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d wave(s) per SIMD, %d v_max3 per MFMA: %.1f ticks per MFMA per wave; %.3f ms = %.2f POP/s, %.0f ticks per us\n", waves / 4, NV,
           (double)c / (8.0 * iters), ms, 256.0 * waves * 8.0 * iters * 65536.0 / (ms * 1e-3) / 1e15, (double)c / (ms * 1e3));
  }
}

int main() {
  int* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&cyc, 8);
  run<0>(out, cyc);
  run<2>(out, cyc);
  run<5>(out, cyc);
  run<8>(out, cyc);
  run2<2>(out, cyc);
  run2<5>(out, cyc);
  return 0;
}
