// Micro-benchmark (diagnostic, gfx950): what one pivot of a Cholesky chain costs a lone wave -- v_readlane of the pivot, the
// reciprocal square root with its third-order correction, the scaling, the update of the next column -- and which part of it.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rsqrt_h(double d) {
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
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
  double a = 4.0 + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      double d = a;
      if (MODE == 0 || MODE == 2) d = readlane_f64(a, kk);        // the pivot from its lane
      double r;
      if (MODE == 0 || MODE == 1) r = rsqrt_h(d);                 // full reciprocal square root
      else if (MODE == 2) r = __builtin_amdgcn_rsq(d);            // raw v_rsq_f64 only
      else r = d * 0.25;                                          // MODE 3: one multiplication in its place
      const double l = a * r;                                     // the column
      double lm = l;
      if (MODE == 0) lm = readlane_f64(l, kk + 1);                // the multiplier of the next column from its lane
      b = __builtin_fma(-l, lm, b + 4.5);                         // the next pivot's entry
      a = b;
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
static void run(const char* name) {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 512); (void)hipMalloc(&cyc, 8);
  k<MODE><<<1, 64>>>(out, cyc, 10);
  k<MODE><<<1, 64>>>(out, cyc, 1000);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-70s %.1f cycles per pivot\n", name, (double)h / 8000.0);
}
int main() {
  run<0>("readlane(pivot) + rsq + third-order step + scale + readlane(multiplier) + fma");
  run<1>("no lane exchange: rsq + third-order step + scale + fma");
  run<2>("readlane(pivot) + raw v_rsq_f64 + scale + fma");
  run<3>("one multiplication in the root's place: scale + fma");
  return 0;
}
