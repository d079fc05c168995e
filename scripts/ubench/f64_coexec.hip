// Micro-benchmark (diagnostic, gfx950): do f64 MFMAs and f64 vector instructions co-execute on a SIMD, or do they share one
// budget?  A: 8 independent v_mfma_f64_16x16x4 per round; B: 32 independent v_fma_f64 per round; C: both interleaved (one MFMA
// per four FMAs) in ONE wave; D: two waves per SIMD, one of A, one of B.  Cycles from s_memtime of one wave, one workgroup per CU.
// Also: the 16-lane row sum by two v_mfma_f64_4x4x4_4b against ones (layout check).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  v4d acc[8];
  double x[32];
  for (int e = 0; e < 8; ++e) acc[e] = v4d{1.0, 2.0, 3.0, 4.0};
  for (int e = 0; e < 32; ++e) x[e] = threadIdx.x * 0.001 + e;
  const double a = 1.0000001, b = 1e-9;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4), do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (do_m) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[e]) : "v"(a), "v"(b));
      if (do_v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[4 * e + j]) : "v"(a), "v"(b));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
  for (int e = 0; e < 8; ++e) s += acc[e][0] + acc[e][3];
  for (int e = 0; e < 32; ++e) s += x[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
template <int MODE>
static void run(const char* name, int threads) {
  double* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * 256 * 512);
  (void)hipMalloc(&cyc, 64);
  const int iters = 2000;
  k<MODE><<<256, threads>>>(out, cyc, 100);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-56s cycles per round (8 MFMA and/or 32 FMA): wave0 %.1f  last wave %.1f\n", name, (double)h[0] / iters, (double)h[threads / 64 - 1] / iters);
  (void)hipFree(out);
  (void)hipFree(cyc);
}
__global__ void rowsum(const double* in, double* o1, double* o2, double* o3) {
  const double x = in[threadIdx.x], one = 1.0;
  double d1 = 0, d2 = 0, d3 = 0;
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=v"(d1) : "v"(x), "v"(one));
  asm volatile("s_nop 7\n\ts_nop 7");
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=v"(d2) : "v"(d1), "v"(one));  // stage 1 as the A operand again
  asm volatile("s_nop 7\n\ts_nop 7");
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=v"(d3) : "v"(one), "v"(d1));  // stage 1 as the B operand
  asm volatile("s_nop 7\n\ts_nop 7");
  o1[threadIdx.x] = d1;
  o2[threadIdx.x] = d2;
  o3[threadIdx.x] = d3;
}
int main() {
  run<0>("A  8 MFMA f64 16x16x4, 1 wave/SIMD", 256);
  run<1>("B  32 v_fma_f64, 1 wave/SIMD", 256);
  run<2>("C  both interleaved in one wave, 1 wave/SIMD", 256);
  run<3>("D  2 waves/SIMD: waves 0-3 MFMA, waves 4-7 FMA", 512);
  run<0>("A2 8 MFMA, 2 waves/SIMD", 512);
  run<1>("B2 32 FMA, 2 waves/SIMD", 512);
  run<2>("C2 both interleaved, 2 waves/SIMD", 512);
  double h[64], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = (double)(1 << (i & 15)) + 65536.0 * (i >> 4);
  (void)hipMalloc(&d, 512);
  (void)hipMalloc(&o, 3 * 512);
  (void)hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  rowsum<<<1, 64>>>(d, o, o + 64, o + 128);
  double r[192];
  (void)hipMemcpy(r, o, 3 * 512, hipMemcpyDeviceToHost);
  for (int s = 0; s < 3; ++s) {
    printf("rowsum stage %d:", s);
    for (int i = 0; i < 20; ++i) printf(" %.0f", r[64 * s + i]);
    printf(" ... lane 63: %.0f\n", r[64 * s + 63]);
  }
  printf("(want per row of 16 lanes: 65535 + 16*65536*row = 65535, 1114111, 2162687, 3211263)\n");
  return 0;
}
