// Micro-benchmark (diagnostic): per-op issue rate of the matcher's epilogue ops on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, int mul) {
  unsigned x[8];
  for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 977 + e * 131;
  unsigned y = threadIdx.x * 31 + 7, z = threadIdx.x ^ 0x5555;
  unsigned long long xx[8], yy = y, zz = z;
  typedef double v4d __attribute__((ext_vector_type(4)));
  v4d m4 = {1.0, 2.0, 3.0, 4.0};
  for (int e = 0; e < 8; ++e) xx[e] = x[e];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(xx[e]) : "v"(yy), "v"(zz));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(xx[e]) : "v"(yy));
        if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(xx[e]) : "v"(yy));
        if (OP == 3) asm volatile("v_rsq_f64 %0, %0" : "+v"(xx[e]));
        if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(xx[e]));
        if (OP == 5) { int t_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t_) : "v"(x[e])); asm volatile("" :: "s"(t_)); }
        if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(xx[0]) : "v"(yy), "v"(zz));
        if (OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(xx[0]) : "v"(yy));
        if (OP == 8) asm volatile("v_rsq_f64 %0, %0" : "+v"(xx[0]));
        if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(y), "v"(z));
        if (OP == 10) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m4) : "v"(yy), "v"(zz));
      }
    }
  }
  unsigned s = 0;
  for (int e = 0; e < 8; ++e) s ^= x[e] ^ (unsigned)xx[e];
  s ^= (unsigned)m4[0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
static void run(const char* name) {
  unsigned* out;
  (void)hipMalloc(&out, sizeof(unsigned) * 1024 * 256);
  for (int w = 1; w <= 2; w *= 2) {
    const int grid = 256 * w, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<grid, 256>>>(out, 100, -512);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<grid, 256>>>(out, iters, -512);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s waves/SIMD %d: %.3f ns per wave-instruction per SIMD\n", name, w, ms * 1e6 / iters / 48 / w);
  }
  (void)hipFree(out);
}
int main() {
  run<0>("v_fma_f64");
  run<1>("v_mul_f64");
  run<2>("v_add_f64");
  run<3>("v_rsq_f64");
  run<4>("v_rcp_f64");
  run<5>("v_readlane");
  run<6>("v_fma_f64 dep chain");
  run<7>("v_mul_f64 dep chain");
  run<8>("v_rsq_f64 dep chain");
  run<9>("v_fma_f32 dep chain");
  run<10>("v_mfma_f64_16x16x4 dep");
  return 0;
}
