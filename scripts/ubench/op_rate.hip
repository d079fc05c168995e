// Micro-benchmark (diagnostic): per-op issue rate of the matcher's epilogue ops on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, int mul) {
  unsigned x[8];
  for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 977 + e * 131;
  unsigned y = threadIdx.x * 31 + 7, z = threadIdx.x ^ 0x5555;
  unsigned long long xx[8], yy = y, zz = z;
  for (int e = 0; e < 8; ++e) xx[e] = x[e];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 3) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[e]));
        if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[e]) : "v"(y) : "vcc");
        if (OP == 5) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 6) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 9) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xx[e]) : "v"(yy), "v"(zz));
        if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 15) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 16) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x[e]) : "v"(y));
        if (OP == 17) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 18) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 19) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 20) asm volatile("v_mov_b32 %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 21) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(x[e]) : "v"(y) : "vcc");
        if (OP == 22) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 23) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 24) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(x[e]));
        if (OP == 25) asm volatile("v_min_f16 %0, %0, %1" : "+v"(x[e]) : "v"(y));
        if (OP == 26) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
        if (OP == 27) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(x[e]) : "v"(y), "v"(z));
      }
    }
  }
  unsigned s = 0;
  for (int e = 0; e < 8; ++e) s ^= x[e] ^ (unsigned)xx[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
static void run(const char* name) {
  unsigned* out;
  (void)hipMalloc(&out, sizeof(unsigned) * 1024 * 256);
  for (int w = 2; w <= 4; w *= 2) {
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
  run<0>("v_add_u32");
  run<1>("v_and_b32");
  run<2>("v_xor_b32");
  run<3>("v_lshlrev_b32");
  run<4>("v_cndmask");
  run<5>("v_min_i32");
  run<6>("v_min_f32");
  run<7>("v_max_f32");
  run<8>("v_med3_f32");
  run<9>("v_min3_f32");
  run<10>("v_add_f32");
  run<11>("v_fma_f32");
  run<12>("v_pk_fma_f32");
  run<13>("v_add3_u32");
  run<14>("v_lshl_or_b32");
  run<15>("v_perm_b32");
  run<16>("v_sub_i32? subrev");
  run<17>("v_min_u16");
  run<18>("v_pk_min_i16");
  run<19>("v_pk_add_u16");
  run<20>("v_mov_b32");
  run<21>("v_cmp_lt_u32");
  run<22>("v_mad_u32_u24");
  run<23>("v_mul_u32_u24");
  run<24>("v_bfe_u32");
  run<25>("v_min_f16");
  run<26>("v_max3_u32");
  run<27>("v_sad_u32");
  return 0;
}
