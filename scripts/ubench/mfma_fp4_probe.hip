// probe: operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (E2M1) A and B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* a, const unsigned* b, float* d, int scale_a, int scale_b) {
  const int l = threadIdx.x;
  v8i A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (int)a[l * 8 + i]; B[i] = (int)b[l * 8 + i]; }
  v4f C = {0, 0, 0, 0};
  v4f D = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, C, 4, 4, 0, scale_a, 0, scale_b);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = D[r];
}
int main() {
  // logical matrices: Am[16][128], Bm[128][16] with values in {-1, +1}
  std::vector<int> Am(16 * 128), Bm(128 * 16);
  srand(7);
  for (auto& v : Am) v = (rand() & 1) ? 1 : -1;
  for (auto& v : Bm) v = (rand() & 1) ? 1 : -1;
  auto enc = [](int v) -> unsigned { return v > 0 ? 0x2u : 0xAu; };
  float ref[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int kk = 0; kk < 128; ++kk) s += Am[i * 128 + kk] * Bm[kk * 16 + j]; ref[i][j] = (float)s; }
  unsigned *da, *db; float* dd;
  hipMalloc(&da, 64 * 8 * 4); hipMalloc(&db, 64 * 8 * 4); hipMalloc(&dd, 64 * 4 * 4);
  for (int hyp = 0; hyp < 4; ++hyp) {
    std::vector<unsigned> a(64 * 8, 0), b(64 * 8, 0);
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        int kk;
        if (hyp == 0) kk = 32 * (l >> 4) + j;                       // 32 consecutive k per lane
        else if (hyp == 1) kk = (j < 16) ? 16 * (l >> 4) + j : 64 + 16 * (l >> 4) + (j - 16);  // two K=64 halves
        else if (hyp == 2) kk = 4 * j + (l >> 4);                    // interleaved
        else kk = 8 * (l >> 4) + (j & 7) + 32 * (j >> 3);            // four K=32 quarters
        const int row = l & 15;
        a[l * 8 + j / 8] |= enc(Am[row * 128 + kk]) << (4 * (j % 8));
        b[l * 8 + j / 8] |= enc(Bm[kk * 16 + row]) << (4 * (j % 8));
      }
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    for (int sb = 0; sb < 2; ++sb) {
      k<<<1, 64>>>(da, db, dd, 127, sb ? 133 : 127);
      std::vector<float> d(256);
      hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
      int bad = 0, badT = 0;
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int col = l & 15, row = (l >> 4) * 4 + r;
        const float want = ref[row][col] * (sb ? 64.f : 1.f);
        if (d[l * 4 + r] != want) ++bad;
        if (d[l * 4 + r] != ref[col][row] * (sb ? 64.f : 1.f)) ++badT;
      }
      printf("hyp %d scale_b %d: mismatches %d (transposed C map: %d)  d[0..3] = %g %g %g %g  ref[0][0] %g ref[1][0] %g\n", hyp, sb ? 133 : 127, bad, badT, d[0], d[1], d[2], d[3], ref[0][0], ref[1][0]);
    }
  }
  return 0;
}
