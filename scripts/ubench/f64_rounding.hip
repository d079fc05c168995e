// f64_rounding.hip -- are the device's f64 sqrt / divide / fma correctly rounded, i.e. bit-equal to the host's?  And how far is
// ocml's hypot from the FMA kernel of glibc >= 2.35's hypot?  (The five-point solver of score.hip must follow its CPU checker
// operation for operation: ill-conditioned samples amplify a last-bit difference to 1e-3 in E.)
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o f64_rounding f64_rounding.hip && ./f64_rounding
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
__global__ void k(const double* a, const double* b, double* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o[i] = sqrt(a[i]);
  o[n + i] = a[i] / b[i];
  o[2 * n + i] = fma(a[i], b[i], -a[i]);
  o[3 * n + i] = hypot(a[i], b[i]);
  const double x = fabs(a[i]), y = fabs(b[i]);
  const double ax = x < y ? y : x, ay = x < y ? x : y;
  const double t1 = ay + ay, t2 = ax - ay;
  o[4 * n + i] = t1 >= ax ? sqrt(fma(t1, ax, t2 * t2)) : sqrt(fma(ax, ax, ay * ay));
  o[5 * n + i] = 1.0 / b[i];
}
int main() {
  const int n = 1 << 22;
  std::mt19937_64 g(7);
  std::vector<double> a(n), b(n), o(6 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    a[i] = std::ldexp(std::generate_canonical<double, 53>(g) + 0.5, (int)(g() % 40) - 20);
    b[i] = std::ldexp(std::generate_canonical<double, 53>(g) + 0.5, (int)(g() % 40) - 20) * ((g() & 1) ? 1 : -1);
  }
  double *da, *db, *dout;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(da, db, dout, n);
  hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  long bad[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double x = std::fabs(a[i]), y = std::fabs(b[i]);
    const double ax = x < y ? y : x, ay = x < y ? x : y, t1 = ay + ay, t2 = ax - ay;
    const double h[6] = {std::sqrt(a[i]), a[i] / b[i], std::fma(a[i], b[i], -a[i]), std::hypot(a[i], b[i]),
                         t1 >= ax ? std::sqrt(std::fma(t1, ax, t2 * t2)) : std::sqrt(std::fma(ax, ax, ay * ay)), 1.0 / b[i]};
    for (int j = 0; j < 6; ++j) bad[j] += std::memcmp(&h[j], &o[(size_t)j * n + i], 8) != 0;
  }
  printf("n %d  differing results: sqrt %ld  div %ld  fma %ld  hypot(ocml vs libm) %ld  hypot(fma kernel, device vs host) %ld  rcp %ld\n", n, bad[0], bad[1],
         bad[2], bad[3], bad[4], bad[5]);
  long hk = 0;
  for (int i = 0; i < n; ++i) {
    const double x = std::fabs(a[i]), y = std::fabs(b[i]);
    const double ax = x < y ? y : x, ay = x < y ? x : y, t1 = ay + ay, t2 = ax - ay;
    const double kk = t1 >= ax ? std::sqrt(std::fma(t1, ax, t2 * t2)) : std::sqrt(std::fma(ax, ax, ay * ay)), l = std::hypot(a[i], b[i]);
    hk += std::memcmp(&kk, &l, 8) != 0;
  }
  printf("host: fma kernel vs libm hypot differ on %ld\n", hk);
  return 0;
}
