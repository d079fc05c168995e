// Micro-benchmark (diagnostic): what a dependent kernel costs in a stream and in a replayed hipGraph.  A chain of N kernels
// that each read what the one before wrote (one workgroup, a few hundred bytes): stream launches, the same chain captured
// into a graph and replayed, and one fused kernel doing the N steps itself.
// Build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void step(double* buf, int k) {
  const int i = threadIdx.x;
  buf[(k + 1) * 64 + i] = buf[k * 64 + i] * 1.0000001 + 1.0;
}
__global__ void fat(double* buf, int k, double* scratch, int n) {  // a kernel with real traffic between the small ones
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) scratch[i] = scratch[i] * 0.5 + buf[k * 64 + (i & 63)];
  if (blockIdx.x == 0) buf[(k + 1) * 64 + threadIdx.x % 64] = buf[k * 64 + threadIdx.x % 64] + 1.0;
}
int main() {
  double *buf, *scratch;
  const int N = 16, nsc = 1 << 20;
  (void)hipMalloc(&buf, sizeof(double) * 64 * (N + 2));
  (void)hipMalloc(&scratch, sizeof(double) * nsc);
  (void)hipMemset(buf, 0, sizeof(double) * 64 * (N + 2));
  (void)hipMemset(scratch, 0, sizeof(double) * nsc);
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  auto run_stream = [&](bool with_fat) {
    for (int k = 0; k < N; ++k) {
      if (with_fat && (k & 3) == 3) hipLaunchKernelGGL(fat, dim3(256), dim3(256), 0, st, buf, k, scratch, nsc);
      else hipLaunchKernelGGL(step, dim3(1), dim3(64), 0, st, buf, k);
    }
  };
  for (int with_fat = 0; with_fat < 2; ++with_fat) {
    for (int i = 0; i < 20; ++i) run_stream(with_fat);
    (void)hipStreamSynchronize(st);
    const int reps = 200;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) run_stream(with_fat);
    (void)hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("stream, %s: %.1f us per chain of %d = %.2f us per kernel\n", with_fat ? "every 4th kernel fat" : "tiny kernels", us, N, us / N);
    hipGraph_t g;
    hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    run_stream(with_fat);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("graph,  %s: %.1f us per chain of %d = %.2f us per kernel\n", with_fat ? "every 4th kernel fat" : "tiny kernels", us, N, us / N);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
  }
  return 0;
}
