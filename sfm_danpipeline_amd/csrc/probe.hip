// probe.hip -- two measurement probes that bench.py runs on the box it benches (include/sfmhip.h, "measurement probes").
// Not on the product path; they exist because a roofline fraction against the NOMINAL i8 MFMA peak (5 POP/s at 2.4 GHz)
// says little on a part that lowers its clock under matrix load: what the chip SUSTAINS for bare v_mfma_i32_32x32x32_i8 on
// random operands, and the clock it holds while the k-NN sweep runs, belong on the bench line next to the fraction.
//   sfmhip_probe_i8_mfma_peak   every SIMD issues chained i8 MFMAs from registers (random operands, two workgroups of four
//                               waves per CU, the sweep's occupancy) for the given time; returns operations per second
//                               and the shader clock the launch held (s_memtime / s_memrealtime of one lane).
//   sfmhip_probe_clock_*        a one-wave kernel that watches the two counters for a given time on ITS stream while the
//                               caller runs what it wants measured on another: the chip's shader clock under that load,
//                               without touching the measured kernel.
// The matcher they qualify: reference src/Sfm.cpp:593-599 (cv::BFMatcher::knnMatch), csrc/match.hip knn_kernel.
#include "common.h"
#include <algorithm>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void probe_mfma_i8(const int* __restrict__ rnd, int* __restrict__ out,
                                                        unsigned long long* __restrict__ clk, int iters) {
  v4i a[4], b[8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    a[k] = v4i{rnd[(threadIdx.x * 16 + 4 * k + blockIdx.x * 131) & 0xFFFFF], rnd[(threadIdx.x * 16 + 4 * k + 1 + blockIdx.x * 131) & 0xFFFFF],
               rnd[(threadIdx.x * 16 + 4 * k + 2 + blockIdx.x * 131) & 0xFFFFF], rnd[(threadIdx.x * 16 + 4 * k + 3 + blockIdx.x * 131) & 0xFFFFF]};
#pragma unroll
  for (int k = 0; k < 8; ++k)
    b[k] = v4i{rnd[(threadIdx.x * 32 + 4 * k + blockIdx.x * 977 + 7) & 0xFFFFF], rnd[(threadIdx.x * 32 + 4 * k + 1 + blockIdx.x * 977 + 7) & 0xFFFFF],
               rnd[(threadIdx.x * 32 + 4 * k + 2 + blockIdx.x * 977 + 7) & 0xFFFFF], rnd[(threadIdx.x * 32 + 4 * k + 3 + blockIdx.x * 977 + 7) & 0xFFFFF]};
  v16i c0, c1;
#pragma unroll
  for (int e = 0; e < 16; ++e) c0[e] = c1[e] = e;
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  for (int it = 0; it < iters; ++it) {
    // two independent accumulator chains of four MFMAs: the matrix pipe never waits for a result
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[k], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[4 + k], c1, 0, 0, 0);
    }
    // (keep the accumulators bounded and the operands live without adding vector work per MFMA)
    if ((it & 255) == 255) {
#pragma unroll
      for (int e = 0; e < 16; ++e) c0[e] &= 0xFFFF, c1[e] &= 0xFFFF;
    }
  }
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  int s = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) s ^= c0[e] ^ c1[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one wave: until `ticks` of the constant 100 MHz counter have passed
__global__ __launch_bounds__(64) void probe_clock(unsigned long long ticks, unsigned long long* __restrict__ out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r = r0;
  while (r - r0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    r = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0) {
    out[0] = __builtin_amdgcn_s_memtime() - t0;
    out[1] = r - r0;
  }
}

}  // namespace

// (the probes are bounded: a caller's typo must not park a kernel on the device until a watchdog fires)
static const double PROBE_MAX_SECONDS = 1.0;

namespace {
struct ProbeBufs {  // freed on every way out of sfmhip_probe_i8_mfma_peak
  int* rnd = nullptr;
  int* out = nullptr;
  unsigned long long* clk = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~ProbeBufs() {
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (rnd) hipFree(rnd);
    if (out) hipFree(out);
    if (clk) hipFree(clk);
  }
};
}  // namespace

extern "C" int sfmhip_probe_i8_mfma_peak(sfmhip_ctx* ctx, double seconds, double* ops_per_s, double* shader_ghz) {
  if (!ctx || !(seconds > 0) || !ops_per_s) return SFMHIP_ERR_ARG;
  seconds = std::min(seconds, PROBE_MAX_SECONDS);
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  const int grid = 2 * ctx->n_cu;
  ProbeBufs pb;
  SFM_TRY(sfm_dev_alloc(&pb.rnd, (size_t)1 << 20));
  SFM_TRY(sfm_dev_alloc(&pb.out, (size_t)grid * 256));
  SFM_TRY(sfm_dev_alloc(&pb.clk, (size_t)grid * 2));
  int* rnd = pb.rnd;
  int* out = pb.out;
  unsigned long long* clk = pb.clk;
  std::vector<int> h((size_t)1 << 20);
  unsigned long long x = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) {  // random bytes: the power an MFMA draws depends on the operands
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    v = (int)(x >> 16);
  }
  SFM_HIP_TRY(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  SFM_HIP_TRY(hipEventCreate(&pb.e0));
  SFM_HIP_TRY(hipEventCreate(&pb.e1));
  hipEvent_t e0 = pb.e0, e1 = pb.e1;
  hipStream_t st = ctx->stream;
  // calibrate (a short launch), then one launch of about the requested length: long enough for the clock to settle
  int iters = 2000;
  float ms = 0;
  for (int pass = 0; pass < 2; ++pass) {
    SFM_HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(probe_mfma_i8, dim3(grid), dim3(256), 0, st, rnd, out, clk, iters);
    SFM_HIP_TRY(hipEventRecord(e1, st));
    SFM_HIP_TRY(hipEventSynchronize(e1));
    SFM_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) iters = (int)std::max(2000.0, std::min(4.0e7, iters * (seconds * 1e3 / std::max(ms, 1e-3f))));
  }
  // per wave and iteration: 8 MFMAs of 32 x 32 x 32 multiply-adds
  const double ops = (double)grid * 4 * (double)iters * 8 * (2.0 * 32 * 32 * 32);
  *ops_per_s = ops / (ms * 1e-3);
  if (shader_ghz) {
    unsigned long long c2[2] = {0, 1};
    SFM_HIP_TRY(hipMemcpy(c2, clk, sizeof c2, hipMemcpyDeviceToHost));
    *shader_ghz = c2[1] ? (double)c2[0] / (double)c2[1] * 0.1 : 0.0;  // (s_memrealtime counts at 100 MHz)
  }
  return SFMHIP_OK;
}

// (the sampler's two counters land in a pinned pair of the probe's own, one per context -- two contexts sampling at once no
// longer share it; the context's pinned scratch belongs to the entry points that copy through it)
#include <map>
#include <mutex>
static std::mutex g_probe_mu;
static std::map<sfmhip_ctx*, unsigned long long*> g_probe_pins;
static int probe_pin_of(sfmhip_ctx* ctx, bool create, unsigned long long** out) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  auto it = g_probe_pins.find(ctx);
  if (it == g_probe_pins.end()) {
    if (!create) return SFMHIP_ERR_ARG;
    unsigned long long* p = nullptr;
    SFM_HIP_TRY(hipHostMalloc((void**)&p, 64, hipHostMallocDefault));
    it = g_probe_pins.emplace(ctx, p).first;
  }
  *out = it->second;
  return SFMHIP_OK;
}

extern "C" int sfmhip_probe_clock_start(sfmhip_ctx* ctx, double seconds) {
  if (!ctx || !(seconds > 0)) return SFMHIP_ERR_ARG;
  seconds = std::min(seconds, PROBE_MAX_SECONDS);
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  unsigned long long* pin = nullptr;
  SFM_TRY(probe_pin_of(ctx, true, &pin));
  pin[0] = pin[1] = 0;
  void* dev = nullptr;
  SFM_HIP_TRY(hipHostGetDevicePointer(&dev, pin, 0));
  hipLaunchKernelGGL(probe_clock, dim3(1), dim3(64), 0, ctx->stream, (unsigned long long)(seconds * 1e8), (unsigned long long*)dev);
  SFM_HIP_TRY(hipGetLastError());
  return SFMHIP_OK;
}

extern "C" int sfmhip_probe_clock_read(sfmhip_ctx* ctx, double* shader_ghz) {
  if (!ctx || !shader_ghz) return SFMHIP_ERR_ARG;
  unsigned long long* pin = nullptr;
  SFM_TRY(probe_pin_of(ctx, false, &pin));
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *shader_ghz = pin[1] ? (double)pin[0] / (double)pin[1] * 0.1 : 0.0;  // (s_memrealtime counts at 100 MHz)
  return SFMHIP_OK;
}
