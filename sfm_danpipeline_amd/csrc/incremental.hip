// incremental.hip -- the index / compare glue of the incremental loop next to the hot path
//  on gfx950:
//   sfmhip_find_2d3d          2D-3D association of find2D3DMatches   reference src/Sfm.cpp:1047-1090
//   sfmhip_merge_new_points   mergeNewPoints                          reference src/Sfm.cpp:1212-1244
// Both are byte/index work bound by HBM (and tiny at the reference's sizes); the reference's
// O(cloud x matches) and O(new x cloud) loops become a hash-free first-match table + ordered
// compaction, and an LDS-tiled all-pairs threshold test.  Results are bit-identical to the
// reference loops (integer logic; the distance test restates cv::norm's operation order, no FMA:
// this file is compiled with -ffp-contract=off).
#include "common.h"
#include <cmath>
#include <vector>

namespace {

constexpr int BLK = 256;
constexpr int EMPTY = 0x7F7F7F7F;  // hipMemset pattern: larger than any match index

// ---------------------------------------------------------------- find_2d3d
// first[f] = index of the first match (in match order) whose key feature is f
__global__ __launch_bounds__(BLK) void first_match_kernel(const int* __restrict__ key, int n_match,
                                                          int* __restrict__ first, int tbl_n) {
  const int m = blockIdx.x * BLK + threadIdx.x;
  if (m < n_match) {
    const int f = key[m];
    if (f >= 0 && f < tbl_n) atomicMin(first + f, m);
  }
}

// per cloud point: the feature it has in done_view (tracks are CSR, ascending view), the first
// match on that feature, the feature on the other side.  Writes flag + value, counts per block.
__global__ __launch_bounds__(BLK) void assoc_kernel(const int* __restrict__ trk_ptr, const int* __restrict__ trk_view,
                                                    const int* __restrict__ trk_feat, int n_cloud, int done_view,
                                                    const int* __restrict__ first, int tbl_n,
                                                    const int* __restrict__ other, int* __restrict__ val,
                                                    int* __restrict__ block_cnt) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  int v = -1;
  if (p < n_cloud) {
    for (int e = trk_ptr[p]; e < trk_ptr[p + 1]; ++e) {
      if (trk_view[e] != done_view) continue;
      const int f = trk_feat[e];
      if (f >= 0 && f < tbl_n) {
        const int m = first[f];
        if (m != EMPTY) v = other[m];
      }
      break;  // idxImage is a map: one entry per view
    }
    val[p] = v;
  }
  __shared__ int wsum[BLK / 64];
  const unsigned long long b = __ballot(v >= 0);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block counts (single workgroup) and the total
__global__ __launch_bounds__(1024) void scan_kernel(int* __restrict__ cnt, int n, int* __restrict__ total) {
  __shared__ int carry;
  __shared__ int wsum[16];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int x = i < n ? cnt[i] : 0;
    int s = x;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(s, o);
      if ((int)(threadIdx.x & 63) >= o) s += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < n) cnt[i] = off + s - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + s;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// ordered compaction: survivors keep the input order
__global__ __launch_bounds__(BLK) void compact_pairs_kernel(const int* __restrict__ val, int n,
                                                            const int* __restrict__ block_off,
                                                            int* __restrict__ out_idx, int* __restrict__ out_val) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  const int v = p < n ? val[p] : -1;
  __shared__ int wsum[BLK / 64];
  const unsigned long long b = __ballot(v >= 0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wsum[wave] = __popcll(b);
  __syncthreads();
  int off = block_off[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += wsum[w];
  if (v >= 0) {
    const int o = off + __popcll(b & ((1ull << lane) - 1ull));
    out_idx[o] = p;
    out_val[o] = v;
  }
}

// ---------------------------------------------------------------- merge_new_points
// sqrt(s) < r  <=>  s <= thr, thr = the largest double whose (correctly rounded) sqrt is < r
__device__ __forceinline__ double dist2(const double* a, double qx, double qy, double qz) {
  const double dx = a[0] - qx, dy = a[1] - qy, dz = a[2] - qz;
  return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));  // cv::norm's order
}

// near[i] = 1 iff some point of the existing cloud is closer than r to new point i.  Grid =
// (blocks of 256 new points) x (slices of CLOUD_SLICE cloud points), so that a few hundred new
// points against a large cloud still fill the chip; slices only ever store 1 (idempotent).
constexpr int CLOUD_SLICE = 1024;
__global__ __launch_bounds__(BLK) void near_cloud_kernel(const double* __restrict__ cloud, int n_cloud,
                                                         const double* __restrict__ pts, int n_new, double thr,
                                                         unsigned char* __restrict__ near) {
  __shared__ double s_c[BLK * 3];
  const int i = blockIdx.x * BLK + threadIdx.x;
  const int ii = i < n_new ? i : 0;
  const double qx = pts[3 * (size_t)ii], qy = pts[3 * (size_t)ii + 1], qz = pts[3 * (size_t)ii + 2];
  const int lo = blockIdx.y * CLOUD_SLICE, hi = min(n_cloud, lo + CLOUD_SLICE);
  bool found = false;
  for (int base = lo; base < hi; base += BLK) {
    const int m = min(BLK, hi - base);
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * m; e += BLK) s_c[e] = cloud[3 * (size_t)base + e];
    __syncthreads();
    if (!found)
      for (int j = 0; j < m; ++j) found = found || dist2(s_c + 3 * j, qx, qy, qz) <= thr;
  }
  if (i < n_new && found) near[i] = 1;
}

// One relaxation sweep of the in-order dependency among the new points: state 0 = undecided,
// 1 = appended, 2 = dropped.  Point i is dropped if an appended earlier point lies within r,
// appended once every earlier point within r is dropped.
__global__ __launch_bounds__(BLK) void resolve_kernel(const double* __restrict__ pts, int n_new, double thr,
                                                      unsigned char* __restrict__ state, int* __restrict__ undecided) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n_new || state[i] != 0) return;
  const double qx = pts[3 * (size_t)i], qy = pts[3 * (size_t)i + 1], qz = pts[3 * (size_t)i + 2];
  bool wait = false;
  for (int j = 0; j < i; ++j) {
    const unsigned char sj = state[j];
    if (sj == 2) continue;
    if (dist2(pts + 3 * (size_t)j, qx, qy, qz) <= thr) {
      if (sj == 1) {
        state[i] = 2;
        return;
      }
      wait = true;  // an undecided earlier neighbour: decide in a later sweep
    }
  }
  if (wait) atomicAdd(undecided, 1);
  else state[i] = 1;
}

__global__ void init_state_kernel(const unsigned char* __restrict__ near, unsigned char* __restrict__ state, int n) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) state[i] = near[i] ? 2 : 0;
}

struct DevBuf {  // frees on scope exit
  std::vector<void*> p;
  ~DevBuf() {
    for (void* q : p) hipFree(q);
  }
  template <typename T>
  int alloc(T** out, size_t n) {
    const int rc = sfm_dev_alloc(out, n);
    if (rc == SFMHIP_OK) p.push_back((void*)*out);
    return rc;
  }
};

}  // namespace

extern "C" int sfmhip_find_2d3d(sfmhip_ctx* ctx, const int32_t* trk_ptr, const int32_t* trk_view,
                                const int32_t* trk_feat, int n_cloud, int done_view, int new_view,
                                const int32_t* match_q, const int32_t* match_t, int n_match, int32_t* out_cloud,
                                int32_t* out_feat, int32_t* n_out) {
  if (!ctx || !n_out || n_cloud < 0 || n_match < 0) return SFMHIP_ERR_ARG;
  *n_out = 0;
  if (n_cloud == 0 || n_match == 0) return SFMHIP_OK;
  if (!trk_ptr || !match_q || !match_t || !out_cloud || !out_feat) return SFMHIP_ERR_ARG;
  const int n_ent = trk_ptr[n_cloud];
  if (n_ent < 0 || (n_ent && (!trk_view || !trk_feat))) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // "originating view is left" iff done_view < new_view (src/Sfm.cpp:1064): the key side of a match
  const bool left = done_view < new_view;
  const int32_t* key_h = left ? match_q : match_t;
  const int32_t* oth_h = left ? match_t : match_q;
  int tbl_n = 0;
  for (int m = 0; m < n_match; ++m) tbl_n = key_h[m] + 1 > tbl_n ? key_h[m] + 1 : tbl_n;
  if (tbl_n <= 0) return SFMHIP_OK;
  const int nblk = (n_cloud + BLK - 1) / BLK;
  DevBuf buf;
  int *d_ptr, *d_view, *d_feat, *d_key, *d_oth, *d_first, *d_val, *d_cnt, *d_total, *d_oi, *d_ov;
  SFM_TRY(buf.alloc(&d_ptr, (size_t)n_cloud + 1));
  SFM_TRY(buf.alloc(&d_view, (size_t)n_ent));
  SFM_TRY(buf.alloc(&d_feat, (size_t)n_ent));
  SFM_TRY(buf.alloc(&d_key, (size_t)n_match));
  SFM_TRY(buf.alloc(&d_oth, (size_t)n_match));
  SFM_TRY(buf.alloc(&d_first, (size_t)tbl_n));
  SFM_TRY(buf.alloc(&d_val, (size_t)n_cloud));
  SFM_TRY(buf.alloc(&d_cnt, (size_t)nblk));
  SFM_TRY(buf.alloc(&d_total, (size_t)1));
  SFM_TRY(buf.alloc(&d_oi, (size_t)n_cloud));
  SFM_TRY(buf.alloc(&d_ov, (size_t)n_cloud));
  SFM_HIP_TRY(hipMemcpyAsync(d_ptr, trk_ptr, sizeof(int) * ((size_t)n_cloud + 1), hipMemcpyHostToDevice, st));
  if (n_ent) {
    SFM_HIP_TRY(hipMemcpyAsync(d_view, trk_view, sizeof(int) * (size_t)n_ent, hipMemcpyHostToDevice, st));
    SFM_HIP_TRY(hipMemcpyAsync(d_feat, trk_feat, sizeof(int) * (size_t)n_ent, hipMemcpyHostToDevice, st));
  }
  SFM_HIP_TRY(hipMemcpyAsync(d_key, key_h, sizeof(int) * (size_t)n_match, hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemcpyAsync(d_oth, oth_h, sizeof(int) * (size_t)n_match, hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemsetAsync(d_first, 0x7F, sizeof(int) * (size_t)tbl_n, st));  // 0x7F7F7F7F > any index
  hipLaunchKernelGGL(first_match_kernel, dim3((n_match + BLK - 1) / BLK), dim3(BLK), 0, st, d_key, n_match, d_first, tbl_n);
  hipLaunchKernelGGL(assoc_kernel, dim3(nblk), dim3(BLK), 0, st, d_ptr, d_view, d_feat, n_cloud, done_view, d_first, tbl_n,
                     d_oth, d_val, d_cnt);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, d_cnt, nblk, d_total);
  hipLaunchKernelGGL(compact_pairs_kernel, dim3(nblk), dim3(BLK), 0, st, d_val, n_cloud, d_cnt, d_oi, d_ov);
  SFM_HIP_TRY(hipGetLastError());
  int total = 0;
  SFM_HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (total > 0) {
    SFM_HIP_TRY(hipMemcpyAsync(out_cloud, d_oi, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(out_feat, d_ov, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
  }
  *n_out = total;
  return SFMHIP_OK;
}

extern "C" int sfmhip_merge_new_points(sfmhip_ctx* ctx, const double* cloud_xyz, int n_cloud, const double* new_xyz,
                                       int n_new, float min_dist, uint8_t* accept, int32_t* n_accepted) {
  if (!ctx || n_cloud < 0 || n_new < 0 || !n_accepted) return SFMHIP_ERR_ARG;
  *n_accepted = 0;
  if (n_new == 0) return SFMHIP_OK;
  if (!new_xyz || !accept || (n_cloud && !cloud_xyz)) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // threshold on the squared norm equivalent to sqrt(s) < r (sqrt is monotone and correctly rounded)
  const double r = (double)min_dist;
  double thr = r * r;
  if (!(r > 0)) {
    thr = -1.0;  // nothing is closer than a non-positive distance
  } else {
    while (thr > 0 && !(std::sqrt(thr) < r)) thr = std::nextafter(thr, 0.0);
    while (std::sqrt(std::nextafter(thr, INFINITY)) < r) thr = std::nextafter(thr, INFINITY);
  }
  DevBuf buf;
  double *d_cloud = nullptr, *d_new = nullptr;
  unsigned char *d_far = nullptr, *d_state = nullptr;
  int* d_und = nullptr;
  SFM_TRY(buf.alloc(&d_cloud, 3 * (size_t)(n_cloud ? n_cloud : 1)));
  SFM_TRY(buf.alloc(&d_new, 3 * (size_t)n_new));
  SFM_TRY(buf.alloc(&d_far, (size_t)n_new));
  SFM_TRY(buf.alloc(&d_state, (size_t)n_new));
  SFM_TRY(buf.alloc(&d_und, (size_t)1));
  if (n_cloud) SFM_HIP_TRY(hipMemcpyAsync(d_cloud, cloud_xyz, sizeof(double) * 3 * (size_t)n_cloud, hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemcpyAsync(d_new, new_xyz, sizeof(double) * 3 * (size_t)n_new, hipMemcpyHostToDevice, st));
  const int nblk = (n_new + BLK - 1) / BLK;
  SFM_HIP_TRY(hipMemsetAsync(d_far, 0, (size_t)n_new, st));
  if (n_cloud)
    hipLaunchKernelGGL(near_cloud_kernel, dim3(nblk, (n_cloud + CLOUD_SLICE - 1) / CLOUD_SLICE), dim3(BLK), 0, st, d_cloud,
                       n_cloud, d_new, n_new, thr, d_far);
  hipLaunchKernelGGL(init_state_kernel, dim3(nblk), dim3(BLK), 0, st, d_far, d_state, n_new);
  SFM_HIP_TRY(hipGetLastError());
  // sweeps until every point is decided: each sweep decides at least the first undecided point
  for (int sweep = 0; sweep <= n_new; ++sweep) {
    SFM_HIP_TRY(hipMemsetAsync(d_und, 0, sizeof(int), st));
    hipLaunchKernelGGL(resolve_kernel, dim3(nblk), dim3(BLK), 0, st, d_new, n_new, thr, d_state, d_und);
    SFM_HIP_TRY(hipGetLastError());
    int und = 0;
    SFM_HIP_TRY(hipMemcpyAsync(&und, d_und, sizeof(int), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    if (und == 0) break;
  }
  std::vector<unsigned char> h(n_new);
  SFM_HIP_TRY(hipMemcpyAsync(h.data(), d_state, (size_t)n_new, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  int n_acc = 0;
  for (int i = 0; i < n_new; ++i) {
    accept[i] = h[i] == 1 ? 1 : 0;
    n_acc += accept[i];
  }
  *n_accepted = n_acc;
  return SFMHIP_OK;
}
