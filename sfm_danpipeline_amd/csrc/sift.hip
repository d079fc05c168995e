// sift.hip -- the detector / descriptor front end of StructFromMotion::getFeature (reference src/Sfm.cpp:300-330):
// cv::xfeatures2d::SIFT::create(0, 3, 0.04, 10, 1.6)->detectAndCompute on gfx950.
//
// OpenCV 3.4.1's float pipeline (xfeatures2d/src/sift.cpp) stage by stage: the doubled, blurred base image; six
// Gaussian images per octave (separable blur, INTER_NEAREST halving); difference of Gaussians; 26-neighbour extrema
// above the contrast floor; adjustLocalExtrema (Newton steps with Cramer's rule in float, contrast and edge tests);
// orientation histogram and peaks; KeyPointsFilter::removeDuplicatedSorted and the rescale of the doubled octave (host);
// the 4 x 4 x 8 descriptor.  The pyramid kernels are HBM-bound streaming kernels; the keypoint stages are one wave
// per candidate / keypoint with the reference's sequential accumulation order (per histogram bin), so that the numbers are those of the
// numpy restatement the tests hold (test infrastructure).  Parity with OpenCV itself is unpinned: it is not in the
// image; where its float results depend on SIMD paths or its own exp / atan2 approximations, one order of operations
// is fixed here (exp / cos / sin / pow: evaluated in double, rounded to float).
#include "common.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int IMG_BORDER = 5, MAX_INTERP_STEPS = 5, ORI_BINS = 36;
constexpr int DW = 4, DB = 8;  // descriptor: DW x DW spatial bins, DB orientation bins

__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
  }
  return p;
}
__device__ __forceinline__ int cv_round(float v) { return (int)__builtin_rint((double)v); }  // cvRound: half to even
__device__ __forceinline__ int cv_round_d(double v) { return (int)__builtin_rint(v); }

// u8 -> float, x2 INTER_LINEAR (horizontal pass, then vertical; float weights) in one kernel
__global__ void sift_base_up2(const unsigned char* __restrict__ src, int h, int w, float* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= 2 * w) return;
  auto tap = [](int d, int dn, int sn, int& i0, int& i1, float& a0, float& a1) {
    float f = (float)(((double)d + 0.5) * ((double)sn / (double)dn) - 0.5);
    int i = (int)floorf(f);
    f -= (float)i;
    if (i < 0) f = 0.f, i = 0;
    if (i >= sn - 1) f = 0.f, i = sn - 1;
    i0 = i;
    i1 = min(i + 1, sn - 1);
    a0 = 1.f - f;
    a1 = f;
  };
  int x0, x1, y0, y1;
  float a0, a1, b0, b1;
  tap(x, 2 * w, w, x0, x1, a0, a1);
  tap(y, 2 * h, h, y0, y1, b0, b1);
  const float h0 = (float)src[(size_t)y0 * w + x0] * a0 + (float)src[(size_t)y0 * w + x1] * a1;
  const float h1 = (float)src[(size_t)y1 * w + x0] * a0 + (float)src[(size_t)y1 * w + x1] * a1;
  dst[(size_t)y * (2 * w) + x] = h0 * b0 + h1 * b1;
}
// row pass: sum_k k[k] * s[x + k - r], k = 0..n-1 in order
__global__ void sift_blur_rows(const float* __restrict__ src, int h, int w, const float* __restrict__ kern, int n,
                               float* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int r = n / 2;
  const float* row = src + (size_t)y * w;
  float s = 0.f;
  for (int k = 0; k < n; ++k) {
    const float t = kern[k] * row[reflect101(x + k - r, w)];
    s = k == 0 ? t : s + t;
  }
  dst[(size_t)y * w + x] = s;
}
// column pass (symmetric kernel): k[r] * s[c], then += k[r + j] * (s[c + j] + s[c - j])
__global__ void sift_blur_cols(const float* __restrict__ src, int h, int w, const float* __restrict__ kern, int n,
                               float* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int r = n / 2;
  float s = kern[r] * src[(size_t)y * w + x];
  for (int j = 1; j <= r; ++j)
    s = s + kern[r + j] * (src[(size_t)reflect101(y + j, h) * w + x] + src[(size_t)reflect101(y - j, h) * w + x]);
  dst[(size_t)y * w + x] = s;
}
__global__ void sift_half_nearest(const float* __restrict__ src, int h, int w, float* __restrict__ dst, int dh, int dw) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dw) return;
  const int sy = min((int)floor((double)y * ((double)h / (double)dh)), h - 1);
  const int sx = min((int)floor((double)x * ((double)w / (double)dw)), w - 1);
  dst[(size_t)y * dw + x] = src[(size_t)sy * w + sx];
}
struct Pyr {  // per octave: image size and the offsets of its Gaussian / DoG images in the two arenas
  int h[16], w[16];
  size_t goff[16], doff[16];
  int n_oct, n_layers;
};
// (the small octaves are a few thousand pixels: one launch for all octaves, a table of first blocks per octave)
struct OctBlocks {
  int first[17];  // first block of octave o; first[n_oct] = all blocks
};
__device__ __forceinline__ int oct_of_block(const OctBlocks& t, int n_oct, int b) {
  int o = 0;
  while (o + 1 < n_oct && b >= t.first[o + 1]) ++o;
  return o;
}

// DoG: D[o][i] = G[o][i + 1] - G[o][i], i < nl + 2, all octaves
__global__ __launch_bounds__(256) void sift_sub_all(Pyr P, OctBlocks t, const float* __restrict__ G, float* __restrict__ D) {
  const int o = oct_of_block(t, P.n_oct, blockIdx.x);
  const size_t isz = (size_t)P.h[o] * P.w[o], n = (size_t)(P.n_layers + 2) * isz;
  const size_t i = (size_t)(blockIdx.x - t.first[o]) * 256 + threadIdx.x;
  if (i < n) D[P.doff[o] + i] = G[P.goff[o] + isz + i] - G[P.goff[o] + i];
}

struct Cand {
  int o, layer, r, c;
};
// 26-neighbour extrema of DoG layer `layer` of one octave (prev, cur, next), |val| > thr
// (all octaves and layers in one launch: block -> octave (table), then layer, row, 64-column block)
__global__ __launch_bounds__(64) void sift_extrema(Pyr P, OctBlocks t, const float* __restrict__ D, float thr,
                                                   Cand* __restrict__ out, int* __restrict__ n_out, int cap) {
  const int o = oct_of_block(t, P.n_oct, blockIdx.x);
  const int h = P.h[o], w = P.w[o];
  if (h <= 2 * IMG_BORDER || w <= 2 * IMG_BORDER) return;
  const int bx = (w - 2 * IMG_BORDER + 63) / 64, by = h - 2 * IMG_BORDER;
  int rem = blockIdx.x - t.first[o];
  const int layer = 1 + rem / (bx * by);
  rem -= (layer - 1) * (bx * by);
  const int yb = rem / bx, xb = rem - yb * bx;
  const size_t isz = (size_t)h * w;
  const float* cur = D + P.doff[o] + (size_t)layer * isz;
  const float *prv = cur - isz, *nxt = cur + isz;
  const int c = IMG_BORDER + xb * 64 + threadIdx.x, r = IMG_BORDER + yb;
  if (c >= w - IMG_BORDER || r >= h - IMG_BORDER) return;
  const float val = cur[(size_t)r * w + c];
  if (!(fabsf(val) > thr)) return;
  bool mx = val > 0, mn = val < 0;
  for (int dy = -1; dy <= 1 && (mx || mn); ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const size_t i = (size_t)(r + dy) * w + c + dx;
      const float a = prv[i], b = cur[i], d = nxt[i];
      mx = mx && val >= a && val >= b && val >= d;
      mn = mn && val <= a && val <= b && val <= d;
    }
  if (mx || mn) {
    const int k = atomicAdd(n_out, 1);
    if (k < cap) out[k] = Cand{o, layer, r, c};
  }
}

// cv::fastAtan2 (degrees): the polynomial of OpenCV's mathfuncs_core
__device__ float fast_atan2_deg(float y, float x) {
  const float p1 = (float)(0.9997878412794807 * 57.29577951308232), p3 = (float)(-0.3258083974640975 * 57.29577951308232),
              p5 = (float)(0.1555786518463281 * 57.29577951308232), p7 = (float)(-0.04432655554792128 * 57.29577951308232);
  const float ax = fabsf(x), ay = fabsf(y), eps = (float)DBL_EPSILON;
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct KeyPt {
  float x, y, size, angle, response;
  int octave;
};

// Matx33f::solve(b, DECOMP_LU): Cramer's rule in float
__device__ bool solve3(const float a[3][3], const float b[3], float x[3]) {
  float d = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
            a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
  if (d == 0) {
    x[0] = x[1] = x[2] = 0.f;
    return false;
  }
  d = 1.f / d;
  x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
              a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
  x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
              a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
  x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
              b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
  return true;
}

// adjustLocalExtrema + calcOrientationHist + the peaks: one WAVE per candidate.  The Newton steps run on every lane
// alike (uniform data, a few dozen loads); the orientation window -- OpenCV fills arrays for all its samples and
// then adds them to the histogram in sample order -- is computed 64 samples at a time, a lane each, and added by
// bin: lane b looks at the 64 (bin, value) pairs in lane order (v_readlane) and adds those of its bin, so every
// bin sees its samples in the reference's order and the sums are the sequential ones bit for bit.
__global__ __launch_bounds__(256) void sift_refine(Pyr P, const float* __restrict__ G, const float* __restrict__ D,
                                                   const Cand* __restrict__ cand, int n_cand, float contrast_thr, float edge_thr,
                                                   float sigma, KeyPt* __restrict__ out, int* __restrict__ n_out, int cap) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= n_cand) return;
  const Cand cd = cand[t];
  const int o = cd.o, nl = P.n_layers, h = P.h[o], w = P.w[o];
  const size_t isz = (size_t)h * w;
  int layer = cd.layer, r = cd.r, c = cd.c;
  const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
  float xi = 0, xr = 0, xc = 0;
  int i = 0;
  auto dog = [&](int l) { return D + P.doff[o] + (size_t)l * isz; };
  for (; i < MAX_INTERP_STEPS; ++i) {
    const float *img = dog(layer), *prv = dog(layer - 1), *nxt = dog(layer + 1);
    const size_t q = (size_t)r * w + c;
    const float dD[3] = {(img[q + 1] - img[q - 1]) * deriv_scale, (img[q + w] - img[q - w]) * deriv_scale,
                         (nxt[q] - prv[q]) * deriv_scale};
    const float v2 = img[q] * 2.f;
    const float dxx = (img[q + 1] + img[q - 1] - v2) * second_scale, dyy = (img[q + w] + img[q - w] - v2) * second_scale,
                dss = (nxt[q] + prv[q] - v2) * second_scale;
    const float dxy = (img[q + w + 1] - img[q + w - 1] - img[q - w + 1] + img[q - w - 1]) * cross_scale;
    const float dxs = (nxt[q + 1] - nxt[q - 1] - prv[q + 1] + prv[q - 1]) * cross_scale;
    const float dys = (nxt[q + w] - nxt[q - w] - prv[q + w] + prv[q - w]) * cross_scale;
    const float Hm[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
    float X[3];
    solve3(Hm, dD, X);
    xi = -X[2];
    xr = -X[1];
    xc = -X[0];
    if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
    const float big = (float)(INT_MAX / 3);
    if (fabsf(xi) > big || fabsf(xr) > big || fabsf(xc) > big) return;
    c += cv_round(xc);
    r += cv_round(xr);
    layer += cv_round(xi);
    if (layer < 1 || layer > nl || c < IMG_BORDER || c >= w - IMG_BORDER || r < IMG_BORDER || r >= h - IMG_BORDER) return;
  }
  if (i >= MAX_INTERP_STEPS) return;
  float contr;
  {
    const float *img = dog(layer), *prv = dog(layer - 1), *nxt = dog(layer + 1);
    const size_t q = (size_t)r * w + c;
    const float dD[3] = {(img[q + 1] - img[q - 1]) * deriv_scale, (img[q + w] - img[q - w]) * deriv_scale,
                         (nxt[q] - prv[q]) * deriv_scale};
    const float tt = dD[0] * xc + dD[1] * xr + dD[2] * xi;
    contr = img[q] * img_scale + tt * 0.5f;
    if (fabsf(contr) * (float)nl < contrast_thr) return;
    const float v2 = img[q] * 2.f;
    const float dxx = (img[q + 1] + img[q - 1] - v2) * second_scale, dyy = (img[q + w] + img[q - w] - v2) * second_scale;
    const float dxy = (img[q + w + 1] - img[q + w - 1] - img[q - w + 1] + img[q - w - 1]) * cross_scale;
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1.f) * (edge_thr + 1.f) * det) return;
  }
  const float sc = (float)(1 << o);
  KeyPt kp;
  kp.x = ((float)c + xc) * sc;
  kp.y = ((float)r + xr) * sc;
  kp.octave = o + (layer << 8) + (cv_round_d(((double)xi + 0.5) * 255) << 16);
  kp.size = sigma * (float)pow(2.0, (double)(((float)layer + xi) / (float)nl)) * sc * 2.f;
  kp.response = fabsf(contr);
  // ---- orientation histogram on the Gaussian image of (o, layer)
  const float scl_octv = kp.size * 0.5f / sc;
  const int radius = cv_round(4.5f * scl_octv);
  const float sg = 1.5f * scl_octv;
  const float expf_scale = -1.f / (2.f * (sg * sg));
  const float* gimg = G + P.goff[o] + (size_t)layer * isz;
  float acc = 0.f;  // lane b < ORI_BINS: bin b of the raw histogram
  {
    const int W = 2 * radius + 1, total = W * W;
    for (int base = 0; base < total; base += 64) {
      const int k = base + lane;
      int bin = -1;
      float val = 0.f;
      if (k < total) {
        const int ii = k / W - radius, jj = k % W - radius;
        const int y = r + ii, x = c + jj;
        if (!(y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1)) {
          const size_t q = (size_t)y * w + x;
          const float dx = gimg[q + 1] - gimg[q - 1], dy = gimg[q - w] - gimg[q + w];
          const float wgt = (float)exp((double)((float)(ii * ii + jj * jj) * expf_scale));
          const float ori = fast_atan2_deg(dy, dx);
          const float mag = sqrtf(dx * dx + dy * dy);
          bin = cv_round((float)(ORI_BINS / 360.0) * ori);
          if (bin >= ORI_BINS) bin -= ORI_BINS;
          if (bin < 0) bin += ORI_BINS;
          val = wgt * mag;
        }
      }
      if (__builtin_amdgcn_ballot_w64(bin >= 0) == 0) continue;
#pragma unroll 8
      for (int sidx = 0; sidx < 64; ++sidx) {
        const int bs = __builtin_amdgcn_readlane(bin, sidx);
        const float vs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), sidx));
        acc = bs == lane ? acc + vs : acc;
      }
    }
  }
  auto T = [&](int k) {  // raw bin k (mod ORI_BINS), on every lane
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), (k + ORI_BINS) % ORI_BINS));
  };
  float hist[ORI_BINS], omax = 0.f;
#pragma unroll
  for (int b = 0; b < ORI_BINS; ++b) {
    hist[b] = (T(b - 2) + T(b + 2)) * (1.f / 16.f) + (T(b - 1) + T(b + 1)) * (4.f / 16.f) + T(b) * (6.f / 16.f);
    omax = b == 0 ? hist[0] : fmaxf(omax, hist[b]);
  }
  const float mag_thr = omax * 0.8f;
  if (lane != 0) return;
#pragma unroll
  for (int j = 0; j < ORI_BINS; ++j) {
    const int l = j > 0 ? j - 1 : ORI_BINS - 1, r2 = j < ORI_BINS - 1 ? j + 1 : 0;
    if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
      float bin = (float)j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2.f * hist[j] + hist[r2]);
      bin = bin < 0 ? (float)ORI_BINS + bin : bin >= (float)ORI_BINS ? bin - (float)ORI_BINS : bin;
      kp.angle = 360.f - (float)(360.0 / ORI_BINS) * bin;
      if (fabsf(kp.angle - 360.f) < FLT_EPSILON) kp.angle = 0.f;
      const int k = atomicAdd(n_out, 1);
      if (k < cap) out[k] = kp;
    }
  }
}

// calcSIFTDescriptor: one WAVE per keypoint.  OpenCV fills arrays for the window's samples (those that fall into the
// 4 x 4 grid and the image), then adds each sample's eight trilinear shares to the histogram in sample order.  Here 64
// window positions at a time are evaluated a lane each, the valid ones compacted in raster order (ballot + prefix
// count) into LDS with their eight shares, and lanes 0..7 -- one per share -- add them sample by sample: the eight
// bins of a sample are distinct and LDS operations of a wave execute in order, so every bin receives its shares in
// the reference's order.  The norms are sequential sums over the 128 values in order (v_readlane).
constexpr int HIST_N = (DW + 2) * (DW + 2) * (DB + 2);
constexpr int DESC_WAVES = 4;
__global__ __launch_bounds__(64 * DESC_WAVES) void sift_describe(Pyr P, const float* __restrict__ G, const KeyPt* __restrict__ kps,
                                                                 int n, float* __restrict__ desc) {
  __shared__ float s_hist_all[DESC_WAVES][HIST_N];
  __shared__ float s_val_all[DESC_WAVES][64 * 8];
  __shared__ int s_idx_all[DESC_WAVES][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * DESC_WAVES + wave;
  if (t >= n) return;  // (no workgroup barrier below: the waves are independent)
  float* const s_hist = s_hist_all[wave];
  float* const s_val = s_val_all[wave];
  int* const s_idx = s_idx_all[wave];
  const KeyPt kp = kps[t];  // (already rescaled by 0.5 and with the first octave folded in, as OpenCV hands it over)
  int octave = kp.octave & 255;
  const int layer = (kp.octave >> 8) & 255;
  octave = octave < 128 ? octave : (-128 | octave);
  const float scale = octave >= 0 ? 1.f / (float)(1 << octave) : (float)(1 << -octave);
  const float size = kp.size * scale;
  const int o = octave + 1;  // firstOctave = -1
  const int rows = P.h[o], cols = P.w[o];
  const float* img = G + P.goff[o] + (size_t)layer * ((size_t)rows * cols);
  float ori = 360.f - kp.angle;
  if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
  const float scl = size * 0.5f;
  const int px = cv_round(kp.x * scale), py = cv_round(kp.y * scale);
  float cos_t = (float)cos((double)(ori * (float)(3.1415926535897932384626433832795 / 180.0))), sin_t = (float)sin((double)(ori * (float)(3.1415926535897932384626433832795 / 180.0)));
  const float bins_per_rad = (float)(DB / 360.0);
  const float exp_scale = -1.f / (float)(DW * DW * 0.5);
  const float hist_width = 3.f * scl;
  int radius = cv_round(hist_width * 1.4142135623730951f * (float)(DW + 1) * 0.5f);
  radius = min(radius, (int)sqrt((double)cols * cols + (double)rows * rows));
  cos_t /= hist_width;
  sin_t /= hist_width;
  for (int k = lane; k < HIST_N; k += 64) s_hist[k] = 0.f;
  const int W = 2 * radius + 1;
  const long long total = (long long)W * W;
  // the share of lane kk = 4 r + 2 c + o goes to bin idx + koff
  const int koff = (lane & 1) + ((lane >> 1) & 1) * (DB + 2) + ((lane >> 2) & 1) * (DW + 2) * (DB + 2);
  for (long long base = 0; base < total; base += 64) {
    const long long k = base + lane;
    bool valid = false;
    int idx = 0;
    float v[8];
    if (k < total) {
      const int i = (int)(k / W) - radius, j = (int)(k % W) - radius;
      const float c_rot = (float)j * cos_t - (float)i * sin_t, r_rot = (float)j * sin_t + (float)i * cos_t;
      const float rbin = r_rot + (float)(DW / 2) - 0.5f, cbin = c_rot + (float)(DW / 2) - 0.5f;
      const int r = py + i, c = px + j;
      if (rbin > -1 && rbin < DW && cbin > -1 && cbin < DW && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
        valid = true;
        const size_t q = (size_t)r * cols + c;
        const float dx = img[q + 1] - img[q - 1], dy = img[q - cols] - img[q + cols];
        const float wgt = (float)exp((double)((c_rot * c_rot + r_rot * r_rot) * exp_scale));
        const float ang = fast_atan2_deg(dy, dx);
        const float mag = sqrtf(dx * dx + dy * dy);
        float obin = (ang - ori) * bins_per_rad;
        const float mg = mag * wgt;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        const float rb = rbin - (float)r0, cb = cbin - (float)c0, ob = obin - (float)o0;
        if (o0 < 0) o0 += DB;
        if (o0 >= DB) o0 -= DB;
        const float v_r1 = mg * rb, v_r0 = mg - v_r1;
        const float v_rc11 = v_r1 * cb, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cb, v_rc00 = v_r0 - v_rc01;
        v[7] = v_rc11 * ob, v[6] = v_rc11 - v[7], v[5] = v_rc10 * ob, v[4] = v_rc10 - v[5];
        v[3] = v_rc01 * ob, v[2] = v_rc01 - v[3], v[1] = v_rc00 * ob, v[0] = v_rc00 - v[1];
        idx = ((r0 + 1) * (DW + 2) + c0 + 1) * (DB + 2) + o0;
      }
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(valid);
    if (m == 0) continue;
    if (valid) {
      const int pos = __builtin_popcountll(m & ((1ull << lane) - 1ull));
      s_idx[pos] = idx;
#pragma unroll
      for (int e = 0; e < 8; ++e) s_val[8 * pos + e] = v[e];
    }
    __builtin_amdgcn_wave_barrier();
    const int nv = __builtin_popcountll(m);
    if (lane < 8) {
      for (int sidx = 0; sidx < nv; ++sidx) {
        const int bi = s_idx[sidx] + koff;
        s_hist[bi] = s_hist[bi] + s_val[8 * sidx + lane];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the circular orientation bins, then the 128 values (two per lane: k = lane, 64 + lane)
  if (lane < DW * DW) {
    const int idx = ((lane / DW + 1) * (DW + 2) + (lane % DW + 1)) * (DB + 2);
    s_hist[idx] = s_hist[idx] + s_hist[idx + DB];
    s_hist[idx + 1] = s_hist[idx + 1] + s_hist[idx + DB + 1];
  }
  __builtin_amdgcn_wave_barrier();
  float d2[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int k = 64 * hh + lane, cell = k / DB, kb = k % DB;
    d2[hh] = s_hist[((cell / DW + 1) * (DW + 2) + (cell % DW + 1)) * (DB + 2) + kb];
  }
  auto ordered_sum_sq = [&](float a, float b) {  // sum_k x_k * x_k, k = 0..127 in order
    const float qa = a * a, qb = b * b;
    float sum = 0.f;
    for (int k = 0; k < 64; ++k) sum = sum + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qa), k));
    for (int k = 0; k < 64; ++k) sum = sum + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qb), k));
    return sum;
  };
  float nrm2 = ordered_sum_sq(d2[0], d2[1]);
  const float thr = sqrtf(nrm2) * 0.2f;
  d2[0] = fminf(d2[0], thr);
  d2[1] = fminf(d2[1], thr);
  nrm2 = ordered_sum_sq(d2[0], d2[1]);
  nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int vv = cv_round(d2[hh] * nrm2);  // saturate_cast<uchar>
    desc[(size_t)t * 128 + 64 * hh + lane] = (float)min(max(vv, 0), 255);
  }
}

// cv::getGaussianKernel(n, sigma, CV_32F) with n = cvRound(sigma * 8 + 1) | 1
static std::vector<float> gaussian_kernel(double sigma) {
  const int n = (int)std::nearbyint(sigma * 4 * 2 + 1) | 1;
  std::vector<float> cf(n);
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    cf[i] = (float)std::exp(scale2x * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) cf[i] = (float)(cf[i] * sum);
  return cf;
}

}  // namespace

// One image through the whole front end on ctx's stream and scratch blocks.  Keypoints (sorted, de-duplicated, rescaled)
// come back in kps; the descriptors go to host memory (descriptors, at most `capacity` rows: SFMHIP_ERR_ARG with the
// count in *n_keypoints when there are more; capacity 0 = a count-only call) and / or stay in HBM (*d_desc_out: a
// hipMalloc'ed n x 128 f32 array the caller owns; no limit).
static int sift_impl(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int n_octave_layers, double contrast_threshold,
                     double edge_threshold, double sigma, int capacity, std::vector<KeyPt>& kps, float* descriptors,
                     void** d_desc_out, int32_t* n_keypoints) {
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int nl = n_octave_layers;
  // device memory: two blocks the context keeps between calls (an image set is extracted image by image), carved
  // here -- block 0 for what the image size fixes, block 1 for what the number of candidates fixes
  auto carve = [](size_t& off, size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  };
  // ---- geometry of the pyramid (base = doubled image; nOctaves = cvRound(log2(min side) - 2) + 1)
  Pyr P{};
  const int bh = 2 * rows, bw = 2 * cols;
  int n_oct = (int)std::nearbyint(std::log((double)std::min(bh, bw)) / std::log(2.0) - 2) + 1;
  n_oct = std::max(1, std::min(n_oct, 16));
  P.n_oct = n_oct;
  P.n_layers = nl;
  size_t gtot = 0, dtot = 0;
  for (int o = 0; o < n_oct; ++o) {
    P.h[o] = o == 0 ? bh : P.h[o - 1] / 2;
    P.w[o] = o == 0 ? bw : P.w[o - 1] / 2;
    if (P.h[o] < 1 || P.w[o] < 1) {
      P.n_oct = n_oct = o;
      break;
    }
    P.goff[o] = gtot;
    P.doff[o] = dtot;
    gtot += (size_t)(nl + 3) * P.h[o] * P.w[o];
    dtot += (size_t)(nl + 2) * P.h[o] * P.w[o];
  }
  unsigned char* d_gray = nullptr;
  float *G = nullptr, *D = nullptr, *tmp = nullptr, *d_kern = nullptr;
  // the kernels: base blur, then sig[1 .. nl + 2]
  std::vector<std::vector<float>> kerns;
  {
    const float sgf = (float)sigma;
    const double sig_diff = (double)std::sqrt(std::max(sgf * sgf - 0.5f * 0.5f * 4, 0.01f));  // sqrtf, float
    kerns.push_back(gaussian_kernel(sig_diff));
    const double k = std::pow(2., 1. / nl);
    for (int i = 1; i < nl + 3; ++i) {
      const double sp = std::pow(k, (double)(i - 1)) * sigma, stt = sp * k;
      kerns.push_back(gaussian_kernel(std::sqrt(stt * stt - sp * sp)));
    }
  }
  size_t koff_total = 0;
  std::vector<size_t> koff;
  for (auto& kk : kerns) {
    koff.push_back(koff_total);
    koff_total += kk.size();
  }
  const int cand_cap = (int)std::min<size_t>((size_t)bh * bw / 4 + 1024, (size_t)1 << 24);
  Cand* d_cand = nullptr;
  int* d_cnt = nullptr;
  {
    size_t off = 0;
    const size_t o_gray = carve(off, (size_t)rows * cols), o_G = carve(off, sizeof(float) * gtot),
                 o_D = carve(off, sizeof(float) * dtot), o_tmp = carve(off, sizeof(float) * (size_t)bh * bw),
                 o_kern = carve(off, sizeof(float) * koff_total), o_cand = carve(off, sizeof(Cand) * cand_cap),
                 o_cnt = carve(off, sizeof(int) * 2);
    char* base = nullptr;
    SFM_TRY(sfm_ctx_dev_scratch(ctx, 0, off, (void**)&base));
    d_gray = (unsigned char*)(base + o_gray);
    G = (float*)(base + o_G);
    D = (float*)(base + o_D);
    tmp = (float*)(base + o_tmp);
    d_kern = (float*)(base + o_kern);
    d_cand = (Cand*)(base + o_cand);
    d_cnt = (int*)(base + o_cnt);
  }
  for (size_t i = 0; i < kerns.size(); ++i)
    SFM_HIP_TRY(hipMemcpyAsync(d_kern + koff[i], kerns[i].data(), sizeof(float) * kerns[i].size(), hipMemcpyHostToDevice, st));
  SFM_HIP_TRY(hipMemcpyAsync(d_gray, gray, (size_t)rows * cols, hipMemcpyHostToDevice, st));
  auto blur = [&](const float* src, float* dst, int h, int w, int ki) {
    const dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(sift_blur_rows, grid, dim3(256), 0, st, src, h, w, d_kern + koff[ki], (int)kerns[ki].size(), tmp);
    hipLaunchKernelGGL(sift_blur_cols, grid, dim3(256), 0, st, (const float*)tmp, h, w, d_kern + koff[ki], (int)kerns[ki].size(), dst);
  };
  // ---- Gaussian and DoG pyramids
  {
    float* up = D;  // (scratch: the DoG arena is free until the pyramid is built; it holds at least one base image)
    hipLaunchKernelGGL(sift_base_up2, dim3((bw + 255) / 256, bh), dim3(256), 0, st, d_gray, rows, cols, up);
    blur(up, G + P.goff[0], bh, bw, 0);
  }
  for (int o = 0; o < n_oct; ++o) {
    const int h = P.h[o], w = P.w[o];
    const size_t isz = (size_t)h * w;
    if (o > 0) {
      const float* src = G + P.goff[o - 1] + (size_t)nl * ((size_t)P.h[o - 1] * P.w[o - 1]);
      hipLaunchKernelGGL(sift_half_nearest, dim3((w + 255) / 256, h), dim3(256), 0, st, src, P.h[o - 1], P.w[o - 1], G + P.goff[o], h, w);
    }
    for (int i = 1; i < nl + 3; ++i) blur(G + P.goff[o] + (size_t)(i - 1) * isz, G + P.goff[o] + (size_t)i * isz, h, w, i);
  }
  {
    OctBlocks t{};
    int nb = 0;
    for (int o = 0; o < n_oct; ++o) {
      t.first[o] = nb;
      nb += (int)(((size_t)(nl + 2) * P.h[o] * P.w[o] + 255) / 256);
    }
    t.first[n_oct] = nb;
    hipLaunchKernelGGL(sift_sub_all, dim3(nb), dim3(256), 0, st, P, t, (const float*)G, D);
  }
  // ---- extrema -> candidates -> keypoints
  SFM_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int) * 2, st));
  const float thr = (float)(int)std::floor(0.5 * contrast_threshold / nl * 255);
  {
    OctBlocks t{};
    int nb = 0;
    for (int o = 0; o < n_oct; ++o) {
      t.first[o] = nb;
      const int h = P.h[o], w = P.w[o];
      if (h > 2 * IMG_BORDER && w > 2 * IMG_BORDER) nb += nl * ((w - 2 * IMG_BORDER + 63) / 64) * (h - 2 * IMG_BORDER);
    }
    t.first[n_oct] = nb;
    if (nb) hipLaunchKernelGGL(sift_extrema, dim3(nb), dim3(64), 0, st, P, t, (const float*)D, thr, d_cand, d_cnt, cand_cap);
  }
  int h_cnt[2] = {0, 0};
  SFM_HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (h_cnt[0] > cand_cap) return SFMHIP_ERR_UNSUPPORTED;
  const int n_cand = h_cnt[0];
  int kp_cap = 4 * n_cand + 16;
  KeyPt* d_kp = nullptr;
  for (int attempt = 0;; ++attempt) {
    // (a candidate can yield several orientation peaks; slots are claimed with atomics, so a list that overflowed
    // would hold an arbitrary subset: the refinement is redone with room for the count it reported)
    SFM_TRY(sfm_ctx_dev_scratch(ctx, 1, sizeof(KeyPt) * kp_cap, (void**)&d_kp));
    if (n_cand > 0)
      hipLaunchKernelGGL(sift_refine, dim3((n_cand + 3) / 4), dim3(256), 0, st, P, (const float*)G, (const float*)D,
                         (const Cand*)d_cand, n_cand, (float)contrast_threshold, (float)edge_threshold, (float)sigma, d_kp,
                         d_cnt + 1, kp_cap);
    SFM_HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    if (h_cnt[1] <= kp_cap) break;
    if (attempt > 0) return SFMHIP_ERR_UNSUPPORTED;
    kp_cap = h_cnt[1];
    SFM_HIP_TRY(hipMemsetAsync(d_cnt + 1, 0, sizeof(int), st));
  }
  int nk = h_cnt[1];
  kps.assign((size_t)nk, KeyPt{});
  if (nk) SFM_HIP_TRY(hipMemcpy(kps.data(), d_kp, sizeof(KeyPt) * nk, hipMemcpyDeviceToHost));
  // ---- KeyPointsFilter::removeDuplicatedSorted, then the rescale of the doubled first octave
  std::sort(kps.begin(), kps.end(), [](const KeyPt& a, const KeyPt& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    return a.octave > b.octave;
  });
  {
    int i = 0;
    for (int j = 1; j < nk; ++j)
      if (kps[i].x != kps[j].x || kps[i].y != kps[j].y || kps[i].size != kps[j].size || kps[i].angle != kps[j].angle) kps[++i] = kps[j];
    if (nk) nk = i + 1;
    kps.resize(nk);
  }
  for (KeyPt& k : kps) {
    k.octave = (k.octave & ~255) | ((k.octave - 1) & 255);  // firstOctave = -1
    k.x *= 0.5f;
    k.y *= 0.5f;
    k.size *= 0.5f;
  }
  *n_keypoints = nk;
  if (d_desc_out) *d_desc_out = nullptr;
  if (descriptors || !d_desc_out)
    if (nk > capacity) return capacity == 0 ? SFMHIP_OK : SFMHIP_ERR_ARG;  // (capacity 0: a count-only call)
  if (nk == 0) return SFMHIP_OK;
  float* d_desc = nullptr;
  {
    // (block 1 again, now that the number of keypoints is known: the device copy of the unsorted keypoints is done
    // with -- the stream is idle -- and a growing call may move the block)
    size_t off = 0;
    const size_t o_kp = carve(off, sizeof(KeyPt) * nk), o_desc = carve(off, d_desc_out ? 0 : sizeof(float) * 128 * (size_t)nk);
    char* base = nullptr;
    SFM_TRY(sfm_ctx_dev_scratch(ctx, 1, off, (void**)&base));
    d_kp = (KeyPt*)(base + o_kp);
    d_desc = (float*)(base + o_desc);
  }
  if (d_desc_out) {  // the descriptors stay in HBM, in a buffer of their own that the caller hands to the matcher
    void* own = nullptr;
    if (hipMalloc(&own, sizeof(float) * 128 * (size_t)nk) != hipSuccess) return SFMHIP_ERR_ALLOC;
    d_desc = (float*)own;
    *d_desc_out = own;
  }
  // (a failure from here on must not leave the caller with a live allocation behind an error code)
  auto finish = [&]() -> int {
    SFM_HIP_TRY(hipMemcpyAsync(d_kp, kps.data(), sizeof(KeyPt) * nk, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(sift_describe, dim3((nk + DESC_WAVES - 1) / DESC_WAVES), dim3(64 * DESC_WAVES), 0, st, P, (const float*)G, (const KeyPt*)d_kp, nk, d_desc);
    SFM_HIP_TRY(hipGetLastError());
    if (descriptors) SFM_HIP_TRY(hipMemcpyAsync(descriptors, d_desc, sizeof(float) * 128 * (size_t)nk, hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    return SFMHIP_OK;
  };
  const int rc = finish();
  if (rc != SFMHIP_OK && d_desc_out && *d_desc_out) {
    hipFree(*d_desc_out);
    *d_desc_out = nullptr;
  }
  return rc;
}

static void sift_pack_keypoints(const std::vector<KeyPt>& kps, float* keypoints) {
  for (size_t i = 0; i < kps.size(); ++i) {
    float* o = keypoints + 6 * i;
    o[0] = kps[i].x;
    o[1] = kps[i].y;
    o[2] = kps[i].size;
    o[3] = kps[i].angle;
    o[4] = kps[i].response;
    std::memcpy(&o[5], &kps[i].octave, 4);
  }
}

static bool sift_args_ok(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int nl, double sigma) {
  return ctx && gray && rows >= 2 && cols >= 2 && nl >= 1 && nl <= 8 && sigma > 0;
}

extern "C" int sfmhip_sift_detect_and_compute(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int n_octave_layers,
                                              double contrast_threshold, double edge_threshold, double sigma, int capacity,
                                              float* keypoints, float* descriptors, int32_t* n_keypoints) {
  if (!sift_args_ok(ctx, gray, rows, cols, n_octave_layers, sigma) || capacity < 0 || !n_keypoints ||
      (capacity > 0 && (!keypoints || !descriptors)))
    return SFMHIP_ERR_ARG;
  std::vector<KeyPt> kps;
  SFM_TRY(sift_impl(ctx, gray, rows, cols, n_octave_layers, contrast_threshold, edge_threshold, sigma, capacity, kps,
                    capacity > 0 ? descriptors : nullptr, nullptr, n_keypoints));
  if (capacity > 0 && *n_keypoints <= capacity) sift_pack_keypoints(kps, keypoints);
  return SFMHIP_OK;
}

extern "C" int sfmhip_sift_detect_and_compute_device(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int n_octave_layers,
                                                     double contrast_threshold, double edge_threshold, double sigma, int capacity,
                                                     float* keypoints, void** d_descriptors, int32_t* n_keypoints) {
  if (!sift_args_ok(ctx, gray, rows, cols, n_octave_layers, sigma) || capacity < 0 || !n_keypoints || !d_descriptors ||
      (capacity > 0 && !keypoints))
    return SFMHIP_ERR_ARG;
  std::vector<KeyPt> kps;
  *d_descriptors = nullptr;
  SFM_TRY(sift_impl(ctx, gray, rows, cols, n_octave_layers, contrast_threshold, edge_threshold, sigma, capacity, kps, nullptr,
                    capacity > 0 ? d_descriptors : nullptr, n_keypoints));
  if (*n_keypoints > capacity) {  // the keypoint array is the caller's: too small -> nothing is kept
    if (*d_descriptors) hipFree(*d_descriptors);
    *d_descriptors = nullptr;
    return capacity == 0 ? SFMHIP_OK : SFMHIP_ERR_ARG;
  }
  sift_pack_keypoints(kps, keypoints);
  return SFMHIP_OK;
}

// A batch of images, several in flight: the front end of one image is ~100 small launches and two host read-backs --
// latency, not throughput -- so the images are dealt to a few worker contexts (a stream and scratch blocks each, kept by
// ctx between calls), one host thread per worker.  Results are those of the one-image entry, image by image.
extern "C" int sfmhip_sift_batch(sfmhip_ctx* ctx, int n_images, const uint8_t* const* gray, const int32_t* rows, const int32_t* cols,
                                 int n_octave_layers, double contrast_threshold, double edge_threshold, double sigma,
                                 float** keypoints, void** d_descriptors, int32_t* n_keypoints) {
  if (!ctx || n_images < 0 || (n_images && (!gray || !rows || !cols || !keypoints || !d_descriptors || !n_keypoints))) return SFMHIP_ERR_ARG;
  for (int i = 0; i < n_images; ++i) {
    if (!sift_args_ok(ctx, gray[i], rows[i], cols[i], n_octave_layers, sigma)) return SFMHIP_ERR_ARG;
    keypoints[i] = nullptr;
    d_descriptors[i] = nullptr;
    n_keypoints[i] = 0;
  }
  if (n_images == 0) return SFMHIP_OK;
  const int n_workers = std::min(n_images, 8);
  while ((int)ctx->workers.size() < n_workers) {
    sfmhip_ctx* w = nullptr;
    SFM_TRY(sfmhip_init(ctx->device, &w));
    ctx->workers.push_back(w);
  }
  std::vector<int> rcs(n_workers, SFMHIP_OK), hip_errs(n_workers, 0);  // (g_sfmhip_last_hip_error is per thread: carried back by hand)
  std::vector<std::thread> threads;
  for (int t = 0; t < n_workers; ++t)
    threads.emplace_back([&, t]() {
      sfmhip_ctx* w = ctx->workers[t];
      for (int i = t; i < n_images && rcs[t] == SFMHIP_OK; i += n_workers) {
        std::vector<KeyPt> kps;
        int32_t n = 0;
        void* dd = nullptr;
        int rc = sift_impl(w, gray[i], rows[i], cols[i], n_octave_layers, contrast_threshold, edge_threshold, sigma, 0, kps, nullptr, &dd, &n);
        if (rc == SFMHIP_OK) {
          float* k = (float*)malloc(sizeof(float) * 6 * (size_t)std::max(n, 1));
          if (!k) rc = SFMHIP_ERR_ALLOC;
          else {
            sift_pack_keypoints(kps, k);
            keypoints[i] = k;
            d_descriptors[i] = dd;
            n_keypoints[i] = n;
          }
        }
        if (rc != SFMHIP_OK) {
          if (dd) hipFree(dd);
          rcs[t] = rc;
          hip_errs[t] = g_sfmhip_last_hip_error;
        }
      }
    });
  for (auto& th : threads) th.join();
  for (int t = 0; t < n_workers; ++t)
    if (rcs[t] != SFMHIP_OK) {
      const int rc = rcs[t];
      if (hip_errs[t]) g_sfmhip_last_hip_error = hip_errs[t];  // (the caller's sfmhip_last_hip_error() reports the worker's)
      for (int i = 0; i < n_images; ++i) {
        free(keypoints[i]);
        if (d_descriptors[i]) hipFree(d_descriptors[i]);
        keypoints[i] = nullptr;
        d_descriptors[i] = nullptr;
        n_keypoints[i] = 0;
      }
      return rc;
    }
  return SFMHIP_OK;
}
