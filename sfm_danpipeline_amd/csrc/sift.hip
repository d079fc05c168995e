// sift.hip -- the detector / descriptor front end of StructFromMotion::getFeature (reference src/Sfm.cpp:300-330):
// cv::xfeatures2d::SIFT::create(0, 3, 0.04, 10, 1.6)->detectAndCompute on gfx950.
//
// OpenCV 3.4.1's float pipeline (xfeatures2d/src/sift.cpp) stage by stage: the doubled, blurred base image; six
// Gaussian images per octave (separable blur, INTER_NEAREST halving); difference of Gaussians; 26-neighbour extrema
// above the contrast floor; adjustLocalExtrema (Newton steps with Cramer's rule in float, contrast and edge tests);
// orientation histogram and peaks; KeyPointsFilter::removeDuplicatedSorted and the rescale of the doubled octave (host);
// the 4 x 4 x 8 descriptor.  The pyramid kernels are HBM-bound streaming kernels; the keypoint stages are one thread
// per candidate / keypoint with the reference's sequential accumulation order, so that the numbers are those of the
// numpy restatement the tests hold (test infrastructure).  Parity with OpenCV itself is unpinned: it is not in the
// image; where its float results depend on SIMD paths or its own exp / atan2 approximations, one order of operations
// is fixed here (exp / cos / sin / pow: evaluated in double, rounded to float).
#include "common.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
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
__global__ void sift_sub(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = a[i] - b[i];
}

struct Cand {
  int o, layer, r, c;
};
// 26-neighbour extrema of DoG layer `layer` of one octave (prev, cur, next), |val| > thr
__global__ void sift_extrema(const float* __restrict__ prv, const float* __restrict__ cur, const float* __restrict__ nxt, int h,
                             int w, int o, int layer, float thr, Cand* __restrict__ out, int* __restrict__ n_out, int cap) {
  const int c = IMG_BORDER + blockIdx.x * blockDim.x + threadIdx.x, r = IMG_BORDER + blockIdx.y;
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

struct Pyr {  // per octave: image size and the offsets of its Gaussian / DoG images in the two arenas
  int h[16], w[16];
  size_t goff[16], doff[16];
  int n_oct, n_layers;
};
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

// adjustLocalExtrema + calcOrientationHist + the peaks: one thread per candidate
__global__ __launch_bounds__(64) void sift_refine(Pyr P, const float* __restrict__ G, const float* __restrict__ D,
                                                  const Cand* __restrict__ cand, int n_cand, float contrast_thr, float edge_thr,
                                                  float sigma, KeyPt* __restrict__ out, int* __restrict__ n_out, int cap) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
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
  float tmp[ORI_BINS];
  for (int b = 0; b < ORI_BINS; ++b) tmp[b] = 0.f;
  for (int ii = -radius; ii <= radius; ++ii) {
    const int y = r + ii;
    if (y <= 0 || y >= h - 1) continue;
    for (int jj = -radius; jj <= radius; ++jj) {
      const int x = c + jj;
      if (x <= 0 || x >= w - 1) continue;
      const size_t q = (size_t)y * w + x;
      const float dx = gimg[q + 1] - gimg[q - 1], dy = gimg[q - w] - gimg[q + w];
      const float wgt = (float)exp((double)((float)(ii * ii + jj * jj) * expf_scale));
      const float ori = fast_atan2_deg(dy, dx);
      const float mag = sqrtf(dx * dx + dy * dy);
      int b = cv_round((float)(ORI_BINS / 360.0) * ori);
      if (b >= ORI_BINS) b -= ORI_BINS;
      if (b < 0) b += ORI_BINS;
      tmp[b] = tmp[b] + wgt * mag;
    }
  }
  float hist[ORI_BINS], omax = 0.f;
  for (int b = 0; b < ORI_BINS; ++b) {
    auto T = [&](int k) { return tmp[(k + ORI_BINS) % ORI_BINS]; };
    hist[b] = (T(b - 2) + T(b + 2)) * (1.f / 16.f) + (T(b - 1) + T(b + 1)) * (4.f / 16.f) + T(b) * (6.f / 16.f);
    omax = b == 0 ? hist[0] : fmaxf(omax, hist[b]);
  }
  const float mag_thr = omax * 0.8f;
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

// calcSIFTDescriptor: one thread per keypoint, the reference's accumulation order
__global__ __launch_bounds__(64) void sift_describe(Pyr P, const float* __restrict__ G, const KeyPt* __restrict__ kps, int n,
                                                    float* __restrict__ desc) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
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
  float hist[(DW + 2) * (DW + 2) * (DB + 2)];
  for (int k = 0; k < (DW + 2) * (DW + 2) * (DB + 2); ++k) hist[k] = 0.f;
  for (int i = -radius; i <= radius; ++i)
    for (int j = -radius; j <= radius; ++j) {
      const float c_rot = (float)j * cos_t - (float)i * sin_t, r_rot = (float)j * sin_t + (float)i * cos_t;
      const float rbin = r_rot + (float)(DW / 2) - 0.5f, cbin = c_rot + (float)(DW / 2) - 0.5f;
      const int r = py + i, c = px + j;
      if (rbin > -1 && rbin < DW && cbin > -1 && cbin < DW && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
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
        const float v111 = v_rc11 * ob, v110 = v_rc11 - v111, v101 = v_rc10 * ob, v100 = v_rc10 - v101;
        const float v011 = v_rc01 * ob, v010 = v_rc01 - v011, v001 = v_rc00 * ob, v000 = v_rc00 - v001;
        const int idx = ((r0 + 1) * (DW + 2) + c0 + 1) * (DB + 2) + o0;
        hist[idx] += v000;
        hist[idx + 1] += v001;
        hist[idx + (DB + 2)] += v010;
        hist[idx + (DB + 3)] += v011;
        hist[idx + (DW + 2) * (DB + 2)] += v100;
        hist[idx + (DW + 2) * (DB + 2) + 1] += v101;
        hist[idx + (DW + 3) * (DB + 2)] += v110;
        hist[idx + (DW + 3) * (DB + 2) + 1] += v111;
      }
    }
  float dst[DW * DW * DB];
  for (int i = 0; i < DW; ++i)
    for (int j = 0; j < DW; ++j) {
      const int idx = ((i + 1) * (DW + 2) + (j + 1)) * (DB + 2);
      hist[idx] += hist[idx + DB];
      hist[idx + 1] += hist[idx + DB + 1];
      for (int k = 0; k < DB; ++k) dst[(i * DW + j) * DB + k] = hist[idx + k];
    }
  float nrm2 = 0.f;
  for (int k = 0; k < DW * DW * DB; ++k) nrm2 = nrm2 + dst[k] * dst[k];
  const float thr = sqrtf(nrm2) * 0.2f;
  nrm2 = 0.f;
  for (int k = 0; k < DW * DW * DB; ++k) {
    const float v = fminf(dst[k], thr);
    dst[k] = v;
    nrm2 = nrm2 + v * v;
  }
  nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
  for (int k = 0; k < DW * DW * DB; ++k) {
    const int v = cv_round(dst[k] * nrm2);  // saturate_cast<uchar>
    desc[(size_t)t * 128 + k] = (float)min(max(v, 0), 255);
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

extern "C" int sfmhip_sift_detect_and_compute(sfmhip_ctx* ctx, const uint8_t* gray, int rows, int cols, int n_octave_layers,
                                              double contrast_threshold, double edge_threshold, double sigma, int capacity,
                                              float* keypoints, float* descriptors, int32_t* n_keypoints) {
  if (!ctx || !gray || rows < 2 || cols < 2 || n_octave_layers < 1 || n_octave_layers > 8 || !(sigma > 0) || capacity < 0 ||
      !n_keypoints || (capacity > 0 && (!keypoints || !descriptors)))
    return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int nl = n_octave_layers;
  struct Bufs {
    std::vector<void*> v;
    ~Bufs() {
      for (void* p : v) hipFree(p);
    }
  } bufs;
  auto dalloc = [&](void** p, size_t bytes) -> int {
    if (hipMalloc(p, bytes ? bytes : 8) != hipSuccess) return SFMHIP_ERR_ALLOC;
    bufs.v.push_back(*p);
    return SFMHIP_OK;
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
  SFM_TRY(dalloc((void**)&d_gray, (size_t)rows * cols));
  SFM_TRY(dalloc((void**)&G, sizeof(float) * gtot));
  SFM_TRY(dalloc((void**)&D, sizeof(float) * dtot));
  SFM_TRY(dalloc((void**)&tmp, sizeof(float) * (size_t)bh * bw));
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
  SFM_TRY(dalloc((void**)&d_kern, sizeof(float) * koff_total));
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
  for (int o = 0; o < n_oct; ++o) {
    const size_t isz = (size_t)P.h[o] * P.w[o], n = (size_t)(nl + 2) * isz;
    hipLaunchKernelGGL(sift_sub, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)(G + P.goff[o] + isz),
                       (const float*)(G + P.goff[o]), D + P.doff[o], n);
  }
  // ---- extrema -> candidates -> keypoints
  const int cand_cap = (int)std::min<size_t>((size_t)bh * bw / 4 + 1024, (size_t)1 << 24);
  Cand* d_cand = nullptr;
  int* d_cnt = nullptr;
  SFM_TRY(dalloc((void**)&d_cand, sizeof(Cand) * cand_cap));
  SFM_TRY(dalloc((void**)&d_cnt, sizeof(int) * 2));
  SFM_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int) * 2, st));
  const float thr = (float)(int)std::floor(0.5 * contrast_threshold / nl * 255);
  for (int o = 0; o < n_oct; ++o) {
    const int h = P.h[o], w = P.w[o];
    if (h <= 2 * IMG_BORDER || w <= 2 * IMG_BORDER) continue;
    const size_t isz = (size_t)h * w;
    for (int i = 1; i <= nl; ++i) {
      const float* cur = D + P.doff[o] + (size_t)i * isz;
      hipLaunchKernelGGL(sift_extrema, dim3((w - 2 * IMG_BORDER + 63) / 64, h - 2 * IMG_BORDER), dim3(64), 0, st, cur - isz, cur,
                         cur + isz, h, w, o, i, thr, d_cand, d_cnt, cand_cap);
    }
  }
  int h_cnt[2] = {0, 0};
  SFM_HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  if (h_cnt[0] > cand_cap) return SFMHIP_ERR_UNSUPPORTED;
  const int n_cand = h_cnt[0];
  const int kp_cap = 4 * n_cand + 16;
  KeyPt* d_kp = nullptr;
  SFM_TRY(dalloc((void**)&d_kp, sizeof(KeyPt) * kp_cap));
  if (n_cand > 0)
    hipLaunchKernelGGL(sift_refine, dim3((n_cand + 63) / 64), dim3(64), 0, st, P, (const float*)G, (const float*)D,
                       (const Cand*)d_cand, n_cand, (float)contrast_threshold, (float)edge_threshold, (float)sigma, d_kp,
                       d_cnt + 1, kp_cap);
  SFM_HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  int nk = std::min(h_cnt[1], kp_cap);
  std::vector<KeyPt> kps(nk);
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
  if (nk > capacity) return capacity == 0 ? SFMHIP_OK : SFMHIP_ERR_ARG;  // (capacity 0: a count-only call)
  if (nk == 0) return SFMHIP_OK;
  float* d_desc = nullptr;
  SFM_TRY(dalloc((void**)&d_desc, sizeof(float) * 128 * (size_t)nk));
  SFM_HIP_TRY(hipMemcpyAsync(d_kp, kps.data(), sizeof(KeyPt) * nk, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(sift_describe, dim3((nk + 63) / 64), dim3(64), 0, st, P, (const float*)G, (const KeyPt*)d_kp, nk, d_desc);
  SFM_HIP_TRY(hipGetLastError());
  SFM_HIP_TRY(hipMemcpyAsync(descriptors, d_desc, sizeof(float) * 128 * (size_t)nk, hipMemcpyDeviceToHost, st));
  SFM_HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < nk; ++i) {
    float* o = keypoints + 6 * (size_t)i;
    o[0] = kps[i].x;
    o[1] = kps[i].y;
    o[2] = kps[i].size;
    o[3] = kps[i].angle;
    o[4] = kps[i].response;
    std::memcpy(&o[5], &kps[i].octave, 4);
  }
  return SFMHIP_OK;
}
