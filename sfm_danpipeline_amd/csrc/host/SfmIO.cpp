// SfmIO.cpp -- image loading, calibration XML and PMVS2 export of the reference's StructFromMotion
// (src/Sfm.cpp:118-252, 1246-1303), without OpenCV / Boost / libpng: a PNG decoder
// (zlib inflate + unfiltering), cv::resize(INTER_LINEAR) and cv::cvtColor(BGR2GRAY) for 8-bit images restated
// from OpenCV 3.4.1's fixed-point arithmetic, a reader for the two opencv-matrix nodes of the calibration file.
// Host code only: this row has no performance content.
#include <dirent.h>
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>
#include "Sfm.h"
#include "hip_backend.h"

namespace {

// ---------------------------------------------------------------- zlib inflate (RFC 1950/1951)
struct BitReader {
  const uint8_t* p;
  size_t n, pos;
  uint32_t bits;
  int nbits;
  bool ok;
  BitReader(const uint8_t* d, size_t len) : p(d), n(len), pos(0), bits(0), nbits(0), ok(true) {}
  uint32_t get(int k) {
    while (nbits < k) {
      if (pos >= n) {
        ok = false;
        return 0;
      }
      bits |= (uint32_t)p[pos++] << nbits;
      nbits += 8;
    }
    const uint32_t v = bits & ((1u << k) - 1u);
    bits >>= k;
    nbits -= k;
    return k ? v : 0;
  }
};
struct Huffman {
  uint16_t count[16], symbol[288];
  void build(const uint8_t* len, int n) {
    for (int i = 0; i < 16; ++i) count[i] = 0;
    for (int i = 0; i < n; ++i) count[len[i]]++;
    count[0] = 0;
    uint16_t offs[16];
    offs[1] = 0;
    for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + count[i];
    for (int i = 0; i < n; ++i)
      if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
  }
  int decode(BitReader& br) const {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
      code |= (int)br.get(1);
      if (!br.ok) return -1;
      const int c = count[l];
      if (code - c < first) return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};
// max_out: the decoded size the caller expects; a stream that expands beyond it is corrupt (or hostile) and is refused
bool inflate_zlib(const std::vector<uint8_t>& in, std::vector<uint8_t>& out, size_t max_out) {
  if (in.size() < 6 || (in[0] & 0x0F) != 8 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return false;
  BitReader br(in.data() + 2, in.size() - 2);
  static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  int last = 0;
  while (!last) {
    last = (int)br.get(1);
    const int type = (int)br.get(2);
    if (!br.ok) return false;
    if (type == 0) {
      br.bits = 0;
      br.nbits = 0;
      if (br.pos + 4 > br.n) return false;
      const unsigned len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
      br.pos += 4;
      if ((len ^ 0xFFFFu) != nlen || br.pos + len > br.n || out.size() + len > max_out) return false;
      out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
      br.pos += len;
      continue;
    }
    if (type == 3) return false;
    Huffman hl, hd;
    uint8_t lens[320];
    if (type == 1) {
      int i = 0;
      for (; i < 144; ++i) lens[i] = 8;
      for (; i < 256; ++i) lens[i] = 9;
      for (; i < 280; ++i) lens[i] = 7;
      for (; i < 288; ++i) lens[i] = 8;
      hl.build(lens, 288);
      for (i = 0; i < 30; ++i) lens[i] = 5;
      hd.build(lens, 30);
    } else {
      const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0};
      for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.get(3);
      if (!br.ok || nlen > 286 || ndist > 30) return false;
      Huffman hc;
      hc.build(cl, 19);
      int idx = 0;
      while (idx < nlen + ndist) {
        const int sym = hc.decode(br);
        if (sym < 0) return false;
        if (sym < 16) lens[idx++] = (uint8_t)sym;
        else {
          int rep = 0, val = 0;
          if (sym == 16) {
            if (idx == 0) return false;
            val = lens[idx - 1];
            rep = 3 + (int)br.get(2);
          } else if (sym == 17) rep = 3 + (int)br.get(3);
          else rep = 11 + (int)br.get(7);
          if (idx + rep > nlen + ndist) return false;
          while (rep--) lens[idx++] = (uint8_t)val;
        }
      }
      hl.build(lens, nlen);
      hd.build(lens + nlen, ndist);
    }
    for (;;) {
      const int sym = hl.decode(br);
      if (sym < 0 || !br.ok) return false;
      if (sym < 256) {
        if (out.size() >= max_out) return false;
        out.push_back((uint8_t)sym);
      } else if (sym == 256) break;
      else {
        const int s = sym - 257;
        if (s >= 29) return false;
        const int len = lbase[s] + (int)br.get(lext[s]);
        const int ds = hd.decode(br);
        if (ds < 0 || ds >= 30) return false;
        const size_t dist = dbase[ds] + br.get(dext[ds]);
        if (!br.ok || dist > out.size() || out.size() + (size_t)len > max_out) return false;
        const size_t from = out.size() - dist;
        for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------- PNG -> BGR (what cv::imread(path) returns)
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
bool decode_png(const std::vector<uint8_t>& f, cv::Mat& bgr, std::string& why) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (f.size() < 8 || memcmp(f.data(), sig, 8) != 0) {
    why = "not a PNG file";
    return false;
  }
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  size_t pos = 8;
  while (pos + 12 <= f.size()) {
    const uint32_t len = be32(&f[pos]);
    const char* tag = (const char*)&f[pos + 4];
    if (pos + 12 + len > f.size()) break;
    const uint8_t* d = &f[pos + 8];
    if (!memcmp(tag, "IHDR", 4) && len >= 13) {
      w = be32(d);
      h = be32(d + 4);
      depth = d[8];
      ctype = d[9];
      interlace = d[12];
    } else if (!memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
    else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
    else if (!memcmp(tag, "IEND", 4)) break;
    pos += 12 + len;
  }
  if (!w || !h || interlace > 1) {
    why = interlace ? "unknown PNG interlace method" : "no IHDR";
    return false;
  }
  const int nch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!nch || (depth != 8 && depth != 16 && !((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
    why = "unsupported colour type / bit depth";
    return false;
  }
  // IHDR is not taken on trust: dimensions beyond 65535 (cv::Mat's int rows / cols, and size_t products that wrap) or a
  // decoded size beyond 1 GiB are refused before anything is inflated, and the inflater stops at the expected size
  if (w > 65535u || h > 65535u) {
    why = "image dimensions out of range";
    return false;
  }
  const size_t bpp_bits = (size_t)nch * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = std::max<size_t>(1, bpp_bits / 8);
  if ((stride + 1) * (size_t)h > ((size_t)1 << 30)) {
    why = "image too large";
    return false;
  }
  // Adam7 (interlace method 1, which libpng / cv::imread read like any other file): seven reduced images, each filtered on
  // its own -- pass p holds the pixels (x0 + i dx, y0 + j dy)
  static const int a7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  struct Pass { uint32_t x0, y0, dx, dy, pw, ph; size_t ps; };
  std::vector<Pass> passes;
  size_t raw_need = 0;
  if (!interlace) {
    passes.push_back({0, 0, 1, 1, w, h, stride});
    raw_need = (stride + 1) * (size_t)h;
  } else {
    for (const auto& a : a7) {
      if (w <= (uint32_t)a[0] || h <= (uint32_t)a[1]) continue;  // (an empty pass carries no rows at all)
      Pass ps{(uint32_t)a[0], (uint32_t)a[1], (uint32_t)a[2], (uint32_t)a[3], (w - a[0] + a[2] - 1) / a[2], (h - a[1] + a[3] - 1) / a[3], 0};
      ps.ps = ((size_t)ps.pw * bpp_bits + 7) / 8;
      raw_need += (ps.ps + 1) * (size_t)ps.ph;
      passes.push_back(ps);
    }
  }
  std::vector<uint8_t> raw;
  if (!inflate_zlib(idat, raw, raw_need)) {
    why = "corrupt zlib stream";
    return false;
  }
  if (raw.size() < raw_need) {
    why = "truncated image data";
    return false;
  }
  std::vector<uint8_t> img(stride * h, 0), zero(stride, 0), sub;
  size_t off = 0;
  for (const Pass& ps : passes) {
    // the pass's rows, unfiltered: straight into the image when it is the only one
    uint8_t* dst = interlace ? (sub.assign(ps.ps * ps.ph, 0), sub.data()) : img.data();
    for (uint32_t y = 0; y < ps.ph; ++y) {
      const uint8_t* in = &raw[off + (ps.ps + 1) * y];
      uint8_t* cur = dst + ps.ps * y;
      const uint8_t* up = y ? dst + ps.ps * (y - 1) : zero.data();
      const int ft = in[0];
      for (size_t x = 0; x < ps.ps; ++x) {
        const int a = x >= bpp ? cur[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
        int v = in[1 + x];
        if (ft == 1) v += a;
        else if (ft == 2) v += b;
        else if (ft == 3) v += (a + b) >> 1;
        else if (ft == 4) v += paeth(a, b, c);
        else if (ft != 0) {
          why = "bad filter type";
          return false;
        }
        cur[x] = (uint8_t)v;
      }
    }
    off += (ps.ps + 1) * (size_t)ps.ph;
    if (!interlace) break;
    for (uint32_t j = 0; j < ps.ph; ++j)
      for (uint32_t i = 0; i < ps.pw; ++i) {
        const size_t X = ps.x0 + (size_t)i * ps.dx, Y = ps.y0 + (size_t)j * ps.dy;
        if (bpp_bits >= 8) {
          memcpy(&img[stride * Y + X * bpp], &sub[ps.ps * j + (size_t)i * bpp], bpp);
        } else {  // 1, 2 or 4 bits per pixel, most significant bits first
          const int per = 8 / (int)bpp_bits, mask = (1 << bpp_bits) - 1;
          const int v = (sub[ps.ps * j + i / per] >> ((per - 1 - (int)(i % per)) * (int)bpp_bits)) & mask;
          img[stride * Y + X / per] |= (uint8_t)(v << ((per - 1 - (int)(X % per)) * (int)bpp_bits));
        }
      }
  }
  bgr = cv::Mat((int)h, (int)w, CV_8UC3);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* r = &img[stride * y];
    uint8_t* o = bgr.ptr() + (size_t)y * w * 3;
    for (uint32_t x = 0; x < w; ++x) {
      uint8_t R, G, B;
      auto sample = [&](size_t idx) -> int {  // idx-th sample of the row, scaled to 8 bit as libpng's strip/expand does
        if (depth == 8) return r[idx];
        if (depth == 16) return r[2 * idx];   // png_set_strip_16: the high byte
        const int per = 8 / depth, sh = (per - 1 - (int)(idx % per)) * depth;
        const int v = (r[idx / per] >> sh) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);  // gray: expand to 8 bit
      };
      if (ctype == 3) {
        const size_t pi = (size_t)sample(x);
        if (3 * pi + 2 >= plte.size()) {
          why = "palette index out of range";
          return false;
        }
        R = plte[3 * pi], G = plte[3 * pi + 1], B = plte[3 * pi + 2];
      } else if (nch <= 2) {
        R = G = B = (uint8_t)sample((size_t)x * nch);
      } else {
        R = (uint8_t)sample((size_t)x * nch), G = (uint8_t)sample((size_t)x * nch + 1), B = (uint8_t)sample((size_t)x * nch + 2);
      }
      o[3 * x] = B, o[3 * x + 1] = G, o[3 * x + 2] = R;  // alpha is dropped (IMREAD_COLOR)
    }
  }
  return true;
}

// ---------------------------------------------------------------- cv::resize(src, dst, Size(), fx, fy, INTER_LINEAR), 8-bit
// OpenCV 3.4.1 imgproc/resize.cpp: dsize = (cvRound(cols*fx), cvRound(rows*fy)); source coordinate of dst x:
// (x + 0.5)/fx - 0.5; 11-bit fixed-point weights (INTER_RESIZE_COEF_SCALE = 2048); horizontal pass in int, vertical
// pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.
int cv_round(double v) { return (int)std::lrint(v); }
cv::Mat resize_linear_8u(const cv::Mat& src, double fx, double fy) {
  const int cn = src.channels(), sw = src.cols, sh = src.rows;
  const int dw = cv_round(sw * fx), dh = cv_round(sh * fy);
  cv::Mat dst(dh, dw, src.type());
  const double scale_x = 1.0 / fx, scale_y = 1.0 / fy;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * (size_t)dw), beta(2 * (size_t)dh);
  for (int dx = 0; dx < dw; ++dx) {
    float f = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(f);
    f -= sx;
    if (sx < 0) f = 0, sx = 0;
    if (sx >= sw - 1) f = 0, sx = sw - 1;
    xofs[dx] = sx;
    const float c0 = 1.f - f, c1 = f;
    alpha[2 * dx] = (short)cv_round(c0 * 2048.f);
    alpha[2 * dx + 1] = (short)cv_round(c1 * 2048.f);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float f = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(f);
    f -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = (short)cv_round((1.f - f) * 2048.f);
    beta[2 * dy + 1] = (short)cv_round(f * 2048.f);
  }
  std::vector<int> row0((size_t)dw * cn), row1((size_t)dw * cn);
  auto hrow = [&](int sy, std::vector<int>& out) {
    sy = std::min(std::max(sy, 0), sh - 1);  // rows beyond the image: border replicate
    const unsigned char* s = src.ptr() + (size_t)sy * sw * cn;
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
      for (int c = 0; c < cn; ++c) out[(size_t)dx * cn + c] = s[sx * cn + c] * alpha[2 * dx] + s[sx1 * cn + c] * alpha[2 * dx + 1];
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    hrow(yofs[dy], row0);
    hrow(yofs[dy] + 1, row1);
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    unsigned char* d = dst.ptr() + (size_t)dy * dw * cn;
    for (size_t i = 0; i < (size_t)dw * cn; ++i)
      d[i] = (unsigned char)((((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2);
  }
  return dst;
}
// cv::cvtColor(BGR2GRAY), 8-bit: (B*1868 + G*9617 + R*4899 + (1 << 13)) >> 14  (color.cpp: yuv_shift = 14)
cv::Mat bgr_to_gray(const cv::Mat& bgr) {
  cv::Mat g(bgr.rows, bgr.cols, CV_8UC1);
  const unsigned char* s = bgr.ptr();
  unsigned char* d = g.ptr();
  for (size_t i = 0, n = (size_t)bgr.rows * bgr.cols; i < n; ++i)
    d[i] = (unsigned char)((s[3 * i] * 1868 + s[3 * i + 1] * 9617 + s[3 * i + 2] * 4899 + (1 << 13)) >> 14);
  return g;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
}
// ---------------------------------------------------------------- JPEG -> BGR (what cv::imread(path) returns)
// Baseline / extended sequential Huffman JPEG (SOF0 / SOF1, 8 bit, 1 or 3 components, sampling factors 1 or 2, restart
// intervals), decoded the way libjpeg (which cv::imread calls with its defaults) decodes it: dequantisation in natural
// order, the accurate integer inverse DCT (jidctint.c "islow": 13-bit constants, PASS1_BITS 2, range-limit table), "fancy"
// triangle-filter upsampling of 2:1 subsampled chroma (jdsample.c: h2v1 (3 a + b + 1|2) >> 2, h2v2 (3 (3 a + b) + (3 c + d)
// + 8|7) >> 4, h1v2 (3 a + b + 1|2) >> 2 down the columns, edge rows / columns replicated), scans interleaved or one per
// component, 0xFF fill bytes before markers, YCbCr -> RGB with the 16-bit fixed-point tables of jdcolor.c.  Checked
// against PIL's decoder (libjpeg-turbo, same algorithms) on generated files (tests/test_host_io.py).  Progressive frames
// (SOF2; round 4): the scans of jdphuff.c -- DC / AC, first pass / refinement, end-of-band runs -- summed into the blocks'
// coefficients, transformed when the last scan is in.  Not decoded: arithmetic coding, 12 bit, CMYK, lossless -- the load
// fails with a message, as cv::imread would return an empty Mat for a file its libjpeg cannot read.
struct JpegHuff {
  uint8_t bits[17];
  uint8_t vals[256];
  int mincode[17], maxcode[18], valptr[17];
  bool present = false;
  // the codes of up to LOOK bits resolved by one table look-up on the next LOOK bits (libjpeg's jdhuff.c does the same with 8):
  // look[bits] = length << 8 | symbol, 0 where the code is longer
  static constexpr int LOOK = 9;
  uint16_t look[1 << LOOK];
  void build() {
    int code = 0, k = 0;
    memset(look, 0, sizeof look);
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      for (int i = 0; i < bits[l] && l <= LOOK; ++i) {
        const int c = code + i;
        if (c >= (1 << l)) break;  // (an over-subscribed table of a corrupt file: the slow path rejects what is wrong with it)
        for (int pad = 0; pad < (1 << (LOOK - l)); ++pad) {
          uint16_t& e = look[(c << (LOOK - l)) | pad];
          if (!e) e = (uint16_t)((l << 8) | vals[k + i]);  // (the shortest code wins, as in the bit-by-bit search)
        }
      }
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};
struct JpegBits {
  const uint8_t* p;
  size_t n, pos;
  uint32_t acc = 0;
  int cnt = 0;
  bool hit_marker = false;
  void fill() {
    while (cnt <= 24) {
      int b = 0;
      if (!hit_marker && pos < n) {
        b = p[pos];
        if (b == 0xFF) {
          while (pos + 1 < n && p[pos + 1] == 0xFF) ++pos;  // fill bytes: any run of 0xFF may precede a marker (T.81 B.1.1.2)
          const int b2 = pos + 1 < n ? p[pos + 1] : 0xD9;
          if (b2 == 0) pos += 2;          // stuffed zero
          else { hit_marker = true; b = 0; }  // a marker: feed zeros (libjpeg does the same past the data)
        } else ++pos;
      }
      acc |= (uint32_t)b << (24 - cnt);
      cnt += 8;
    }
  }
  int get(int nb) {
    if (nb == 0) return 0;
    fill();
    const int v = (int)(acc >> (32 - nb));
    acc <<= nb;
    cnt -= nb;
    return v;
  }
  int decode(const JpegHuff& h) {
    fill();
    const uint16_t e = h.look[acc >> (32 - JpegHuff::LOOK)];
    if (e) {
      acc <<= e >> 8;
      cnt -= e >> 8;
      return e & 255;
    }
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
      code = (code << 1) | (int)(acc >> 31);
      acc <<= 1;
      --cnt;
      if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
  }
  void reset() { acc = 0; cnt = 0; hit_marker = false; }
};
inline int jpeg_extend(int v, int nb) { return v < (1 << (nb - 1)) ? v - (1 << nb) + 1 : v; }
inline uint8_t jpeg_range_limit(int x) {  // libjpeg's range_limit[(x) & RANGE_MASK] for the IDCT output (+128)
  const int i = x & 1023;
  return (uint8_t)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
}
// jidctint.c jpeg_idct_islow on one dequantised block (natural order) -> 8 x 8 samples
void jpeg_idct_islow(const int* in, uint8_t* out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                 F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069,
                 F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };  // (arithmetic shift of a negative value: as libjpeg's RIGHT_SHIFT)
  long ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* ip = in + c;
    if (!ip[8] && !ip[16] && !ip[24] && !ip[32] && !ip[40] && !ip[48] && !ip[56]) {
      const long dc = (long)ip[0] * (1L << P1);
      for (int r = 0; r < 8; ++r) ws[8 * r + c] = dc;
      continue;
    }
    long z2 = ip[16], z3 = ip[48];
    long z1 = (z2 + z3) * F_0_541196100;
    long tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
    z2 = ip[0];
    z3 = ip[32];
    long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);  // (a multiplication: shifting a negative value left is undefined)
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = ip[56];
    tmp1 = ip[40];
    tmp2 = ip[24];
    tmp3 = ip[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336;
    tmp1 *= F_2_053119869;
    tmp2 *= F_3_072711026;
    tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    ws[c] = descale(tmp10 + tmp3, CB - P1);
    ws[56 + c] = descale(tmp10 - tmp3, CB - P1);
    ws[8 + c] = descale(tmp11 + tmp2, CB - P1);
    ws[48 + c] = descale(tmp11 - tmp2, CB - P1);
    ws[16 + c] = descale(tmp12 + tmp1, CB - P1);
    ws[40 + c] = descale(tmp12 - tmp1, CB - P1);
    ws[24 + c] = descale(tmp13 + tmp0, CB - P1);
    ws[32 + c] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; ++r) {
    const long* w = ws + 8 * r;
    uint8_t* o = out + (size_t)r * stride;
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F_0_541196100;
    long tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
    long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7];
    tmp1 = w[5];
    tmp2 = w[3];
    tmp3 = w[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336;
    tmp1 *= F_2_053119869;
    tmp2 *= F_3_072711026;
    tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    constexpr int SH = CB + P1 + 3;
    o[0] = jpeg_range_limit((int)descale(tmp10 + tmp3, SH));
    o[7] = jpeg_range_limit((int)descale(tmp10 - tmp3, SH));
    o[1] = jpeg_range_limit((int)descale(tmp11 + tmp2, SH));
    o[6] = jpeg_range_limit((int)descale(tmp11 - tmp2, SH));
    o[2] = jpeg_range_limit((int)descale(tmp12 + tmp1, SH));
    o[5] = jpeg_range_limit((int)descale(tmp12 - tmp1, SH));
    o[3] = jpeg_range_limit((int)descale(tmp13 + tmp0, SH));
    o[4] = jpeg_range_limit((int)descale(tmp13 - tmp0, SH));
  }
}

bool decode_jpeg(const std::vector<uint8_t>& f, cv::Mat& bgr, std::string& why) {
  static const uint8_t zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) {
    why = "not a JPEG file";
    return false;
  }
  uint16_t qt[4][64];
  bool qt_set[4] = {false, false, false, false};
  JpegHuff hdc[4], hac[4];
  struct Comp {
    int id, h, v, tq, td, ta;
    int bw, bh;  // blocks per row / column of the padded plane
    int dw, dhh; // downsampled_width / height (true sizes)
    bool coded = false;  // a scan has carried this component
    std::vector<uint8_t> plane;
    // progressive frames: the coefficients of every block (natural order), summed up scan by scan; the quantisation table as it
    // stood at the component's first scan (jdinput.c latch_quant_tables)
    std::vector<int32_t> coef;
    uint16_t q[64];
    bool q_latched = false;
  } comp[3];
  bool progressive = false;
  int ncomp = 0, width = 0, height = 0, restart = 0, adobe_transform = -1;
  int hmax = 1, vmax = 1, n_scans = 0;
  bool have_sof = false, planes_ready = false;
  size_t pos = 2;
  while (pos + 4 <= f.size()) {
    if (f[pos] != 0xFF) {
      why = "corrupt JPEG (marker expected)";
      return false;
    }
    const int mk = f[pos + 1];
    if (mk == 0xFF) {
      ++pos;
      continue;
    }
    if (mk == 0xD8 || (mk >= 0xD0 && mk <= 0xD7) || mk == 0x01) {
      pos += 2;
      continue;
    }
    const size_t len = ((size_t)f[pos + 2] << 8) | f[pos + 3];
    if (len < 2 || pos + 2 + len > f.size()) {
      why = "truncated JPEG segment";
      return false;
    }
    const uint8_t* d = &f[pos + 4];
    const size_t dl = len - 2;
    if (mk == 0xDB) {  // DQT
      size_t o = 0;
      while (o < dl) {
        const int pq = d[o] >> 4, tq = d[o] & 15;
        ++o;
        if (tq > 3 || o + (pq ? 128 : 64) > dl) {
          why = "corrupt quantisation table";
          return false;
        }
        for (int k = 0; k < 64; ++k) {
          qt[tq][zz[k]] = pq ? (uint16_t)((d[o] << 8) | d[o + 1]) : d[o];
          o += pq ? 2 : 1;
        }
        qt_set[tq] = true;
      }
    } else if (mk == 0xC4) {  // DHT
      size_t o = 0;
      while (o + 17 <= dl) {
        const int tc = d[o] >> 4, th = d[o] & 15;
        if (tc > 1 || th > 3) {
          why = "corrupt Huffman table";
          return false;
        }
        JpegHuff& h = tc ? hac[th] : hdc[th];
        int total = 0;
        h.bits[0] = 0;
        for (int l = 1; l <= 16; ++l) total += (h.bits[l] = d[o + l]);
        o += 17;
        if (total > 256 || o + total > dl) {
          why = "corrupt Huffman table";
          return false;
        }
        memcpy(h.vals, d + o, total);
        o += total;
        h.build();
        h.present = true;
      }
    } else if (mk == 0xC0 || mk == 0xC1 || mk == 0xC2) {  // SOF0 / SOF1 / SOF2 (progressive, Huffman)
      progressive = mk == 0xC2;
      if (have_sof) {
        why = "corrupt JPEG (second frame header)";
        return false;
      }
      if (dl < 6 || d[0] != 8) {
        why = "unsupported JPEG sample precision";
        return false;
      }
      height = (d[1] << 8) | d[2];
      width = (d[3] << 8) | d[4];
      ncomp = d[5];
      if ((ncomp != 1 && ncomp != 3) || dl < 6 + 3 * (size_t)ncomp || !width || !height) {
        why = "unsupported JPEG component count";
        return false;
      }
      // (a header of a few bytes may claim 65535 x 65535 pixels: the planes -- and, progressive, 256 bytes of coefficients per
      // block -- are only allocated for a frame the file could plausibly hold: even an all-zero block costs a bit per scan)
      if ((uint64_t)width * (uint64_t)height > (uint64_t)1 << 28 || (uint64_t)width * (uint64_t)height / 64 / 8 > (uint64_t)f.size() * 16 + 4096) {
        why = "JPEG frame size implausible for the file's length";
        return false;
      }
      for (int c = 0; c < ncomp; ++c) {
        comp[c].id = d[6 + 3 * c];
        comp[c].h = d[7 + 3 * c] >> 4;
        comp[c].v = d[7 + 3 * c] & 15;
        comp[c].tq = d[8 + 3 * c];
        if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) {
          why = "unsupported JPEG sampling factors";
          return false;
        }
      }
      have_sof = true;
    } else if (mk >= 0xC3 && mk <= 0xCF && mk != 0xC4 && mk != 0xC8 && mk != 0xCC) {
      why = "unsupported JPEG coding process";  // (lossless, hierarchical, arithmetic coding)
      return false;
    } else if (mk == 0xDD) {
      if (dl >= 2) restart = (d[0] << 8) | d[1];
    } else if (mk == 0xEE) {
      if (dl >= 12 && !memcmp(d, "Adobe", 5)) adobe_transform = d[11];
    } else if (mk == 0xDA) {
      // SOS.  A baseline file is one interleaved scan over all components (what cv::imwrite and cameras write) or several
      // scans of fewer components each (jpegtran -scans, some scanners): a scan of ONE component is not interleaved -- its
      // MCU is a single block and the block grid is the component's own ceil(size / 8), not the frame's MCU grid (T.81 A.2.2).
      const int ns = dl >= 1 ? d[0] : 0;
      if (!have_sof || ns < 1 || ns > ncomp || dl < 1 + 2 * (size_t)ns + 3) {
        why = "unsupported JPEG scan layout";
        return false;
      }
      int sc[3];
      for (int k = 0; k < ns; ++k) {
        int c = -1;
        for (int j = 0; j < ncomp; ++j)
          if (comp[j].id == d[1 + 2 * k]) c = j;
        if (c < 0 || (k > 0 && c <= sc[k - 1]) || (comp[c].coded && !progressive)) {  // (frame order; sequential: every component coded once)
          why = "unsupported JPEG scan layout";
          return false;
        }
        sc[k] = c;
        comp[c].td = d[2 + 2 * k] >> 4;
        comp[c].ta = d[2 + 2 * k] & 15;
        if (comp[c].td > 3 || comp[c].ta > 3 || !qt_set[comp[c].tq] ||
            (!progressive && (!hdc[comp[c].td].present || !hac[comp[c].ta].present))) {
          why = "JPEG scan refers to a missing table";
          return false;
        }
      }
      // spectral selection and successive approximation (progressive scans; a sequential scan carries 0, 63, 0, 0)
      const int Ss = d[1 + 2 * ns], Se = d[2 + 2 * ns], Ah = d[3 + 2 * ns] >> 4, Al = d[3 + 2 * ns] & 15;
      if (progressive) {
        bool ok = Ss <= Se && Se <= 63 && Al <= 13 && Ah <= 13 && (Ah == 0 || Ah == Al + 1) && (Ss == 0 ? Se == 0 : ns == 1);
        for (int k = 0; ok && k < ns; ++k)
          ok = Ss == 0 ? (Ah != 0 || hdc[comp[sc[k]].td].present) : hac[comp[sc[k]].ta].present;
        if (!ok) {
          why = "unsupported progressive JPEG scan";
          return false;
        }
      }
      if (!planes_ready) {
        if ((size_t)width * height > ((size_t)1 << 28)) {
          why = "image too large";
          return false;
        }
        for (int c = 0; c < ncomp; ++c) hmax = std::max(hmax, comp[c].h), vmax = std::max(vmax, comp[c].v);
        if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;  // (a single-component frame: 1 x 1 blocks)
        const int fx = (width + 8 * hmax - 1) / (8 * hmax), fy = (height + 8 * vmax - 1) / (8 * vmax);
        for (int c = 0; c < ncomp; ++c) {
          comp[c].bw = fx * comp[c].h;
          comp[c].bh = fy * comp[c].v;
          comp[c].dw = (width * comp[c].h + hmax - 1) / hmax;
          comp[c].dhh = (height * comp[c].v + vmax - 1) / vmax;
          // (blocks no scan reaches stay at the level shift, as libjpeg's zeroed coefficient arrays decode)
          comp[c].plane.assign((size_t)comp[c].bw * 8 * comp[c].bh * 8, 128);
          if (progressive) comp[c].coef.assign((size_t)comp[c].bw * comp[c].bh * 64, 0);
        }
        planes_ready = true;
      }
      for (int k = 0; k < ns; ++k)
        if (!comp[sc[k]].q_latched) {
          memcpy(comp[sc[k]].q, qt[comp[sc[k]].tq], sizeof comp[sc[k]].q);
          comp[sc[k]].q_latched = true;
        }
      const bool inter = ns > 1;
      const int mcux = inter ? (width + 8 * hmax - 1) / (8 * hmax) : (comp[sc[0]].dw + 7) / 8;
      const int mcuy = inter ? (height + 8 * vmax - 1) / (8 * vmax) : (comp[sc[0]].dhh + 7) / 8;
      // ---- entropy-coded data: MCU by MCU
      JpegBits br{f.data(), f.size(), pos + 2 + len};
      int pred[3] = {0, 0, 0}, until_restart = restart;
      unsigned eobrun = 0;  // (progressive AC scans: blocks still to come that the last end-of-band code covers)
      auto sat = [](long v) { return (int32_t)std::max(-(1L << 20), std::min(1L << 20, v)); };  // (corrupt streams must not overflow)
      for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
          if (restart && until_restart == 0) {
            // byte-align, expect RSTn
            br.reset();
            while (br.pos + 1 < f.size() && !(f[br.pos] == 0xFF && f[br.pos + 1] >= 0xD0 && f[br.pos + 1] <= 0xD7)) ++br.pos;
            br.pos += 2;
            pred[0] = pred[1] = pred[2] = 0;
            eobrun = 0;
            until_restart = restart;
          }
          for (int k = 0; k < ns; ++k) {
            const int c = sc[k];
            const int nbx = inter ? comp[c].h : 1, nby = inter ? comp[c].v : 1;
            for (int by = 0; by < nby; ++by)
              for (int bx = 0; bx < nbx; ++bx) {
                if (progressive) {
                  // jdphuff.c: the four kinds of scan, into the block's coefficients
                  int32_t* cf = &comp[c].coef[((size_t)(my * nby + by) * comp[c].bw + (mx * nbx + bx)) * 64];
                  if (Ss == 0) {
                    if (Ah == 0) {  // DC, first pass: the difference, scaled
                      const int t = br.decode(hdc[comp[c].td]);
                      if (t < 0 || t > 15) {
                        why = "corrupt JPEG data (DC code)";
                        return false;
                      }
                      pred[c] = sat((long)pred[c] + (t ? jpeg_extend(br.get(t), t) : 0));
                      cf[0] = sat((long)pred[c] * (1L << Al));
                    } else if (br.get(1)) {  // DC, refinement: one more bit
                      cf[0] |= 1 << Al;
                    }
                    continue;
                  }
                  const int32_t p1 = 1 << Al, m1 = -(1 << Al);
                  if (Ah == 0) {  // AC, first pass
                    if (eobrun > 0) {
                      --eobrun;
                      continue;
                    }
                    for (int kk = Ss; kk <= Se; ++kk) {
                      const int rs = br.decode(hac[comp[c].ta]);
                      if (rs < 0) {
                        why = "corrupt JPEG data (AC code)";
                        return false;
                      }
                      const int r = rs >> 4, sz = rs & 15;
                      if (sz) {
                        kk += r;
                        if (kk > 63) {
                          why = "corrupt JPEG data (run past the block)";
                          return false;
                        }
                        cf[zz[kk]] = sat((long)jpeg_extend(br.get(sz), sz) * (1L << Al));
                      } else if (r == 15) {
                        kk += 15;
                      } else {  // end of band for this and the next eobrun blocks
                        eobrun = 1u << r;
                        if (r) eobrun += (unsigned)br.get(r);
                        --eobrun;
                        break;
                      }
                    }
                    continue;
                  }
                  // AC, refinement: new coefficients of magnitude 1 << Al, and one correction bit for every coefficient that is
                  // already nonzero, in the order the runs pass over them
                  auto correct = [&](int32_t& v) {
                    if (br.get(1) && (v & p1) == 0) v += v >= 0 ? p1 : m1;
                  };
                  int kk = Ss;
                  if (eobrun == 0) {
                    for (; kk <= Se; ++kk) {
                      const int rs = br.decode(hac[comp[c].ta]);
                      if (rs < 0) {
                        why = "corrupt JPEG data (AC code)";
                        return false;
                      }
                      int r = rs >> 4;
                      int32_t nv = 0;
                      if (rs & 15) {
                        nv = br.get(1) ? p1 : m1;  // (the size is 1 in a valid stream)
                      } else if (r != 15) {
                        eobrun = 1u << r;
                        if (r) eobrun += (unsigned)br.get(r);
                        break;
                      }
                      for (; kk <= Se; ++kk) {
                        int32_t& v = cf[zz[kk]];
                        if (v != 0) correct(v);
                        else if (--r < 0) break;
                      }
                      if (nv && kk <= 63) cf[zz[kk]] = nv;
                    }
                  }
                  if (eobrun > 0) {
                    for (; kk <= Se; ++kk)
                      if (cf[zz[kk]] != 0) correct(cf[zz[kk]]);
                    --eobrun;
                  }
                  continue;
                }
                int blk[64];
                memset(blk, 0, sizeof blk);
                const int t = br.decode(hdc[comp[c].td]);
                if (t < 0 || t > 11) {
                  why = "corrupt JPEG data (DC code)";
                  return false;
                }
                const int diff = t ? jpeg_extend(br.get(t), t) : 0;
                pred[c] = std::max(-(1 << 20), std::min(1 << 20, pred[c] + diff));  // (a corrupt stream must not overflow; real DC values fit 12 bits)
                blk[0] = (int)std::max(-(1L << 26), std::min(1L << 26, (long)pred[c] * qt[comp[c].tq][0]));
                for (int kk = 1; kk < 64;) {
                  const int rs = br.decode(hac[comp[c].ta]);
                  if (rs < 0) {
                    why = "corrupt JPEG data (AC code)";
                    return false;
                  }
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r != 15) break;  // EOB
                    kk += 16;
                    continue;
                  }
                  kk += r;
                  if (kk > 63) {
                    why = "corrupt JPEG data (run past the block)";
                    return false;
                  }
                  blk[zz[kk]] = (int)std::max(-(1L << 26), std::min(1L << 26, (long)jpeg_extend(br.get(sz), sz) * qt[comp[c].tq][zz[kk]]));
                  ++kk;
                }
                const int px = (mx * nbx + bx) * 8, py = (my * nby + by) * 8;
                jpeg_idct_islow(blk, &comp[c].plane[(size_t)py * comp[c].bw * 8 + px], comp[c].bw * 8);
              }
          }
          if (restart) --until_restart;
        }
      for (int k = 0; k < ns; ++k) comp[sc[k]].coded = true;
      if (++n_scans > 500) {  // (libjpeg-turbo's own limit for fuzzed files: every scan walks all blocks)
        why = "corrupt JPEG (too many scans)";
        return false;
      }
      // the next marker: the reader never steps over one, so it lies at or behind its position
      pos = br.pos;
      while (pos + 1 < f.size() && !(f[pos] == 0xFF && f[pos + 1] != 0 && f[pos + 1] != 0xFF && !(f[pos + 1] >= 0xD0 && f[pos + 1] <= 0xD7))) ++pos;
      bool all = true;
      for (int c = 0; c < ncomp; ++c) all = all && comp[c].coded;
      if (all && !progressive) break;  // (a progressive frame goes on until EOI)
      continue;
    } else if (mk == 0xD9) {
      break;
    }
    pos += 2 + len;
  }
  if (!n_scans) {
    why = "JPEG without a scan";
    return false;
  }
  if (progressive) {
    // every scan is in: dequantise and transform what the blocks hold.  (libjpeg smooths blocks whose first five AC
    // coefficients are not fully refined when output starts, jdcoefct.c; for a file that carries all its scans -- what
    // encoders write -- there are none, and the output is that of the sequential path on the same coefficients.)
    for (int c = 0; c < ncomp; ++c)
      for (int by = 0; by < comp[c].bh; ++by)
        for (int bx = 0; bx < comp[c].bw; ++bx) {
          const int32_t* cf = &comp[c].coef[((size_t)by * comp[c].bw + bx) * 64];
          int blk[64];
          for (int j = 0; j < 64; ++j) blk[j] = (int)std::max(-(1L << 26), std::min(1L << 26, (long)cf[j] * comp[c].q[j]));
          jpeg_idct_islow(blk, &comp[c].plane[(size_t)by * 8 * comp[c].bw * 8 + bx * 8], comp[c].bw * 8);
        }
  }
  // (a file that ends before every component was coded decodes like libjpeg's premature-EOI case: what is missing is gray)
  // ---- upsampling (jdsample.c) to full resolution planes of width x height
  std::vector<uint8_t> full[3];
  for (int c = 0; c < ncomp; ++c) {
    const int hs = hmax / comp[c].h, vs = vmax / comp[c].v;  // expansion factors: 1 or 2
    const int W = comp[c].dw, Hh = comp[c].dhh, pitch = comp[c].bw * 8;
    const uint8_t* src = comp[c].plane.data();
    full[c].assign((size_t)width * height, 0);
    const bool fancy = W > 2;  // (libjpeg: fancy upsampling only when the component is more than two samples wide)
    for (int y = 0; y < height; ++y) {
      uint8_t* o = &full[c][(size_t)y * width];
      if (hs == 1 && vs == 1) {
        memcpy(o, src + (size_t)y * pitch, width);
      } else if (hs == 2 && vs == 1) {
        const uint8_t* in = src + (size_t)y * pitch;
        for (int x = 0; x < width; ++x) {
          const int i = x >> 1;
          if (!fancy) o[x] = in[i];
          else if ((x & 1) == 0) o[x] = i == 0 ? in[0] : (uint8_t)((in[i] * 3 + in[i - 1] + 1) >> 2);
          else o[x] = i == W - 1 ? in[i] : (uint8_t)((in[i] * 3 + in[i + 1] + 2) >> 2);
        }
      } else if (hs == 2 && vs == 2 && fancy) {
        const int iy = y >> 1, ny = (y & 1) ? std::min(iy + 1, Hh - 1) : std::max(iy - 1, 0);
        const uint8_t* in0 = src + (size_t)iy * pitch;
        const uint8_t* in1 = src + (size_t)ny * pitch;
        for (int x = 0; x < width; ++x) {
          const int i = x >> 1;
          const int cur = in0[i] * 3 + in1[i];
          if ((x & 1) == 0) {
            if (i == 0) o[x] = (uint8_t)((cur * 4 + 8) >> 4);
            else o[x] = (uint8_t)((cur * 3 + (in0[i - 1] * 3 + in1[i - 1]) + 8) >> 4);
          } else {
            if (i == W - 1) o[x] = (uint8_t)((cur * 4 + 7) >> 4);
            else o[x] = (uint8_t)((cur * 3 + (in0[i + 1] * 3 + in1[i + 1]) + 7) >> 4);
          }
        }
      } else if (hs == 1 && vs == 2) {
        // h1v2 (a 4:2:2 frame turned by 90 degrees): libjpeg-turbo's h1v2_fancy_upsample, the triangle filter down the
        // columns -- 3/4 of the nearer row, 1/4 of the farther one, rounding bias 1 above and 2 below; no width condition
        const int iy = y >> 1, ny = (y & 1) ? std::min(iy + 1, Hh - 1) : std::max(iy - 1, 0);
        const uint8_t* in0 = src + (size_t)iy * pitch;
        const uint8_t* in1 = src + (size_t)ny * pitch;
        const int bias = (y & 1) ? 2 : 1;
        for (int x = 0; x < width; ++x) o[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
      } else {  // replication (a component too narrow for the triangle filter)
        const uint8_t* in = src + (size_t)(y / vs) * pitch;
        for (int x = 0; x < width; ++x) o[x] = in[x / hs];
      }
    }
  }
  // ---- colour conversion (jdcolor.c): gray -> B = G = R; YCbCr -> RGB with the 16-bit tables
  bgr = cv::Mat(height, width, CV_8UC3);
  uint8_t* out = bgr.ptr();
  if (ncomp == 1) {
    for (size_t i = 0; i < (size_t)width * height; ++i) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = full[0][i];
    return true;
  }
  if (adobe_transform == 0) {  // Adobe marker: components are R, G, B
    for (size_t i = 0; i < (size_t)width * height; ++i) {
      out[3 * i] = full[2][i];
      out[3 * i + 1] = full[1][i];
      out[3 * i + 2] = full[0][i];
    }
    return true;
  }
  int cr_r[256], cb_b[256];
  long cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) {
    const long x = i - 128;
    cr_r[i] = (int)((91881L * x + 32768) >> 16);   // FIX(1.40200)
    cb_b[i] = (int)((116130L * x + 32768) >> 16);  // FIX(1.77200)
    cr_g[i] = -46802L * x;                         // FIX(0.71414)
    cb_g[i] = -22554L * x + 32768;                 // FIX(0.34414) + ONE_HALF
  }
  auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
  for (size_t i = 0; i < (size_t)width * height; ++i) {
    const int y = full[0][i], cb = full[1][i], cr = full[2][i];
    out[3 * i + 2] = clamp8(y + cr_r[cr]);
    out[3 * i + 1] = clamp8(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
    out[3 * i] = clamp8(y + cb_b[cb]);
  }
  return true;
}

// cv::imread(path) turns a JPEG the way its EXIF orientation says (OpenCV >= 3.1, loadsave.cpp ApplyExifOrientation; the
// reference's call, src/Sfm.cpp:150, carries no IMREAD_IGNORE_ORIENTATION): phone and camera photos taken upright come
// back upright.  The tag is found as modules/imgcodecs/src/exif.cpp finds it: the markers are walked from the start of the
// file, the FIRST APP1 segment is taken, six bytes ("Exif\0\0") are skipped, and behind them a TIFF header (II / MM, 42, the
// offset of IFD0) leads to the directory whose entry 0x0112 holds the orientation as a SHORT.  Returns 1 (nothing to do)
// whenever anything is missing or out of bounds.
int jpeg_exif_orientation(const std::vector<uint8_t>& f) {
  size_t pos = 0;
  const uint8_t* t = nullptr;
  size_t tn = 0;
  while (pos + 2 <= f.size()) {
    const int mk = f[pos + 1];
    pos += 2;
    if (mk == 0xD8 || mk == 0xD9) continue;  // SOI, EOI: no size field
    const bool sized = mk == 0xC0 || mk == 0xC2 || mk == 0xC4 || mk == 0xDB || mk == 0xDD || mk == 0xDA || (mk >= 0xD0 && mk <= 0xD7) ||
                       mk == 0xE0 || (mk >= 0xE2 && mk <= 0xEF) || mk == 0xFE;
    if (!sized && mk != 0xE1) return 1;  // (any other marker ends the search)
    if (pos + 2 > f.size()) return 1;
    const size_t len = ((size_t)f[pos] << 8) | f[pos + 1];
    if (len < 2) return 1;
    if (mk != 0xE1) {
      pos += len;
      continue;
    }
    if (len <= 6 + 2 || pos + len > f.size()) return 1;
    t = &f[pos + 2 + 6];
    tn = len - 2 - 6;
    break;
  }
  if (!t || tn < 8 || t[0] != t[1] || (t[0] != 'I' && t[0] != 'M')) return 1;
  const bool le = t[0] == 'I';
  auto u16 = [&](size_t o) -> uint32_t { return le ? (uint32_t)(t[o] | (t[o + 1] << 8)) : (uint32_t)((t[o] << 8) | t[o + 1]); };
  auto u32 = [&](size_t o) -> uint32_t { return le ? (u16(o) | (u16(o + 2) << 16)) : ((u16(o) << 16) | u16(o + 2)); };
  if (u16(2) != 0x002A) return 1;
  size_t off = u32(4);
  if (off + 2 > tn) return 1;
  const size_t n = u16(off);
  off += 2;
  for (size_t i = 0; i < n && off + 12 <= tn; ++i, off += 12)
    if (u16(off) == 0x0112) {
      const int o = (int)u16(off + 8);
      return o >= 1 && o <= 8 ? o : 1;
    }
  return 1;
}
// loadsave.cpp ExifTransform: 2 mirror, 3 half turn, 4 upside down, 5 transpose, 6 transpose + mirror (a quarter turn clockwise),
// 7 transpose + half turn, 8 transpose + upside down
void apply_exif_orientation(cv::Mat& img, int o) {
  if (o <= 1 || o > 8) return;
  const int H = img.rows, W = img.cols;
  const bool tr = o >= 5;
  const int fl = o == 2 || o == 6 ? 1 : o == 3 || o == 7 ? -1 : o == 4 || o == 8 ? 0 : 2;  // cv::flip code (2: none)
  cv::Mat out(tr ? W : H, tr ? H : W, CV_8UC3);
  const uint8_t* s = img.ptr();
  uint8_t* d = out.ptr();
  for (int y = 0; y < out.rows; ++y)
    for (int x = 0; x < out.cols; ++x) {
      // undo the flip, then the transpose: where the output pixel comes from
      int ty = y, tx = x;
      if (fl == 1 || fl == -1) tx = out.cols - 1 - x;
      if (fl == 0 || fl == -1) ty = out.rows - 1 - y;
      const int sy = tr ? tx : ty, sx = tr ? ty : tx;
      memcpy(d + ((size_t)y * out.cols + x) * 3, s + ((size_t)sy * W + sx) * 3, 3);
    }
  img = out;
}

std::string lower_ext(const std::string& p) {
  const size_t dot = p.find_last_of('.'), slash = p.find_last_of('/');
  if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
  std::string e = p.substr(dot);
  for (char& c : e) c = (char)std::tolower((unsigned char)c);
  return e;
}

// one <Name type_id="opencv-matrix"> node: rows, cols, data (numbers as text)
bool xml_matrix(const std::string& xml, const std::string& name, cv::Mat_<double>& out) {
  const size_t a = xml.find("<" + name);
  if (a == std::string::npos) return false;
  const size_t b = xml.find("</" + name + ">", a);
  if (b == std::string::npos) return false;
  const std::string node = xml.substr(a, b - a);
  auto field = [&](const char* tag, std::string& v) {
    const std::string open = std::string("<") + tag + ">", close = std::string("</") + tag + ">";
    const size_t i = node.find(open), j = node.find(close);
    if (i == std::string::npos || j == std::string::npos || j < i) return false;
    v = node.substr(i + open.size(), j - i - open.size());
    return true;
  };
  std::string r, c, d;
  if (!field("rows", r) || !field("cols", c) || !field("data", d)) return false;
  const int rows = atoi(r.c_str()), cols = atoi(c.c_str());
  if (rows <= 0 || cols <= 0) return false;
  out = cv::Mat_<double>(rows, cols);
  std::istringstream is(d);
  for (int i = 0; i < rows * cols; ++i)
    if (!(is >> out.data[i])) return false;
  return true;
}
void mkdirs(const std::string& p) {
  for (size_t i = 1; i <= p.size(); ++i)
    if (i == p.size() || p[i] == '/') mkdir(p.substr(0, i).c_str(), 0777);
}

}  // namespace

// reference src/Sfm.cpp:257-296
void StructFromMotion::extractFeature() {
  std::cout << "Extracting features from all images..." << std::endl;
  nCameraPoses.resize(mGrayImages.size());
  imagesKeypoints.resize(mGrayImages.size(), std::vector<cv::KeyPoint>());
  imagesDescriptors.resize(mGrayImages.size(), cv::Mat());
  imagesPts2D.resize(mGrayImages.size(), std::vector<cv::Point2d>());
  if (detector == 1)
    std::cout << "No detector choose. Using default:" << "SIFT(Scale-Invariant Feature Transform) detector." << "\n"
              << "Parameters:" << "\n" << "nFeatures = 0\n" << "nOctaveLayers = 3\n" << "contrastThreshold = 0.04\n"
              << "edgeThreshold = 10\n" << "sigma = 1.6" << std::endl;
  std::cout << "*-- Features --*" << std::endl;
  releaseDeviceDescriptors();
  bool batched = false;
  if (detector == 1 && !mGrayImages.empty()) {
    // the loop of :283-290 as ONE call: the images go through the device front end several at a time, and the descriptor
    // rows stay in HBM, where the matcher adopts them (no download / upload between extraction and matching)
    const int n = (int)mGrayImages.size();
    std::vector<const uint8_t*> ptrs(n);
    std::vector<int32_t> rows(n), cols(n), nk(n, 0);
    std::vector<float*> kp(n, nullptr);
    std::vector<void*> dd(n, nullptr);
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      const cv::Mat& g = mGrayImages[i];
      ok = ok && g.channels() == 1 && g.depth == CV_8U && !g.empty();
      ptrs[i] = g.ptr();
      rows[i] = g.rows;
      cols[i] = g.cols;
    }
    const int rc = ok ? sfmhip_sift_batch(sfm_hip_context(), n, ptrs.data(), rows.data(), cols.data(), 3, 0.04, 10, 1.6, kp.data(),
                                          dd.data(), nk.data())
                      : SFMHIP_ERR_ARG;
    if (rc == SFMHIP_OK) {
      batched = true;
      devDescriptors = dd;
      for (int i = 0; i < n; ++i) {
        setKeypoints(i, kp[i], nk[i]);
        imagesDescriptors[i] = cv::Mat();  // the shape only: the rows are in HBM until descriptors() is asked
        imagesDescriptors[i].rows = nk[i];
        imagesDescriptors[i].cols = 128;
        imagesDescriptors[i].depth = CV_32F;
        sfmhip_host_free(kp[i]);
        std::cout << "Image:" << i << " --> " << imagesKeypoints.at(i).size() << " kps" << std::endl;
      }
    } else if (ok) {
      std::cerr << "extractFeature: " << sfmhip_error_string(rc) << std::endl;
    }
  }
  for (size_t n = 0; !batched && n < mGrayImages.size(); n++) {
    getFeature(mGrayImages.at(n), (int)n);
    std::cout << "Image:" << n << " --> " << imagesKeypoints.at(n).size() << " kps" << std::endl;
  }
  releaseDeviceSet();  // (new descriptors: what the matcher holds on the device is stale)
  clearPairCache();
}

// keypoint records of the C ABI (x, y, size, angle, response, octave bits) -> imagesKeypoints / imagesPts2D
void StructFromMotion::setKeypoints(int numImage, const float* kp, int n) {
  std::vector<cv::KeyPoint> kps((size_t)n);
  for (int i = 0; i < n; ++i) {
    cv::KeyPoint& k = kps[i];
    k.pt.x = kp[6 * i];
    k.pt.y = kp[6 * i + 1];
    k.size = kp[6 * i + 2];
    k.angle = kp[6 * i + 3];
    k.response = kp[6 * i + 4];
    std::memcpy(&k.octave, &kp[6 * i + 5], 4);
    k.class_id = -1;
  }
  std::vector<cv::Point2d> points2d;
  keypointstoPoints(kps, points2d);
  imagesKeypoints[numImage] = kps;
  imagesPts2D[numImage] = points2d;
}

// reference src/Sfm.cpp:300-403
void StructFromMotion::getFeature(const cv::Mat& image, const int& numImage) {
  if (detector != 1) {
    std::cerr << "getFeature: only the SIFT detector (1) is built" << std::endl;
    return;
  }
  if (image.channels() != 1 || image.depth != CV_8U || image.empty()) {
    std::cerr << "getFeature: an 8-bit gray image is expected" << std::endl;
    return;
  }
  int32_t n = 0;
  sfmhip_ctx* ctx = sfm_hip_context();
  int cap = std::max(1024, image.rows * image.cols / 48);  // one pass when the guess holds
  std::vector<float> kp(6 * (size_t)cap), desc(128 * (size_t)cap);
  int rc = sfmhip_sift_detect_and_compute(ctx, image.ptr(), image.rows, image.cols, 3, 0.04, 10, 1.6, cap, kp.data(), desc.data(), &n);
  if (rc == SFMHIP_ERR_ARG && n > cap) {  // more keypoints than guessed: again with the count the call reported
    cap = n;
    kp.resize(6 * (size_t)cap);
    desc.resize(128 * (size_t)cap);
    rc = sfmhip_sift_detect_and_compute(ctx, image.ptr(), image.rows, image.cols, 3, 0.04, 10, 1.6, cap, kp.data(), desc.data(), &n);
  }
  if (rc != SFMHIP_OK) {
    std::cerr << "getFeature: " << sfmhip_error_string(rc) << std::endl;
    return;
  }
  setKeypoints(numImage, kp.data(), n);
  // new rows for this image: what the matcher holds on the device (a set that may have ADOPTED the old rows' pointer) and
  // every cached pair list are stale -- let go of them before the old device copy is freed
  releaseDeviceSet();
  clearPairCache();
  if ((size_t)numImage < devDescriptors.size() && devDescriptors[numImage]) {  // (a device copy of older rows)
    sfmhip_device_free(devDescriptors[numImage]);
    devDescriptors[numImage] = nullptr;
  }
  imagesDescriptors[numImage] = cv::Mat(n, 128, CV_32F, n ? desc.data() : nullptr);
}

// reference src/Sfm.cpp:397-403
void StructFromMotion::keypointstoPoints(std::vector<cv::KeyPoint>& keypoints, Points2d& points2D) {
  points2D.clear();
  for (const cv::KeyPoint& kp : keypoints) points2D.push_back(cv::Point2d(kp.pt.x, kp.pt.y));
}

// reference src/Sfm.cpp:118-198
bool StructFromMotion::imagesLOAD(const std::string& directoryPath) {
  std::cout << "Getting images..." << std::flush;
  pathImages = directoryPath;
  nImagesPath.clear();
  nImages.clear();
  mColorImages.clear();
  mGrayImages.clear();
  DIR* dir = opendir(directoryPath.c_str());
  if (!dir) {
    std::cerr << "Cannot open directory: " << directoryPath << std::endl;
    return false;
  }
  const std::string base = directoryPath + (directoryPath.empty() || directoryPath.back() == '/' ? "" : "/");
  while (dirent* e = readdir(dir)) {
    const std::string ext = lower_ext(e->d_name);
    if (ext == ".jpg" || ext == ".png") nImagesPath.push_back(base + e->d_name);  // :129-135
  }
  closedir(dir);
  std::sort(nImagesPath.begin(), nImagesPath.end());  // :138
  if (nImagesPath.empty()) {
    std::cerr << "Unable to find valid files in images directory (\"" << directoryPath << "\")." << std::endl;
    return false;
  }
  std::cout << "Found " << nImagesPath.size() << " image files in directory." << std::endl;
  // the loop of :146-170, the files decoded side by side (the decoders here are plain scalar code: a camera JPEG takes a few
  // hundred milliseconds); the images are kept, and the first unreadable one is reported, in the sorted order of the loop
  const size_t nfiles = nImagesPath.size();
  std::vector<cv::Mat> decoded(nfiles);
  std::vector<std::string> whys(nfiles, "cannot read the file");
  std::vector<char> oks(nfiles, 0);
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (size_t i = next.fetch_add(1); i < nfiles; i = next.fetch_add(1)) try {
      std::vector<uint8_t> bytes;
      cv::Mat image;
      bool ok = read_file(nImagesPath[i], bytes);
      // (cv::imread picks the decoder by the file's signature, not by its name)
      const bool is_jpeg = bytes.size() >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8;
      if (ok && is_jpeg) {
        ok = decode_jpeg(bytes, image, whys[i]);
        if (ok) apply_exif_orientation(image, jpeg_exif_orientation(bytes));
      } else if (ok) {
        ok = decode_png(bytes, image, whys[i]);
      }
      if (ok) decoded[i] = image.rows > 480 && image.cols > 640 ? resize_linear_8u(image, 0.60, 0.60) : image;  // :153-155
      oks[i] = ok ? 1 : 0;
    } catch (const std::exception& e) {  // (bad_alloc / length_error of a decoder: the reference's "Unable to read image", not std::terminate)
      oks[i] = 0;
      whys[i] = std::string("decoder failed: ") + e.what();
    }
  };
  {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nthreads = std::min<size_t>(std::min<size_t>(hw, 16), nfiles);
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
  }
  for (size_t i = 0; i < nfiles; ++i) {
    if (!oks[i]) {  // cv::imread returns an empty Mat; the reference then reports and fails (:158-161)
      std::cerr << "[x]" << "\n" << "Unable to read image from file: " << nImagesPath[i] << " (" << whys[i] << ")" << std::endl;
      return false;
    }
    nImages.push_back(decoded[i]);
  }
  if (nImages.size() < 2) {
    std::cerr << "Sorry. is not enough images, 6 minimum" << std::endl;  // :172-175
    return false;
  }
  for (size_t i = 0; i < nImages.size(); ++i) {  // :177-195: decoded images are CV_8UC3 here, so the copy branch
    mColorImages.push_back(nImages[i]);
    mGrayImages.push_back(bgr_to_gray(mColorImages[i]));
  }
  return true;
}

// reference src/Sfm.cpp:203-252
bool StructFromMotion::getCameraMatrix(const std::string str) {
  std::cout << "Getting camera matrix..." << std::endl;
  std::vector<uint8_t> bytes;
  cv::Mat_<double> intrinsics, cameraDistCoeffs;
  if (read_file(str, bytes)) {
    const std::string xml(bytes.begin(), bytes.end());
    xml_matrix(xml, "Camera_Matrix", intrinsics);
    xml_matrix(xml, "Distortion_Coefficients", cameraDistCoeffs);
  }
  if (intrinsics.rows != 3 || intrinsics.cols != 3 || intrinsics(2, 0) != 0) {  // :216
    std::cerr << "Error: no found or invalid camera calibration file.xml" << std::endl;
    return false;
  }
  cv::Mat_<double> cam_matrix(3, 3);
  cam_matrix(0, 0) = intrinsics(0, 0);
  cam_matrix(1, 1) = intrinsics(1, 1);
  cam_matrix(0, 2) = intrinsics(0, 2);
  cam_matrix(1, 2) = intrinsics(1, 2);
  cam_matrix(2, 2) = 1;
  cv::Mat_<double> distortionC(1, 5);
  for (int i = 0; i < 5; ++i)  // slots in file order (the reference labels them k1,k2,k3,p1,p2: :230-236)
    distortionC(0, i) = (cameraDistCoeffs.rows * cameraDistCoeffs.cols > i) ? cameraDistCoeffs.data[i] : 0.0;
  cameraMatrix.K = cam_matrix;
  cameraMatrix.distCoef = distortionC;
  std::cout << "Camera matrix:" << "\n";
  for (int r = 0; r < 3; ++r) std::cout << cam_matrix(r, 0) << " " << cam_matrix(r, 1) << " " << cam_matrix(r, 2) << "\n";
  return true;
}

// reference src/Sfm.cpp:1246-1303
void StructFromMotion::PMVS2() {
  std::cout << "Creating folders for PMVS2..." << std::endl;
  mkdirs("denseCloud/visualize");
  mkdirs("denseCloud/txt");
  mkdirs("denseCloud/models");
  {
    std::ofstream option("denseCloud/options.txt");
    option << "minImageNum 5" << std::endl;
    option << "CPU 4" << std::endl;
    option << "timages  -1 " << 0 << " " << (nImages.size() - 1) << std::endl;
    option << "oimages 0" << std::endl;
    option << "level 1" << std::endl;
  }
  for (size_t i = 0; i < nCameraPoses.size(); ++i) {
    char str[256];
    if (i < nImagesPath.size()) {  // `cp -f <input> denseCloud/visualize/%04d.jpg`: a byte copy whatever the format
      std::vector<uint8_t> bytes;
      std::snprintf(str, sizeof str, "denseCloud/visualize/%04d.jpg", (int)i);
      if (read_file(nImagesPath[i], bytes)) {
        FILE* o = fopen(str, "wb");
        if (o) {
          fwrite(bytes.data(), 1, bytes.size(), o);
          fclose(o);
        }
      }
    }
    std::snprintf(str, sizeof str, "denseCloud/txt/%04d.txt", (int)i);
    std::ofstream ofs(str);
    const cv::Matx34d& P = nCameraPoses[i];
    cv::Matx34d pose;  // K*P
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) {
        double v = 0;
        for (int k = 0; k < 3; ++k) v += cameraMatrix.K(r, k) * P(k, c);
        pose(r, c) = v;
      }
    ofs << "CONTOUR" << std::endl;
    ofs << pose(0, 0) << " " << pose(0, 1) << " " << pose(0, 2) << " " << pose(0, 3) << "\n"
        << pose(1, 0) << " " << pose(1, 1) << " " << pose(1, 2) << " " << pose(1, 3) << "\n"
        << pose(2, 0) << " " << pose(2, 1) << " " << pose(2, 2) << " " << pose(2, 3) << std::endl;
    ofs << std::endl;
  }
  std::cout << "Camera poses saved." << "\n" << "Camera images saved." << std::endl;
}

// reference src/Sfm.cpp:69-81
size_t StructFromMotion::convertPLYtoPCD(const std::string& plyPath, const std::string& pcdPath) {
  std::vector<uint8_t> f;
  if (!read_file(plyPath, f)) return 0;
  // header: lines up to "end_header"
  size_t pos = 0;
  auto line = [&](std::string& out) {
    if (pos >= f.size()) return false;
    size_t e = pos;
    while (e < f.size() && f[e] != '\n') ++e;
    out.assign((const char*)&f[pos], e - pos);
    if (!out.empty() && out.back() == '\r') out.pop_back();
    pos = e + 1;
    return true;
  };
  std::string l;
  if (!line(l) || l != "ply") return 0;
  struct Prop {
    std::string type, name;
  };
  std::vector<Prop> props;
  size_t nvert = 0;
  bool binary = false, in_vertex = false, done = false;
  while (line(l)) {
    std::istringstream is(l);
    std::string w;
    is >> w;
    if (w == "format") {
      is >> w;
      if (w == "binary_little_endian") binary = true;
      else if (w != "ascii") return 0;
    } else if (w == "element") {
      std::string nm;
      size_t cnt = 0;
      is >> nm >> cnt;
      in_vertex = nm == "vertex";
      if (in_vertex) nvert = cnt;
    } else if (w == "property" && in_vertex) {
      Prop p;
      is >> p.type >> p.name;
      if (p.type == "list") return 0;
      props.push_back(p);
    } else if (w == "end_header") {
      done = true;
      break;
    }
  }
  if (!done || !nvert || props.empty()) return 0;
  auto size_of = [](const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "double" || t == "float64") return 8;
    return 4;
  };
  int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1;
  for (size_t i = 0; i < props.size(); ++i) {
    const std::string& n = props[i].name;
    if (n == "x") ix = (int)i;
    else if (n == "y") iy = (int)i;
    else if (n == "z") iz = (int)i;
    else if (n == "red" || n == "diffuse_red") ir = (int)i;
    else if (n == "green" || n == "diffuse_green") ig = (int)i;
    else if (n == "blue" || n == "diffuse_blue") ib = (int)i;
  }
  if (ix < 0 || iy < 0 || iz < 0) return 0;
  std::vector<float> xyz;
  std::vector<uint32_t> rgb;
  xyz.reserve(3 * nvert);
  rgb.reserve(nvert);
  std::vector<double> v(props.size());
  std::istringstream body;
  if (!binary) body.str(std::string((const char*)f.data() + std::min(pos, f.size()), f.size() - std::min(pos, f.size())));
  for (size_t k = 0; k < nvert; ++k) {
    for (size_t i = 0; i < props.size(); ++i) {
      if (!binary) {
        if (!(body >> v[i])) return 0;
        continue;
      }
      const int sz = size_of(props[i].type);
      if (pos + sz > f.size()) return 0;
      const uint8_t* p = &f[pos];
      const std::string& t = props[i].type;
      if (t == "float" || t == "float32") {
        float x;
        memcpy(&x, p, 4);
        v[i] = x;
      } else if (sz == 8) memcpy(&v[i], p, 8);
      else if (sz == 1) v[i] = (t == "char" || t == "int8") ? (double)(int8_t)p[0] : (double)p[0];
      else if (sz == 2) {
        uint16_t x;
        memcpy(&x, p, 2);
        v[i] = (t == "short" || t == "int16") ? (double)(int16_t)x : (double)x;
      } else {
        uint32_t x;
        memcpy(&x, p, 4);
        v[i] = (t == "int" || t == "int32") ? (double)(int32_t)x : (double)x;
      }
      pos += sz;
    }
    xyz.push_back((float)v[ix]);
    xyz.push_back((float)v[iy]);
    xyz.push_back((float)v[iz]);
    const uint32_t r = ir >= 0 ? (uint32_t)v[ir] & 255u : 0, g = ig >= 0 ? (uint32_t)v[ig] & 255u : 0, b = ib >= 0 ? (uint32_t)v[ib] & 255u : 0;
    rgb.push_back((r << 16) | (g << 8) | b);
  }
  std::ofstream o(pcdPath);
  if (!o) return 0;
  o << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
    << "WIDTH " << nvert << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << nvert << "\nDATA ascii\n";
  o.precision(8);
  auto put = [&](float x) {
    if (std::isfinite(x)) o << x;
    else o << "nan";
  };
  for (size_t k = 0; k < nvert; ++k) {
    float packed;
    memcpy(&packed, &rgb[k], 4);
    put(xyz[3 * k]);
    o << " ";
    put(xyz[3 * k + 1]);
    o << " ";
    put(xyz[3 * k + 2]);
    o << " ";
    put(packed);
    o << "\n";
  }
  return nvert;
}
