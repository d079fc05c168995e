"""ORACLE (test infrastructure, not product): cv::xfeatures2d::SIFT::create(0, 3, 0.04, 10, 1.6)->detectAndCompute, the
detector / descriptor front end of StructFromMotion::getFeature (reference src/Sfm.cpp:300-330).

PARITY UNPINNED.  The algorithm lives in OpenCV 3.4.1 (opencv_contrib xfeatures2d/src/sift.cpp, imgproc's GaussianBlur /
resize), which is neither under /root/reference nor in this image, and the reference holds no keypoint or descriptor
fixture.  Restated from the published sources (float pipeline: sift_wt = float, SIFT_FIXPT_SCALE = 1):
createInitialImage (x2 INTER_LINEAR, blur to sigma 1.6 assuming 0.5 in the input), buildGaussianPyramid (nOctaves =
round(log2(min side)) - 2 + 1, 6 images per octave, INTER_NEAREST halving), buildDoGPyramid, findScaleSpaceExtrema
(threshold floor(0.5 * 0.04 / 3 * 255) = 1, 26 neighbours, adjustLocalExtrema: 5 Newton steps with Cramer's rule in
float, contrast and edge tests), calcOrientationHist (36 bins, smoothing 1-4-6-4-1, peaks >= 0.8 max, parabolic bin),
KeyPointsFilter::removeDuplicatedSorted, the x0.5 rescale of the doubled first octave, calcSIFTDescriptor (4 x 4 x 8,
trilinear, clip 0.2, x512 saturated to 8 bit, stored as float).  Where OpenCV's exact float results depend on its SIMD
paths or its own exp / atan2 approximations, this file fixes ONE order of operations (stated at each place) that the HIP
code follows too (exp / cos / sin / pow: evaluated in double and rounded to float, where OpenCV calls its own float
approximations); keypoints and descriptors can therefore differ from OpenCV's in the last bits, and near thresholds in
membership.
"""
import numpy as np

F = np.float32
SIFT_INIT_SIGMA, IMG_BORDER, MAX_INTERP_STEPS = 0.5, 5, 5
ORI_BINS, ORI_SIG_FCTR, ORI_RADIUS, ORI_PEAK_RATIO = 36, F(1.5), F(3 * 1.5), F(0.8)
DESCR_WIDTH, DESCR_BINS, DESCR_SCL_FCTR, DESCR_MAG_THR, INT_DESCR_FCTR = 4, 8, F(3.0), F(0.2), F(512.0)
FLT_EPSILON = F(np.finfo(np.float32).eps)


def cv_round(x):
    return int(np.rint(np.float64(x)))


def gaussian_kernel(sigma):
    """cv::getGaussianKernel(n, sigma, CV_32F), n = cvRound(sigma * 4 * 2 + 1) | 1 (GaussianBlur on a float image)"""
    n = cv_round(sigma * 4 * 2 + 1) | 1
    scale2x = -0.5 / (np.float64(sigma) * np.float64(sigma))
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    cf = np.exp(scale2x * x * x).astype(np.float32)
    s = 1.0 / np.sum(cf.astype(np.float64))          # (the sum is accumulated in double from the float values)
    return (cf.astype(np.float64) * s).astype(np.float32)


def _reflect101(i, n):
    """cv::borderInterpolate(BORDER_REFLECT_101), repeated for borders wider than the image"""
    i = np.array(i, np.int64)
    if n == 1:
        return np.zeros_like(i)
    while ((i < 0) | (i >= n)).any():
        i = np.where(i < 0, -i, i)
        i = np.where(i >= n, 2 * (n - 1) - i, i)
    return i


def gaussian_blur(img, sigma):
    """separable, BORDER_REFLECT_101; row pass: sum_k k[k] * s[x + k - r] accumulated left to right; column pass
    (symmetric kernel): k[r] * s[c], then += k[r + j] * (s[c + j] + s[c - j]), j = 1..r -- all float32"""
    k = gaussian_kernel(sigma)
    r = len(k) // 2
    h, w = img.shape
    cols = _reflect101(np.arange(-r, w + r), w)
    pad = img[:, cols]
    tmp = np.zeros((h, w), np.float32)
    for j in range(len(k)):
        term = k[j] * pad[:, j:j + w]
        tmp = term if j == 0 else tmp + term
    rows = _reflect101(np.arange(-r, h + r), h)
    pad = tmp[rows, :]
    out = k[r] * pad[r:r + h, :]
    for j in range(1, r + 1):
        out = out + k[r + j] * (pad[r + j:r + j + h, :] + pad[r - j:r - j + h, :])
    return out.astype(np.float32)


def resize_linear_x2(img):
    """cv::resize(src, dst, Size(2w, 2h), INTER_LINEAR) for CV_32F: horizontal pass then vertical, float weights"""
    h, w = img.shape

    def taps(d, s):
        f = ((np.arange(d) + 0.5) * (s / float(d)) - 0.5).astype(np.float32)       # (float)((dx + 0.5) * scale - 0.5)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        lo = i < 0
        f[lo], i[lo] = 0, 0
        hi = i >= s - 1
        f[hi], i[hi] = 0, s - 1
        return i, (F(1.0) - f).astype(np.float32), f
    ix, a0, a1 = taps(2 * w, w)
    iy, b0, b1 = taps(2 * h, h)
    ix1 = np.minimum(ix + 1, w - 1)
    hz = (img[:, ix] * a0[None, :] + img[:, ix1] * a1[None, :]).astype(np.float32)
    iy1 = np.minimum(iy + 1, h - 1)
    return (hz[iy, :] * b0[:, None] + hz[iy1, :] * b1[:, None]).astype(np.float32)


def resize_nearest_half(img):
    h, w = img.shape
    dh, dw = h // 2, w // 2
    ys = np.minimum(np.floor(np.arange(dh) * (h / float(dh))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(dw) * (w / float(dw))).astype(np.int64), w - 1)
    return img[np.ix_(ys, xs)].copy()


def build_pyramids(gray_u8, n_layers=3, sigma=1.6):
    base = gray_u8.astype(np.float32)
    sig_diff = float(np.sqrt(max(F(F(F(sigma) * F(sigma)) - F(SIFT_INIT_SIGMA * SIFT_INIT_SIGMA * 4)), F(0.01))))   # sqrtf, float
    base = gaussian_blur(resize_linear_x2(base), sig_diff)
    n_oct = cv_round(np.log(float(min(base.shape))) / np.log(2.0) - 2) + 1          # - firstOctave (= -1)
    sig = [sigma]
    k = 2.0 ** (1.0 / n_layers)
    for i in range(1, n_layers + 3):
        sp = (k ** (i - 1)) * sigma
        st = sp * k
        sig.append(float(np.sqrt(st * st - sp * sp)))
    gp = []
    for o in range(n_oct):
        for i in range(n_layers + 3):
            if o == 0 and i == 0:
                gp.append(base)
            elif i == 0:
                gp.append(resize_nearest_half(gp[(o - 1) * (n_layers + 3) + n_layers]))
            else:
                gp.append(gaussian_blur(gp[-1], sig[i]))
    dog = []
    for o in range(n_oct):
        for i in range(n_layers + 2):
            dog.append((gp[o * (n_layers + 3) + i + 1] - gp[o * (n_layers + 3) + i]).astype(np.float32))
    return gp, dog, n_oct


def fast_atan2_deg(y, x):
    """cv::fastAtan2 (degrees, 0..360): the polynomial of OpenCV's mathfuncs_core"""
    p1, p3, p5, p7 = (F(0.9997878412794807 * 57.29577951308232), F(-0.3258083974640975 * 57.29577951308232),
                      F(0.1555786518463281 * 57.29577951308232), F(-0.04432655554792128 * 57.29577951308232))
    ax, ay = F(abs(x)), F(abs(y))
    eps = F(2.220446049250313e-16)                     # (float)DBL_EPSILON
    if ax >= ay:
        c = F(ay / F(ax + eps))
        c2 = F(c * c)
        a = F(F(F(F(F(F(p7 * c2) + p5) * c2) + p3) * c2 + p1) * c)
    else:
        c = F(ax / F(ay + eps))
        c2 = F(c * c)
        a = F(F(90.0) - F(F(F(F(F(F(p7 * c2) + p5) * c2) + p3) * c2 + p1) * c))
    if x < 0:
        a = F(F(180.0) - a)
    if y < 0:
        a = F(F(360.0) - a)
    return a


def _solve3(H, b):
    """Matx33f::solve(b, DECOMP_LU): Cramer's rule in float (Matx_FastSolveOp<float, 3, 1>)"""
    a = H
    d = F(a[0, 0] * F(a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) - a[0, 1] * F(a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
          + a[0, 2] * F(a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]))
    if d == 0:
        return np.zeros(3, np.float32)
    d = F(F(1.0) / d)
    x0 = F(d * F(b[0] * F(a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) - a[0, 1] * F(b[1] * a[2, 2] - a[1, 2] * b[2])
                 + a[0, 2] * F(b[1] * a[2, 1] - a[1, 1] * b[2])))
    x1 = F(d * F(a[0, 0] * F(b[1] * a[2, 2] - a[1, 2] * b[2]) - b[0] * F(a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
                 + a[0, 2] * F(a[1, 0] * b[2] - b[1] * a[2, 0])))
    x2 = F(d * F(a[0, 0] * F(a[1, 1] * b[2] - b[1] * a[2, 1]) - a[0, 1] * F(a[1, 0] * b[2] - b[1] * a[2, 0])
                 + b[0] * F(a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0])))
    return np.array([x0, x1, x2], np.float32)


def adjust_local_extrema(dog, octv, layer, r, c, n_layers, contrast_thr, edge_thr, sigma):
    img_scale = F(1.0 / 255.0)
    deriv_scale, second_scale, cross_scale = F(img_scale * F(0.5)), img_scale, F(img_scale * F(0.25))
    xi = xr = xc = F(0)
    i = 0
    while i < MAX_INTERP_STEPS:
        idx = octv * (n_layers + 2) + layer
        img, prv, nxt = dog[idx], dog[idx - 1], dog[idx + 1]
        dD = np.array([F(F(img[r, c + 1] - img[r, c - 1]) * deriv_scale), F(F(img[r + 1, c] - img[r - 1, c]) * deriv_scale),
                       F(F(nxt[r, c] - prv[r, c]) * deriv_scale)], np.float32)
        v2 = F(img[r, c] * F(2))
        dxx = F(F(F(img[r, c + 1] + img[r, c - 1]) - v2) * second_scale)
        dyy = F(F(F(img[r + 1, c] + img[r - 1, c]) - v2) * second_scale)
        dss = F(F(F(nxt[r, c] + prv[r, c]) - v2) * second_scale)
        dxy = F(F(F(F(img[r + 1, c + 1] - img[r + 1, c - 1]) - img[r - 1, c + 1]) + img[r - 1, c - 1]) * cross_scale)
        dxs = F(F(F(F(nxt[r, c + 1] - nxt[r, c - 1]) - prv[r, c + 1]) + prv[r, c - 1]) * cross_scale)
        dys = F(F(F(F(nxt[r + 1, c] - nxt[r - 1, c]) - prv[r + 1, c]) + prv[r - 1, c]) * cross_scale)
        X = _solve3(np.array([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]], np.float32), dD)
        xi, xr, xc = F(-X[2]), F(-X[1]), F(-X[0])
        if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
            break
        big = float(2147483647 // 3)
        if abs(xi) > big or abs(xr) > big or abs(xc) > big:
            return None
        c += cv_round(xc)
        r += cv_round(xr)
        layer += cv_round(xi)
        if (layer < 1 or layer > n_layers or c < IMG_BORDER or c >= img.shape[1] - IMG_BORDER or r < IMG_BORDER
                or r >= img.shape[0] - IMG_BORDER):
            return None
        i += 1
    if i >= MAX_INTERP_STEPS:
        return None
    idx = octv * (n_layers + 2) + layer
    img, prv, nxt = dog[idx], dog[idx - 1], dog[idx + 1]
    dD = np.array([F(F(img[r, c + 1] - img[r, c - 1]) * deriv_scale), F(F(img[r + 1, c] - img[r - 1, c]) * deriv_scale),
                   F(F(nxt[r, c] - prv[r, c]) * deriv_scale)], np.float32)
    t = F(F(F(dD[0] * xc) + F(dD[1] * xr)) + F(dD[2] * xi))
    contr = F(F(img[r, c] * img_scale) + F(t * F(0.5)))
    if F(abs(contr) * F(n_layers)) < F(contrast_thr):
        return None
    v2 = F(img[r, c] * F(2))
    dxx = F(F(F(img[r, c + 1] + img[r, c - 1]) - v2) * second_scale)
    dyy = F(F(F(img[r + 1, c] + img[r - 1, c]) - v2) * second_scale)
    dxy = F(F(F(F(img[r + 1, c + 1] - img[r + 1, c - 1]) - img[r - 1, c + 1]) + img[r - 1, c - 1]) * cross_scale)
    tr = F(dxx + dyy)
    det = F(F(dxx * dyy) - F(dxy * dxy))
    e = F(edge_thr)
    if det <= 0 or F(F(tr * tr) * e) >= F(F(F(e + F(1)) * F(e + F(1))) * det):
        return None
    sc = F(1 << octv)
    kp = dict(x=F(F(F(c) + xc) * sc), y=F(F(F(r) + xr) * sc), octave=octv + (layer << 8) + (cv_round((float(xi) + 0.5) * 255) << 16),
              size=F(F(F(F(sigma) * F(np.power(2.0, np.float64(F(F(F(layer) + xi) / F(n_layers)))))) * sc) * F(2)), response=F(abs(contr)))
    return kp, r, c, layer


def orientation_hist(img, px, py, radius, sigma):
    n = ORI_BINS
    expf_scale = F(F(-1.0) / F(F(2.0) * F(sigma * sigma)))
    tmp = np.zeros(n, np.float32)
    for i in range(-radius, radius + 1):
        y = py + i
        if y <= 0 or y >= img.shape[0] - 1:
            continue
        for j in range(-radius, radius + 1):
            x = px + j
            if x <= 0 or x >= img.shape[1] - 1:
                continue
            dx = F(img[y, x + 1] - img[y, x - 1])
            dy = F(img[y - 1, x] - img[y + 1, x])
            w = F(np.exp(np.float64(F(F(i * i + j * j) * expf_scale))))     # (float)exp((double)x): see the header
            ori = fast_atan2_deg(dy, dx)
            mag = F(np.sqrt(F(F(dx * dx) + F(dy * dy))))
            b = cv_round(F(F(n / 360.0) * ori))
            if b >= n:
                b -= n
            if b < 0:
                b += n
            tmp[b] = F(tmp[b] + F(w * mag))
    t = np.concatenate([tmp[-2:], tmp, tmp[:2]])
    hist = np.zeros(n, np.float32)
    for i in range(n):
        hist[i] = F(F(F(F(t[i] + t[i + 4]) * F(1.0 / 16.0)) + F(F(t[i + 1] + t[i + 3]) * F(4.0 / 16.0))) + F(t[i + 2] * F(6.0 / 16.0)))
    return hist, F(hist.max())


def find_keypoints(gp, dog, n_oct, n_layers=3, contrast_thr=0.04, edge_thr=10.0, sigma=1.6):
    thr = int(np.floor(0.5 * contrast_thr / n_layers * 255))
    kps = []
    for o in range(n_oct):
        for i in range(1, n_layers + 1):
            idx = o * (n_layers + 2) + i
            img, prv, nxt = dog[idx], dog[idx - 1], dog[idx + 1]
            h, w = img.shape
            if h <= 2 * IMG_BORDER or w <= 2 * IMG_BORDER:
                continue
            cube = np.stack([m[IMG_BORDER - 1 + dy:h - IMG_BORDER - 1 + dy + 1 - 0, IMG_BORDER - 1 + dx:w - IMG_BORDER - 1 + dx + 1]
                             for m in (prv, img, nxt) for dy in range(3) for dx in range(3)], 0)
            cube = cube[:, :h - 2 * IMG_BORDER, :w - 2 * IMG_BORDER]
            val = img[IMG_BORDER:h - IMG_BORDER, IMG_BORDER:w - IMG_BORDER]
            ismax = (val > 0) & (val >= cube.max(0))
            ismin = (val < 0) & (val <= cube.min(0))
            cand = (np.abs(val) > thr) & (ismax | ismin)
            for r0, c0 in zip(*np.nonzero(cand)):                    # row-major: OpenCV's scan order
                res = adjust_local_extrema(dog, o, i, int(r0) + IMG_BORDER, int(c0) + IMG_BORDER, n_layers, contrast_thr, edge_thr, sigma)
                if res is None:
                    continue
                kp, r1, c1, layer = res
                scl_octv = F(F(kp["size"] * F(0.5)) / F(1 << o))
                hist, omax = orientation_hist(gp[o * (n_layers + 3) + layer], c1, r1, cv_round(F(ORI_RADIUS * scl_octv)),
                                              F(ORI_SIG_FCTR * scl_octv))
                mag_thr = F(omax * ORI_PEAK_RATIO)
                n = ORI_BINS
                for j in range(n):
                    l, r2 = (j - 1) % n, (j + 1) % n
                    if hist[j] > hist[l] and hist[j] > hist[r2] and hist[j] >= mag_thr:
                        b = F(F(j) + F(F(F(0.5) * F(hist[l] - hist[r2])) / F(F(hist[l] - F(F(2) * hist[j])) + hist[r2])))
                        b = F(n + b) if b < 0 else (F(b - n) if b >= n else b)
                        ang = F(F(360.0) - F(F(360.0 / n) * b))
                        if abs(F(ang - F(360.0))) < FLT_EPSILON:
                            ang = F(0)
                        kps.append(dict(kp, angle=ang))
    return kps


def remove_duplicated_sorted(kps):
    key = lambda k: (float(k["x"]), float(k["y"]), -float(k["size"]), float(k["angle"]), -float(k["response"]), -k["octave"])
    kps = sorted(kps, key=key)
    out = []
    for k in kps:
        if not out or (k["x"], k["y"], k["size"], k["angle"]) != (out[-1]["x"], out[-1]["y"], out[-1]["size"], out[-1]["angle"]):
            out.append(k)
    return out


def unpack_octave(k):
    octave = k["octave"] & 255
    layer = (k["octave"] >> 8) & 255
    octave = octave if octave < 128 else (-128 | octave)
    scale = F(1.0 / (1 << octave)) if octave >= 0 else F(1 << -octave)
    return octave, layer, scale


def descriptor(img, ptx, pty, ori, scl):
    d, n = DESCR_WIDTH, DESCR_BINS
    px, py = cv_round(ptx), cv_round(pty)
    cos_t = F(np.cos(np.float64(F(ori * F(np.pi / 180.0)))))
    sin_t = F(np.sin(np.float64(F(ori * F(np.pi / 180.0)))))
    bins_per_rad = F(n / 360.0)
    exp_scale = F(F(-1.0) / F(d * d * 0.5))
    hist_width = F(DESCR_SCL_FCTR * scl)
    radius = cv_round(F(F(F(hist_width * F(1.4142135623730951)) * F(d + 1)) * F(0.5)))
    radius = min(radius, int(np.sqrt(float(img.shape[1]) ** 2 + float(img.shape[0]) ** 2)))
    cos_t = F(cos_t / hist_width)
    sin_t = F(sin_t / hist_width)
    hist = np.zeros((d + 2, d + 2, n + 2), np.float32)
    rows, cols = img.shape
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            c_rot = F(F(j * cos_t) - F(i * sin_t))
            r_rot = F(F(j * sin_t) + F(i * cos_t))
            rbin = F(F(r_rot + F(d // 2)) - F(0.5))
            cbin = F(F(c_rot + F(d // 2)) - F(0.5))
            r, c = py + i, px + j
            if -1 < rbin < d and -1 < cbin < d and 0 < r < rows - 1 and 0 < c < cols - 1:
                dx = F(img[r, c + 1] - img[r, c - 1])
                dy = F(img[r - 1, c] - img[r + 1, c])
                w = F(np.exp(np.float64(F(F(F(c_rot * c_rot) + F(r_rot * r_rot)) * exp_scale))))
                ang = fast_atan2_deg(dy, dx)
                mag = F(np.sqrt(F(F(dx * dx) + F(dy * dy))))
                obin = F(F(ang - ori) * bins_per_rad)
                mg = F(mag * w)
                r0, c0, o0 = int(np.floor(rbin)), int(np.floor(cbin)), int(np.floor(obin))
                rb, cb, ob = F(rbin - F(r0)), F(cbin - F(c0)), F(obin - F(o0))
                if o0 < 0:
                    o0 += n
                if o0 >= n:
                    o0 -= n
                v_r1 = F(mg * rb)
                v_r0 = F(mg - v_r1)
                v_rc11 = F(v_r1 * cb)
                v_rc10 = F(v_r1 - v_rc11)
                v_rc01 = F(v_r0 * cb)
                v_rc00 = F(v_r0 - v_rc01)
                v111, v011, v101, v001 = F(v_rc11 * ob), F(v_rc01 * ob), F(v_rc10 * ob), F(v_rc00 * ob)
                v110, v010, v100, v000 = F(v_rc11 - v111), F(v_rc01 - v011), F(v_rc10 - v101), F(v_rc00 - v001)
                hist[r0 + 1, c0 + 1, o0] += v000
                hist[r0 + 1, c0 + 1, o0 + 1] += v001
                hist[r0 + 1, c0 + 2, o0] += v010
                hist[r0 + 1, c0 + 2, o0 + 1] += v011
                hist[r0 + 2, c0 + 1, o0] += v100
                hist[r0 + 2, c0 + 1, o0 + 1] += v101
                hist[r0 + 2, c0 + 2, o0] += v110
                hist[r0 + 2, c0 + 2, o0 + 1] += v111
    dst = np.zeros(d * d * n, np.float32)
    for i in range(d):
        for j in range(d):
            hist[i + 1, j + 1, 0] = F(hist[i + 1, j + 1, 0] + hist[i + 1, j + 1, n])
            hist[i + 1, j + 1, 1] = F(hist[i + 1, j + 1, 1] + hist[i + 1, j + 1, n + 1])
            dst[(i * d + j) * n:(i * d + j + 1) * n] = hist[i + 1, j + 1, :n]
    nrm2 = F(0)
    for v in dst:
        nrm2 = F(nrm2 + F(v * v))
    thr = F(F(np.sqrt(nrm2)) * DESCR_MAG_THR)
    nrm2 = F(0)
    for k in range(len(dst)):
        v = min(dst[k], thr)
        dst[k] = v
        nrm2 = F(nrm2 + F(v * v))
    nrm2 = F(INT_DESCR_FCTR / max(F(np.sqrt(nrm2)), FLT_EPSILON))
    return np.clip(np.rint(dst * nrm2), 0, 255).astype(np.float32)       # saturate_cast<uchar>(val * nrm2)


def detect_and_compute(gray_u8, n_layers=3, contrast_thr=0.04, edge_thr=10.0, sigma=1.6):
    """Returns (keypoints [n x 6 float32: x, y, size, angle, response, octave-as-int-bits], descriptors n x 128 float32)."""
    gray_u8 = np.ascontiguousarray(gray_u8, np.uint8)
    gp, dog, n_oct = build_pyramids(gray_u8, n_layers, sigma)
    kps = remove_duplicated_sorted(find_keypoints(gp, dog, n_oct, n_layers, contrast_thr, edge_thr, sigma))
    first_octave = -1
    out, desc = [], []
    for k in kps:
        # descriptors are computed with the unscaled keypoint against the pyramid; OpenCV rescales first and undoes
        # it inside calcDescriptors -- the same numbers
        k = dict(k)
        k["octave"] = (k["octave"] & ~255) | ((k["octave"] + first_octave) & 255)
        k["x"], k["y"], k["size"] = F(k["x"] * F(0.5)), F(k["y"] * F(0.5)), F(k["size"] * F(0.5))
        octave, layer, scale = unpack_octave(k)
        size = F(k["size"] * scale)
        img = gp[(octave - first_octave) * (n_layers + 3) + layer]
        angle = F(F(360.0) - k["angle"])
        if abs(F(angle - F(360.0))) < FLT_EPSILON:
            angle = F(0)
        desc.append(descriptor(img, F(k["x"] * scale), F(k["y"] * scale), angle, F(size * F(0.5))))
        out.append([k["x"], k["y"], k["size"], k["angle"], k["response"], np.int32(k["octave"]).view(np.float32)])
    K = np.array(out, np.float32).reshape(-1, 6)
    D = np.array(desc, np.float32).reshape(-1, 128)
    return K, D
