"""Ad-hoc GPU measurement (run through gpurun): sfmhip_sift_detect_and_compute on a 640 x 480 textured image (the size
of the reference's temple frames), wall time of the whole call; the numpy restatement on a 160 x 120 crop for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import sfm_oracle_sift as S
from sfm_danpipeline_amd import features, _lib

ctx = _lib.default_context()
rng = np.random.default_rng(0)
h, w = 480, 640
yy, xx = np.mgrid[0:h, 0:w]
img = np.zeros((h, w))
for _ in range(900):
    cx, cy, s, a = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1.2, 7), rng.uniform(30, 160)
    y0, y1, x0, x1 = int(max(cy - 5 * s, 0)), int(min(cy + 5 * s, h)), int(max(cx - 5 * s, 0)), int(min(cx + 5 * s, w))
    img[y0:y1, x0:x1] += a * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * s * s))
img = np.clip(img + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)
features.sift_detect_and_compute(img[:64, :64], ctx=ctx)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); K, D = features.sift_detect_and_compute(img, ctx=ctx); ts.append(time.perf_counter() - t0)
print(f"640x480: {len(K)} keypoints, {min(ts)*1e3:.1f} ms per image (best of 5, whole call: upload, pyramid, keypoints, descriptors, download)", flush=True)
t0 = time.perf_counter(); Ko, Do = S.detect_and_compute(img[:120, :160]); dt = time.perf_counter() - t0
Kc, Dc = features.sift_detect_and_compute(img[:120, :160], ctx=ctx)
print(f"160x120 crop: numpy restatement {dt:.1f} s for {len(Ko)} keypoints; device finds {len(Kc)}, same octave bits: "
      f"{len(Kc) == len(Ko) and np.array_equal(Kc[:, 5].view(np.int32), Ko[:, 5].view(np.int32))}", flush=True)
