"""Generates tests/golden/temple_sift.npz: the SIFT front end's expected output on BASELINE.json's cfg1 input (the ten
640 x 480 frames of the reference's data/temple, committed as DATA under tests/golden/temple/ together with its
camera_calibration_template.xml).

Per frame: PIL decode -> BGR -> OpenCV 3.4.1's 8-bit BGR2GRAY arithmetic (no resize: imagesLOAD resizes only images
larger than 640 x 480, reference src/Sfm.cpp:155) -> oracle/sfm_oracle_sift.py (the numpy restatement of
cv::xfeatures2d::SIFT::create(0, 3, 0.04, 10, 1.6)->detectAndCompute, src/Sfm.cpp:315-320; PARITY UNPINNED, see its
header).  The restatement takes ~40 s per frame in numpy, which is why its output is a committed fixture and not
recomputed by the test.  Run from the repo root:  python tests/golden/make_temple_golden.py
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
TEMPLE = os.path.join(OUT, "temple")


def cv_gray(bgr):
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def one(name):
    from PIL import Image
    from oracle import sfm_oracle_sift as S
    bgr = np.asarray(Image.open(os.path.join(TEMPLE, name)).convert("RGB"))[:, :, ::-1]
    K, D = S.detect_and_compute(cv_gray(bgr))
    assert np.array_equal(D, np.rint(D)) and D.min() >= 0 and D.max() <= 255
    return name, K.astype(np.float32), D.astype(np.uint8)


def main():
    names = sorted(f for f in os.listdir(TEMPLE) if f.lower().endswith(".png"))
    with ProcessPoolExecutor(max_workers=5) as ex:
        res = list(ex.map(one, names))
    out = {"names": np.array(names)}
    for i, (name, K, D) in enumerate(res):
        out[f"kp{i}"] = K          # (n, 6) float32: x, y, size, angle, response, octave (bit-copied int32)
        out[f"desc{i}"] = D        # (n, 128) uint8 (the float descriptors hold these integers)
        print(name, K.shape)
    np.savez_compressed(os.path.join(OUT, "temple_sift.npz"), **out)


if __name__ == "__main__":
    main()
