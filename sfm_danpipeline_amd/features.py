"""The detector / descriptor front end of StructFromMotion::getFeature (reference src/Sfm.cpp:300-330) over
sfmhip_sift_detect_and_compute: SIFT(0, 3, 0.04, 10, 1.6) keypoints and 128-float descriptors of a gray image."""
import ctypes as C

import numpy as np

from ._lib import check, default_context, lib


def sift_detect_and_compute(gray, n_octave_layers=3, contrast_threshold=0.04, edge_threshold=10.0, sigma=1.6, ctx=None):
    """gray: rows x cols uint8.  Returns (keypoints n x 6 float32 [x, y, size, angle, response, octave bits],
    descriptors n x 128 float32) in OpenCV's keypoint order."""
    ctx = ctx or default_context()
    g = np.ascontiguousarray(gray, np.uint8)
    assert g.ndim == 2
    n = C.c_int32(0)
    args = (ctx.h, g.ctypes.data, g.shape[0], g.shape[1], int(n_octave_layers), float(contrast_threshold), float(edge_threshold),
            float(sigma))
    cap = max(1024, g.size // 48)            # one pass when the guess holds; the call reports the count if it does not
    for _ in range(2):
        k = np.zeros((cap, 6), np.float32)
        d = np.zeros((cap, 128), np.float32)
        rc = lib().sfmhip_sift_detect_and_compute(*args, cap, k.ctypes.data, d.ctypes.data, C.addressof(n))
        if rc == 0 or n.value <= cap:
            break
        cap = n.value
    check(rc, "sfmhip_sift_detect_and_compute")
    return k[:n.value].copy(), d[:n.value].copy()


def keypoints_to_points(kps):
    """keypointstoPoints (src/Sfm.cpp:1476-1482): pt as Point2d"""
    return np.asarray(kps, np.float32)[:, :2].astype(np.float64)
