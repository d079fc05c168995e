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
    check(lib().sfmhip_sift_detect_and_compute(*args, 0, None, None, C.addressof(n)), "sfmhip_sift_detect_and_compute")
    k = np.zeros((max(n.value, 1), 6), np.float32)
    d = np.zeros((max(n.value, 1), 128), np.float32)
    check(lib().sfmhip_sift_detect_and_compute(*args, n.value, k.ctypes.data, d.ctypes.data, C.addressof(n)),
          "sfmhip_sift_detect_and_compute")
    return k[:n.value], d[:n.value]


def keypoints_to_points(kps):
    """keypointstoPoints (src/Sfm.cpp:1476-1482): pt as Point2d"""
    return np.asarray(kps, np.float32)[:, :2].astype(np.float64)
