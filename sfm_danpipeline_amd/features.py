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


class DeviceDescriptors:
    """n x 128 float32 descriptor rows that sfmhip_sift_batch left in HBM: .ptr for ImageSet.adopt_device, .download()
    for a host copy; freed with the object."""

    def __init__(self, ptr, n, ctx):
        self.ptr, self.n, self.ctx = ptr, int(n), ctx

    def download(self):
        out = np.empty((self.n, 128), np.float32)
        if self.n:
            check(lib().sfmhip_device_download(self.ctx.h, out.ctypes.data, C.c_void_p(self.ptr), out.nbytes), "sfmhip_device_download")
        return out

    def close(self):
        if self.ptr:
            lib().sfmhip_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sift_batch(grays, n_octave_layers=3, contrast_threshold=0.04, edge_threshold=10.0, sigma=1.6, ctx=None):
    """extractFeature's loop as one call (sfmhip_sift_batch): a list of gray images, several in flight.  Returns
    [(keypoints n x 6 float32, DeviceDescriptors)] -- the descriptor rows stay in HBM."""
    ctx = ctx or default_context()
    imgs = [np.ascontiguousarray(g, np.uint8) for g in grays]
    n = len(imgs)
    ptrs = (C.c_void_p * max(n, 1))(*[g.ctypes.data for g in imgs])
    rows = np.array([g.shape[0] for g in imgs], np.int32)
    cols = np.array([g.shape[1] for g in imgs], np.int32)
    kp = (C.c_void_p * max(n, 1))()
    dd = (C.c_void_p * max(n, 1))()
    nk = np.zeros(max(n, 1), np.int32)
    check(lib().sfmhip_sift_batch(ctx.h, n, ptrs, rows.ctypes.data, cols.ctypes.data, int(n_octave_layers), float(contrast_threshold),
                                  float(edge_threshold), float(sigma), kp, dd, nk.ctypes.data), "sfmhip_sift_batch")
    out = []
    for i in range(n):
        k = np.frombuffer((C.c_float * (6 * max(int(nk[i]), 1))).from_address(kp[i]), np.float32, 6 * int(nk[i])).reshape(-1, 6).copy()
        lib().sfmhip_host_free(C.c_void_p(kp[i]))
        out.append((k, DeviceDescriptors(dd[i], nk[i], ctx)))
    return out


def keypoints_to_points(kps):
    """keypointstoPoints (src/Sfm.cpp:1476-1482): pt as Point2d"""
    return np.asarray(kps, np.float32)[:, :2].astype(np.float64)
