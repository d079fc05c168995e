"""Host-side mirror of the reference's matching interface over the C ABI.

`get_matching` has the argument meaning of StructFromMotion::getMatching (reference
src/Sfm.cpp:590-608): two descriptor matrices in, the ratio-filtered `knn[i][0]` matches out in
ascending queryIdx.  `ImageSet` / `MatchPlan` are the batched form of the findBestPair all-pairs
loop (reference src/Sfm.cpp:511-515) with descriptors resident in HBM.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import F32, HAMMING, L2, U8, check, lib

NN_MATCH_RATIO = 0.8  # reference include/Sfm.h:60


def _dtype_code(a):
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.uint8:
        return U8
    raise TypeError("descriptors must be float32 (SIFT, CV_32F) or uint8 (ORB/AKAZE, CV_8U) rows")


def get_matching(query_desc, train_desc, ratio=NN_MATCH_RATIO, norm=L2, ctx=None):
    """One pair, host arrays.  Returns (queryIdx, trainIdx, distance) numpy arrays; imgIdx is 0 for
    every DMatch as in the reference."""
    ctx = ctx or _lib.default_context()
    q = np.ascontiguousarray(query_desc)
    t = np.ascontiguousarray(train_desc)
    if q.dtype != t.dtype:
        raise TypeError("query/train descriptor types differ")
    dt = _dtype_code(q)
    dim = q.shape[1] if q.ndim == 2 and q.shape[0] else t.shape[1]
    nq, nt = q.shape[0], t.shape[0]
    oq = np.empty(max(nq, 1), np.int32)
    ot = np.empty(max(nq, 1), np.int32)
    od = np.empty(max(nq, 1), np.float32)
    on = C.c_int32(0)
    check(lib().sfmhip_match_knn2(ctx.h, q.ctypes.data, nq, t.ctypes.data, nt, dim, dt, norm, ratio,
                                  oq.ctypes.data, ot.ctypes.data, od.ctypes.data, C.addressof(on)),
          "sfmhip_match_knn2")
    n = on.value
    return oq[:n].copy(), ot[:n].copy(), od[:n].copy()


class ImageSet:
    """imagesDescriptors (reference include/Sfm.h:29) resident in HBM."""

    def __init__(self, descriptors=None, norm=L2, ctx=None, n_rows=None, dim=None, dtype=None):
        self.ctx = ctx or _lib.default_context()
        self.h = C.c_void_p()
        if descriptors is not None:
            descriptors = [np.ascontiguousarray(d) for d in descriptors]
            n_rows = [d.shape[0] for d in descriptors]
            dim = descriptors[0].shape[1]
            dtype = _dtype_code(descriptors[0])
            for i, d in enumerate(descriptors):   # the upload copies n_rows * dim * elemsize bytes of EVERY matrix
                if d.ndim != 2 or d.shape[1] != dim or d.dtype != descriptors[0].dtype:
                    raise ValueError(f"descriptor matrix {i}: shape {d.shape} / {d.dtype}, expected (*, {dim}) "
                                     f"{descriptors[0].dtype} like matrix 0")
        self.n_rows = np.ascontiguousarray(n_rows, np.int32)
        self.n_images = len(self.n_rows)
        self.dim, self.dtype, self.norm = int(dim), int(dtype), int(norm)
        check(lib().sfmhip_imageset_create(self.ctx.h, self.n_images, self.n_rows.ctypes.data, self.dim, self.dtype,
                                           self.norm, C.byref(self.h)), "sfmhip_imageset_create")
        self._keep = []
        if descriptors is not None:
            for i, d in enumerate(descriptors):
                check(lib().sfmhip_imageset_upload(self.h, i, d.ctypes.data), "sfmhip_imageset_upload")

    def adopt_device(self, image, device_ptr, keepalive=None):
        """Use descriptor rows that already live in HBM (e.g. a torch tensor's data_ptr())."""
        check(lib().sfmhip_imageset_adopt_device(self.h, image, C.c_void_p(device_ptr)), "sfmhip_imageset_adopt_device")
        if keepalive is not None:
            self._keep.append(keepalive)

    def prepare_async(self):
        check(lib().sfmhip_imageset_prepare_async(self.h), "sfmhip_imageset_prepare_async")

    def close(self):
        if self.h:
            lib().sfmhip_imageset_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchPlan:
    """A list of (queryImage, trainImage) pairs matched by one batched launch."""

    def __init__(self, image_set, pairs):
        self.set = image_set
        self.pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self.n_pairs = self.pairs.shape[0]
        self.h = C.c_void_p()
        check(lib().sfmhip_matchplan_create(image_set.h, self.pairs.ctypes.data, self.n_pairs, C.byref(self.h)),
              "sfmhip_matchplan_create")

    def set_pairs(self, pairs):
        """Re-target the plan (at most as many pairs as it was created with); buffers are reused."""
        self.pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self.n_pairs = self.pairs.shape[0]
        check(lib().sfmhip_matchplan_set_pairs(self.h, self.pairs.ctypes.data, self.n_pairs), "sfmhip_matchplan_set_pairs")

    def run_async(self, ratio=NN_MATCH_RATIO):
        check(lib().sfmhip_matchplan_run_async(self.h, ratio), "sfmhip_matchplan_run_async")

    def counts(self):
        cnt = np.zeros(max(self.n_pairs, 1), np.int32)
        tot = C.c_int64(0)
        check(lib().sfmhip_matchplan_fetch(self.h, cnt.ctypes.data, None, None, None, 0, C.byref(tot)),
              "sfmhip_matchplan_fetch")
        return cnt[:self.n_pairs]

    def fetch(self):
        """Returns counts (n_pairs,), and the concatenated (queryIdx, trainIdx, distance) lists."""
        cnt = self.counts()
        total = int(cnt.sum())
        oq = np.empty(max(total, 1), np.int32)
        ot = np.empty(max(total, 1), np.int32)
        od = np.empty(max(total, 1), np.float32)
        tot = C.c_int64(0)
        c2 = np.zeros(max(self.n_pairs, 1), np.int32)
        check(lib().sfmhip_matchplan_fetch(self.h, c2.ctypes.data, oq.ctypes.data, ot.ctypes.data, od.ctypes.data,
                                           total, C.byref(tot)), "sfmhip_matchplan_fetch")
        return cnt, oq[:total], ot[:total], od[:total]

    def pipeline(self, capacity=0):
        """Switch the pipelined fetch on (sfmhip_matchplan_pipeline): every run_async is followed by a pass on a second
        stream that packs the lists into one of two pinned host buffers; fetch_wait hands them out."""
        check(lib().sfmhip_matchplan_pipeline(self.h, int(capacity)), "sfmhip_matchplan_pipeline")

    def fetch_wait(self, back=0):
        """counts and (queryIdx, trainIdx, distance) of the latest run (back=0) or the one before (back=1) as numpy
        VIEWS of the pinned buffer: valid until two more runs have been enqueued (copy what must live longer)."""
        pc, pq, pt, pd = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        tot = C.c_int64(0)
        check(lib().sfmhip_matchplan_fetch_wait(self.h, back, C.byref(pc), C.byref(pq), C.byref(pt), C.byref(pd), C.byref(tot)),
              "sfmhip_matchplan_fetch_wait")
        n = int(tot.value)
        view = lambda ptr, ct, dt, k: np.frombuffer((ct * max(k, 1)).from_address(ptr.value), dt, k)
        return (view(pc, C.c_int32, np.int32, self.n_pairs), view(pq, C.c_int32, np.int32, n), view(pt, C.c_int32, np.int32, n),
                view(pd, C.c_float, np.float32, n))

    def fetch_pair(self, p):
        cnt, oq, ot, od = self.fetch()
        off = int(cnt[:p].sum())
        n = int(cnt[p])
        return oq[off:off + n], ot[off:off + n], od[off:off + n]

    def fetch_knn(self, p):
        nq = int(self.set.n_rows[self.pairs[p, 0]])
        idx = np.empty((max(nq, 1), 2), np.int32)
        dist = np.empty((max(nq, 1), 2), np.float32)
        check(lib().sfmhip_matchplan_fetch_knn(self.h, p, idx.ctypes.data, dist.ctypes.data),
              "sfmhip_matchplan_fetch_knn")
        return idx[:nq], dist[:nq]

    def last_timing(self):
        t = np.zeros(3, np.float64)
        check(lib().sfmhip_matchplan_last_timing(self.h, t.ctypes.data), "sfmhip_matchplan_last_timing")
        k = C.c_double(0)
        check(lib().sfmhip_matchplan_last_knn_kernel_time(self.h, C.byref(k)), "sfmhip_matchplan_last_knn_kernel_time")
        return dict(prepare_s=t[0], knn_s=t[1], compact_s=t[2], knn_kernel_s=k.value)

    def close(self):
        if self.h:
            lib().sfmhip_matchplan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def match_checksum(counts, oq, ot):
    """Order-sensitive checksum of match lists (bench/parity at full size)."""
    h = np.uint64(1469598103934665603)
    cs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    pid = np.repeat(np.arange(len(counts), dtype=np.uint64), counts)
    pos = (np.arange(len(oq), dtype=np.uint64) - cs[:-1].astype(np.uint64)[pid.astype(np.int64)]) if len(oq) else np.zeros(0, np.uint64)
    v = (pid * np.uint64(0x9E3779B97F4A7C15)) ^ (pos * np.uint64(0xC2B2AE3D27D4EB4F)) ^ \
        (oq.astype(np.uint64) << np.uint64(32)) ^ ot.astype(np.uint64)
    with np.errstate(over="ignore"):
        v = v * np.uint64(0xFF51AFD7ED558CCD)
        return int(np.bitwise_xor.reduce(v) ^ h) if len(v) else int(h), int(np.sum(v, dtype=np.uint64)) if len(v) else 0
