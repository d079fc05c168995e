"""Host-side mirror of the incremental-loop glue next to the hot path over
the C ABI: the 2D-3D association of StructFromMotion::find2D3DMatches (reference
src/Sfm.cpp:1011-1095) and StructFromMotion::mergeNewPoints (src/Sfm.cpp:1212-1244).  Clouds are
lists of dicts {pt, idxImage, pt2D} as produced by triangulate.triangulate_views (the reference's
Point3D, include/Utilities.h:37-43)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE = 0.01  # float literal, reference src/Sfm.cpp:1216


def tracks_csr(cloud):
    """Point3D::idxImage of every cloud point as CSR (ascending view = std::map order)."""
    ptr = np.zeros(len(cloud) + 1, np.int32)
    views, feats = [], []
    for i, p in enumerate(cloud):
        for v in sorted(p["idxImage"]):
            views.append(v)
            feats.append(p["idxImage"][v])
        ptr[i + 1] = len(views)
    return ptr, np.asarray(views, np.int32), np.asarray(feats, np.int32)


def find_2d3d(trk_ptr, trk_view, trk_feat, done_view, new_view, match_q, match_t, ctx=None):
    """Returns (cloud indices, feature indices in the new view), in cloud order."""
    ctx = ctx or _lib.default_context()
    trk_ptr = np.ascontiguousarray(trk_ptr, np.int32)
    trk_view = np.ascontiguousarray(trk_view, np.int32)
    trk_feat = np.ascontiguousarray(trk_feat, np.int32)
    mq = np.ascontiguousarray(match_q, np.int32)
    mt = np.ascontiguousarray(match_t, np.int32)
    n_cloud = len(trk_ptr) - 1
    oc = np.zeros(max(n_cloud, 1), np.int32)
    of = np.zeros(max(n_cloud, 1), np.int32)
    n = C.c_int32(0)
    check(lib().sfmhip_find_2d3d(ctx.h, trk_ptr.ctypes.data, trk_view.ctypes.data, trk_feat.ctypes.data, n_cloud,
                                 int(done_view), int(new_view), mq.ctypes.data, mt.ctypes.data, len(mq),
                                 oc.ctypes.data, of.ctypes.data, C.byref(n)), "sfmhip_find_2d3d")
    return oc[:n.value].copy(), of[:n.value].copy()


def find_2d3d_matches(cloud, new_view, done_view, best_q, best_t, new_view_pts2d, ctx=None):
    """The association half of find2D3DMatches (:1047-1090) given the best match list between
    done_view and new_view (queryIdx/trainIdx of getMatching(min(view), max(view))).
    Returns (points3D, points2D) like the reference's output vectors."""
    ptr, views, feats = tracks_csr(cloud)
    ci, nf = find_2d3d(ptr, views, feats, done_view, new_view, best_q, best_t, ctx=ctx)
    pts3 = np.array([cloud[i]["pt"] for i in ci], np.float64).reshape(-1, 3)
    pts2 = np.asarray(new_view_pts2d, np.float64).reshape(-1, 2)[nf]
    return pts3, pts2


def merge_accept(cloud_xyz, new_xyz, min_dist=MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE, ctx=None):
    ctx = ctx or _lib.default_context()
    cloud = np.ascontiguousarray(cloud_xyz, np.float64).reshape(-1, 3)
    new = np.ascontiguousarray(new_xyz, np.float64).reshape(-1, 3)
    acc = np.zeros(max(len(new), 1), np.uint8)
    n = C.c_int32(0)
    check(lib().sfmhip_merge_new_points(ctx.h, cloud.ctypes.data, len(cloud), new.ctypes.data, len(new),
                                        C.c_float(min_dist), acc.ctypes.data, C.byref(n)), "sfmhip_merge_new_points")
    return acc[:len(new)].astype(bool), n.value


def merge_new_points(cloud, new_cloud, ctx=None):
    """mergeNewPoints: appends the accepted new points to `cloud` in place (no track merging,
    like the reference: foundAnyMatchingExistingViews is constant false, :1225).  Returns the
    number of points added."""
    if not new_cloud:
        return 0
    acc, n = merge_accept([p["pt"] for p in cloud], [p["pt"] for p in new_cloud], ctx=ctx)
    for p, a in zip(new_cloud, acc):
        if a:
            cloud.append(p)
    return n
