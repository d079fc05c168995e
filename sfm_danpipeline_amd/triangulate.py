"""Host-side mirror of StructFromMotion::triangulateViews (reference src/Sfm.cpp:804-878) over
the C ABI: gather (AlignedPoints, :694-711) -> device DLT + reprojection filter -> Point3D
records with their two-view tracks (:862-873)."""
import numpy as np

from . import _lib
from ._lib import check, lib

MIN_REPROJECTION_ERROR = 6.0  # reference src/Sfm.cpp:850


def aligned_points(query_pts, train_pts, q_idx, t_idx):
    """AlignedPoints (src/Sfm.cpp:700-711): gathered 2-D points + back references."""
    q_idx = np.asarray(q_idx, np.int64)
    t_idx = np.asarray(t_idx, np.int64)
    return (np.ascontiguousarray(np.asarray(query_pts, np.float64)[q_idx]),
            np.ascontiguousarray(np.asarray(train_pts, np.float64)[t_idx]), q_idx.copy(), t_idx.copy())


def triangulate_points(P1, P2, K, dist, xy1, xy2, max_err=MIN_REPROJECTION_ERROR, ctx=None):
    ctx = ctx or _lib.default_context()
    P1 = np.ascontiguousarray(P1, np.float64).reshape(12)
    P2 = np.ascontiguousarray(P2, np.float64).reshape(12)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    dist = np.ascontiguousarray(dist, np.float64).reshape(-1)
    if dist.size != 5:
        raise ValueError("distCoef must hold 5 coefficients (include/Utilities.h:30-35)")
    xy1 = np.ascontiguousarray(xy1, np.float64).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.float64).reshape(-1, 2)
    m = xy1.shape[0]
    if xy2.shape[0] != m:
        raise ValueError("left/right point counts differ")
    X = np.empty((max(m, 1), 3), np.float64)
    err = np.empty((max(m, 1), 2), np.float32)
    keep = np.empty(max(m, 1), np.uint8)
    check(lib().sfmhip_triangulate(ctx.h, P1.ctypes.data, P2.ctypes.data, K.ctypes.data, dist.ctypes.data,
                                   xy1.ctypes.data, xy2.ctypes.data, m, max_err, X.ctypes.data, err.ctypes.data,
                                   keep.ctypes.data), "sfmhip_triangulate")
    return X[:m], err[:m], keep[:m]


def triangulate_views(query_pts, train_pts, P1, P2, match_q, match_t, K, dist, image_pair, ctx=None):
    """triangulateViews: returns the new point cloud as a list of dicts
    {pt:(x,y,z), idxImage:{view:featIdx}, pt2D:{view:(u,v)}} in match order (the reference's
    Point3D, include/Utilities.h:37-43).  Always 'succeeds' like the reference (:877)."""
    aq, at, lref, rref = aligned_points(query_pts, train_pts, match_q, match_t)
    X, _, keep = triangulate_points(P1, P2, K, dist, aq, at, ctx=ctx)
    cloud = []
    a, b = int(image_pair[0]), int(image_pair[1])
    for i in np.nonzero(keep)[0]:
        cloud.append(dict(pt=(float(X[i, 0]), float(X[i, 1]), float(X[i, 2])),
                          idxImage={a: int(lref[i]), b: int(rref[i])},
                          pt2D={a: (float(aq[i, 0]), float(aq[i, 1])), b: (float(at[i, 0]), float(at[i, 1]))}))
    return cloud
