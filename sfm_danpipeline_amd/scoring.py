"""The scoring half of StructFromMotion::findBestPair (reference src/Sfm.cpp:533-569) over sfmhip_score_essential:
per pair with >= 120 matches the pose-inlier ratio of cv::findEssentialMat(RANSAC, 0.999, 1.0), collected in the
reference's std::map<float, pair> (ascending keys; equal keys keep the last pair inserted)."""
import ctypes as C

import numpy as np

from ._lib import check, default_context, lib


def score_essential(pairs_points, K, prob=0.999, threshold=1.0, want_mask=False, ctx=None):
    """pairs_points: list of (left n x 2, right n x 2) pixel coordinates.  Returns (inliers[int32], masks or None,
    iterations[int32])."""
    ctx = ctx or default_context()
    n = len(pairs_points)
    counts = np.array([len(a) for a, _ in pairs_points], np.int32)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    cat = lambda k: (np.ascontiguousarray(np.concatenate([np.asarray(p[k], np.float64).reshape(-1, 2) for p in pairs_points]))
                     if n and offsets[-1] else np.zeros((0, 2)))
    left, right = cat(0), cat(1)
    inl = np.zeros(max(n, 1), np.int32)
    its = np.zeros(max(n, 1), np.int32)
    mask = np.zeros(max(int(offsets[-1]), 1), np.uint8) if want_mask else None
    K = np.asarray(K, np.float64)
    check(lib().sfmhip_score_essential(ctx.h, n, offsets.ctypes.data, left.ctypes.data, right.ctypes.data, float(K[0, 0]),
                                       float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), float(prob), float(threshold),
                                       inl.ctypes.data, mask.ctypes.data if want_mask else None, its.ctypes.data),
          "sfmhip_score_essential")
    masks = [mask[offsets[i]:offsets[i + 1]].copy() for i in range(n)] if want_mask else None
    return inl[:n], masks, its[:n]


def five_point(q1, q2, ctx=None):
    """sfmhip_score_five_point: q1, q2 (n, 5, 2) normalised points -> (models (n, 10, 3, 3), counts (n,), flags (n,))"""
    ctx = ctx or default_context()
    q1 = np.ascontiguousarray(q1, np.float64).reshape(-1, 5, 2)
    q2 = np.ascontiguousarray(q2, np.float64).reshape(-1, 5, 2)
    n = len(q1)
    models = np.zeros((max(n, 1), 10, 3, 3))
    nm = np.zeros(max(n, 1), np.int32)
    check(lib().sfmhip_score_five_point(ctx.h, n, q1.ctypes.data, q2.ctypes.data, models.ctypes.data, nm.ctypes.data),
          "sfmhip_score_five_point")
    return models[:n], nm[:n] & 0xff, nm[:n] >> 8


def homography_kernel(M, m, ctx=None):
    """sfmhip_score_homography_kernel: M, m (n, 4, 2) float32 -> (H (n, 3, 3), ok (n,))"""
    ctx = ctx or default_context()
    M = np.ascontiguousarray(M, np.float32).reshape(-1, 4, 2)
    m = np.ascontiguousarray(m, np.float32).reshape(-1, 4, 2)
    n = len(M)
    H = np.zeros((max(n, 1), 3, 3))
    ok = np.zeros(max(n, 1), np.int32)
    check(lib().sfmhip_score_homography_kernel(ctx.h, n, M.ctypes.data, m.ctypes.data, H.ctypes.data, ok.ctypes.data),
          "sfmhip_score_homography_kernel")
    return H[:n], ok[:n]


def last_flags(ctx=None):
    """sfmhip_score_last_flags: non-zero when a five-point sample of the last score_essential call reached a corner of
    cv::solvePoly whose library behaviour is not reproduced (include/sfmhip.h)."""
    return int(lib().sfmhip_score_last_flags((ctx or default_context()).h))


def score_homography(pairs_points, thresholds=None, confidence=0.995, max_iters=2000, want_mask=False, ctx=None):
    """findHomographyInliers (src/Sfm.cpp:667-689) for a batch: the RANSAC inlier count of cv::findHomography(left,
    right, RANSAC, threshold).  thresholds None: the reference's 0.004 * (largest coordinate of the pair's left points)."""
    ctx = ctx or default_context()
    n = len(pairs_points)
    counts = np.array([len(a) for a, _ in pairs_points], np.int32)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    cat = lambda k: (np.ascontiguousarray(np.concatenate([np.asarray(p[k], np.float64).reshape(-1, 2) for p in pairs_points]))
                     if n and offsets[-1] else np.zeros((0, 2)))
    left, right = cat(0), cat(1)
    if thresholds is None:
        thresholds = [0.004 * float(np.max(a)) if len(a) else 0.0 for a, _ in pairs_points]   # cv::minMaxIdx over x and y
    thr = np.ascontiguousarray(np.asarray(thresholds, np.float64).reshape(-1)) if n else np.zeros(1)
    inl = np.zeros(max(n, 1), np.int32)
    its = np.zeros(max(n, 1), np.int32)
    mask = np.zeros(max(int(offsets[-1]), 1), np.uint8) if want_mask else None
    check(lib().sfmhip_score_homography(ctx.h, n, offsets.ctypes.data, left.ctypes.data, right.ctypes.data, thr.ctypes.data,
                                        float(confidence), int(max_iters), inl.ctypes.data,
                                        mask.ctypes.data if want_mask else None, its.ctypes.data), "sfmhip_score_homography")
    masks = [mask[offsets[i]:offsets[i + 1]].copy() for i in range(n)] if want_mask else None
    return inl[:n], masks, its[:n]


def find_best_pair(pair_ids, pairs_points, K, min_matches=120, ctx=None):
    """src/Sfm.cpp:511-569 after the matching: pair_ids in the loop's order; pairs with fewer than `min_matches`
    matches are skipped (:533).  Returns [(float32 ratio, (q, t))] ascending -- the iteration order of the map."""
    keep = [i for i, (a, _) in enumerate(pairs_points) if len(a) >= min_matches]
    inl, _, _ = score_essential([pairs_points[i] for i in keep], K, ctx=ctx)
    m = {}
    for k, i in enumerate(keep):
        ratio = np.float32(np.float32(inl[k]) / np.float32(len(pairs_points[i][0])))   # (float)pruned / (float)matches
        m[ratio] = tuple(pair_ids[i])                                                      # equal keys: overwritten (:569)
    return sorted(m.items(), key=lambda kv: kv[0])
