"""ORACLE (test infrastructure, not product): the scoring half of StructFromMotion::findBestPair (reference
src/Sfm.cpp:533-569) -- cv::findEssentialMat(alignedLeft, alignedRight, K, RANSAC, 0.999, 1.0, mask) per image pair
with >= 120 matches, poseInliersRatio = (float)inliers / (float)matches, std::map<float, pair> (ascending keys, equal
keys overwrite) -- and findHomographyInliers (src/Sfm.cpp:667-689).

The essential-matrix RANSAC is restated in C (oracle/sfm_oracle_score.c) along OpenCV 3.4.1's own route and ONLY that
route: Jacobi-SVD null space with the library's RNG-filled singular vectors, the 10 x 20 elimination by Mat::inv(), the
tenth-degree polynomial, cv::solvePoly's Durand-Kerner iteration, the |imag| <= 1e-10 real-root test, SVD::solveZ, the
models in root order.  That file's header lists what is restated and what (at last-bit level) is not.  This module
holds the Python-side pieces around it: cv::RNG and getSubset (the sample tables the device code is given), the
std::map bookkeeping, and the homography path in numpy.

PARITY UNPINNED.  OpenCV 3.4.1 is not in /root/reference and not in this image, and the reference holds no test or
golden vector for this path.  oracle/np_check.py carries an independent action-matrix five-point solver used by
tests/test_oracle_score.py to check the restatement's models (same matrices on well-conditioned samples); it is a check
of the oracle, never an alternative route for a parity test.
"""
import numpy as np

from . import orc

CV_RNG_COEFF = 4164903690
MASK64 = (1 << 64) - 1


class CvRNG:
    def __init__(self, state=MASK64):           # RNG rng((uint64)-1)
        self.state = state & MASK64

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * CV_RNG_COEFF + (self.state >> 32)) & MASK64
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else a + self.next() % (b - a)


def get_subset(rng, count, model_points=5):
    """ptsetreg.cpp getSubset without checkSubset (EMEstimatorCallback has none): distinct indices, redraw on a duplicate."""
    idx = []
    while len(idx) < model_points:
        while True:
            v = rng.uniform(0, count)
            if v not in idx:
                break
        idx.append(v)
    return idx


def subset_table(count, n_iters, model_points=5):
    """The samples of the first n_iters RANSAC iterations: they depend on the match count alone."""
    rng = CvRNG()
    return np.array([get_subset(rng, count, model_points) for _ in range(n_iters)], np.int32)


def ransac_update_num_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))            # cvRound: half to even


def find_essential_mat_ransac(pts1, pts2, K, prob=0.999, threshold=1.0, max_iters=1000):
    """cv::findEssentialMat(pts1, pts2, K, RANSAC, prob, threshold, mask): (inlier count, mask, E or None, iterations
    run).  Raises if a sample reached a corner of cv::solvePoly that the restatement does not cover (sfm_oracle_score.c)."""
    cnt, mask, E, it, flags = orc.find_essential_mat(pts1, pts2, K, prob, threshold, max_iters)
    if flags:
        raise RuntimeError(f"five-point restatement: solvePoly corner reached (flags {flags})")
    return cnt, mask, E, it


def find_best_pair_scores(pair_points, K, min_matches=120):
    """findBestPair's map: pair_points = [((q, t), pts_q (n x 2), pts_t (n x 2))] in the loop's order (q < t ascending).
    Returns the std::map<float, pair> as a list ascending in the key: equal float keys keep the LAST pair inserted."""
    m = {}
    for (q, t), a, b in pair_points:
        if len(a) < min_matches:
            continue
        cnt = find_essential_mat_ransac(a, b, K)[0]
        m[np.float32(np.float32(cnt) / np.float32(len(a)))] = (q, t)
    return sorted(m.items(), key=lambda kv: kv[0])


# ------------------------------------------------------------------ findHomographyInliers (reference src/Sfm.cpp:667-689)
# cv::findHomography(query_points, train_points, CV_RANSAC, 0.004 * maxVal, inliersMask): restated in C along OpenCV
# 3.4.1's own route (oracle/sfm_oracle_score.c: float points, checkSubset, the normalised DLT with cv::eigen's Jacobi,
# float error arithmetic, 0.995 / 2000).  An independent numpy version (numpy's eigh for the eigenvector) lives in
# oracle/np_check.py and checks it (tests/test_oracle_score.py).
def find_homography_ransac(pts1, pts2, threshold, confidence=0.995, max_iters=2000):
    """(inlier count, mask, iterations run) of cv::findHomography(pts1, pts2, RANSAC, threshold, mask)"""
    return orc.find_homography(pts1, pts2, threshold, confidence, max_iters)


def homography_kernel(M, m):
    """HomographyEstimatorCallback::runKernel: M -> m, float points in, 3x3 double out (None if degenerate)"""
    return orc.homography_kernel(M, m)


def find_homography_inliers(query_pts, train_pts):
    """StructFromMotion::findHomographyInliers: threshold 0.004 * (largest coordinate of the query points)"""
    q = np.asarray(query_pts, np.float64).reshape(-1, 2)
    if len(q) < 4:
        return 0
    return find_homography_ransac(q, train_pts, 0.004 * float(q.max()))[0]
