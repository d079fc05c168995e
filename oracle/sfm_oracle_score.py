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
# cv::findHomography(query_points, train_points, CV_RANSAC, 0.004 * maxVal, inliersMask) as OpenCV 3.4.1 (calib3d/
# fundam.cpp) runs it: points converted to float; RANSAC with 4-point samples (a sample is drawn again when
# checkSubset rejects it: the last point collinear with two others, or the orientation of the four triples not
# preserved), confidence 0.995, at most 2000 iterations; model = normalised DLT, the eigenvector of the smallest
# eigenvalue of L^T L, scaled to H[2][2] = 1; error in FLOAT arithmetic; the mask returned is the RANSAC mask (the
# refit on the inliers and the LM refinement change H only).  PARITY UNPINNED like the rest of this file; restated
# from the published library, with numpy's eigh where OpenCV runs its Jacobi eigen().
FLT_EPSILON = float(np.finfo(np.float32).eps)


def _have_collinear(p, count):
    i = count - 1
    for j in range(i):
        dx1, dy1 = float(p[j][0]) - float(p[i][0]), float(p[j][1]) - float(p[i][1])
        for k in range(j):
            dx2, dy2 = float(p[k][0]) - float(p[i][0]), float(p[k][1]) - float(p[i][1])
            if abs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def homography_check_subset(s1, s2):
    """HomographyEstimatorCallback::checkSubset for four float correspondences"""
    if _have_collinear(s1, 4) or _have_collinear(s2, 4):
        return False
    negative = 0
    for t in ((0, 1, 2), (1, 2, 3), (0, 2, 3), (0, 1, 3)):
        A = np.array([[s1[k][0], s1[k][1], 1.0] for k in t], np.float64)
        B = np.array([[s2[k][0], s2[k][1], 1.0] for k in t], np.float64)
        negative += _det3(A) * _det3(B) < 0
    return negative in (0, 4)


def _det3(m):          # cv::determinant for Matx33d: cofactor expansion along the first row
    return (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
            + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))


def homography_subset(rng, m1, m2, max_attempts=10000):
    count = len(m1)
    for _ in range(max_attempts):
        idx = get_subset(rng, count, 4)
        if homography_check_subset(m1[idx], m2[idx]):
            return idx
    return None


def homography_kernel(M, m):
    """HomographyEstimatorCallback::runKernel: M -> m, float points in, 3x3 double out (None if degenerate)"""
    M = M.astype(np.float64)
    m = m.astype(np.float64)
    n = len(M)
    cM, cm = M.sum(0) / n, m.sum(0) / n
    sM, sm = np.abs(M - cM).sum(0), np.abs(m - cm).sum(0)
    eps = np.finfo(np.float64).eps
    if min(abs(sm[0]), abs(sm[1]), abs(sM[0]), abs(sM[1])) < eps:
        return None
    sm, sM = n / sm, n / sM
    inv_hnorm = np.array([[1.0 / sm[0], 0, cm[0]], [0, 1.0 / sm[1], cm[1]], [0, 0, 1]])
    hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    LtL = np.zeros((9, 9))
    for i in range(n):
        x, y = (m[i] - cm) * sm
        X, Y = (M[i] - cM) * sM
        Lx = np.array([X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x])
        Ly = np.array([0, 0, 0, X, Y, 1, -y * X, -y * Y, -y])
        LtL += np.outer(Lx, Lx) + np.outer(Ly, Ly)
    w, V = np.linalg.eigh(LtL)
    H0 = V[:, 0].reshape(3, 3)                  # the smallest eigenvalue's vector (cv::eigen sorts descending: row 8)
    H = inv_hnorm @ H0 @ hnorm2
    return H / H[2, 2]


def homography_error(H, M, m):
    """computeError: float arithmetic throughout"""
    Hf = H.astype(np.float32).reshape(-1)
    Mx, My, mx, my = (a.astype(np.float32) for a in (M[:, 0], M[:, 1], m[:, 0], m[:, 1]))
    one = np.float32(1.0)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        ww = one / (Hf[6] * Mx + Hf[7] * My + one)
        dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - mx
        dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - my
        return dx * dx + dy * dy


def find_homography_ransac(pts1, pts2, threshold, confidence=0.995, max_iters=2000):
    """(inlier count, mask, iterations run) of cv::findHomography(pts1, pts2, RANSAC, threshold, mask)"""
    m1 = np.asarray(pts1, np.float64).reshape(-1, 2).astype(np.float32)
    m2 = np.asarray(pts2, np.float64).reshape(-1, 2).astype(np.float32)
    count = len(m1)
    if threshold <= 0:
        threshold = 3.0
    if count < 4:
        return 0, np.zeros(count, np.uint8), 0
    if count == 4:
        return (4, np.ones(4, np.uint8), 0) if homography_kernel(m1, m2) is not None else (0, np.zeros(4, np.uint8), 0)
    t = np.float32(threshold * threshold)
    rng = CvRNG()
    niters, best, best_mask, it = max(max_iters, 1), 0, np.zeros(count, np.uint8), 0
    while it < niters:
        idx = homography_subset(rng, m1, m2)
        if idx is None:
            break                                # (iter == 0: run() returns false; later: the loop ends)
        H = homography_kernel(m1[idx], m2[idx])
        if H is not None:
            mask = homography_error(H, m1, m2) <= t
            good = int(mask.sum())
            if good > max(best, 3):
                best, best_mask = good, mask.astype(np.uint8)
                niters = ransac_update_num_iters(confidence, (count - good) / count, 4, niters)
        it += 1
    return best, best_mask, it


def find_homography_inliers(query_pts, train_pts):
    """StructFromMotion::findHomographyInliers: threshold 0.004 * (largest coordinate of the query points)"""
    q = np.asarray(query_pts, np.float64).reshape(-1, 2)
    if len(q) < 4:
        return 0
    return find_homography_ransac(q, train_pts, 0.004 * float(q.max()))[0]
