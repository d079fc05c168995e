"""ORACLE (test infrastructure, not product): the scoring half of StructFromMotion::findBestPair (reference
src/Sfm.cpp:533-569) -- cv::findEssentialMat(alignedLeft, alignedRight, K, RANSAC, 0.999, 1.0, mask) per image pair
with >= 120 matches, poseInliersRatio = (float)inliers / (float)matches, std::map<float, pair> (ascending keys, equal
keys overwrite).

PARITY UNPINNED.  The algorithm lives in OpenCV 3.4.1 (calib3d: five-point.cpp, ptsetreg.cpp; core: cv::RNG), which is
not in /root/reference and not in this image, and the reference holds no test or golden vector for this path.  What is
restated from the published library:
  * cv::RNG (multiply-with-carry, coefficient 4164903690, state (uint64)-1 per findEssentialMat call) and
    RNG::uniform(int a, int b) = a + next() % (b - a);
  * RANSACPointSetRegistrator::getSubset (five distinct indices, a duplicate is drawn again), ::run (models of a
    sample in order; goodCount > max(maxGoodCount, 4) updates the best and niters), RANSACUpdateNumIters,
    findInliers (float error <= (float)(t*t)), maxIters 1000;
  * findEssentialMat's normalisation ((p - c) / f per axis, threshold / ((fx + fy) / 2)) and EMEstimatorCallback::
    computeError (squared epipolar residual over the four squared line coefficients, cast to float).
What is NOT OpenCV's code path: the five-point solver itself.  OpenCV expands Nister's constraints with generated code
and finds the roots of the tenth-degree polynomial with solvePoly; here the same ten cubic constraints are expanded
numerically and solved through the action matrix (Stewenius).  Both yield the same essential matrices up to scale and
rounding; the ORDER of a sample's models may differ (here: canonical_order below), which cannot change the final inlier count
(the best count of a sample wins whatever the order, and niters depends on the count alone) but can change which of two
equally good models supplies the mask.
"""
import numpy as np

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


# ------------------------------------------------------------------ five-point solver
MONO = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3),
        (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]


def _pmul(a, b):
    """product of two polynomials in (x, y, z) stored as c[i, j, k] (4 x 4 x 4: total degree <= 3 is all we form)"""
    out = np.zeros((4, 4, 4))
    for (i, j, k), v in np.ndenumerate(a):
        if v != 0.0:
            for (p, q, r), w in np.ndenumerate(b):
                if w != 0.0 and i + p < 4 and j + q < 4 and k + r < 4:
                    out[i + p, j + q, k + r] += v * w
    return out


def constraint_matrix(X, Y, Z, W):
    """10 x 20 coefficients (monomials MONO) of det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0, E = xX + yY + zZ + W."""
    E = np.empty((3, 3), object)
    for a in range(3):
        for b in range(3):
            c = np.zeros((4, 4, 4))
            c[1, 0, 0], c[0, 1, 0], c[0, 0, 1], c[0, 0, 0] = X[a, b], Y[a, b], Z[a, b], W[a, b]
            E[a, b] = c
    det = (_pmul(_pmul(E[0, 0], E[1, 1]), E[2, 2]) + _pmul(_pmul(E[0, 1], E[1, 2]), E[2, 0]) + _pmul(_pmul(E[0, 2], E[1, 0]), E[2, 1])
           - _pmul(_pmul(E[0, 2], E[1, 1]), E[2, 0]) - _pmul(_pmul(E[0, 1], E[1, 0]), E[2, 2]) - _pmul(_pmul(E[0, 0], E[1, 2]), E[2, 1]))
    EEt = np.empty((3, 3), object)
    for a in range(3):
        for b in range(3):
            EEt[a, b] = sum(_pmul(E[a, k], E[b, k]) for k in range(3))
    tr = EEt[0, 0] + EEt[1, 1] + EEt[2, 2]
    rows = [det]
    for a in range(3):
        for b in range(3):
            rows.append(2.0 * sum(_pmul(EEt[a, k], E[k, b]) for k in range(3)) - _pmul(tr, E[a, b]))
    return np.array([[r[m] for m in MONO] for r in rows])


def canonical_order(models):
    """A sample's models in an order that does not depend on the null-space basis: by the first entry of E / ||E||_F,
    sign fixed so that the entry of largest magnitude is positive."""
    def key(E):
        n = E / np.linalg.norm(E)
        k = np.argmax(np.abs(n))
        return (n * np.sign(n.flat[k]))[0, 0]
    return sorted(models, key=key)


def five_point(q1, q2):
    """Essential matrices E (q2^T E q1 = 0 for the five pairs, q = normalised image points), in canonical_order."""
    x1, y1, x2, y2 = q1[:, 0], q1[:, 1], q2[:, 0], q2[:, 1]
    Q = np.stack([x1 * x2, x2 * y1, x2, x1 * y2, y1 * y2, y2, x1, y1, np.ones(5)], axis=1)
    _, _, Vt = np.linalg.svd(Q)
    X, Y, Z, W = (Vt[5 + k].reshape(3, 3) for k in range(4))
    M = constraint_matrix(X, Y, Z, W)
    try:
        B = np.linalg.solve(M[:, :10], M[:, 10:])
    except np.linalg.LinAlgError:
        return []
    A = np.zeros((10, 10))
    A[:6] = -B[:6]
    A[6, 0] = A[7, 1] = A[8, 2] = A[9, 6] = 1.0
    w, V = np.linalg.eig(A)
    sols = []
    for k in range(10):
        if abs(w[k].imag) > 1e-10 * max(1.0, abs(w[k].real)):
            continue
        v = V[:, k].real
        if v[9] == 0:
            continue
        x, y, z = v[6] / v[9], v[7] / v[9], v[8] / v[9]
        sols.append(x * X + y * Y + z * Z + W)
    return canonical_order(sols)


def compute_error(E, p1, p2):
    """EMEstimatorCallback::computeError: float per correspondence (p = normalised points, n x 2)."""
    x1 = np.concatenate([p1, np.ones((len(p1), 1))], axis=1)
    x2 = np.concatenate([p2, np.ones((len(p2), 1))], axis=1)
    Ex1 = x1 @ E.T
    Etx2 = x2 @ E
    x2tEx1 = np.sum(x2 * Ex1, axis=1)
    den = Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        return (x2tEx1 * x2tEx1 / den).astype(np.float32)


def find_essential_mat_ransac(pts1, pts2, K, prob=0.999, threshold=1.0, max_iters=1000, solver=None):
    """cv::findEssentialMat(pts1, pts2, K, RANSAC, prob, threshold, mask): (inlier count, mask, E or None, iterations run).
    solver: five_point (default; action matrix) or five_point_hidden_variable (the device's route).  On well-conditioned
    samples the two give the same matrices; over hundreds of iterations on data with few inliers an ill-conditioned
    sample can give one route a root the other misses, and with it a different best count (seen: 21 vs 20 of 57)."""
    solver = solver or five_point
    pts1 = np.asarray(pts1, np.float64).reshape(-1, 2)
    pts2 = np.asarray(pts2, np.float64).reshape(-1, 2)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    p1 = np.stack([(pts1[:, 0] - cx) / fx, (pts1[:, 1] - cy) / fy], axis=1)
    p2 = np.stack([(pts2[:, 0] - cx) / fx, (pts2[:, 1] - cy) / fy], axis=1)
    thr = threshold / ((fx + fy) / 2)
    count = len(p1)
    t = np.float32(thr * thr)
    if count < 5:
        return 0, np.zeros(count, np.uint8), None, 0
    if count == 5:
        models = solver(p1, p2)
        return (5, np.ones(5, np.uint8), models[0], 0) if models else (0, np.zeros(5, np.uint8), None, 0)
    rng = CvRNG()
    niters, best_count, best_mask, best_E, it = max(max_iters, 1), 0, np.zeros(count, np.uint8), None, 0
    while it < niters:
        idx = get_subset(rng, count)
        for E in solver(p1[idx], p2[idx]):
            mask = compute_error(E, p1, p2) <= t          # (NaN compares false, as in C)
            good = int(mask.sum())
            if good > max(best_count, 4):
                best_count, best_mask, best_E = good, mask.astype(np.uint8), E
                niters = ransac_update_num_iters(prob, (count - good) / count, 5, niters)
        it += 1
    return best_count, best_mask, best_E, it


def find_best_pair_scores(pair_points, K, min_matches=120, solver=None):
    """findBestPair's map: pair_points = [((q, t), pts_q (n x 2), pts_t (n x 2))] in the loop's order (q < t ascending).
    Returns the std::map<float, pair> as a list ascending in the key: equal float keys keep the LAST pair inserted."""
    m = {}
    for (q, t), a, b in pair_points:
        if len(a) < min_matches:
            continue
        cnt = find_essential_mat_ransac(a, b, K, solver=solver)[0]
        m[np.float32(np.float32(cnt) / np.float32(len(a)))] = (q, t)
    return sorted(m.items(), key=lambda kv: kv[0])


# ------------------------------------------------------------------ the same solver the way the device code does it
# (Gauss-Jordan null space, Nister's elimination order, the tenth-degree polynomial in z, real roots by the
# interlacing of the derivatives' roots): a second, independent route to the same matrices -- tests compare the two.
NISTER = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
          (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]


def _null_space_gj(Q):
    """4 orthonormalised null vectors of the 5 x 9 matrix by Gauss-Jordan with full pivoting + modified Gram-Schmidt."""
    A = Q.astype(np.float64).copy()
    cols = list(range(9))
    for i in range(5):
        sub = np.abs(A[i:, i:])
        r, c = np.unravel_index(np.argmax(sub), sub.shape)
        A[[i, i + r]] = A[[i + r, i]]
        A[:, [i, i + c]] = A[:, [i + c, i]]
        cols[i], cols[i + c] = cols[i + c], cols[i]
        if A[i, i] == 0.0:
            return None
        A[i] /= A[i, i]
        for k in range(5):
            if k != i:
                A[k] -= A[k, i] * A[i]
    basis = []
    for f in range(5, 9):
        v = np.zeros(9)
        v[cols[f]] = 1.0
        for i in range(5):
            v[cols[i]] = -A[i, f]
        for b in basis:
            v -= (v @ b) * b
        basis.append(v / np.linalg.norm(v))
    return basis


def _real_roots(c):
    """all real roots of sum c[k] z^k, ascending: the roots of each derivative bracket the roots of the one before"""
    n = len(c) - 1
    while n > 0 and c[n] == 0:
        n -= 1
    if n == 0:
        return []
    c = np.asarray(c[:n + 1], np.float64)
    bound = 1.0 + np.max(np.abs(c[:-1] / c[-1]))
    derivs = [c]
    for _ in range(n - 1):
        d = derivs[-1]
        derivs.append(d[1:] * np.arange(1, len(d)))
    roots = []                                  # of the current (lowest-degree) derivative
    for d in reversed(derivs):
        ev = lambda z: np.polyval(d[::-1], z)
        pts = [-bound] + roots + [bound]
        new = []
        for a, b in zip(pts[:-1], pts[1:]):
            fa, fb = ev(a), ev(b)
            if fa == 0.0:
                new.append(a)
                continue
            if fa * fb > 0 or b <= a:
                continue
            lo, hi = a, b
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                if mid == lo or mid == hi:
                    break
                fm = ev(mid)
                if (fm > 0) == (fa > 0):
                    lo = mid
                else:
                    hi = mid
            new.append(0.5 * (lo + hi))
        roots = new
    return roots


def five_point_hidden_variable(q1, q2):
    x1, y1, x2, y2 = q1[:, 0], q1[:, 1], q2[:, 0], q2[:, 1]
    Q = np.stack([x1 * x2, x2 * y1, x2, x1 * y2, y1 * y2, y2, x1, y1, np.ones(5)], axis=1)
    basis = _null_space_gj(Q)
    if basis is None:
        return []
    X, Y, Z, W = (v.reshape(3, 3) for v in basis)
    M20 = constraint_matrix(X, Y, Z, W)
    perm = [MONO.index(m) for m in NISTER]
    M = M20[:, perm]
    # Gauss-Jordan on the first ten columns, partial pivoting
    for i in range(10):
        p = i + int(np.argmax(np.abs(M[i:, i])))
        if M[p, i] == 0.0:
            return []
        M[[i, p]] = M[[p, i]]
        M[i] /= M[i, i]
        for k in range(10):
            if k != i:
                M[k] -= M[k, i] * M[i]
    R = M[:, 10:]

    def brow(hi, lo):       # <row hi> - z <row lo>: [p(z) (deg 3), q(z) (deg 3), r(z) (deg 4)], ascending powers
        a, b = R[hi], R[lo]
        p = [a[2], a[1] - b[2], a[0] - b[1], -b[0]]
        q = [a[5], a[4] - b[5], a[3] - b[4], -b[3]]
        r = [a[9], a[8] - b[9], a[7] - b[8], a[6] - b[7], -b[6]]
        return [np.array(p), np.array(q), np.array(r)]
    Bk, Bl, Bm = brow(4, 5), brow(6, 7), brow(8, 9)
    pm = np.polynomial.polynomial.polymul
    sub = np.polynomial.polynomial.polysub
    det = np.polynomial.polynomial.polyadd(
        sub(pm(Bk[0], sub(pm(Bl[1], Bm[2]), pm(Bm[1], Bl[2]))), pm(Bk[1], sub(pm(Bl[0], Bm[2]), pm(Bm[0], Bl[2])))),
        pm(Bk[2], sub(pm(Bl[0], Bm[1]), pm(Bm[0], Bl[1]))))
    det = np.concatenate([det, np.zeros(11 - len(det))])
    out = []
    for z in _real_roots(det):
        rows = [np.array([np.polyval(b[::-1], z) for b in B]) for B in (Bk, Bl, Bm)]
        best = None
        for a, b in ((0, 1), (0, 2), (1, 2)):
            c = np.cross(rows[a], rows[b])
            if best is None or abs(c[2]) > abs(best[2]):
                best = c
        if best[2] == 0.0:
            continue
        x, y = best[0] / best[2], best[1] / best[2]
        out.append(x * X + y * Y + z * Z + W)
    return canonical_order(out)


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
