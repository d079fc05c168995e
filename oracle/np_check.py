"""Independent numpy / scipy restatements used to pin the C oracle.

TEST INFRASTRUCTURE ONLY.  None of this is reference code: the reference ships no tests or
fixtures, so these second, independently written derivations (brute-force sort for k-NN,
numpy.linalg.svd for the DLT, complex-step / finite-difference Jacobians, a dense normal
-equations LM step and scipy.optimize.least_squares optima) are what the oracle is checked
against, and what tests/golden/*.npz are generated from (tests/golden/make_golden.py).
"""
import numpy as np


# ---------------------------------------------------------------- matcher
def knn2_bruteforce(q, t, norm="l2"):
    """k=2 nearest train rows per query row with cv::batchDistance's ordering: ascending
    (float distance, train index).  Returns idx (nq,2) int32 (-1 padded), dist (nq,2) float32."""
    q = np.asarray(q)
    t = np.asarray(t)
    nq, nt = q.shape[0], t.shape[0]
    idx = -np.ones((nq, 2), np.int32)
    dist = np.full((nq, 2), np.finfo(np.float32).max, np.float32)
    if nt == 0:
        return idx, dist
    for i in range(nq):
        if norm == "hamming":
            x = np.bitwise_xor(q[i][None, :], t)
            d = np.unpackbits(x, axis=1).sum(axis=1).astype(np.float32)
        else:
            diff = q[i].astype(np.int64)[None, :] - t.astype(np.int64) if q.dtype == np.uint8 else None
            if diff is not None:
                s = (diff * diff).sum(axis=1)
            else:
                df = q[i].astype(np.float64)[None, :] - t.astype(np.float64)
                s = (df * df).sum(axis=1)  # exact for integer-valued rows
            d = np.sqrt(s.astype(np.float32)).astype(np.float32)
        order = np.lexsort((np.arange(nt), d))  # stable: distance, then lower index
        k = min(2, nt)
        idx[i, :k] = order[:k]
        dist[i, :k] = d[order[:k]]
    return idx, dist


def ratio_filter(idx, dist, ratio=0.8):
    r = np.float32(ratio)
    ok = (idx[:, 1] >= 0) & (dist[:, 0] <= r * dist[:, 1])
    qi = np.nonzero(ok)[0].astype(np.int32)
    return qi, idx[ok, 0].copy(), dist[ok, 0].copy()


# ---------------------------------------------------------------- triangulation
def triangulate_svd(P1, P2, K, xy1, xy2, max_err=6.0):
    """DLT by numpy.linalg.svd, zero distortion.  Returns X, err (m,2) float64, keep."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    m = xy1.shape[0]
    X = np.zeros((m, 3))
    err = np.zeros((m, 2))
    for i in range(m):
        x1, y1 = (xy1[i, 0] - cx) / fx, (xy1[i, 1] - cy) / fy
        x2, y2 = (xy2[i, 0] - cx) / fx, (xy2[i, 1] - cy) / fy
        A = np.stack([x1 * P1[2] - P1[0], y1 * P1[2] - P1[1], x2 * P2[2] - P2[0], y2 * P2[2] - P2[1]])
        v = np.linalg.svd(A)[2][3]
        X[i] = v[:3] / v[3]
        for j, (P, xy) in enumerate(((P1, xy1[i]), (P2, xy2[i]))):
            p = P[:, :3] @ X[i] + P[:, 3]
            u = np.array([fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy])
            err[i, j] = np.linalg.norm(u - xy)
    keep = ~((err[:, 0] > max_err) | (err[:, 1] > max_err))
    return X, err, keep


# ---------------------------------------------------------------- bundle adjustment
def rotate_aa(aa, X):
    """ceres::AngleAxisRotatePoint, both branches, generic over real/complex (complex-step)."""
    th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]
    if np.real(th2) > np.finfo(np.float64).eps:
        th = np.sqrt(th2)
        w = aa / th
        c, s = np.cos(th), np.sin(th)
        return X * c + np.cross(w, X) * s + w * (w @ X) * (1 - c)
    return X + np.cross(aa, X)


def residual(cam, X, focal, obs):
    p = rotate_aa(cam[:3], X) + cam[3:]
    return np.array([focal * p[0] / p[2] - obs[0], focal * p[1] / p[2] - obs[1]])


def jacobian_complex_step(cam, X, focal, obs, h=1e-30):
    """d r / d [cam(6), X(3), focal] by complex-step differentiation: exact to rounding."""
    x0 = np.concatenate([cam, X, [focal]]).astype(np.complex128)
    J = np.zeros((2, 10))
    for j in range(10):
        x = x0.copy()
        x[j] += 1j * h
        J[:, j] = np.imag(residual(x[:6], x[6:9], x[9], obs)) / h
    return J


def all_residuals(cams, pts, focal, obs_cam, obs_pt, obs_xy):
    r = np.zeros((len(obs_cam), 2))
    for o in range(len(obs_cam)):
        r[o] = residual(cams[obs_cam[o]], pts[obs_pt[o]], focal, obs_xy[o])
    return r


def dense_jacobian(cams, pts, focal, obs_cam, obs_pt, obs_xy):
    """Full dense J with column order [cams(6 each), focal, points(3 each)] -- small problems."""
    nc, npt, no = len(cams), len(pts), len(obs_cam)
    J = np.zeros((2 * no, 6 * nc + 1 + 3 * npt))
    r = np.zeros(2 * no)
    for o in range(no):
        c, p = obs_cam[o], obs_pt[o]
        Jo = jacobian_complex_step(cams[c], pts[p], focal, obs_xy[o])
        J[2 * o:2 * o + 2, 6 * c:6 * c + 6] = Jo[:, :6]
        J[2 * o:2 * o + 2, 6 * nc] = Jo[:, 9]
        J[2 * o:2 * o + 2, 6 * nc + 1 + 3 * p:6 * nc + 4 + 3 * p] = Jo[:, 6:9]
        r[2 * o:2 * o + 2] = residual(cams[c], pts[p], focal, obs_xy[o])
    return J, r


def reduced_system_dense(cams, pts, focal, obs_cam, obs_pt, obs_xy, radius=1e4):
    """Ceres' first LM linear system by dense linear algebra: Jacobi scaling 1/(1+||col||),
    LM diagonal clamp [1e-6,1e32], Schur complement onto [cams, focal].  Returns S, g, scale
    (in the oracle's order [cams, points, focal]) and the full scaled-space step."""
    nc, npt = len(cams), len(pts)
    J, r = dense_jacobian(cams, pts, focal, obs_cam, obs_pt, obs_xy)
    scale = 1.0 / (1.0 + np.linalg.norm(J, axis=0))
    Js = J * scale
    diag = np.clip((Js * Js).sum(axis=0), 1e-6, 1e32)
    H = Js.T @ Js + np.diag(diag / radius)
    b = Js.T @ r
    nf = 6 * nc + 1
    B, E, Cm = H[:nf, :nf], H[:nf, nf:], H[nf:, nf:]
    Ci = np.linalg.inv(Cm)
    S = B - E @ Ci @ E.T
    g = b[:nf] - E @ Ci @ b[nf:]
    step = -np.linalg.solve(H, b)
    scale_orc = np.concatenate([scale[:6 * nc], scale[nf:], scale[6 * nc:nf]])
    return S, g, scale_orc, step, 0.5 * float(r @ r)


def solve_scipy(cams0, pts0, focal0, obs_cam, obs_pt, obs_xy, **kw):
    """Independent optimum by scipy.optimize.least_squares (trust-region reflective)."""
    from scipy.optimize import least_squares
    from scipy.sparse import lil_matrix

    nc, npt, no = len(cams0), len(pts0), len(obs_cam)

    def unpack(x):
        return x[:6 * nc].reshape(nc, 6), x[6 * nc:6 * nc + 3 * npt].reshape(npt, 3), x[-1]

    def fun(x):
        c, p, f = unpack(x)
        return all_residuals(c, p, f, obs_cam, obs_pt, obs_xy).reshape(-1)

    sp = lil_matrix((2 * no, 6 * nc + 3 * npt + 1), dtype=int)
    for o in range(no):
        sp[2 * o:2 * o + 2, 6 * obs_cam[o]:6 * obs_cam[o] + 6] = 1
        sp[2 * o:2 * o + 2, 6 * nc + 3 * obs_pt[o]:6 * nc + 3 * obs_pt[o] + 3] = 1
        sp[2 * o:2 * o + 2, -1] = 1
    x0 = np.concatenate([np.ravel(cams0), np.ravel(pts0), [focal0]])
    res = least_squares(fun, x0, jac_sparsity=sp, x_scale="jac", method="trf", ftol=1e-12, xtol=1e-12,
                        gtol=1e-12, **kw)
    c, p, f = unpack(res.x)
    return c.copy(), p.copy(), float(f), float(res.cost)


# ------------------------------------------------------------------ five-point problem, action-matrix route
# (an independent check of the C restatement of OpenCV's polynomial route, oracle/sfm_oracle_score.c)
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


def five_point_action_matrix(q1, q2):
    """Essential matrices E (q2^T E q1 = 0 for the five pairs, q = normalised image points) through the action matrix
    of the ten cubic constraints (Stewenius): an independent check of oracle/sfm_oracle_score.c, in canonical_order."""
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


# ------------------------------------------------------------------ cv::findHomography(RANSAC), numpy route
# (an independent check of the C restatement in oracle/sfm_oracle_score.c: the same bookkeeping, numpy's eigh where the
# library -- and the C restatement -- run cv::eigen's Jacobi)
class _CvRNG:
    def __init__(self, state=(1 << 64) - 1):
        self.state = state & ((1 << 64) - 1)

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & ((1 << 64) - 1)
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else a + self.next() % (b - a)


def _get_subset(rng, count, model_points):
    idx = []
    while len(idx) < model_points:
        while True:
            v = rng.uniform(0, count)
            if v not in idx:
                break
        idx.append(v)
    return idx


def _ransac_update_num_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))


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
        idx = _get_subset(rng, count, 4)
        if homography_check_subset(m1[idx], m2[idx]):
            return idx
    return None


def homography_kernel_np(M, m):
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


def find_homography_ransac_np(pts1, pts2, threshold, confidence=0.995, max_iters=2000):
    """(inlier count, mask, iterations run) of cv::findHomography(pts1, pts2, RANSAC, threshold, mask)"""
    m1 = np.asarray(pts1, np.float64).reshape(-1, 2).astype(np.float32)
    m2 = np.asarray(pts2, np.float64).reshape(-1, 2).astype(np.float32)
    count = len(m1)
    if threshold <= 0:
        threshold = 3.0
    if count < 4:
        return 0, np.zeros(count, np.uint8), 0
    if count == 4:
        return (4, np.ones(4, np.uint8), 0) if homography_kernel_np(m1, m2) is not None else (0, np.zeros(4, np.uint8), 0)
    t = np.float32(threshold * threshold)
    rng = _CvRNG()
    niters, best, best_mask, it = max(max_iters, 1), 0, np.zeros(count, np.uint8), 0
    while it < niters:
        idx = homography_subset(rng, m1, m2)
        if idx is None:
            break                                # (iter == 0: run() returns false; later: the loop ends)
        H = homography_kernel_np(m1[idx], m2[idx])
        if H is not None:
            mask = homography_error(H, m1, m2) <= t
            good = int(mask.sum())
            if good > max(best, 3):
                best, best_mask = good, mask.astype(np.uint8)
                niters = _ransac_update_num_iters(confidence, (count - good) / count, 4, niters)
        it += 1
    return best, best_mask, it
