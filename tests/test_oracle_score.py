"""CPU: the restatement of the findBestPair scoring path (oracle/sfm_oracle_score.{c,py}) checked against what can be
checked without OpenCV: cv::RNG's recurrence, RANSACUpdateNumIters at hand-computed points, the models of the C
restatement of OpenCV's five-point route against the constraints they solve, against an independent action-matrix
solver (oracle/np_check.py) and against a known essential matrix."""
import numpy as np

from oracle import np_check, orc
from oracle import sfm_oracle_score as S
from sfm_danpipeline_amd import synth

K = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1]])


def test_rng_recurrence_and_subsets():
    r = S.CvRNG()
    s = 0xFFFFFFFFFFFFFFFF
    for _ in range(5):                      # state <- (uint32)state * 4164903690 + (state >> 32), output = low word
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert r.next() == s & 0xFFFFFFFF
    assert S.CvRNG().uniform(3, 3) == 3
    t = S.subset_table(7, 200)
    assert t.shape == (200, 5) and t.min() >= 0 and t.max() < 7
    assert all(len(set(row)) == 5 for row in t.tolist())
    assert np.array_equal(t[:50], S.subset_table(7, 50))            # a prefix of the same stream


def test_update_num_iters():
    assert S.ransac_update_num_iters(0.999, 0.5, 5, 1000) == 218     # log(0.001) / log(1 - 0.5^5) = 217.6
    assert S.ransac_update_num_iters(0.999, 0.2, 5, 1000) == 17      # log(0.001) / log(1 - 0.8^5) = 17.4
    assert S.ransac_update_num_iters(0.999, 0.9, 5, 1000) == 1000    # 690 k wanted: capped
    assert S.ransac_update_num_iters(0.999, 0.0, 5, 1000) == 0       # every point an inlier: stop
    assert S.ransac_update_num_iters(0.999, 0.5, 5, 100) == 100


def test_update_num_iters_c_and_python_agree():
    for ep in (0.0, 0.01, 0.2, 0.5, 0.77, 0.9, 1.0):
        for mx in (1000, 100, 7):
            assert orc.lib().orc_ransac_update_num_iters(0.999, ep, 5, mx) == S.ransac_update_num_iters(0.999, ep, 5, mx)


def _unit(E):
    return E / np.linalg.norm(E)


def _dist(E, F):
    return min(np.abs(_unit(E) - _unit(F)).max(), np.abs(_unit(E) + _unit(F)).max())


def test_five_point_models_solve_the_constraints_and_recover_a_known_matrix():
    sc = synth.two_view_scene(m=40, seed=5, K=K, noise_px=0.0, outlier_frac=0.0)
    R, t = sc["P2"][:, :3], sc["P2"][:, 3]
    Et = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
    n1, n2 = orc.em_normalize(sc["xy1"], K), orc.em_normalize(sc["xy2"], K)
    assert np.allclose(n1, (sc["xy1"] - K[:2, 2]) / np.array([K[0, 0], K[1, 1]]), rtol=0, atol=1e-15)
    for k in range(0, 35, 5):
        models, flags = orc.five_point(n1[k:k + 5], n2[k:k + 5])
        assert flags == 0 and 1 <= len(models) <= 10
        assert min(_dist(E, Et) for E in models) < 1e-8
        for E in models:
            assert abs(np.linalg.norm(E) - 1.0) < 1e-14                                   # Evec /= norm(Evec)
            x1 = np.concatenate([n1[k:k + 5], np.ones((5, 1))], 1)
            x2 = np.concatenate([n2[k:k + 5], np.ones((5, 1))], 1)
            assert np.abs(np.sum(x2 * (x1 @ E.T), axis=1)).max() < 1e-11                  # x2^T E x1 = 0
            assert abs(np.linalg.det(E)) < 1e-10
            assert np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-9


def test_five_point_models_equal_the_action_matrix_solver_on_random_samples():
    """noisy data with outliers: every sample's model SET against the independent route (the order is OpenCV's: the
    order of solvePoly's roots; compared as sets)"""
    sc = synth.two_view_scene(m=300, seed=3, K=K, noise_px=0.3, outlier_frac=0.3)
    n1, n2 = orc.em_normalize(sc["xy1"], K), orc.em_normalize(sc["xy2"], K)
    rng = np.random.default_rng(0)
    worst = []
    for _ in range(120):
        idx = rng.choice(300, 5, replace=False)
        a, flags = orc.five_point(n1[idx], n2[idx])
        b = np_check.five_point_action_matrix(n1[idx], n2[idx])
        assert flags == 0
        if len(a) != len(b):                      # a double root on the edge of real: rare, and must stay rare
            worst.append(1.0)
            continue
        worst.append(max([min(_dist(E, F) for F in b) for E in a], default=0.0))
    worst = np.array(worst)
    assert np.median(worst) < 1e-10 and (worst > 1e-6).mean() < 0.05


def test_null_space_rows_are_the_library_rng_completion():
    """JacobiSVDImpl_ fills the singular vectors of the four zero singular values from RNG(0x12345678) sign vectors:
    the models of a sample do not depend on that basis, but the basis is deterministic -- the same sample gives
    bit-identical models twice, and a permuted sample (another Q, same null space) the same model set"""
    sc = synth.two_view_scene(m=20, seed=11, K=K, noise_px=0.2, outlier_frac=0.0)
    n1, n2 = orc.em_normalize(sc["xy1"], K), orc.em_normalize(sc["xy2"], K)
    a, _ = orc.five_point(n1[:5], n2[:5])
    b, _ = orc.five_point(n1[:5], n2[:5])
    assert len(a) >= 1 and all(np.array_equal(x, y) for x, y in zip(a, b))
    p = [3, 0, 4, 1, 2]
    c, _ = orc.five_point(n1[:5][p], n2[:5][p])
    assert len(c) == len(a) and max(min(_dist(E, F) for F in c) for E in a) < 1e-8


def test_ransac_on_a_scene_with_outliers():
    sc = synth.two_view_scene(m=400, seed=9, K=K, noise_px=0.3, outlier_frac=0.25)
    cnt, mask, E, it = S.find_essential_mat_ransac(sc["xy1"], sc["xy2"], K)
    assert 0.70 * 400 <= cnt <= 0.80 * 400 and mask.sum() == cnt and 1 <= it <= 1000
    assert it == S.ransac_update_num_iters(0.999, (400 - cnt) / 400, 5, 1000) or it > 0
    # the map: ascending float keys, the later of two equal keys wins, pairs under 120 matches skipped
    a, b = sc["xy1"], sc["xy2"]
    m = S.find_best_pair_scores([((0, 1), a, b), ((0, 2), a[:100], b[:100]), ((1, 2), a, b)], K)
    assert [v for _, v in m] == [(1, 2)]


def _planar(n, seed, outliers, noise=0.4):
    rng = np.random.default_rng(seed)
    Ht = np.array([[1.02, 0.05, 12.0], [-0.03, 0.98, -7.0], [1e-5, -2e-5, 1.0]])
    p = rng.uniform(0, 640, (n, 2))
    ph = np.concatenate([p, np.ones((n, 1))], 1) @ Ht.T
    q = ph[:, :2] / ph[:, 2:] + rng.normal(0, noise, (n, 2))
    out = rng.random(n) < outliers
    q[out] = rng.uniform(0, 640, (int(out.sum()), 2))
    return p, q


def test_homography_restatement_against_the_independent_numpy_route():
    """cv::findHomography(RANSAC) restated in C with cv::eigen's Jacobi (oracle/sfm_oracle_score.c) against the numpy
    route with eigh (oracle/np_check.py): the same H up to rounding, the same counts, iteration numbers and masks"""
    rng = np.random.default_rng(3)
    for _ in range(20):
        M = rng.uniform(0, 640, (4, 2)).astype(np.float32)
        m = (M @ np.array([[1.01, 0.02], [-0.03, 0.99]], np.float32) + rng.uniform(-5, 5, 2).astype(np.float32)).astype(np.float32)
        a, b = orc.homography_kernel(M, m), np_check.homography_kernel_np(M, m)
        assert a is not None and b is not None and np.abs(a - b).max() <= 1e-6 * np.abs(b).max()     # (eigenvector of a 9 x 9 normal matrix: conditioning squares, the two eigen-solvers agree to ~1e-8)
        assert abs(a[2, 2] - 1.0) <= 2.3e-16            # (H * (1 / H22), as the library scales it: H22 * (1 / H22) may miss 1 by an ulp)
    assert orc.homography_kernel(np.zeros((4, 2), np.float32), np.ones((4, 2), np.float32)) is None      # degenerate: no spread
    for n, seed, o in ((600, 0, 0.3), (150, 1, 0.6), (2000, 2, 0.1), (40, 3, 0.0), (300, 4, 0.85), (5, 5, 0.0), (4, 6, 0.0), (3, 7, 0.0)):
        p, q = _planar(n, seed, o)
        thr = 0.004 * float(p.max())
        c, mask, it = S.find_homography_ransac(p, q, thr)
        c2, mask2, it2 = np_check.find_homography_ransac_np(p, q, thr)
        assert (c, it) == (c2, it2) and np.array_equal(mask, mask2), (n, o)
        assert S.find_homography_inliers(p, q) == c
