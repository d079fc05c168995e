"""GPU: the scoring half of findBestPair (reference src/Sfm.cpp:533-569) --
sfmhip_score_essential against the restatement of OpenCV 3.4.1's findEssentialMat(RANSAC) along the library's own
five-point route (oracle/sfm_oracle_score.c, the only route; PARITY UNPINNED: OpenCV is not in the image, see that
file's header), sfmhip_score_homography against the numpy restatement in oracle/sfm_oracle_score.py."""
import numpy as np
import pytest

from oracle import sfm_oracle_score as S
from sfm_danpipeline_amd import scoring, synth

pytestmark = pytest.mark.gpu
K = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1]])


def _scene(m, seed, outliers=0.1, noise=0.3):
    sc = synth.two_view_scene(m=m, seed=seed, K=K, noise_px=noise, outlier_frac=outliers)
    return sc["xy1"], sc["xy2"]


def test_inlier_counts_iterations_and_masks_match_the_restatement(ctx):
    pairs = [_scene(500, 99), _scene(150, 3, outliers=0.3), _scene(2000, 7, outliers=0.5, noise=0.5), _scene(121, 11, outliers=0.0),
             _scene(40, 5), _scene(777, 21, outliers=0.7)]
    inl, masks, its = scoring.score_essential(pairs, K, want_mask=True, ctx=ctx)
    for i, (a, b) in enumerate(pairs):
        cnt, mask, E, it = S.find_essential_mat_ransac(a, b, K)
        assert (int(inl[i]), int(its[i])) == (cnt, it), i
        assert int(masks[i].sum()) == cnt
        assert np.array_equal(masks[i], mask), i          # (the best model is unique here: no ties between models)


def test_degenerate_inputs(ctx):
    a, b = _scene(60, 1)
    pairs = [(a[:0], b[:0]), (a[:4], b[:4]), (a[:5], b[:5]), (a[:6], b[:6]), (a[:13], b[:13])]
    inl, masks, its = scoring.score_essential(pairs, K, want_mask=True, ctx=ctx)
    assert inl[0] == 0 and inl[1] == 0                     # fewer than five matches: no model
    assert inl[2] in (0, 5) and (inl[2] == 0 or masks[2].all())   # exactly five: every match an inlier when a model exists
    for i in (3, 4):
        cnt, mask, E, it = S.find_essential_mat_ransac(pairs[i][0], pairs[i][1], K)
        assert (int(inl[i]), int(its[i])) == (cnt, it)
    inl0, _, _ = scoring.score_essential([], K, ctx=ctx)
    assert len(inl0) == 0


def test_many_pairs_with_few_inliers_every_pair_reported(ctx):
    """the hard regime -- 50 to 85 % wrong matches, hundreds of RANSAC iterations per pair, ill-conditioned samples among
    them: every pair's count, iteration number and mask against the one oracle route; all disagreements are listed"""
    pairs = [_scene(150 + 37 * i, 300 + i, outliers=0.5 + 0.05 * (i % 8), noise=0.4) for i in range(40)]
    inl, masks, its = scoring.score_essential(pairs, K, want_mask=True, ctx=ctx)
    assert scoring.last_flags(ctx) == 0
    bad = []
    for i, (a, b) in enumerate(pairs):
        cnt, mask, E, it = S.find_essential_mat_ransac(a, b, K)
        if (int(inl[i]), int(its[i])) != (cnt, it) or not np.array_equal(masks[i], mask):
            bad.append((i, len(a), (int(inl[i]), int(its[i])), (cnt, it)))
    assert not bad, bad


def test_five_point_models_are_the_restatements_bit_for_bit(ctx):
    """sfmhip_score_five_point against the C restatement on thousands of samples of noisy scenes with outliers, a few
    degenerate ones among them: the same number of models, in the same order, with the same bits.  (Ill-conditioned
    samples amplify a last-bit difference of any intermediate to 1e-3 of E: anything short of following the checker
    operation for operation -- the host libm's hypot included, csrc/hypot_glibc.h -- shows up here.)"""
    from oracle import orc
    rng = np.random.default_rng(5)
    q1, q2 = [], []
    for seed in range(6):
        a, b = _scene(400, 700 + seed, outliers=0.4, noise=0.5)
        n1, n2 = orc.em_normalize(a, K), orc.em_normalize(b, K)
        for _ in range(500):
            idx = rng.choice(400, 5, replace=False)
            q1.append(n1[idx])
            q2.append(n2[idx])
    # degenerate samples: a repeated correspondence, five times the same one, collinear points
    q1.append(np.concatenate([q1[0][:4], q1[0][3:4]])); q2.append(np.concatenate([q2[0][:4], q2[0][3:4]]))
    q1.append(np.repeat(q1[1][:1], 5, 0)); q2.append(np.repeat(q2[1][:1], 5, 0))
    q1.append(np.stack([np.linspace(-0.1, 0.1, 5), np.linspace(-0.05, 0.05, 5)], 1)); q2.append(q1[-1] + 0.01)
    q1, q2 = np.array(q1), np.array(q2)
    models, counts, flags = scoring.five_point(q1, q2, ctx=ctx)
    differ = []
    for i in range(len(q1)):
        want, fl = orc.five_point(q1[i], q2[i])
        same = counts[i] == len(want) and flags[i] == fl and all(
            np.array_equal(models[i, m].view(np.uint64), want[m].view(np.uint64)) or
            (np.isnan(models[i, m]).all() and np.isnan(want[m]).all()) for m in range(len(want)))
        if not same:
            differ.append(i)
    assert not differ, (len(differ), differ[:10])
    assert counts[:3000].max() <= 10 and counts[:3000].mean() > 2


def test_find_best_pair_map_semantics(ctx):
    """std::map<float, pair>: ascending keys, equal keys keep the last pair; pairs below 120 matches are skipped."""
    a, b = _scene(300, 2)
    c, d = _scene(200, 4, outliers=0.4)
    ids = [(0, 1), (0, 2), (1, 2), (1, 3)]
    pts = [(a, b), (c, d), (a, b), (a[:100], b[:100])]     # pairs 0 and 2 score the same ratio; pair 3 is too small
    got = scoring.find_best_pair(ids, pts, K, ctx=ctx)
    want = S.find_best_pair_scores(list(zip(ids, [p[0] for p in pts], [p[1] for p in pts])), K)
    assert [(float(k), v) for k, v in got] == [(float(k), v) for k, v in want]
    assert (1, 2) in [v for _, v in got] and (0, 1) not in [v for _, v in got] and (1, 3) not in [v for _, v in got]
    assert [k for k, _ in got] == sorted(k for k, _ in got)


def test_noise_free_scenes_end_at_the_first_sample(ctx):
    for trial in range(6):
        a, b = _scene(400, 100 + trial, outliers=0.0, noise=0.0)
        cnt, mask, E, it = S.find_essential_mat_ransac(a, b, K)
        inl, _, its = scoring.score_essential([(a, b)], K, ctx=ctx)
        assert int(inl[0]) == cnt == 400 and int(its[0]) == it      # noise-free: the first sample explains everything


def _planar(n, seed, outliers, noise=0.4):
    rng = np.random.default_rng(seed)
    Ht = np.array([[1.02, 0.05, 12.0], [-0.03, 0.98, -7.0], [1e-5, -2e-5, 1.0]])
    p = rng.uniform(0, 640, (n, 2))
    ph = np.concatenate([p, np.ones((n, 1))], 1) @ Ht.T
    q = ph[:, :2] / ph[:, 2:] + rng.normal(0, noise, (n, 2))
    out = rng.random(n) < outliers
    q[out] = rng.uniform(0, 640, (int(out.sum()), 2))
    return p, q


def test_homography_inliers_match_the_restatement(ctx):
    """findHomographyInliers (src/Sfm.cpp:667-689): counts, iteration numbers and masks against the numpy
    restatement of cv::findHomography(RANSAC) -- float points, checkSubset, float error arithmetic."""
    a, b = _scene(400, 31, outliers=0.2)                      # a general scene: few points agree with any homography
    pairs = [_planar(600, 0, 0.3), _planar(150, 1, 0.6), _planar(2000, 2, 0.1), (a, b), _planar(40, 3, 0.0), _planar(300, 4, 0.85)]
    inl, masks, its = scoring.score_homography(pairs, want_mask=True, ctx=ctx)
    for i, (p, q) in enumerate(pairs):
        cnt, mask, it = S.find_homography_ransac(p, q, 0.004 * float(np.max(p)))
        assert (int(inl[i]), int(its[i])) == (cnt, it), i
        assert np.array_equal(masks[i], mask), i
        assert S.find_homography_inliers(p, q) == cnt
    assert inl[0] > 0.6 * 600 and inl[3] < 0.5 * 400


def test_homography_models_are_the_restatements_bit_for_bit(ctx):
    """sfmhip_score_homography_kernel (normalised DLT + cv::eigen's Jacobi on the device) against the C restatement on
    2000 random 4-point samples, near-degenerate ones among them: the same nine doubles, bit for bit"""
    from oracle import orc
    rng = np.random.default_rng(12)
    M = rng.uniform(0, 640, (2000, 4, 2)).astype(np.float32)
    A = np.array([[1.02, 0.05], [-0.03, 0.98]], np.float32)
    m = (M @ A + rng.normal(0, 3.0, M.shape)).astype(np.float32)
    M[10, 3] = M[10, 2] + np.float32(1e-3)          # two points almost on top of each other
    M[11] = M[11, 0] + np.arange(4, dtype=np.float32)[:, None] * np.float32([1.0, 2.0])   # collinear
    m[12] = m[12, 0]                                 # all four train points equal: no spread -> degenerate
    H, ok = scoring.homography_kernel(M, m, ctx=ctx)
    differ = []
    for i in range(len(M)):
        want = orc.homography_kernel(M[i], m[i])
        if (want is None) != (ok[i] == 0) or (want is not None and not np.array_equal(H[i].view(np.uint64), want.view(np.uint64))):
            if not (want is not None and np.isnan(want).all() and np.isnan(H[i]).all()):
                differ.append(i)
    assert not differ, (len(differ), differ[:10])
    assert ok[12] == 0 and ok.sum() >= 1990


def test_homography_degenerate_inputs(ctx):
    p, q = _planar(50, 9, 0.0)
    line = np.stack([np.arange(30.0), 2 * np.arange(30.0)], 1)      # every sample is collinear: getSubset gives up
    pairs = [(p[:0], q[:0]), (p[:3], q[:3]), (p[:4], q[:4]), (p[:5], q[:5]), (line, line + 1.0)]
    inl, masks, its = scoring.score_homography(pairs, want_mask=True, ctx=ctx)
    assert inl[0] == 0 and inl[1] == 0
    assert inl[2] == 4 and masks[2].all()                     # npoints == 4: runKernel on all of them, the mask all ones
    cnt, mask, it = S.find_homography_ransac(p[:5], q[:5], 0.004 * float(np.max(p[:5])))
    assert (int(inl[3]), int(its[3])) == (cnt, it)
    assert inl[4] == 0 and its[4] == 0                        # run() returns false at iteration 0
