"""CPU: BASELINE.json configs[0] through the ORACLE only -- "data/temple sequence, CPU reference path (plumbing, no
GPU)": the committed SIFT fixture of the ten temple frames (tests/golden/temple_sift.npz, made by
tests/golden/make_temple_golden.py from the PNGs under tests/golden/temple/) through the C restatements of getMatching
and findBestPair's scoring.  The numbers below are the oracle's own (a regression pin of the checker, recorded when the
five-point restatement became OpenCV's polynomial route); the GPU run of the same sequence is tests/test_gpu_cfg1.py."""
import os

import numpy as np

from oracle import sfm_oracle_score as SC

HERE = os.path.dirname(os.path.abspath(__file__))
K = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1.0]])
# pair: (matches, E inliers, E iterations, H inliers, H iterations) for the pairs that pass the 120-match cut (:533)
PINNED = {
    (0, 1): (480, 452, 7, 299, 32), (0, 2): (315, 280, 13, 123, 225), (0, 3): (221, 182, 15, 67, 646),
    (0, 4): (141, 100, 35, 45, 508), (1, 2): (415, 389, 5, 232, 52), (1, 3): (281, 257, 7, 139, 86),
    (1, 4): (158, 132, 14, 65, 182), (2, 3): (476, 432, 7, 214, 127), (2, 4): (286, 238, 14, 99, 614),
    (2, 5): (195, 148, 24, 54, 898), (3, 4): (511, 457, 8, 227, 133), (3, 5): (332, 279, 13, 101, 780),
    (4, 5): (525, 489, 6, 203, 234), (6, 7): (421, 377, 8, 297, 47), (6, 8): (332, 302, 7, 163, 89),
    (6, 9): (220, 188, 18, 94, 162), (7, 8): (438, 411, 5, 303, 20), (7, 9): (320, 285, 8, 176, 55),
    (8, 9): (452, 430, 5, 264, 43),
}


def test_cfg1_oracle_chain_on_the_temple_fixture(orc):
    g = np.load(os.path.join(HERE, "golden", "temple_sift.npz"))
    assert [str(n) for n in g["names"]] == ["temple%04d.png" % i for i in range(1, 11)]
    desc = [g[f"desc{i}"].astype(np.float32) for i in range(10)]
    pts = [g[f"kp{i}"][:, :2].astype(np.float64) for i in range(10)]
    assert [len(d) for d in desc] == [808, 801, 845, 914, 918, 971, 776, 747, 798, 850]
    scored, pair_points = {}, []
    for q in range(9):
        for t in range(q + 1, 10):
            rq, rt, _ = orc.match_knn2(desc[q], desc[t])
            a, b = pts[q][rq], pts[t][rt]
            pair_points.append(((q, t), a, b))
            if len(rq) < 120:
                continue
            cnt, mask, E, it = SC.find_essential_mat_ransac(a, b, K)
            hc, hmask, hit = SC.find_homography_ransac(a, b, 0.004 * float(a.max()))
            assert mask.sum() == cnt and hmask.sum() == hc
            scored[(q, t)] = (len(rq), cnt, it, hc, hit)
    assert scored == PINNED
    best = SC.find_best_pair_scores(pair_points, K)
    assert [v for _, v in best][0] == (0, 4) and [v for _, v in best][-1] == (8, 9)     # ascending: worst ratio first
    assert len(best) == len({np.float32(np.float32(c) / np.float32(n)) for n, c, *_ in PINNED.values()})
