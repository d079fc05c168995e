"""CPU: the C oracle's matcher against the golden vectors and the numpy brute force."""
import numpy as np
import pytest

from oracle import np_check as nc
from sfm_danpipeline_amd import synth


def _check_case(orc, q, t, idx, dist, mq, mt, md, norm):
    r = orc.match_knn2(q, t, norm=norm, want_knn=True)
    assert np.array_equal(r[3], idx)
    assert np.array_equal(r[4].view(np.uint32), dist.view(np.uint32))
    assert np.array_equal(r[0], mq) and np.array_equal(r[1], mt)
    assert np.array_equal(r[2].view(np.uint32), md.view(np.uint32))


def test_golden_l2_u8_and_f32(orc, golden):
    g = golden["match_l2"]
    for name in g["names"]:
        q, t = g[f"{name}_q"], g[f"{name}_t"]
        args = (g[f"{name}_idx"], g[f"{name}_dist"], g[f"{name}_mq"], g[f"{name}_mt"], g[f"{name}_md"])
        _check_case(orc, q, t, *args, orc.NORM_L2)                                        # CV_8U rows
        _check_case(orc, q.astype(np.float32), t.astype(np.float32), *args, orc.NORM_L2)  # CV_32F rows (SIFT)


def test_golden_hamming(orc, golden):
    g = golden["match_hamming"]
    for name in g["names"]:
        q, t = g[f"{name}_q"], g[f"{name}_t"]
        _check_case(orc, q, t, g[f"{name}_idx"], g[f"{name}_dist"], g[f"{name}_mq"], g[f"{name}_mt"], g[f"{name}_md"],
                    orc.NORM_HAMMING)
        r = orc.match_knn2(q, t, norm=orc.NORM_L2, want_knn=True)  # reference-literal L2 on binary rows
        assert np.array_equal(r[3], g[f"{name}_l2idx"])
        assert np.array_equal(r[4].view(np.uint32), g[f"{name}_l2dist"].view(np.uint32))


def test_sqrt_collision_prefers_lower_index(orc, golden):
    g = golden["match_l2"]
    assert g["sqrt_collision_dist"][0, 0] == g["sqrt_collision_dist"][0, 1]
    r = orc.match_knn2(g["sqrt_collision_q"], g["sqrt_collision_t"], want_knn=True)
    assert r[3].tolist() == [[1, 2]]


def test_ratio_boundary_is_inclusive(orc, golden):
    g = golden["match_l2"]
    r = orc.match_knn2(g["ratio_boundary_q"], g["ratio_boundary_t"])
    assert r[0].tolist() == [0] and r[1].tolist() == [1] and r[2].tolist() == [4.0]


@pytest.mark.parametrize("nt", [0, 1])
def test_too_few_train_rows_emit_nothing(orc, nt):
    q = np.zeros((4, 128), np.float32)
    t = np.ones((nt, 128), np.float32)
    r = orc.match_knn2(q, t)
    assert len(r[0]) == 0


def test_random_vs_bruteforce_and_threads(orc):
    imgs = synth.sift_image_set(2, 150, 128, bank=200, seed=3)
    idx, dist = nc.knn2_bruteforce(imgs[0], imgs[1])
    for threads in (1, 4):
        r = orc.match_knn2(imgs[0], imgs[1], want_knn=True, threads=threads)
        assert np.array_equal(r[3], idx) and np.array_equal(r[4], dist)
    assert np.all(np.diff(r[0]) > 0)  # ascending queryIdx


def test_non_integer_rows_keep_documented_order(orc):
    rng = np.random.default_rng(0)
    q = rng.random((20, 128), dtype=np.float32) * 255
    t = rng.random((30, 128), dtype=np.float32) * 255
    r = orc.match_knn2(q, t, want_knn=True)
    ref = np.sqrt(((q[:, None, :].astype(np.float64) - t[None, :, :]) ** 2).sum(-1))
    assert np.array_equal(r[3][:, 0], ref.argmin(1))
    assert np.allclose(r[4][:, 0], ref.min(1), rtol=1e-6)


def test_blocked_baseline_matcher_gives_the_same_lists(orc):
    """bench.py's CPU-baseline matcher (query blocks x train tiles, SIMD, per-thread merge) against the row-by-row
    restatement: counts and per-pair checksums, ragged sizes around the block / tile / SIMD-group edges"""
    imgs = synth.sift_image_set(5, 150, 128, bank=200, seed=7)
    imgs = [imgs[0], imgs[1][:129], imgs[2][:65], imgs[3][:2], imgs[4][:1]]
    pairs = np.array([[a, b] for a in range(5) for b in range(5) if a != b], np.int32)
    c0, cs0 = orc.match_many_checksum(imgs, pairs, threads=1)
    for threads in (1, 3):
        c1, cs1 = orc.match_many_blocked(imgs, pairs, threads=threads)
        assert np.array_equal(c0, c1) and np.array_equal(cs0, cs1)
    assert c0.sum() > 50


def test_product_side_checksum_helper_equals_the_checkers(orc):
    """sfm_danpipeline_amd.synth.pair_checksums (what bench.py's cfg5 leg prints) against orc_match_mix in C"""
    imgs = synth.sift_image_set(3, 120, 128, bank=150, seed=9)
    pairs = np.array([[0, 1], [1, 2], [0, 2]], np.int32)
    cnt, cs = orc.match_many_checksum(imgs, pairs)
    lists = [orc.match_knn2(imgs[a], imgs[b]) for a, b in pairs]
    q, t, d = (np.concatenate([l[k] for l in lists]) for k in range(3))
    assert np.array_equal(synth.pair_checksums(cnt, q, t, d), cs)
    assert np.array_equal(orc.pair_checksums(cnt, q, t, d), cs)
