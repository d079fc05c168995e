"""CPU property tests (hypothesis) of the oracle's semantics that the GPU parity tests lean on:
the k-NN tie rule, the ratio boundary, and the two incremental-glue operators."""
import numpy as np
from hypothesis import given, settings, strategies as st

from tests.test_oracle_incremental import py_find_2d3d, py_merge
from sfm_danpipeline_amd import synth


def brute_knn2(q, t):
    """cv::batchDistance's insertion rule, literally: strict < on entry,
    strict > while bubbling, float sqrt of the exact integer squared distance."""
    out = []
    for i in range(len(q)):
        d = [np.float32(3.402823466e38)] * 2
        idx = [-1, -1]
        for j in range(len(t)):
            diff = q[i].astype(np.int64) - t[j].astype(np.int64)
            b = np.sqrt(np.float32(int((diff * diff).sum())))
            if b < d[1]:
                k = 0
                while k >= 0 and d[k] > b:
                    d[k + 1], idx[k + 1] = d[k], idx[k]
                    k -= 1
                d[k + 1], idx[k + 1] = b, j
        out.append((idx[0], idx[1], d[0], d[1]))
    return out


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 12), st.integers(0, 9), st.integers(0, 2**31 - 1), st.integers(1, 3))
def test_knn2_tie_rule_on_low_entropy_rows(orc, nq, nt, seed, levels):
    """Rows drawn from very few distinct values: many exact distance ties, which must go to the
    lower train index, and a later equal candidate must not displace the 2nd best."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, levels + 1, (nq, 8)).astype(np.float32) * 40
    t = rng.integers(0, levels + 1, (nt, 8)).astype(np.float32) * 40
    mq, mt, md, kidx, kd = orc.match_knn2(q, t, want_knn=True)
    ref = brute_knn2(q, t)
    for i, (j0, j1, d0, d1) in enumerate(ref):
        assert (kidx[i, 0], kidx[i, 1]) == (j0, j1)
        if j0 >= 0:
            assert kd[i, 0] == d0
        if j1 >= 0:
            assert kd[i, 1] == d1
    keep = [i for i, (j0, j1, d0, d1) in enumerate(ref) if nt >= 2 and d0 <= np.float32(0.8) * d1]
    assert list(mq) == keep and list(mt) == [ref[i][0] for i in keep]


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 40), st.integers(0, 30), st.integers(0, 2**31 - 1), st.integers(0, 5), st.integers(0, 5))
def test_find_2d3d_property(orc, n_cloud, n_matches, seed, done, new):
    cloud, matches = synth.random_tracks_and_matches(n_cloud, 6, n_matches, seed=seed, done_view=done, n_feat=24)
    ptr, views, feats = synth.tracks_to_csr(cloud)
    oc, of = orc.find_2d3d(ptr, views, feats, done, new, [m[0] for m in matches], [m[1] for m in matches])
    assert [(int(a), int(b)) for a, b in zip(oc, of)] == py_find_2d3d(cloud, done, new, matches)


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 30), st.integers(0, 30), st.integers(0, 2**31 - 1))
def test_merge_property(orc, n_cloud, n_new, seed):
    rng = np.random.default_rng(seed)
    grid = lambda n: rng.integers(0, 6, (n, 3)) * 0.004          # many points within / at / beyond 0.01
    cloud, new = grid(n_cloud), grid(n_new)
    acc, n = orc.merge_new_points(cloud, new)
    ref = py_merge(cloud, new) if n_new else np.zeros(0, bool)
    assert np.array_equal(acc, ref) and n == int(ref.sum())
