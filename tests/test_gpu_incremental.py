"""GPU: find2D3DMatches association and mergeNewPoints through the C ABI, bit-exact against the
oracle."""
import numpy as np
import pytest

from sfm_danpipeline_amd import incremental, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_cloud,n_matches,seed,done,new", [(1, 1, 1, 0, 1), (300, 120, 2, 2, 5), (300, 120, 3, 5, 2),
                                                             (5000, 390, 4, 1, 7), (100000, 400, 5, 6, 0)])
def test_find_2d3d_vs_oracle(ctx, orc, n_cloud, n_matches, seed, done, new):
    cloud, matches = synth.random_tracks_and_matches(n_cloud, 8, n_matches, seed=seed, done_view=done)
    ptr, views, feats = synth.tracks_to_csr(cloud)
    mq, mt = [m[0] for m in matches], [m[1] for m in matches]
    oc, of = incremental.find_2d3d(ptr, views, feats, done, new, mq, mt, ctx=ctx)
    rc, rf = orc.find_2d3d(ptr, views, feats, done, new, mq, mt)
    assert np.array_equal(oc, rc) and np.array_equal(of, rf)          # bit-exact, cloud order


def test_find_2d3d_edges(ctx, orc):
    cloud = [{0: 7, 1: 3}, {1: 3}, {0: 9}, {1: 4, 2: 8}]
    ptr, views, feats = synth.tracks_to_csr(cloud)
    oc, of = incremental.find_2d3d(ptr, views, feats, 1, 0, [5, 6, 2], [3, 3, 4], ctx=ctx)
    assert list(oc) == [0, 1, 3] and list(of) == [5, 5, 2]            # first match on a repeated trainIdx wins
    oc, of = incremental.find_2d3d(ptr, views, feats, 1, 0, [], [], ctx=ctx)
    assert len(oc) == 0
    oc, of = incremental.find_2d3d([0], [], [], 1, 0, [1], [2], ctx=ctx)
    assert len(oc) == 0
    oc, of = incremental.find_2d3d(ptr, views, feats, 5, 0, [5, 6, 2], [3, 3, 4], ctx=ctx)   # nobody saw view 5
    assert len(oc) == 0


def test_find_2d3d_matches_builds_the_reference_vectors(ctx, orc):
    sc = synth.two_view_scene(50, seed=3)
    cloud = [dict(pt=tuple(np.array([i, 2 * i, 3 * i], float)), idxImage={0: i, 1: 49 - i}, pt2D={}) for i in range(50)]
    q = np.arange(0, 50, 2)
    t = 49 - q
    p3, p2 = incremental.find_2d3d_matches(cloud, 1, 0, q, t, sc["xy2"], ctx=ctx)
    assert np.array_equal(p3[:, 0], q.astype(float)) and np.array_equal(p2, sc["xy2"][t])


@pytest.mark.parametrize("n_cloud,n_new,seed", [(0, 5, 1), (1, 1, 2), (255, 257, 3), (4000, 3000, 4), (50000, 2000, 5)])
def test_merge_vs_oracle(ctx, orc, n_cloud, n_new, seed):
    rng = np.random.default_rng(seed)
    cloud = rng.uniform(-1, 1, (n_cloud, 3))
    k = min(n_cloud, n_new // 3)
    near = cloud[rng.choice(n_cloud, k, replace=False)] + rng.normal(0, 0.004, (k, 3)) if k else np.zeros((0, 3))
    dup = rng.uniform(2, 3, (max(n_new // 6, 1), 3))
    new = np.concatenate([near, dup, dup + rng.normal(0, 0.004, dup.shape), rng.uniform(-1, 1, (max(n_new - 2 * len(dup) - k, 0), 3))])
    rng.shuffle(new)
    acc, n = incremental.merge_accept(cloud, new, ctx=ctx)
    racc, rn = orc.merge_new_points(cloud, new)
    assert np.array_equal(acc, racc) and n == rn
    assert 0 < n < len(new) or len(new) <= 1


def test_merge_chain_and_threshold(ctx, orc):
    chain = np.stack([np.arange(40) * 0.006 + 3.0, np.zeros(40), np.zeros(40)], 1)
    acc, n = incremental.merge_accept(np.zeros((1, 3)), chain, ctx=ctx)
    assert np.array_equal(acc, orc.merge_new_points(np.zeros((1, 3)), chain)[0])
    assert list(acc) == [True, False] * 20                           # 40-deep in-order dependency
    r = np.float64(np.float32(0.01))
    pts = np.array([[r, 0, 0], [0, np.nextafter(r, 0), 0], [0, 0, 0.01]])
    acc, _ = incremental.merge_accept(np.zeros((1, 3)), pts, ctx=ctx)
    assert list(acc) == [True, False, True]                           # < (double)(float)0.01, like cv::norm(..) < 0.01f
    acc, n = incremental.merge_accept(np.zeros((3, 3)), np.zeros((0, 3)), ctx=ctx)
    assert n == 0


def test_merge_new_points_appends_in_order(ctx):
    cloud = [dict(pt=(0.0, 0.0, 0.0), idxImage={0: 1, 1: 2}, pt2D={})]
    new = [dict(pt=(0.0, 0.0, 0.005), idxImage={0: 3, 2: 4}, pt2D={}), dict(pt=(1.0, 0.0, 0.0), idxImage={1: 5, 2: 6}, pt2D={}),
           dict(pt=(1.0, 0.0, 0.002), idxImage={1: 7, 2: 8}, pt2D={})]
    assert incremental.merge_new_points(cloud, new, ctx=ctx) == 1
    assert len(cloud) == 2 and cloud[1]["idxImage"] == {1: 5, 2: 6}   # tracks are never merged (src/Sfm.cpp:1225)
