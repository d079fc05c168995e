"""GPU: the HIP matcher through the C ABI against the oracle, the golden vectors, and -- at the
BASELINE cfg2 size -- size-independent properties."""
import numpy as np
import pytest

from sfm_danpipeline_amd import _lib, matcher, synth

pytestmark = pytest.mark.gpu


def _plan(ctx, imgs, pairs, norm=_lib.L2):
    s = matcher.ImageSet(imgs, norm=norm, ctx=ctx)
    s.prepare_async()
    pl = matcher.MatchPlan(s, pairs)
    pl.run_async(0.8)
    return s, pl


def _assert_pair(orc, pl, p, q, t, norm):
    ki, kd = pl.fetch_knn(p)
    r = orc.match_knn2(q, t, norm=(orc.NORM_HAMMING if norm == _lib.HAMMING else orc.NORM_L2), want_knn=True, threads=8)
    assert np.array_equal(ki, r[3])
    assert np.array_equal(kd.view(np.uint32), r[4].view(np.uint32))
    a = pl.fetch_pair(p)
    assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])
    assert np.array_equal(a[2].view(np.uint32), r[2].view(np.uint32))


def test_golden_l2(ctx, golden):
    g = golden["match_l2"]
    for name in g["names"]:
        for cast in (np.uint8, np.float32):
            q, t = g[f"{name}_q"].astype(cast), g[f"{name}_t"].astype(cast)
            mq, mt, md = matcher.get_matching(q, t, ctx=ctx)          # the getMatching drop-in
            assert np.array_equal(mq, g[f"{name}_mq"]) and np.array_equal(mt, g[f"{name}_mt"]), name
            assert np.array_equal(md.view(np.uint32), g[f"{name}_md"].view(np.uint32)), name
            if q.shape[0] and t.shape[0]:
                s, pl = _plan(ctx, [q, t], [[0, 1]])
                ki, kd = pl.fetch_knn(0)
                assert np.array_equal(ki, g[f"{name}_idx"]), name
                assert np.array_equal(kd.view(np.uint32), g[f"{name}_dist"].view(np.uint32)), name


def test_golden_hamming_and_reference_literal_l2(ctx, golden):
    g = golden["match_hamming"]
    for name in g["names"]:
        q, t = g[f"{name}_q"], g[f"{name}_t"]
        mq, mt, md = matcher.get_matching(q, t, norm=_lib.HAMMING, ctx=ctx)
        assert np.array_equal(mq, g[f"{name}_mq"]) and np.array_equal(mt, g[f"{name}_mt"])
        assert np.array_equal(md, g[f"{name}_md"])
        s, pl = _plan(ctx, [q, t], [[0, 1]], _lib.L2)  # cv::NORM_L2 on CV_8U rows, src/Sfm.cpp:593
        ki, kd = pl.fetch_knn(0)
        assert np.array_equal(ki, g[f"{name}_l2idx"])
        assert np.array_equal(kd.view(np.uint32), g[f"{name}_l2dist"].view(np.uint32))


@pytest.mark.parametrize("nq,nt", [(1, 2), (33, 257), (700, 700), (300, 2049), (64, 32)])
def test_sift_shapes_vs_oracle(ctx, orc, nq, nt):
    imgs = synth.sift_image_set(2, max(nq, nt), 128, bank=max(nq, nt) + 200, seed=nq * 7 + nt)
    q, t = imgs[0][:nq], imgs[1][:nt]
    s, pl = _plan(ctx, [q, t], [[0, 1]])
    _assert_pair(orc, pl, 0, q, t, _lib.L2)


def test_extreme_values_take_the_exact_fixup_path(ctx, orc):
    rng = np.random.default_rng(3)
    q = (rng.integers(0, 2, (200, 128)) * 255).astype(np.float32)
    t = (rng.integers(0, 2, (300, 128)) * 255).astype(np.float32)   # s ~ 4.2M: sqrtf collisions
    s, pl = _plan(ctx, [q, t], [[0, 1]])
    _assert_pair(orc, pl, 0, q, t, _lib.L2)


def test_more_flagged_queries_than_the_fixup_list_holds(ctx, orc):
    """Every query lands in the sqrtf-merge range (2nd-best squared distance >= 2^22): 600 pairs x 2000
    queries = 1.2 M flagged, more than the 2^20 entries of the fix-up list.  Nothing may be dropped:
    what does not fit the list stays flagged in-band and is redone by the compaction kernel."""
    import os
    rng = np.random.default_rng(77)
    qs = [(rng.random((2000, 128)) < 0.1).astype(np.float32) * 255 for _ in range(30)]     # mostly 0
    ts = [(rng.random((256, 128)) < 0.9).astype(np.float32) * 255 for _ in range(20)]      # mostly 255
    imgs = qs + ts
    pairs = np.array([[a, 30 + b] for a in range(30) for b in range(20)], np.int32)
    s, pl = _plan(ctx, imgs, pairs)
    cnt, oq, ot, od = pl.fetch()
    for p in (0, 313, 599):
        ki, kd = pl.fetch_knn(p)
        assert np.all(kd[:, 1] >= 2048.0)                  # all of them in the merge range
        _assert_pair(orc, pl, p, imgs[pairs[p, 0]], imgs[pairs[p, 1]], _lib.L2)
    ocnt, ocs = orc.match_many_checksum(imgs, pairs, threads=os.cpu_count() or 8)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(orc.pair_checksums(cnt, oq, ot, od), ocs)
    # a k-NN checksum too (the ratio test passes few of these): every pair's raw lists on sampled pairs
    for p in range(7, 600, 37):
        ki, kd = pl.fetch_knn(p)
        r = orc.match_knn2(imgs[pairs[p, 0]], imgs[pairs[p, 1]], want_knn=True, threads=8)
        assert np.array_equal(ki, r[3]) and np.array_equal(kd.view(np.uint32), r[4].view(np.uint32))


def test_non_integer_rows_use_the_exact_kernel(ctx, orc):
    imgs = synth.sift_image_set(3, 200, 128, bank=300, seed=5)
    mixed = [imgs[0] + 0.25, imgs[1], imgs[2]]        # one non-integer image poisons only its pairs
    s, pl = _plan(ctx, mixed, [[0, 1], [1, 2], [2, 0]])
    for p, (a, b) in enumerate([(0, 1), (1, 2), (2, 0)]):
        _assert_pair(orc, pl, p, mixed[a], mixed[b], _lib.L2)


@pytest.mark.parametrize("value", [300.0, -1.0, 256.0, 1e6])
def test_integer_valued_rows_outside_the_sift_range_use_the_exact_kernel(ctx, orc, value):
    """f32 rows that ARE integers but do not fit the centred-i8 operand (the range half of the device's guard): their
    pairs go through the literal-f32 kernel and equal the oracle bit for bit"""
    imgs = synth.sift_image_set(3, 200, 128, bank=300, seed=6)
    bad = imgs[1].copy()
    bad[17, 5] = value                                 # a single element out of 0..255
    bad[150, :] = value if abs(value) < 1e5 else 255.0
    mixed = [imgs[0], bad, imgs[2]]
    s, pl = _plan(ctx, mixed, [[0, 1], [1, 2], [2, 0], [1, 0]])
    for p, (a, b) in enumerate([(0, 1), (1, 2), (2, 0), (1, 0)]):
        _assert_pair(orc, pl, p, mixed[a], mixed[b], _lib.L2)


@pytest.mark.parametrize("dim", [32, 61, 64, 96, 200, 300])
def test_other_descriptor_widths(ctx, orc, dim):
    rng = np.random.default_rng(dim)
    q = rng.integers(0, 256, (150, dim), dtype=np.uint8)
    t = rng.integers(0, 256, (170, dim), dtype=np.uint8)
    s, pl = _plan(ctx, [q, t], [[0, 1]])
    _assert_pair(orc, pl, 0, q, t, _lib.L2)


def test_hamming_vs_oracle_and_ragged_set(ctx, orc):
    orbs = synth.orb_image_set(4, 900, bank=1200, seed=8)
    ragged = [orbs[0], orbs[1][:513], orbs[2][:2], orbs[3][:0]]
    pairs = [[0, 1], [1, 0], [0, 2], [2, 1], [0, 3], [3, 0]]
    s, pl = _plan(ctx, ragged, pairs, _lib.HAMMING)
    for p, (a, b) in enumerate(pairs):
        if len(ragged[a]) and len(ragged[b]):
            _assert_pair(orc, pl, p, ragged[a], ragged[b], _lib.HAMMING)
    assert pl.counts()[4] == 0 and pl.counts()[5] == 0


def test_batched_plan_equals_single_pair_calls(ctx):
    imgs = synth.sift_image_set(6, 400, 128, bank=600, seed=9)
    pairs = synth.all_pairs(6)
    s, pl = _plan(ctx, imgs, pairs)
    cnt, oq, ot, od = pl.fetch()
    off = 0
    for p, (a, b) in enumerate(pairs):
        mq, mt, md = matcher.get_matching(imgs[a], imgs[b], ctx=ctx)
        n = cnt[p]
        assert n == len(mq) and np.array_equal(oq[off:off + n], mq) and np.array_equal(ot[off:off + n], mt)
        assert np.array_equal(od[off:off + n], md)
        off += n


def test_cfg2_full_size_properties(ctx, orc):
    """BASELINE cfg2 (50 x 2000 x SIFT-128, 1225 pairs): determinism, ordering, planted-match
    recovery, and three pairs bit-checked against the oracle."""
    imgs = synth.sift_image_set()
    pairs = synth.all_pairs(len(imgs))
    s, pl = _plan(ctx, imgs, pairs)
    cnt, oq, ot, od = pl.fetch()
    s.prepare_async()
    pl.run_async(0.8)
    cnt2, oq2, ot2, od2 = pl.fetch()
    assert np.array_equal(cnt, cnt2) and np.array_equal(oq, oq2) and np.array_equal(ot, ot2) and np.array_equal(od, od2)
    off = np.concatenate([[0], np.cumsum(cnt)])
    for p in range(len(pairs)):
        qs = oq[off[p]:off[p + 1]]
        assert np.all(np.diff(qs) > 0)                     # ascending queryIdx inside a pair
        assert len(np.unique(ot[off[p]:off[p + 1]])) >= 0.9 * cnt[p]
    assert 150 < cnt.mean() < 260                          # ~10 % of 2000 rows are shared per pair
    assert od.max() < 400                                  # true matches sit at ~sigma*sqrt(2*128)
    for p in (0, 611, 1224):
        _assert_pair(orc, pl, p, imgs[pairs[p, 0]], imgs[pairs[p, 1]], _lib.L2)
    # ALL 1225 pairs against the oracle: counts and a per-pair checksum of (queryIdx, trainIdx, distance bits)
    import os
    ocnt, ocs = orc.match_many_checksum(imgs, pairs, threads=os.cpu_count() or 8)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(orc.pair_checksums(cnt, oq, ot, od), ocs)


def test_train_permutation_property(ctx):
    """Permuting train rows permutes trainIdx and nothing else (no ties in this data)."""
    imgs = synth.sift_image_set(2, 1500, 128, bank=2500, seed=12)
    perm = np.random.default_rng(0).permutation(1500)
    a = matcher.get_matching(imgs[0], imgs[1], ctx=ctx)
    b = matcher.get_matching(imgs[0], imgs[1][perm], ctx=ctx)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], perm[b[1]]) and np.array_equal(a[2], b[2])


def test_cfg5_orb_shard_properties(ctx, orc):
    """BASELINE cfg5 (500 x 5000 x ORB-256 bit, Hamming): one rank's share of the 124 750 pairs under
    the 8-way pair sharding -- determinism, ordering, planted-match recovery, and three pairs
    bit-checked against the oracle."""
    import time
    from sfm_danpipeline_amd import sharding
    imgs = synth.orb_image_set()
    pairs = synth.all_pairs(len(imgs))
    shards = sharding.shard_pairs(pairs, [len(a) for a in imgs], 8)
    mine = pairs[shards[0]]
    assert 124750 // 8 - 5 <= len(mine) <= 124750 // 8 + 5
    t0 = time.time()
    s, pl = _plan(ctx, imgs, mine, _lib.HAMMING)
    cnt, oq, ot, od = pl.fetch()
    dt = time.time() - t0
    s.prepare_async()
    pl.run_async(0.8)
    cnt2, oq2, ot2, od2 = pl.fetch()
    assert np.array_equal(cnt, cnt2) and np.array_equal(oq, oq2) and np.array_equal(ot, ot2) and np.array_equal(od, od2)
    off = np.concatenate([[0], np.cumsum(cnt)])
    for p in range(0, len(mine), 97):
        assert np.all(np.diff(oq[off[p]:off[p + 1]]) > 0)  # ascending queryIdx inside a pair
    # two images share ~5000*5000/100000 = 250 bank rows; their copies differ in ~2*0.04*256 = 20 bits,
    # impostors in ~128: the ratio test keeps the shared rows
    assert 150 < cnt.mean() < 350 and od.max() < 110 and np.all(od == np.round(od))
    for p in (0, len(mine) // 2, len(mine) - 1):
        _assert_pair(orc, pl, p, imgs[mine[p, 0]], imgs[mine[p, 1]], _lib.HAMMING)
    # 32 sampled pairs of the shard against the oracle by per-pair checksum (counts + every match)
    import os
    sample = np.random.default_rng(11).choice(len(mine), 32, replace=False)
    ocnt, ocs = orc.match_many_checksum(imgs, mine[sample], norm=orc.NORM_HAMMING, threads=os.cpu_count() or 8)
    gcs = orc.pair_checksums(cnt, oq, ot, od)
    assert np.array_equal(cnt[sample], ocnt) and np.array_equal(gcs[sample], ocs)
    print(f"[cfg5 shard] {len(mine)} pairs incl. upload + first run: {dt:.2f} s")


def test_cfg5_every_pair_against_the_oracle(ctx, golden):
    """BASELINE cfg5 at full size, ALL 124 750 pairs of the 500 x 5000 x ORB-256 set under Hamming k-NN-2 + ratio, against
    the C restatement's answer for every single pair: match count and [sum, xor] of the match mix (queryIdx, trainIdx,
    distance bits).  The oracle's side was computed once in the build container (tests/golden/make_cfg5_checksums.py,
    half an hour on 8 cores) and is committed as data; the device's side is one sweep."""
    gold = golden["cfg5_checksums"]
    imgs = synth.orb_image_set()
    pairs = synth.all_pairs(len(imgs))
    assert len(pairs) == 124750 == len(gold["counts"])
    s, pl = _plan(ctx, imgs, pairs, _lib.HAMMING)
    cnt, oq, ot, od = pl.fetch()
    cs = synth.pair_checksums(cnt, oq, ot, od)
    assert np.array_equal(cnt, gold["counts"])
    bad = np.nonzero(np.any(cs != gold["checksums"], axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} pairs differ, first {bad[:5]}"
    assert int(cnt.sum()) == int(gold["counts"].sum())


@pytest.mark.parametrize("nq,nt,levels,seed", [(300, 700, 1, 1), (65, 513, 2, 2), (1000, 1031, 3, 3)])
def test_heavy_ties_across_tiles_and_chunks(ctx, orc, nq, nt, levels, seed):
    """Rows from very few distinct values: thousands of exact distance ties per query, spread over
    32-row tiles, the two half-waves and the key chunks.  The k-NN lists must follow
    cv::batchDistance's rule (lower train index first) bit for bit."""
    rng = np.random.default_rng(seed)
    q = (rng.integers(0, levels + 1, (nq, 128)) * 60).astype(np.float32)
    t = (rng.integers(0, levels + 1, (nt, 128)) * 60).astype(np.float32)
    t[nt // 2:] = t[: nt - nt // 2]                       # every row has a twin 300+ rows later
    s, pl = _plan(ctx, [q, t], [[0, 1]])
    _assert_pair(orc, pl, 0, q, t, _lib.L2)
    orb = rng.integers(0, 2, (nt, 32)).astype(np.uint8) * 255   # bytes 0x00 / 0xFF only
    qb = rng.integers(0, 2, (nq, 32)).astype(np.uint8) * 255
    s2, pl2 = _plan(ctx, [qb, orb], [[0, 1]], _lib.HAMMING)
    _assert_pair(orc, pl2, 0, qb, orb, _lib.HAMMING)


@pytest.mark.parametrize("nq,nt,seed", [(200, 1024, 21), (97, 1025, 22), (300, 2100, 23), (64, 3073, 24), (129, 4096, 25)])
def test_hamming_ties_across_the_key_chunks(ctx, orc, nq, nt, seed):
    """The Hamming key holds ten index bits since round 6 (1024 train rows per chunk; the lane's two keys are folded into a
    running (distance, row) pair at every chunk's end and behind the last, partial one): descriptors of 0x00 / 0xFF bytes tie by
    the thousand, and every row has twins one row, one tile, one chunk and several chunks away -- the lists must still follow
    cv::batchDistance's rule (lower train index first) bit for bit, at sizes on, one past and well past a chunk's end."""
    rng = np.random.default_rng(seed)
    orb = rng.integers(0, 2, (nt, 32)).astype(np.uint8) * 255
    for off in (1, 32, 1023, 1024, 1025, 2048):
        if off < nt:
            src = rng.integers(0, nt - off, max(nt // 8, 1))
            orb[src + off] = orb[src]
    qb = orb[rng.integers(0, nt, nq)].copy()
    flip = rng.integers(0, 32, nq)
    qb[np.arange(nq), flip] ^= rng.choice(np.array([0, 1, 3, 255], np.uint8), nq)   # some queries exact copies, some a few bits off
    s2, pl2 = _plan(ctx, [qb, orb], [[0, 1]], _lib.HAMMING)
    _assert_pair(orc, pl2, 0, qb, orb, _lib.HAMMING)


@pytest.mark.parametrize("seed", range(12))
def test_ties_at_the_second_best_value_inside_a_lane(ctx, orc, seed):
    """The resolve of round 6 reads one or two candidate rows where it read four: where the second-best value of a lane is both its
    slot's and its tile's maximum, the rows (first tile, other slot), (second tile, either slot) are told apart by position, and one
    of them is entered with a value that was never read.  Here the second-best distance of most queries is held by SEVERAL train
    rows -- exact copies placed one row, one tile (32 rows) and several tiles away, before and behind the best row -- and the best
    distance by copies too: every list must still be cv::batchDistance's (lower train index first), bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    nt = int(rng.integers(96, 900))
    nq = int(rng.integers(40, 260))
    centres = rng.integers(0, 256, (max(nt // 6, 4), 128))
    t = centres[rng.integers(0, len(centres), nt)].copy()
    # a few dimensions differ by small steps inside a cluster: distances tie often, but not always
    for _ in range(3):
        d_ = rng.integers(0, 128, nt)
        t[np.arange(nt), d_] = np.clip(t[np.arange(nt), d_] + rng.integers(-2, 3, nt), 0, 255)
    for off in (1, 2, 31, 32, 33, 64, 97, 160):          # exact copies at lane / tile distances
        src = rng.integers(0, nt, nt // 10)
        dst = np.clip(src + rng.choice([-off, off], len(src)), 0, nt - 1)
        t[dst] = t[src]
    q = t[rng.integers(0, nt, nq)].copy()
    dq = rng.integers(0, 128, nq)
    q[np.arange(nq), dq] = np.clip(q[np.arange(nq), dq] + rng.integers(-1, 2, nq), 0, 255)
    for cast in (np.float32, np.uint8):
        s, pl = _plan(ctx, [q.astype(cast), t.astype(cast)], [[0, 1]])
        _assert_pair(orc, pl, 0, q.astype(cast), t.astype(cast), _lib.L2)


@pytest.mark.parametrize("nq,nt,dup", [(300, 8193, False), (200, 9000, True), (130, 17000, True)])
def test_train_images_longer_than_an_epoch(ctx, orc, nq, nt, dup):
    """Train images beyond 8192 rows are swept in epochs of 256 tiles (the tile tag has 8 bits), each resolved on its own and merged
    through the k-NN buffer: the best row of one epoch and the second-best of another, ties across the epoch boundary (the earlier
    epoch's row has the lower index and stays), queries flagged in one epoch only."""
    imgs = synth.sift_image_set(2, max(nq, nt), 128, bank=max(nq, nt) + 500, seed=nq + nt)
    q, t = imgs[0][:nq].copy(), imgs[1][:nt].copy()
    if dup:
        rng = np.random.default_rng(nt)
        src = rng.integers(0, 8000, 400)
        t[8192 + rng.integers(0, nt - 8192, 400)] = t[src]                                                    # copies in a later epoch
        q[: nq // 2] = t[rng.integers(0, nt, nq // 2)]                                                        # exact hits, some with copies
    s, pl = _plan(ctx, [q, t], [[0, 1]])
    _assert_pair(orc, pl, 0, q, t, _lib.L2)


def test_one_pair_plan_retargeted_over_a_resident_set(ctx):
    """sfmhip_matchplan_set_pairs: a one-pair plan pointed at each pair in turn (the getMatching
    drop-in over descriptors resident in HBM) returns what the batched all-pairs plan returns."""
    imgs = synth.sift_image_set(5, 700, 128, bank=1000, seed=21)
    imgs[3] = imgs[3][:130]                               # ragged
    pairs = np.array([[a, b] for a in range(5) for b in range(5) if a != b], np.int32)
    s, pl = _plan(ctx, imgs, pairs)
    cnt, oq, ot, od = pl.fetch()
    off = np.concatenate([[0], np.cumsum(cnt)])
    one = matcher.MatchPlan(s, pairs[:1])
    for p in np.random.default_rng(0).permutation(len(pairs)):
        one.set_pairs(pairs[p:p + 1])
        one.run_async(0.8)
        c1, q1, t1, d1 = one.fetch()
        assert c1[0] == cnt[p]
        assert np.array_equal(q1, oq[off[p]:off[p + 1]]) and np.array_equal(t1, ot[off[p]:off[p + 1]])
        assert np.array_equal(d1, od[off[p]:off[p + 1]])
    with pytest.raises(Exception):
        one.set_pairs(pairs[:2])                          # more pairs than the plan was created for


def test_stage_timing_is_opt_in(ctx):
    rng = np.random.default_rng(5)
    descs = [rng.integers(0, 256, size=(300, 128)).astype(np.float32) for _ in range(3)]
    iset = matcher.ImageSet(descs, ctx=ctx)
    plan = matcher.MatchPlan(iset, [(0, 1), (0, 2), (1, 2)])
    try:
        ctx.set_timing(False)
        iset.prepare_async()
        plan.run_async(0.8)
        ctx.synchronize()
        assert all(v == 0 for v in plan.last_timing().values())
        ctx.set_timing(True)
        iset.prepare_async()
        plan.run_async(0.8)
        tm = plan.last_timing()
        assert tm["knn_s"] > 0 and tm["prepare_s"] > 0 and tm["compact_s"] > 0
    finally:
        ctx.set_timing(False)
        plan.close()
        iset.close()


def test_pipelined_fetch_hands_out_the_same_lists(ctx, orc):
    """sfmhip_matchplan_pipeline / _fetch_wait: a second stream packs every run's lists into one of two pinned host
    buffers; the lists of the latest run and of the run before it equal the copying fetch's, run after run, also after
    the plan is pointed at other pairs, and a run that outgrows the buffers says so instead of truncating"""
    imgs = synth.sift_image_set(5, 300, 128, bank=420, seed=13)
    pairs = synth.all_pairs(5)
    s, pl = _plan(ctx, imgs, pairs)
    want = {}
    for ratio in (0.8, 0.6, 0.95):
        pl.run_async(ratio)
        want[ratio] = [a.copy() for a in pl.fetch()]
    pl.pipeline()
    prev = None
    for ratio in (0.8, 0.6, 0.95, 0.6):
        s.prepare_async()
        pl.run_async(ratio)
        got = pl.fetch_wait(0)
        assert all(np.array_equal(a, b) for a, b in zip(got, want[ratio])), ratio
        assert got[3].view(np.uint32).tolist() == want[ratio][3].view(np.uint32).tolist()
        if prev is not None:
            before = pl.fetch_wait(1)
            assert all(np.array_equal(a, b) for a, b in zip(before, want[prev])), (ratio, prev)
        prev = ratio
    # two runs in flight before the host looks: back = 1 is the older one
    pl.run_async(0.8)
    pl.run_async(0.95)
    assert all(np.array_equal(a, b) for a, b in zip(pl.fetch_wait(1), want[0.8]))
    assert all(np.array_equal(a, b) for a, b in zip(pl.fetch_wait(0), want[0.95]))
    # re-targeted plan (fewer pairs), still pipelined
    pl.set_pairs(pairs[3:6])
    pl.run_async(0.8)
    cnt, q, t, d = pl.fetch_wait(0)
    off = 0
    for p, (a, b) in enumerate(pairs[3:6]):
        r = orc.match_knn2(imgs[a], imgs[b])
        assert cnt[p] == len(r[0]) and np.array_equal(q[off:off + cnt[p]], r[0]) and np.array_equal(t[off:off + cnt[p]], r[1])
        off += cnt[p]
    # a buffer too small for the run: reported, the copying fetch still works, and a larger pipeline recovers
    pl.set_pairs(pairs)
    pl.pipeline(-1)
    pl.pipeline(capacity=16)
    pl.run_async(0.8)
    with pytest.raises(_lib.SfmHipError):
        pl.fetch_wait(0)
    assert all(np.array_equal(a, b) for a, b in zip(pl.fetch(), want[0.8]))
    pl.pipeline(capacity=int(want[0.8][0].sum()))
    pl.run_async(0.8)
    assert all(np.array_equal(a, b) for a, b in zip(pl.fetch_wait(0), want[0.8]))
    pl.pipeline(-1)
    with pytest.raises(_lib.SfmHipError):
        pl.fetch_wait(0)
