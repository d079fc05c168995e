"""GPU: triangulation and bundle adjustment through the C ABI against the oracle / goldens."""
import numpy as np
import pytest

from sfm_danpipeline_amd import _lib, bundle, synth, triangulate

pytestmark = pytest.mark.gpu

# f64 results of the two implementations differ by rounding only (different but equivalent
# operation order in hypot / Jacobian assembly): stated tolerances
TRI_ATOL = 1e-11          # 3-D points, scene scale ~1..10
BA_PARAM_RTOL = 1e-6      # optimised cameras / points / focal at Ceres' default tolerances
BA_COST_RTOL = 1e-9


def test_triangulate_golden(ctx, golden):
    g = golden["triangulate"]
    X, err, keep = triangulate.triangulate_points(g["P1"], g["P2"], g["K"], g["dist"], g["xy1"], g["xy2"], ctx=ctx)
    assert np.array_equal(keep.astype(bool), g["keep"])
    assert np.allclose(X, g["X"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("m", [1, 63, 64, 65, 5000])
def test_triangulate_vs_oracle(ctx, orc, m):
    sc = synth.two_view_scene(m, seed=m)
    X, err, keep = triangulate.triangulate_points(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"], sc["xy2"], ctx=ctx)
    Xo, erro, keepo = orc.triangulate(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"], sc["xy2"])
    assert np.array_equal(keep, keepo)                      # bit-exact visibility
    # and, since round 3, bit-exact points and errors: the device follows the checker operation for operation (same
    # Jacobi SVD, no contraction, the host libm's hypot restated in csrc/hypot_glibc.h; gfx950's f64 +, *, /, sqrt are
    # correctly rounded: scripts/ubench/f64_rounding.hip)
    assert np.array_equal(X.view(np.uint64), Xo.view(np.uint64))
    assert np.array_equal(err.view(np.uint32), erro.view(np.uint32))


def test_triangulate_with_distortion_and_empty(ctx, orc):
    sc = synth.two_view_scene(300, seed=2)
    dist = np.array([-0.05, 0.01, 1e-4, -2e-4, 0.001])
    X, err, keep = triangulate.triangulate_points(sc["P1"], sc["P2"], sc["K"], dist, sc["xy1"], sc["xy2"], ctx=ctx)
    Xo, erro, keepo = orc.triangulate(sc["P1"], sc["P2"], sc["K"], dist, sc["xy1"], sc["xy2"])
    assert np.array_equal(keep, keepo) and np.array_equal(X.view(np.uint64), Xo.view(np.uint64))      # the 5-iteration undistortion too
    assert np.array_equal(err.view(np.uint32), erro.view(np.uint32))
    X, err, keep = triangulate.triangulate_points(sc["P1"], sc["P2"], sc["K"], dist, np.zeros((0, 2)), np.zeros((0, 2)), ctx=ctx)
    assert X.shape == (0, 3) and keep.shape == (0,)


def test_triangulate_views_builds_two_view_tracks(ctx, orc):
    sc = synth.two_view_scene(200, seed=5)
    mq = np.arange(200)[::-1].copy()
    mt = np.arange(200)[::-1].copy()
    cloud = triangulate.triangulate_views(sc["xy1"], sc["xy2"], sc["P1"], sc["P2"], mq, mt, sc["K"], sc["dist"], (3, 7), ctx=ctx)
    Xo, erro, keepo = orc.triangulate(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"][mq], sc["xy2"][mt])
    kept = np.nonzero(keepo)[0]
    assert len(cloud) == len(kept)
    for p, i in zip(cloud, kept):                                 # match order, src/Sfm.cpp:862-873
        assert p["idxImage"] == {3: int(mq[i]), 7: int(mt[i])}
        assert p["pt2D"][3] == tuple(sc["xy1"][mq[i]]) and p["pt2D"][7] == tuple(sc["xy2"][mt[i]])


def _ba_args(pb):
    return pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"]


def test_reduced_system_golden(ctx, golden):
    g = golden["ba"]
    prob = bundle.BaProblem(len(g["cams0"]), len(g["pts0"]), g["obs_cam"], g["obs_pt"], g["obs_xy"], ctx=ctx)
    prob.set_params(g["cams0"], g["pts0"], float(g["focal0"]))
    S, gg, cost = prob.reduced_system(1e4)
    assert abs(cost - float(g["cost0"])) < 1e-9 * cost
    assert np.abs(S - g["S"]).max() < 1e-11 * np.abs(g["S"]).max()
    assert np.abs(gg - g["g"]).max() < 1e-10 * np.abs(g["g"]).max()


@pytest.mark.parametrize("nc,npt,k,seed", [(6, 60, 4, 5), (9, 400, 2, 6), (20, 2000, 10, 7), (12, 500, 12, 8),
                                           (10, 300, 7, 9), (10, 300, 8, 10)])
def test_reduced_system_vs_oracle(ctx, orc, nc, npt, k, seed):
    pb = synth.ba_problem(nc, npt, k, seed=seed)
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    for radius in (1e4, 3.0):
        S, g, cost = prob.reduced_system(radius)
        So, go, costo, _ = orc.ba_reduced_system(*_ba_args(pb), radius=radius)
        assert abs(cost - costo) <= 1e-12 * costo
        assert np.abs(S - So).max() <= 1e-11 * np.abs(So).max()
        assert np.abs(g - go).max() <= 1e-10 * np.abs(go).max()


def test_mixed_signatures_unsorted_and_repeated_cameras(ctx, orc):
    """Ragged tracks, shuffled observation order, a camera seen twice by one point, an
    unobserved camera and an unobserved point."""
    rng = np.random.default_rng(4)
    pb = synth.ba_problem(9, 250, 6, seed=11)
    keep = rng.random(pb["n_obs"]) < 0.7
    keep[:6] = True
    oc, op, xy = pb["obs_cam"][keep], pb["obs_pt"][keep], pb["obs_xy"][keep]
    sel = ~((oc == 8) | (op == 17))                  # camera 8 and point 17 lose every observation
    oc, op, xy = oc[sel], op[sel], xy[sel]
    oc = np.concatenate([oc, oc[:1]])                # point op[0] sees camera oc[0] twice
    op = np.concatenate([op, op[:1]])
    xy = np.concatenate([xy, xy[:1] + 0.3])
    perm = rng.permutation(len(oc))
    oc, op, xy = oc[perm], op[perm], xy[perm]
    prob = bundle.BaProblem(9, 250, oc, op, xy, ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    S, g, cost = prob.reduced_system(1e4)
    So, go, costo, _ = orc.ba_reduced_system(pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy, radius=1e4)
    assert abs(cost - costo) <= 1e-12 * costo
    assert np.abs(S - So).max() <= 1e-11 * np.abs(So).max() and np.abs(g - go).max() <= 1e-10 * np.abs(go).max()
    c, p, f, s = bundle.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy, opts=bundle.default_opts(max_time_s=0.0), ctx=ctx)
    co, po, fo, so = orc.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy, orc.default_opts(max_time_s=0.0))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert np.array_equal(p[17], pb["pts0"][17]) and np.array_equal(c[8], pb["cams0"][8])   # untouched blocks
    assert np.allclose(c, co, rtol=BA_PARAM_RTOL, atol=1e-9) and np.allclose(p, po, rtol=BA_PARAM_RTOL, atol=1e-9)


# reduced system of 6*nc+1 columns in 32-column tiles, padded to an even count: 2, 4, 6, 10, 18 tiles
# (one, two, three ... two-panel launches; the backward substitution in groups of 8 tiles: 8+2, 8+8+2)
@pytest.mark.parametrize("nc,npt,k,seed", [(6, 60, 4, 5), (11, 600, 5, 11), (20, 2000, 10, 7), (27, 1500, 6, 27),
                                           (43, 2500, 6, 43), (86, 4000, 6, 86), (50, 20000, 10, 777)])
def test_solve_matches_oracle_trajectory(ctx, orc, nc, npt, k, seed):
    pb = synth.ba_problem(nc, npt, k, seed=seed)            # (50, 20000, 10, 777) is BASELINE cfg3
    c, p, f, s = bundle.ba_solve(*_ba_args(pb), opts=bundle.default_opts(max_time_s=0.0), ctx=ctx)
    co, po, fo, so = orc.ba_solve(*_ba_args(pb), opts=orc.default_opts(max_time_s=0.0))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert abs(s.initial_cost - so.initial_cost) <= 1e-12 * so.initial_cost
    assert abs(s.final_cost - so.final_cost) <= BA_COST_RTOL * so.final_cost
    assert np.allclose(c, co, rtol=BA_PARAM_RTOL, atol=1e-9)
    assert np.allclose(p, po, rtol=BA_PARAM_RTOL, atol=1e-9)
    assert abs(f - fo) <= BA_PARAM_RTOL * fo


def test_cfg4_sized_reduced_system_matches_oracle_iterates(ctx, orc):
    """200 cameras (BASELINE cfg4's reduced system: 1201 columns = 38 tiles, 19 two-panel launches,
    five back-substitution groups) with a fifth of cfg4's points: three LM iterations, iterate by
    iterate against the oracle."""
    pb = synth.ba_problem(200, 20000, 10, seed=778)
    opts = dict(max_time_s=0.0, max_iterations=3)
    c, p, f, s = bundle.ba_solve(*_ba_args(pb), opts=bundle.default_opts(**opts), ctx=ctx)
    co, po, fo, so = orc.ba_solve(*_ba_args(pb), opts=orc.default_opts(**opts))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert abs(s.final_cost - so.final_cost) <= BA_COST_RTOL * so.final_cost
    assert np.allclose(c, co, rtol=BA_PARAM_RTOL, atol=1e-9) and np.allclose(p, po, rtol=BA_PARAM_RTOL, atol=1e-9)
    assert abs(f - fo) <= BA_PARAM_RTOL * fo


def test_dissected_reduced_system_walks_the_dense_iterates(ctx, orc, monkeypatch):
    """The reduced camera system factored as independent chains + separator (ring capture: the camera graph has
    small separators) against the dense factorisation of the same system, iterate by iterate, and both against
    the oracle; the layout query says which one ran."""
    pb = synth.ba_problem(96, 12000, 6, seed=5)
    opts = dict(max_time_s=0.0, max_iterations=6)
    out = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("SFMHIP_BA_ND", mode)      # read when the problem plans its first solve
        prob = bundle.BaProblem(96, 12000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        s = prob.run(bundle.default_opts(**opts))
        out[mode] = (s, prob.get_params(), prob.reduced_layout())
        prob.close()
    (s0, (c0, p0, f0), l0), (s1, (c1, p1, f1), l1), (s2, (c2, p2, f2), l2) = out["0"], out["1"], out["2"]
    assert l2["chains"] == 0                       # (the front tree; its shape is asserted in test_reduced_layout_...)
    assert (s0.termination, s0.iterations, s0.successful_steps) == (s2.termination, s2.iterations, s2.successful_steps)
    assert abs(s0.final_cost - s2.final_cost) <= 1e-11 * s0.final_cost
    assert np.allclose(c0, c2, rtol=1e-9, atol=1e-12) and np.allclose(p0, p2, rtol=1e-8, atol=1e-10) and abs(f0 - f2) <= 1e-10 * f0
    assert l0["chains"] == 0 and l0["dense_tiles"] == 20
    assert l1["chains"] >= 2 and l1["chain_tiles"] + l1["separator_tiles"] < l1["dense_tiles"]
    assert (s0.termination, s0.iterations, s0.successful_steps) == (s1.termination, s1.iterations, s1.successful_steps)
    assert abs(s0.final_cost - s1.final_cost) <= 1e-11 * s0.final_cost
    assert np.allclose(c0, c1, rtol=1e-9, atol=1e-12) and np.allclose(p0, p1, rtol=1e-8, atol=1e-10) and abs(f0 - f1) <= 1e-10 * f0
    co, po, fo, so = orc.ba_solve(*_ba_args(pb), opts=orc.default_opts(**opts))
    assert (s1.termination, s1.iterations, s1.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert abs(s1.final_cost - so.final_cost) <= BA_COST_RTOL * so.final_cost
    assert np.allclose(c1, co, rtol=BA_PARAM_RTOL, atol=1e-9) and np.allclose(p1, po, rtol=BA_PARAM_RTOL, atol=1e-9)


@pytest.mark.parametrize("shape,nd", [((96, 6000, 6), "0"), ((96, 6000, 6), "1"), ((180, 8000, 8), "0"), ((560, 8000, 8), "1"),
                                      ((260, 6000, 8), "0"), ((350, 6000, 8), "0"), ((560, 8000, 8), "0"), ((1400, 9000, 8), "0"), ((6, 300, 4), "2"), ((50, 5000, 10), "2"), ((96, 6000, 6), "2"),
                                      ((200, 20000, 10), "2"), ((560, 8000, 8), "2"), ((1400, 9000, 8), "2")])
def test_reduced_step_solves_the_reduced_system(ctx, monkeypatch, shape, nd):
    """(S + D/r) z = g, z from the solver's own factorisation (dense; dissected: chains + separator; the front tree),
    against numpy on the system the solver hands out.  560 cameras: six chains whose launches exceed one round of
    workgroups; 1400 cameras dense: 266 panel workgroups on 256 CUs (no workgroup of a launch may depend on another
    one's being resident: until round 2 the owner overwrote the diagonal tiles the others read); the front tree from one
    front (6 cameras) to 127 fronts in seven levels (1400 cameras).  Dense from 48 tile columns on: X only inside diagonal
    blocks of 8 tile columns, block-by-block backward substitution, trailing tiles visited every 2nd (260 / 350 cameras: 50 / 66
    tile columns, a last block of two) or 4th launch (560: 106 columns; 1400: 264) with the panels they missed folded at once."""
    monkeypatch.setenv("SFMHIP_BA_ND", nd)
    nc, npt, k = shape
    pb = synth.ba_problem(nc, npt, k, seed=5)
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    z, failed = prob.reduced_step(1e4)
    S, g, _ = prob.reduced_system(1e4)
    lay, tree = prob.reduced_layout(), prob.reduced_tree()
    assert failed == 0 and (lay["chains"] >= 2) == (nd == "1") and (tree["fronts"] >= 1) == (nd == "2")
    assert np.linalg.norm(S @ z - g) <= 1e-12 * np.linalg.norm(g)
    zr = np.linalg.solve(S, g)
    assert np.abs(z - zr).max() <= 1e-9 * np.abs(zr).max()
    prob.close()


def test_large_dense_solve_gives_the_same_answer_every_time(ctx, monkeypatch):
    """The block-by-block backward substitution adds its parts in a fixed order and the deferred trailing updates fold the
    panels a tile missed in panel order: the same system must give the same bits, launch after launch."""
    monkeypatch.setenv("SFMHIP_BA_ND", "0")
    for (nc, npt, k) in ((350, 6000, 8), (560, 8000, 8)):
        pb = synth.ba_problem(nc, npt, k, seed=5)
        prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        z0, failed = prob.reduced_step(1e4)
        S, g, _ = prob.reduced_system(1e4)
        assert failed == 0 and np.linalg.norm(S @ z0 - g) <= 1e-12 * np.linalg.norm(g)
        for rep in range(20):
            z, failed = prob.reduced_step(1e4)
            assert failed == 0 and np.array_equal(z, z0), (nc, rep)
        prob.close()


def test_reduced_solve_on_poisoned_allocations():
    """SFMHIP_POISON=1 (csrc/common.h: fresh device memory holds 0xFF bytes) in a process of its own: whatever a reduced solve
    reads it must have written or zeroed itself.  (Round 4: with X kept in diagonal blocks only, the panel workgroups of a
    block's first panels read the block's rows of X at the pending columns of the block before -- memory nothing had ever
    written; right by luck while the driver hands out zeroed pages.)"""
    import subprocess, sys, os
    code = (
        "import os, sys, numpy as np\n"
        "from sfm_danpipeline_amd import synth, bundle, _lib\n"
        "ctx = _lib.default_context()\n"
        "for nd, nc, npt, k in (('0', 350, 5000, 6), ('0', 560, 6000, 8), ('0', 180, 6000, 8), ('1', 96, 6000, 6), ('2', 260, 4000, 4), ('2', 560, 6000, 8)):\n"
        "    os.environ['SFMHIP_BA_ND'] = nd\n"
        "    pb = synth.ba_problem(nc, npt, k, seed=5)\n"
        "    prob = bundle.BaProblem(nc, npt, pb['obs_cam'], pb['obs_pt'], pb['obs_xy'], ctx=ctx)\n"
        "    prob.set_params(pb['cams0'], pb['pts0'], pb['focal0'])\n"
        "    S, g, _ = prob.reduced_system(1e4)\n"
        "    for rep in range(4):\n"
        "        z, failed = prob.reduced_step(1e4)\n"
        "        assert failed == 0 and np.linalg.norm(S @ z - g) <= 1e-12 * np.linalg.norm(g), (nd, nc, rep, failed)\n"
        "    s = prob.iterate(4)\n"
        "    assert np.isfinite(s.final_cost) and s.final_cost < s.initial_cost, (nd, nc)\n"
        "    prob.close()\n"
        "print('poisoned solves ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root,
                       env=dict(os.environ, SFMHIP_POISON="1", PYTHONPATH=root))
    assert r.returncode == 0 and "poisoned solves ok" in r.stdout, r.stderr[-2000:]


def test_front_tree_gives_the_same_answer_every_time(ctx, monkeypatch):
    """The front tree's workgroups hand tiles to each other inside one launch, and inside a workgroup twelve waves share two
    panel generations through counters: 150 solves of the same system must be the same bit pattern every time, and right.
    (Round 4 found a wave with no work in a step adding its count for the step at once, so that the counter reached a full
    round while the right-hand side's wave was still reading the panel the next step overwrote -- one solve in ten at 200
    cameras once the children's right-hand sides arrived late.)"""
    monkeypatch.setenv("SFMHIP_BA_ND", "2")
    for (nc, npt, k) in ((200, 20000, 10), (560, 8000, 8)):
        pb = synth.ba_problem(nc, npt, k, seed=5)
        prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        S, g, _ = prob.reduced_system(1e4)
        zr = np.linalg.solve(S, g)
        z0 = None
        for rep in range(150):
            z, failed = prob.reduced_step(1e4)
            assert failed == 0
            if z0 is None:
                z0 = z.copy()
                assert np.abs(z - zr).max() <= 1e-9 * np.abs(zr).max()
            assert np.array_equal(z, z0), rep
        prob.close()


def test_reduced_solve_on_many_camera_graphs(ctx, monkeypatch):
    """Whatever plan a camera graph gets -- front trees of one to six levels, chains + separator, dense --: the step solves the
    system the solver hands out, and is the same bit pattern when asked again.  Rings and open bands of several widths,
    irregular tracks, long-range edges (scripts/gpu_tree_fuzz.py runs hundreds of these)."""
    monkeypatch.delenv("SFMHIP_BA_ND", raising=False)
    rng = np.random.default_rng(1)
    plans = set()
    for case in range(14):
        nc = int(rng.choice([12, 24, 40, 64, 90, 130, 200, 260, 330, 420]))
        k = int(rng.integers(3, 11))
        npt = int(rng.integers(20, 60)) * nc
        pb = synth.ba_problem(nc, npt, k, seed=int(rng.integers(1 << 30)))
        oc, op, xy = pb["obs_cam"].copy(), pb["obs_pt"].copy(), pb["obs_xy"].copy()
        kind = case % 4
        kk = min(k, nc)
        if kind == 1:      # an open band: the tracks that wrap around the ring go
            cams = oc.reshape(-1, kk)
            keep = np.repeat((cams[:, -1] - cams[:, 0]) < kk, kk)
            oc, op, xy = oc[keep], op[keep], xy[keep]
        elif kind == 2:    # irregular tracks
            keep = rng.random(len(oc)) < 0.8
            keep[np.unique(op, return_index=True)[1]] = True
            oc, op, xy = oc[keep], op[keep], xy[keep]
        elif kind == 3:    # a few points also seen from the other side of the ring: long-range edges
            idx = np.nonzero((np.arange(npt) % 7 == 0) & (rng.random(npt) < 0.15))[0]
            fc = ((oc.reshape(-1, kk)[idx, 0] + nc // 2) % nc).astype(np.int32)
            oc, op, xy = np.concatenate([oc, fc]), np.concatenate([op, idx.astype(np.int32)]), np.concatenate([xy, rng.normal(0, 50, (len(idx), 2))])
        prob = bundle.BaProblem(nc, npt, oc, op, xy, ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        S, g, _ = prob.reduced_system(1e3)
        zr = np.linalg.solve(S, g)
        z0 = None
        for rep in range(6):
            z, failed = prob.reduced_step(1e3)
            assert failed == 0, (case, nc, k, kind)
            assert np.abs(z - zr).max() <= 1e-8 * np.abs(zr).max(), (case, nc, k, kind)
            z0 = z.copy() if z0 is None else z0
            assert np.array_equal(z, z0), (case, nc, k, kind, rep)
        tree, lay = prob.reduced_tree(), prob.reduced_layout()
        plans.add("tree" if tree["fronts"] else "chains" if lay["chains"] else "dense")
        prob.close()
    assert "tree" in plans and len(plans) >= 2


def test_reduced_layout_follows_the_camera_graph(ctx, monkeypatch):
    """cfg4's co-visibility (a ring, every point seen by 10 consecutive cameras) is dissected by default; cameras
    that all see each other (random visibility) leave no separator and keep the dense factorisation; small
    systems are not worth it."""
    monkeypatch.delenv("SFMHIP_BA_ND", raising=False)
    pb = synth.ba_problem(200, 20000, 10, seed=3)
    ring = bundle.BaProblem(200, 20000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    ring.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    assert ring.reduced_layout()["dense_tiles"] == 0          # not planned yet
    ring.iterate(1)
    lay, tree = ring.reduced_layout(), ring.reduced_tree()
    # the front tree: 16 leaves of one tile, three levels of 2-tile separators, a 4-tile root -- 11 tile steps on the chain
    assert lay["chains"] == 0 and lay["dense_tiles"] == 38
    assert tree == dict(fronts=31, levels=5, chain_tiles=11, max_front_tiles=7)
    monkeypatch.setenv("SFMHIP_BA_ND", "1")                   # the chains + separator plan of round 2
    ring1 = bundle.BaProblem(200, 20000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    ring1.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    ring1.iterate(1)
    lay = ring1.reduced_layout()
    assert lay["chains"] == 4 and lay["chain_tiles"] + lay["separator_tiles"] <= 16 and ring1.reduced_tree()["fronts"] == 0
    monkeypatch.delenv("SFMHIP_BA_ND", raising=False)
    rng = np.random.default_rng(0)
    oc = np.concatenate([np.sort(rng.choice(200, 10, replace=False)) for _ in range(20000)]).astype(np.int32)
    rnd = bundle.BaProblem(200, 20000, oc, pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    rnd.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    rnd.iterate(1)
    assert rnd.reduced_layout()["chains"] == 0 and rnd.reduced_tree()["fronts"] == 0
    small = synth.ba_problem(50, 5000, 10, seed=4)
    sm = bundle.BaProblem(50, 5000, small["obs_cam"], small["obs_pt"], small["obs_xy"], ctx=ctx)
    sm.set_params(small["cams0"], small["pts0"], small["focal0"])
    sm.iterate(1)
    assert sm.reduced_layout()["chains"] == 0 and sm.reduced_tree() == dict(fronts=3, levels=2, chain_tiles=7, max_front_tiles=7)


def test_tracks_longer_than_a_wave(ctx, orc):
    """Points seen by 70 of 80 cameras, and by 260 of 300 (the pair path: the reference, Ceres, has no limit on a
    track's length) against the oracle."""
    pb = synth.ba_problem(80, 120, 70, seed=21)
    opts = dict(max_time_s=0.0, max_iterations=4)
    c, p, f, s = bundle.ba_solve(*_ba_args(pb), opts=bundle.default_opts(**opts), ctx=ctx)
    co, po, fo, so = orc.ba_solve(*_ba_args(pb), opts=orc.default_opts(**opts))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert abs(s.final_cost - so.final_cost) <= BA_COST_RTOL * so.final_cost
    assert np.allclose(c, co, rtol=BA_PARAM_RTOL, atol=1e-9) and np.allclose(p, po, rtol=BA_PARAM_RTOL, atol=1e-9)
    big = synth.ba_problem(300, 6, 260, seed=22)
    o2 = dict(max_time_s=0.0, max_iterations=2)
    c, p, f, s = bundle.ba_solve(*_ba_args(big), opts=bundle.default_opts(**o2), ctx=ctx)
    co, po, fo, so = orc.ba_solve(*_ba_args(big), opts=orc.default_opts(**o2))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert np.allclose(c, co, rtol=BA_PARAM_RTOL, atol=1e-9) and np.allclose(p, po, rtol=BA_PARAM_RTOL, atol=1e-9)


def test_noise_free_scene_converges_to_zero_cost(ctx):
    pb = synth.ba_problem(8, 300, 5, seed=3, noise_px=0.0)
    c, p, f, s = bundle.ba_solve(*_ba_args(pb), ctx=ctx, opts=bundle.default_opts(
        max_time_s=0.0, function_tolerance=1e-16, parameter_tolerance=1e-16, max_iterations=200))
    assert s.final_cost < 1e-12 * s.initial_cost


def test_termination_policies(ctx):
    pb = synth.ba_problem(6, 80, 4, seed=9)
    _, _, _, s = bundle.ba_solve(*_ba_args(pb), ctx=ctx, opts=bundle.default_opts(
        max_time_s=0.0, max_iterations=1, function_tolerance=0.0, parameter_tolerance=0.0))
    assert s.termination == _lib.BA_NO_CONVERGENCE and s.iterations == 1
    _, _, _, s = bundle.ba_solve(*_ba_args(pb), ctx=ctx, opts=bundle.default_opts(max_time_s=1e-9, function_tolerance=0.0,
                                                                                   parameter_tolerance=0.0))
    assert s.termination == _lib.BA_NO_CONVERGENCE          # the 10 s rule, src/BundleAdjustment.cpp:120


def test_adjust_bundle_end_to_end(ctx, orc):
    """The adjustBundle drop-in on reference-shaped containers, device solver vs oracle solver."""
    import copy
    from tests.test_host_logic import _orc_solver, _scene
    pb, cloud, poses, K, feats = _scene(n_cam=7, n_pt=150, k=4, seed=33)
    cloud2, poses2, K2 = copy.deepcopy(cloud), copy.deepcopy(poses), K.copy()
    s = bundle.adjust_bundle(cloud, poses, K, feats, ctx=ctx)
    s2 = bundle.adjust_bundle(cloud2, poses2, K2, feats, solver=_orc_solver(orc))
    assert s.termination == s2.termination == _lib.BA_CONVERGENCE
    assert abs(K[0, 0] - K2[0, 0]) <= BA_PARAM_RTOL * K2[0, 0]
    assert np.allclose([q["pt"] for q in cloud], [q["pt"] for q in cloud2], rtol=BA_PARAM_RTOL, atol=1e-9)
    assert all(np.allclose(a, b, rtol=BA_PARAM_RTOL, atol=1e-9) for a, b in zip(poses, poses2))


def test_allreduce_hook_with_a_mirrored_rank(ctx):
    """The N>1 data path on one device: a fake 2-rank job whose all-reduce doubles the buffer
    (= a second rank holding an identical point block) must walk the same LM iterates as one
    rank solving the problem with every point duplicated."""
    import torch
    from sfm_danpipeline_amd import sharding
    pb = synth.ba_problem(10, 1500, 6, seed=15)
    view = sharding.TorchAllReduce(device=f"cuda:{ctx.device}")
    calls = []

    def mirrored(ptr, count):
        ctx.synchronize()
        view._view(ptr, count).mul_(2.0)
        torch.cuda.synchronize()
        calls.append(count)

    two = bundle.BaProblem(10, 1500, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    two.set_allreduce(mirrored, 0, 2)
    two.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    s2 = two.iterate(4)
    dup = bundle.BaProblem(10, 3000, np.concatenate([pb["obs_cam"]] * 2),
                           np.concatenate([pb["obs_pt"], pb["obs_pt"] + 1500]), np.concatenate([pb["obs_xy"]] * 2), ctx=ctx)
    dup.set_params(pb["cams0"], np.concatenate([pb["pts0"]] * 2), pb["focal0"])
    s1 = dup.iterate(4)
    ld = (6 * 10 + 1 + 63) // 64 * 64                                # row stride of S: dim rounded up to 64
    # [packed upper triangle of S | g | F^T b | diag | scalars + rank slots] and the step evaluation's sums (8 doubles: the
    # back-substitution's workgroups are added up on the device, in a fixed order, before the exchange)
    assert ld * (ld + 1) // 2 + 3 * ld + 16 + 2 in calls and 8 in calls
    assert s1.successful_steps == s2.successful_steps
    assert abs(s1.initial_cost - s2.initial_cost) <= 1e-12 * s1.initial_cost
    assert abs(s1.final_cost - s2.final_cost) <= 1e-9 * s1.final_cost
    c2, p2, f2 = two.get_params()
    c1, p1, f1 = dup.get_params()
    assert np.allclose(c1, c2, rtol=1e-9, atol=1e-12) and np.allclose(p1[:1500], p2, rtol=1e-9, atol=1e-12)
    assert abs(f1 - f2) <= 1e-9 * f1


def test_residual_and_jacobian_per_observation(ctx, orc, golden):
    """The device linearisation of single observations (a-4, src/BundleAdjustment.cpp:10-35) against the
    complex-step golden Jacobians in BOTH theta^2 branches -- the first-order branch is the one the base
    camera P_left = I always takes (src/Sfm.cpp:432,772) -- and against the oracle on random observations."""
    g = golden["ba"]
    n = len(g["jac_cams"])
    r, Jc, Jp, Jf = bundle.linearize_obs(g["jac_cams"], np.tile(g["jac_X"], (n, 1)), 1500.0, np.tile(g["jac_obs"], (n, 1)), ctx=ctx)
    for i in range(n):
        J = g["jac_J"][i]
        assert np.allclose(r[i], g["jac_r"][i], atol=1e-12)
        assert np.allclose(Jc[i], J[:, :6], rtol=1e-11, atol=1e-11)
        assert np.allclose(Jp[i], J[:, 6:9], rtol=1e-11, atol=1e-11)
        assert np.allclose(Jf[i], J[:, 9], rtol=1e-13, atol=1e-13)
    rng = np.random.default_rng(4)
    m = 500
    cams = np.concatenate([rng.normal(0, 0.4, (m, 3)), rng.normal(0, 0.3, (m, 2)), rng.uniform(4, 8, (m, 1))], axis=1)
    cams[::5, :3] = 0.0                                   # exactly the identity rotation: theta^2 == 0
    cams[1::5, :3] *= 1e-9                                # theta^2 below DBL_EPSILON: still the first-order branch
    X = rng.uniform(-1, 1, (m, 3))
    xy = rng.normal(0, 50, (m, 2))
    r, Jc, Jp, Jf = bundle.linearize_obs(cams, X, 1520.0, xy, ctx=ctx)
    for i in range(m):
        ro, Jco, Jpo, Jfo = orc.ba_residual(cams[i], X[i], 1520.0, xy[i])
        assert np.allclose(r[i], ro, rtol=1e-12, atol=1e-10)
        assert np.allclose(Jc[i], Jco, rtol=1e-10, atol=1e-9) and np.allclose(Jp[i], Jpo, rtol=1e-10, atol=1e-9)
        assert np.allclose(Jf[i], Jfo, rtol=1e-12, atol=1e-12)


def test_cfg4_full_size_three_iterations_vs_oracle(ctx, orc):
    """BASELINE cfg4 at full size (200 cams / 100k points / 1M obs): the first three LM iterations,
    device solver against the oracle -- termination, iteration and accepted-step counts, costs, parameters."""
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    kw = dict(max_iterations=3, max_time_s=0.0)
    c, p, f, s = bundle.ba_solve(*_ba_args(pb), ctx=ctx, opts=bundle.default_opts(**kw))
    co, po, fo, so = orc.ba_solve(*_ba_args(pb), opts=orc.default_opts(**kw))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert abs(s.initial_cost - so.initial_cost) <= 1e-12 * so.initial_cost
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.allclose(c, co, rtol=1e-6, atol=1e-9) and np.allclose(p, po, rtol=1e-6, atol=1e-9)
    assert abs(f - fo) <= 1e-6 * fo


def test_cfg4_full_size_twelve_iterations_vs_oracle(ctx, orc):
    """BASELINE cfg4 at full size, twelve LM iterations (accepted and rejected steps, radius going both ways): the device
    solver's decisions and numbers against the oracle's."""
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    kw = dict(max_iterations=12, max_time_s=0.0)
    c, p, f, s = bundle.ba_solve(*_ba_args(pb), ctx=ctx, opts=bundle.default_opts(**kw))
    co, po, fo, so = orc.ba_solve(*_ba_args(pb), opts=orc.default_opts(**kw))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert 0 < so.successful_steps < so.iterations          # (the run does contain rejected steps)
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.allclose(c, co, rtol=1e-6, atol=1e-9) and np.allclose(p, po, rtol=1e-6, atol=1e-9)
    assert abs(f - fo) <= 1e-6 * fo


def test_cfg4_iterations_decrease_cost(ctx):
    """BASELINE cfg4 shape (200 cams / 100k points / 1M obs): LM iterations run and descend."""
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    s = prob.iterate(3)
    s2 = prob.iterate(3)
    assert s2.iterations == 6 and s2.final_cost <= s.final_cost < s.initial_cost
    assert np.isfinite(s2.final_cost)


def test_degenerate_problems_do_not_crash(ctx, orc):
    """No observations at all, a point nobody observes, cameras nobody uses, one observation only:
    the solver must return the oracle's verdict (or at least terminate cleanly) on all of them."""
    pb = synth.ba_problem(5, 40, 3, seed=21)
    cams, pts, f = pb["cams0"], pb["pts0"], pb["focal0"]
    e_i, e_f = np.zeros(0, np.int32), np.zeros((0, 2))
    c, p, ff, s = bundle.ba_solve(cams, pts, f, e_i, e_i, e_f, ctx=ctx)                     # nothing to fit
    assert np.array_equal(c, cams) and np.array_equal(p, pts) and ff == f and s.initial_cost == 0.0
    c, p, ff, s = bundle.ba_solve(cams, np.zeros((0, 3)), f, e_i, e_i, e_f, ctx=ctx)        # no points either
    assert s.initial_cost == 0.0
    # drop every observation of camera 4 and of point 7: unused blocks keep their values
    keep = (pb["obs_cam"] != 4) & (pb["obs_pt"] != 7)
    args = (cams, pts, f, pb["obs_cam"][keep], pb["obs_pt"][keep], pb["obs_xy"][keep])
    c, p, ff, s = bundle.ba_solve(*args, opts=bundle.default_opts(max_time_s=0.0), ctx=ctx)
    co, po, fo, so = orc.ba_solve(*args, opts=orc.default_opts(max_time_s=0.0))
    assert (s.termination, s.iterations) == (so.termination, so.iterations)
    assert np.array_equal(c[4], cams[4]) and np.array_equal(p[7], pts[7])
    assert np.allclose(c, co, rtol=1e-6, atol=1e-9) and np.allclose(p, po, rtol=1e-6, atol=1e-9)
    # a single observation: rank-deficient everywhere but the LM diagonal keeps it solvable
    one = (cams, pts, f, pb["obs_cam"][:1], pb["obs_pt"][:1], pb["obs_xy"][:1])
    c, p, ff, s = bundle.ba_solve(*one, opts=bundle.default_opts(max_time_s=0.0, max_iterations=20), ctx=ctx)
    co, po, fo, so = orc.ba_solve(*one, opts=orc.default_opts(max_time_s=0.0, max_iterations=20))
    assert s.termination == so.termination and np.isfinite(s.final_cost)
    assert abs(s.final_cost - so.final_cost) <= 1e-6 * max(so.final_cost, 1e-12) + 1e-12


def _two_rank_worker(rank, world, port, out_dir, shape=(16, 6000, 6, 44)):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from sfm_danpipeline_amd import _lib as L, bundle as B, sharding, synth as S
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    pb = S.ba_problem(shape[0], shape[1], shape[2], seed=shape[3])
    loc = sharding.local_ba_problem(pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], pb["pts0"], rank, world)
    prob = B.BaProblem(shape[0], len(loc["pts"]), loc["obs_cam"], loc["obs_pt"], loc["obs_xy"], ctx=ctx)
    ar = sharding.StagedAllReduce(device="cuda:0")
    prob.set_allreduce(ar, rank, world)
    prob.set_params(pb["cams0"], loc["pts"], pb["focal0"])
    s = prob.iterate(6)
    c, p, f = prob.get_params()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), c=c, p=p, f=f, lo=loc["lo"], hi=loc["hi"], cost=s.final_cost,
             steps=s.successful_steps, chains=prob.reduced_layout()["chains"], counts=np.array(ar.counts))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,nd", [((16, 6000, 6, 44), None), ((96, 9000, 5, 45), "1"), ((350, 5000, 6, 46), "0")])
def test_two_ranks_share_the_device_and_run_the_sharded_solver(ctx, tmp_path, monkeypatch, shape, nd):
    """(second case: the dissected reduced system, whose camera graph is the union over the ranks; third: a large dense
    one -- X in diagonal blocks, block-by-block backward substitution, deferred trailing updates -- replicated on both ranks)
    The N>1 data path with the device kernels as the per-rank engine: two processes (gloo, the all-reduce
    staged through the host because RCCL refuses two ranks on one device) each hold half the points; pack,
    exchange, unpack, redundant reduced solve and per-rank back-substitution must walk the iterates of one
    process holding everything, and the replicated cameras must agree between the ranks."""
    import socket
    import torch.multiprocessing as mp
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    if nd is None:
        monkeypatch.delenv("SFMHIP_BA_ND", raising=False)
    else:
        monkeypatch.setenv("SFMHIP_BA_ND", nd)
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), shape), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert (int(r0["chains"]) >= 2) == (nd == "1") and int(r0["chains"]) == int(r1["chains"])
    # the linearisations' exchange: the blocks of co-visible camera pairs + focal column + tail, not the packed
    # triangle (cameras that see each other across at most `views` positions of the ring: a sparse graph)
    ld = (6 * shape[0] + 1 + 63) // 64 * 64
    dense = ld * (ld + 1) // 2 + 3 * ld + 16 + 2
    pb = synth.ba_problem(shape[0], shape[1], shape[2], seed=shape[3])
    seen = {(a, a) for a in range(shape[0])}
    for cams in {tuple(sorted(set(c))) for c in pb["obs_cam"].reshape(-1, shape[2]).tolist()}:
        seen.update((a, b) for i, a in enumerate(cams) for b in cams[i:])
    sparse = 36 * len(seen) + (6 * shape[0] + 1) + 3 * ld + 16 + 2
    # (the sparse form whenever it is less than half the packed triangle -- since round 4 the camera graph is kept at every size)
    want, other = (sparse, dense) if 36 * len(seen) + 6 * shape[0] + 1 < (ld * (ld + 1) // 2) // 2 else (dense, sparse)
    assert want in r0["counts"] and other not in r0["counts"] and np.array_equal(r0["counts"], r1["counts"]), \
        (sparse, dense, sorted(set(r0["counts"].tolist())))
    one = bundle.BaProblem(shape[0], shape[1], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    one.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    s1 = one.iterate(6)
    c1, p1, f1 = one.get_params()
    assert int(r0["steps"]) == int(r1["steps"]) == s1.successful_steps
    assert abs(float(r0["cost"]) - s1.final_cost) <= 1e-9 * s1.final_cost and float(r0["cost"]) == float(r1["cost"])
    assert np.allclose(r0["c"], c1, rtol=1e-9, atol=1e-12) and abs(float(r0["f"]) - f1) <= 1e-9 * f1
    # replicas are bitwise equal: after the all-reduce both ranks hold the same S and g, and nothing in the reduced
    # solve sums in a run-dependent order (chol_apply_inverse is atomic-free since round 3)
    assert np.array_equal(r0["c"], r1["c"]) and float(r0["f"]) == float(r1["f"])
    # (points: absolute tolerance -- coordinates near zero carry the f64 summation-order noise of the exchange)
    assert np.allclose(np.concatenate([r0["p"], r1["p"]]), p1, rtol=1e-8, atol=1e-9)


def _eq_bits(x, y):
    return np.array_equal(np.ascontiguousarray(x, np.float64).view(np.uint64), np.ascontiguousarray(y, np.float64).view(np.uint64))


def test_one_shot_solve_keeps_its_plan_for_a_call_of_the_same_structure(ctx):
    """sfmhip_ba_solve -- what BundleAdjustment::adjustBundle maps to (reference include/BundleAdjustment.h:19-20: a static
    one-shot function, called again and again, src/Sfm.cpp:883-888) -- keeps the problem it built: a second call with the same
    observation structure skips the set-up (plan_reused) and returns the SAME BITS as the first; new measurements on the same
    structure give the bits of a freshly built problem; another structure replaces the kept one."""
    c = _lib.Context(0)                       # (a context of its own: the kept problem is the context's)
    pb = synth.ba_problem(24, 3000, 6, seed=91)
    a = _ba_args(pb)
    r1 = bundle.ba_solve(*a, ctx=c)
    p1 = bundle.last_solve_profile(c)
    r2 = bundle.ba_solve(*a, ctx=c)
    p2 = bundle.last_solve_profile(c)
    assert (p1["plan_reused"], p2["plan_reused"]) == (0, 1) and p2["create_ms"] < p1["create_ms"]
    assert _eq_bits(r1[0], r2[0]) and _eq_bits(r1[1], r2[1]) and r1[2] == r2[2]
    assert (r1[3].iterations, r1[3].final_cost, r1[3].termination) == (r2[3].iterations, r2[3].final_cost, r2[3].termination)
    # the same structure, other measurements (and another start): the kept plan with the new data == a problem built for them
    xy2 = a[5] + np.random.default_rng(5).normal(0, 0.3, a[5].shape)
    b = (a[0] * (1 + 1e-4), a[1], a[2], a[3], a[4], xy2)
    r3 = bundle.ba_solve(*b, ctx=c)
    assert bundle.last_solve_profile(c)["plan_reused"] == 1
    fresh = bundle.BaProblem(24, 3000, b[3], b[4], b[5], ctx=ctx)
    fresh.set_params(b[0], b[1], b[2])
    s4 = fresh.run(bundle.default_opts())
    c4, q4, f4 = fresh.get_params()
    assert _eq_bits(r3[0], c4) and _eq_bits(r3[1], q4) and r3[2] == f4 and r3[3].iterations == s4.iterations
    assert not _eq_bits(r3[0], r1[0])
    # another structure (one observation less): built anew, and the next call of THAT structure finds it kept
    d = (a[0], a[1], a[2], a[3][:-1], a[4][:-1], a[5][:-1])
    r5 = bundle.ba_solve(*d, ctx=c)
    assert bundle.last_solve_profile(c)["plan_reused"] == 0
    r6 = bundle.ba_solve(*d, ctx=c)
    assert bundle.last_solve_profile(c)["plan_reused"] == 1 and _eq_bits(r5[0], r6[0]) and _eq_bits(r5[1], r6[1])
    # same sizes, one camera index changed: compared element for element, not by size
    oc = a[3].copy()
    oc[5] = (oc[5] + 12) % 24
    bundle.ba_solve(a[0], a[1], a[2], oc, a[4], a[5], ctx=c)
    assert bundle.last_solve_profile(c)["plan_reused"] == 0
    c.close()


def test_one_shot_solve_keeps_the_front_tree_of_an_unchanged_camera_graph(ctx):
    """The per-view call pattern of src/Sfm.cpp:996 changes the tracks from call to call and the camera graph hardly ever: a NEW
    structure over the same camera graph is set up anew but takes the kept front tree (front_plan_reused), and returns the bits
    of a problem built from nothing; a structure with another camera graph plans its own tree -- same check."""
    c = _lib.Context(0)
    pb = synth.ba_problem(96, 6000, 6, seed=92)
    a = _ba_args(pb)

    def fresh_bits(args):
        pr = bundle.BaProblem(96, 6000, args[3], args[4], args[5], ctx=ctx)   # (sfmhip_ba_create: nothing kept, nothing re-used)
        pr.set_params(args[0], args[1], args[2])
        s = pr.run(bundle.default_opts())
        out = pr.get_params() + (s, pr.reduced_tree())
        pr.close()
        return out

    bundle.ba_solve(*a, ctx=c)
    p0 = bundle.last_solve_profile(c)
    assert (p0["plan_reused"], p0["front_plan_reused"]) == (0, 0)
    d = (a[0], a[1], a[2], a[3][:-1], a[4][:-1], a[5][:-1])   # one observation less of a six-view track: the same camera graph
    r = bundle.ba_solve(*d, ctx=c)
    p1 = bundle.last_solve_profile(c)
    assert (p1["plan_reused"], p1["front_plan_reused"]) == (0, 1)
    f = fresh_bits(d)
    assert f[4]["fronts"] > 1, "the check wants a front tree"
    assert _eq_bits(r[0], f[0]) and _eq_bits(r[1], f[1]) and r[2] == f[2] and r[3].iterations == f[3].iterations
    oc = a[3].copy()
    oc[5] = (oc[5] + 48) % 96                                  # an edge the ring does not have: another graph, another tree
    e = (a[0], a[1], a[2], oc, a[4], a[5])
    r = bundle.ba_solve(*e, ctx=c)
    p2 = bundle.last_solve_profile(c)
    assert (p2["plan_reused"], p2["front_plan_reused"]) == (0, 0)
    f = fresh_bits(e)
    assert _eq_bits(r[0], f[0]) and _eq_bits(r[1], f[1]) and r[2] == f[2] and r[3].iterations == f[3].iterations
    c.close()


def _logical_ranks_solve(pb, world, opts=None, iterate=None):
    """BASELINE cfg4's split, `world` logical ranks on the one device: a BaProblem per rank over its point block (all cameras, the
    focal), a context + stream + host thread each, the exchange summed on the device in rank order (sharding.InProcessRanks)."""
    from sfm_danpipeline_amd import _lib as L, sharding
    nc = pb["n_cam"]
    grp = sharding.InProcessRanks(world)

    def rank(r):
        c = L.Context(0)
        loc = sharding.local_ba_problem(pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], pb["pts0"], r, world)
        prob = bundle.BaProblem(nc, len(loc["pts"]), loc["obs_cam"], loc["obs_pt"], loc["obs_xy"], ctx=c)
        prob.set_allreduce(grp.allreduce(r, c), r, world)
        prob.set_params(pb["cams0"], loc["pts"], pb["focal0"])
        s = prob.iterate(iterate) if iterate is not None else prob.run(opts or bundle.default_opts())
        cams, pts, f = prob.get_params()
        tree = prob.reduced_tree()
        prob.close()
        c.close()
        return dict(s=s, c=cams, p=pts, f=f, lo=loc["lo"], hi=loc["hi"], tree=tree)
    return grp.run(rank), grp


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 8])
def test_cfg4_eight_logical_ranks_walk_the_single_rank_solve(ctx, world):
    """BASELINE cfg4 as it is stated -- 200 cameras / 100 000 points / 10^6 observations, "point-block sharded across 8 x MI355X
    with RCCL all-reduce of camera normal equations" (reference src/BundleAdjustment.cpp:83-123 is what gets sharded) -- with the
    eight (and four) ranks LOGICAL: eight problems in one process on the one device, each holding an eighth of the points, the
    sparse exchange list rebuilt for world = 8, eight packed buffers summed per linearisation.  Run to termination with the
    reference's options: termination type, iterations and accepted steps equal to the single-rank solve, cost to 1e-9, parameters
    to 1e-6, the replicated cameras and focal bitwise equal between the ranks ("world = 1, 2, 4, 8 logical
    shards on one device")."""
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    c1, p1, f1, s1 = bundle.ba_solve(*_ba_args(pb), opts=bundle.default_opts(), ctx=ctx)
    assert s1.termination == _lib.BA_CONVERGENCE
    res, grp = _logical_ranks_solve(pb, world)
    for r in res:
        s = r["s"]
        assert (s.termination, s.iterations, s.successful_steps) == (s1.termination, s1.iterations, s1.successful_steps), \
            (world, s.termination, s.iterations, s.successful_steps, s1.iterations, s1.successful_steps)
        assert abs(s.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost and abs(s.initial_cost - s1.initial_cost) <= 1e-12 * s1.initial_cost
        assert np.allclose(r["c"], c1, rtol=1e-6, atol=1e-9) and abs(r["f"] - f1) <= 1e-6 * f1
        assert np.array_equal(r["c"], res[0]["c"]) and r["f"] == res[0]["f"]          # replicas: the same bits on every rank
        assert s.final_cost == res[0]["s"].final_cost and s.spin_timeouts == 0
        assert r["tree"] == res[0]["tree"] and r["tree"]["fronts"] > 0               # the front tree of the union graph, on every rank
    assert [(r["lo"], r["hi"]) for r in res] == [sharding_block(100000, k, world) for k in range(world)]
    assert np.allclose(np.concatenate([r["p"] for r in res]), p1, rtol=1e-6, atol=1e-8)
    # every rank issued the same exchanges; the linearisations' is the sparse list (the 6 x 6 blocks of co-visible cameras), not
    # the packed triangle
    assert all(c == grp.counts[0] for c in grp.counts)
    ld = (6 * 200 + 1 + 63) // 64 * 64
    assert max(grp.counts[0]) < (ld * (ld + 1) // 2) // 2 and 8 in grp.counts[0]


def sharding_block(n, r, w):
    from sfm_danpipeline_amd import sharding
    return sharding.point_block(n, r, w)


@pytest.mark.gpu
def test_solver_on_a_busy_device_walks_the_same_iterates(ctx):
    """The reduced solve with the CUs contended: a second stream keeps the device full of k-NN sweeps while the LM
    iterations run, so the factorisation's workgroups do not start together.  (Until late in round 2 the owner
    workgroup of a chol_step2 launch overwrote the diagonal tiles the other panel workgroups read at their start: a
    panel workgroup that started late read L for D and the step came out wrong or NaN -- two processes on one GPU,
    or a second stream, were enough.)"""
    import torch
    from sfm_danpipeline_amd import matcher
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    ctx2 = _lib.Context(0, stream=side.cuda_stream)
    imgs = synth.sift_image_set(24, 2000, 128, seed=99)
    iset = matcher.ImageSet(imgs, ctx=ctx2)
    plan = matcher.MatchPlan(iset, synth.all_pairs(24))
    for shape in ((120, 12000, 8), (16, 6000, 6)):
        pb = synth.ba_problem(shape[0], shape[1], shape[2], seed=31)
        prob = bundle.BaProblem(shape[0], shape[1], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        quiet = prob.iterate(8)
        cq, pq, fq = prob.get_params()
        for rep in range(3):
            prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
            iset.prepare_async()
            for _ in range(12):                       # ~8 ms of sweeps queued on the other stream
                plan.run_async(0.8)
            busy = prob.iterate(8)
            cb, pb_, fb = prob.get_params()
            ctx2.synchronize()
            assert (busy.successful_steps, busy.iterations) == (quiet.successful_steps, quiet.iterations)
            assert abs(busy.final_cost - quiet.final_cost) <= 1e-9 * quiet.final_cost
            assert np.allclose(cb, cq, rtol=1e-9, atol=1e-12) and abs(fb - fq) <= 1e-9 * abs(fq)
        prob.close()
    plan.close()
    iset.close()
    ctx2.close()                                      # (after everything that lives on it)


def test_front_tree_under_real_contention_walks_the_same_bits(ctx):
    """The front tree's hand-offs lean on child fronts being resident when their parents poll (dispatch order is
    not a promise), and every front needs a whole compute unit's LDS.  Here the device is kept full of cfg2-sized k-NN sweeps on a
    second stream -- two workgroups of 67 KB of LDS on every compute unit -- while a 96-camera tree (7 fronts) runs its LM
    iterations: a front only gets a compute unit when a sweep workgroup leaves one.  Either no bounded poll runs out
    (spin_timeouts == 0) or the solve falls back to one launch per tree level; BOTH ways the iterates are the quiet run's, bit
    for bit -- a time-out is a status, never a trajectory."""
    import torch
    from sfm_danpipeline_amd import matcher
    side = torch.cuda.Stream(torch.device("cuda:0"))
    ctx2 = _lib.Context(0, stream=side.cuda_stream)
    imgs = synth.sift_image_set(50, 2000, 128, seed=98)
    iset = matcher.ImageSet(imgs, ctx=ctx2)
    plan = matcher.MatchPlan(iset, synth.all_pairs(50))          # 1225 pairs: 9800 workgroups per sweep, ~0.6 ms
    pb = synth.ba_problem(96, 9000, 5, seed=45)
    prob = bundle.BaProblem(96, 9000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    quiet = prob.iterate(10)
    assert prob.reduced_tree()["fronts"] >= 3 and quiet.spin_timeouts == 0
    cq, pq, fq = prob.get_params()
    seen = []
    for rep in range(4):
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        iset.prepare_async()
        for _ in range(16):                                       # ~10 ms of sweeps queued on the other stream
            plan.run_async(0.8)
        busy = prob.iterate(10)
        cb, pb_, fb = prob.get_params()
        seen.append(busy.spin_timeouts)
        assert (busy.successful_steps, busy.iterations) == (quiet.successful_steps, quiet.iterations), seen
        assert _eq_bits(cb, cq) and _eq_bits(pb_, pq) and fb == fq and busy.final_cost == quiet.final_cost, seen
    ctx2.synchronize()
    print("spin time-outs under contention:", seen)
    prob.close()
    plan.close()
    iset.close()
    ctx2.close()


def test_second_reduced_system_buffer_changes_nothing(ctx, monkeypatch):
    """The dissected solve's gather zeroes a second [S | g | ...] buffer on the side and the next linearisation swaps the two
    instead of filling one (SFMHIP_BA_PREZERO=0: a memset per linearisation): the same systems bit for bit, the same LM run --
    through accepted and rejected steps, a parameter reset in between, and a look at the reduced system afterwards"""
    pb = synth.ba_problem(80, 40000, 10, seed=23)
    monkeypatch.delenv("SFMHIP_BA_PREZERO", raising=False)
    a = bundle.BaProblem(80, 40000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    monkeypatch.setenv("SFMHIP_BA_PREZERO", "0")
    b = bundle.BaProblem(80, 40000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    monkeypatch.delenv("SFMHIP_BA_PREZERO")
    for prob in (a, b):
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    sa, sb = a.iterate(9), b.iterate(9)
    assert 0 < sa.successful_steps and (sa.successful_steps, sa.iterations) == (sb.successful_steps, sb.iterations)
    assert abs(sa.final_cost - sb.final_cost) <= 1e-12 * sb.final_cost
    for prob in (a, b):                                # a new start on the same objects
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    sa, sb = a.iterate(4), b.iterate(4)
    assert abs(sa.final_cost - sb.final_cost) <= 1e-12 * sb.final_cost
    (Sa, ga, ca), (Sb, gb, cb) = a.reduced_system(1e4), b.reduced_system(1e4)
    assert np.abs(Sa - Sb).max() <= 1e-12 * np.abs(Sb).max() and np.abs(ga - gb).max() <= 1e-12 * np.abs(gb).max()
    assert abs(ca - cb) <= 1e-12 * cb
    a.close()
    b.close()


def test_linearisation_is_bitwise_reproducible(ctx, monkeypatch):
    """The default epilogue: the elimination's workgroups store their sums in slabs and ba_gather_rows adds them in a fixed
    order -- S, g and the cost come out as the same bit patterns on every run; SFMHIP_BA_DETERMINISTIC=0 scatters with f64
    atomics instead (same values to ~1e-15, not the same bits), and the two agree"""
    pb = synth.ba_problem(60, 30000, 10, seed=19)
    monkeypatch.delenv("SFMHIP_BA_DETERMINISTIC", raising=False)
    det = bundle.BaProblem(60, 30000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    monkeypatch.setenv("SFMHIP_BA_DETERMINISTIC", "0")
    ref = bundle.BaProblem(60, 30000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    monkeypatch.delenv("SFMHIP_BA_DETERMINISTIC")
    runs = []
    for _ in range(3):
        det.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        runs.append(det.reduced_system(1e4))
    for S, g, cost in runs[1:]:
        assert np.array_equal(S.view(np.uint64), runs[0][0].view(np.uint64)) and np.array_equal(g.view(np.uint64), runs[0][1].view(np.uint64))
        assert cost == runs[0][2]
    ref.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    S1, g1, c1 = ref.reduced_system(1e4)
    assert np.abs(S1 - runs[0][0]).max() <= 1e-12 * np.abs(S1).max() and np.abs(g1 - runs[0][1]).max() <= 1e-12 * np.abs(g1).max()
    assert abs(c1 - runs[0][2]) <= 1e-13 * c1
    # and the LM iterates follow the atomic path's
    det.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    ref.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    sd, sr = det.iterate(5), ref.iterate(5)
    assert sd.successful_steps == sr.successful_steps and abs(sd.final_cost - sr.final_cost) <= 1e-10 * sr.final_cost
    det.close()
    ref.close()


def test_threaded_setup_groups_the_points_as_one_thread_does(ctx, monkeypatch):
    """sfmhip_ba_create groups the points by camera list on the host threads (per-block tables that meet in block order, a
    counting sort from per-block positions): with SFMHIP_BA_CHECK_SETUP the library recomputes the grouping on one thread, the
    way it was written before, and refuses the problem when run ids, order or offsets differ.  Shapes: one camera list per start
    camera (cfg3 / cfg4's), ragged lists of 2-10 cameras in thousands of combinations, unsorted input with points that have no
    observation."""
    monkeypatch.setenv("SFMHIP_BA_CHECK_SETUP", "1")
    rng = np.random.default_rng(12)
    pb = synth.ba_problem(60, 30000, 10, seed=19)
    bundle.BaProblem(60, 30000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx).close()
    # ragged: every point keeps a random subset (>= 2) of its ten views -> thousands of camera lists, most of them short runs
    keep = rng.random(len(pb["obs_cam"])) < 0.6
    keep[0::10] = True
    keep[1::10] = True
    bundle.BaProblem(60, 30000, pb["obs_cam"][keep], pb["obs_pt"][keep], pb["obs_xy"][keep], ctx=ctx).close()
    # the same observations in random order, every third point without any
    sel = np.flatnonzero(keep & (pb["obs_pt"] % 3 != 0))
    rng.shuffle(sel)
    pr = bundle.BaProblem(60, 30000, pb["obs_cam"][sel], pb["obs_pt"][sel], pb["obs_xy"][sel], ctx=ctx)
    pr.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    s = pr.iterate(2)
    assert np.isfinite(s.final_cost)
    pr.close()


def test_a_few_short_runs_become_pieces_of_the_elimination(ctx, orc, monkeypatch):
    """A camera list that only a handful of points share ("short run") used to send its points to the pair path -- four more
    launches per iteration for what may be ONE point (the per-view pattern of src/Sfm.cpp:996 produces exactly that).  While
    nothing else needs the pair path, up to 512 short runs are pieces of ba_eliminate_mfma instead: the same reduced system
    as the pair path's (SFMHIP_BA_SHORT_PIECES=0) to rounding, the oracle's trajectory, fewer launches."""
    pb = synth.ba_problem(43, 2500, 6, seed=43)
    keep = np.ones(len(pb["obs_cam"]), bool)
    keep[[6 * 17 + 2, 6 * 900 + 0, 6 * 2499 + 5]] = False          # three points lose one of their six views
    oc, op, xy = pb["obs_cam"][keep], pb["obs_pt"][keep], pb["obs_xy"][keep]
    out = {}
    for tag, env in (("pieces", None), ("pair_path", "0")):
        if env is None:
            monkeypatch.delenv("SFMHIP_BA_SHORT_PIECES", raising=False)
        else:
            monkeypatch.setenv("SFMHIP_BA_SHORT_PIECES", env)
        pr = bundle.BaProblem(43, 2500, oc, op, xy, ctx=ctx)
        pr.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        S, g, cost = pr.reduced_system(1e4)
        pr.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        s = pr.iterate(4)
        out[tag] = (S, g, cost, s, pr.last_timing()["launches"])
        pr.close()
    monkeypatch.delenv("SFMHIP_BA_SHORT_PIECES", raising=False)
    a, b = out["pieces"], out["pair_path"]
    assert np.abs(a[0] - b[0]).max() <= 1e-12 * np.abs(b[0]).max() and np.abs(a[1] - b[1]).max() <= 1e-12 * np.abs(b[1]).max()
    assert abs(a[2] - b[2]) <= 1e-13 * b[2]
    assert a[3].successful_steps == b[3].successful_steps and abs(a[3].final_cost - b[3].final_cost) <= 1e-10 * b[3].final_cost
    assert a[4] < b[4], (a[4], b[4])
    args = (pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy)
    c, p, f, s = bundle.ba_solve(*args, opts=bundle.default_opts(max_time_s=0.0), ctx=ctx)
    co, po, fo, so = orc.ba_solve(*args, opts=orc.default_opts(max_time_s=0.0))
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert abs(s.final_cost - so.final_cost) <= BA_COST_RTOL * so.final_cost
    assert np.allclose(c, co, rtol=BA_PARAM_RTOL, atol=1e-9) and np.allclose(p, po, rtol=BA_PARAM_RTOL, atol=1e-9)


@pytest.mark.parametrize("case", ["ragged", "many_per_camera"])
def test_linearisation_of_ragged_tracks_is_bitwise_reproducible(ctx, monkeypatch, case):
    """The same for the points the runs do not take (the pair path: ba_pp_points / ba_pp_pairs / ba_cam_blocks): ragged,
    shuffled tracks with a camera seen twice, and random visibility at 12 cameras x 60 000 points -- 25 000 observations per
    camera, which ba_cam_blocks cuts into slices of a workgroup each (their sums meet in slice order, whichever finishes
    last; one addend per entry of S from each of the three kernels, which follow each other in stream order)."""
    monkeypatch.delenv("SFMHIP_BA_DETERMINISTIC", raising=False)
    rng = np.random.default_rng(23)
    if case == "ragged":
        pb = synth.ba_problem(40, 20000, 8, seed=31)
        keep = rng.random(pb["n_obs"]) < 0.7
        oc, op, xy = pb["obs_cam"][keep], pb["obs_pt"][keep], pb["obs_xy"][keep]
        oc, op, xy = np.concatenate([oc, oc[:3]]), np.concatenate([op, op[:3]]), np.concatenate([xy, xy[:3] + 0.25])
        perm = rng.permutation(len(oc))
        oc, op, xy = oc[perm], op[perm], xy[perm]
        nc, npt = 40, 20000
    else:
        nc, npt = 12, 60000
        pb = synth.ba_problem(nc, npt, 5, seed=37)
        perm = rng.permutation(pb["n_obs"])          # (unsorted tracks: none of these points is taken by a run)
        oc, op, xy = pb["obs_cam"][perm], pb["obs_pt"][perm], pb["obs_xy"][perm]
    prob = bundle.BaProblem(nc, npt, oc, op, xy, ctx=ctx)
    runs = []
    for _ in range(4):
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        runs.append(prob.reduced_system(1e4))
    for S, g, cost in runs[1:]:
        assert np.array_equal(S.view(np.uint64), runs[0][0].view(np.uint64)) and np.array_equal(g.view(np.uint64), runs[0][1].view(np.uint64))
        assert cost == runs[0][2]
    prob.close()


# ---------------------------------------------------------------- the LM loop on the device (round 5)
def _solve_both_loops(monkeypatch, ctx, args, **kw):
    """The same solve with the trust-region decision on the device (default) and on the host (SFMHIP_BA_HOST_LOOP=1)."""
    out = []
    for host in ("0", "1"):
        monkeypatch.setenv("SFMHIP_BA_HOST_LOOP", host)
        out.append(bundle.ba_solve(*args, opts=bundle.default_opts(**kw), ctx=ctx))
    monkeypatch.delenv("SFMHIP_BA_HOST_LOOP")
    return out


def _same_bits(a, b):
    (c, p, f, s), (c2, p2, f2, s2) = a, b
    assert (s.termination, s.iterations, s.successful_steps) == (s2.termination, s2.iterations, s2.successful_steps)
    for k in ("initial_cost", "final_cost", "final_radius", "gradient_max_norm"):
        assert np.float64(getattr(s, k)).view(np.uint64) == np.float64(getattr(s2, k)).view(np.uint64), k
    assert np.array_equal(c.view(np.uint64), c2.view(np.uint64)) and np.array_equal(p.view(np.uint64), p2.view(np.uint64))
    assert np.float64(f).view(np.uint64) == np.float64(f2).view(np.uint64)
    assert s.spin_timeouts == 0 and s2.spin_timeouts == 0


@pytest.mark.parametrize("shape", [(6, 60, 4, 5), (20, 2000, 10, 7), (43, 2500, 6, 43), (50, 20000, 10, 777), (200, 20000, 10, 778)])
def test_device_loop_equals_host_loop(ctx, monkeypatch, shape):
    """lm_decide is ONE function compiled for both sides, the step evaluation's sums are added in a fixed order: the loop that
    runs on the device (the last workgroup of the step evaluation decides, the host reads records behind the GPU) and the
    host's loop walk the same trajectory BIT FOR BIT, to the same termination (TrustRegionMinimizer behind
    reference src/BundleAdjustment.cpp:115-123)."""
    nc, npt, k, seed = shape
    pb = synth.ba_problem(nc, npt, k, seed=seed)
    dev, host = _solve_both_loops(monkeypatch, ctx, _ba_args(pb), max_time_s=0.0)
    _same_bits(dev, host)
    assert dev[3].termination == _lib.BA_CONVERGENCE and dev[3].iterations >= 3


def test_device_loop_takes_every_exit_the_host_loop_takes(ctx, monkeypatch, orc):
    """Every termination path of the loop, device against host (bits) and against the oracle (counts): the iteration limit at
    1 ... 12 (between accepted and rejected steps, i.e. with and without an unread linearisation behind the last step), the
    function, parameter and gradient tolerances, the radius floor, invalid steps until FAILURE."""
    pb = synth.ba_problem(11, 600, 5, seed=11)
    args = _ba_args(pb)
    # the iteration limit on a start far enough from the optimum, under a radius large enough, that LM REJECTS steps for real
    # (oracle: iterations 1-3 accepted, 4-10 rejected at rho = -0.83 -- the cost nearly doubles --, 11-12 accepted): a limit that
    # falls on a solve already converged to the last bit compares rounding, not loops (the candidate cost equal to the cost to the
    # last bit is "function tolerance reached" even at tolerance 0, Ceres 1.13 trust_region_minimizer.cc, FunctionToleranceReached)
    far = _ba_args(synth.ba_problem(11, 600, 5, seed=13, pt_sigma=1.5, cam_sigma=0.4, focal_factor=1.0))
    for n in range(1, 13):
        kw = dict(max_iterations=n, function_tolerance=0.0, parameter_tolerance=0.0, initial_radius=1e10)
        dev, host = _solve_both_loops(monkeypatch, ctx, far, max_time_s=0.0, **kw)
        _same_bits(dev, host)
        so = orc.ba_solve(*far, opts=orc.default_opts(max_time_s=0.0, **kw))[3]
        assert (dev[3].termination, dev[3].iterations, dev[3].successful_steps) == (so.termination, so.iterations, so.successful_steps), kw
        assert so.termination == _lib.BA_NO_CONVERGENCE and so.iterations == n and so.successful_steps == (n if n < 4 else 3 if n < 11 else n - 7)
    cases = [dict(max_iterations=n, function_tolerance=0.0, parameter_tolerance=0.0) for n in range(1, 5)]
    cases += [dict(function_tolerance=1e-2), dict(parameter_tolerance=1e-3, function_tolerance=0.0),
              dict(gradient_tolerance=1e3, function_tolerance=0.0, parameter_tolerance=0.0),
              dict(gradient_tolerance=1e9),                       # (met at x0: no iteration at all)
              dict(min_radius=1e5),                               # (the initial radius is below the floor)
              dict(min_radius=3e4, function_tolerance=0.0, parameter_tolerance=0.0, max_iterations=30),
              dict(max_iterations=0)]
    seen = set()
    for kw in cases:
        dev, host = _solve_both_loops(monkeypatch, ctx, args, max_time_s=0.0, **kw)
        _same_bits(dev, host)
        so = orc.ba_solve(*args, opts=orc.default_opts(max_time_s=0.0, **kw))[3]
        assert (dev[3].termination, dev[3].iterations, dev[3].successful_steps) == (so.termination, so.iterations, so.successful_steps), kw
        seen.add(dev[3].termination)
    assert seen == {_lib.BA_CONVERGENCE, _lib.BA_NO_CONVERGENCE}
    # invalid steps: an observation that is not a number makes every candidate cost NaN
    bad = [a.copy() if isinstance(a, np.ndarray) else a for a in args]
    bad[5][3, 0] = np.nan
    dev, host = _solve_both_loops(monkeypatch, ctx, bad, max_time_s=0.0)
    assert dev[3].termination == host[3].termination == _lib.BA_FAILURE
    assert dev[3].iterations == host[3].iterations == 5 and dev[3].successful_steps == 0
    assert np.array_equal(dev[0], args[0]) and np.array_equal(dev[1], args[1])       # nothing was accepted


def test_device_loop_with_ragged_tracks_and_resumed_iterations(ctx, monkeypatch):
    """The pair path's points (ba_backsub next to ba_backsub_runs: the decision is ba_decide's then) and sfmhip_ba_iterate called
    a few iterations at a time: device loop == host loop, bit for bit."""
    rng = np.random.default_rng(5)
    pb = synth.ba_problem(24, 3000, 6, seed=31)
    keep = rng.random(len(pb["obs_cam"])) > 0.15                 # ragged: most signatures are no longer shared
    res = []
    for host in ("0", "1"):
        monkeypatch.setenv("SFMHIP_BA_HOST_LOOP", host)
        prob = bundle.BaProblem(24, 3000, pb["obs_cam"][keep], pb["obs_pt"][keep], pb["obs_xy"][keep], ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        sums = [prob.iterate(n) for n in (1, 3, 0, 5, 2)]
        res.append((prob.get_params(), sums))
        prob.close()
    monkeypatch.delenv("SFMHIP_BA_HOST_LOOP")
    (pa, sa), (pb_, sb) = res
    for x, y in zip(pa, pb_):
        assert np.array_equal(np.asarray(x).view(np.uint64), np.asarray(y).view(np.uint64))
    for x, y in zip(sa, sb):
        assert (x.iterations, x.successful_steps) == (y.iterations, y.successful_steps)
        assert np.float64(x.final_cost).view(np.uint64) == np.float64(y.final_cost).view(np.uint64)
        assert np.float64(x.final_radius).view(np.uint64) == np.float64(y.final_radius).view(np.uint64)
    assert sa[-1].iterations == 11 and sa[-1].final_cost < sa[0].initial_cost


def test_cfg4_full_size_runs_to_termination_vs_oracle(ctx, orc):
    """BASELINE cfg4 at full size, to the END of the solve: the reference's policy lives there (src/BundleAdjustment.cpp:118-129:
    500 iterations / 10 s / write-back only on CONVERGENCE).  Termination type, iterations, accepted steps, final cost (1e-9),
    parameters (1e-6), against the oracle."""
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    kw = dict(max_time_s=0.0)
    c, p, f, s = bundle.ba_solve(*_ba_args(pb), ctx=ctx, opts=bundle.default_opts(**kw))
    co, po, fo, so = orc.ba_solve(*_ba_args(pb), opts=orc.default_opts(**kw))
    assert so.termination == _lib.BA_CONVERGENCE
    assert (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    assert abs(s.final_cost - so.final_cost) <= BA_COST_RTOL * so.final_cost
    assert np.allclose(c, co, rtol=BA_PARAM_RTOL, atol=1e-9) and np.allclose(p, po, rtol=BA_PARAM_RTOL, atol=1e-9)
    assert abs(f - fo) <= BA_PARAM_RTOL * fo


_TIMEOUT_CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
from sfm_danpipeline_amd import _lib, bundle, synth
ctx = _lib.default_context()
nc, npt, k, seed, loop = (int(a) for a in sys.argv[1:6])
os.environ["SFMHIP_BA_HOST_LOOP"] = str(loop)
pb = synth.ba_problem(nc, npt, k, seed=seed)
args = (pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
try:
    c, p, f, s = bundle.ba_solve(*args, opts=bundle.default_opts(max_time_s=0.0, max_iterations=6), ctx=ctx)
    print("RESULT", s.termination, s.iterations, s.successful_steps, repr(s.final_cost), s.spin_timeouts, float(np.abs(c).sum()).hex())
except _lib.SfmHipError as e:
    print("ERROR", str(e))
'''


def _run_with_library(so, shape, host_loop, **extra_env):
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    env.pop("SFMHIP_SO", None)
    if so:
        env["SFMHIP_SO"] = so
    out = subprocess.run([sys.executable, "-c", _TIMEOUT_CHILD % root] + [str(v) for v in shape] + [str(host_loop)], env=env,
                         capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith(("RESULT", "ERROR"))]
    assert lines, out.stderr[-2000:]
    return lines[0]


@pytest.mark.parametrize("host_loop", [0, 1])
def test_a_timed_out_hand_off_between_fronts_is_repeated_level_by_level(host_loop):
    """A bounded spin that runs out is a scheduling artefact, not a matrix that is not positive definite (VERDICT round 4, weak 9;
    advisor): the radius is NOT shrunk.  In a diagnostic build whose first front never raises its flag in the one-launch form of
    the up-sweep, the solve reports it (summary.spin_timeouts), falls back to one launch per tree level, and walks the product
    build's trajectory bit for bit -- with the decision on the device and on the host."""
    import os
    from sfm_danpipeline_amd import build
    so = [p for p in build.build_timeout_diag() if p.endswith("breakfront.so")][0]
    shape = (96, 6000, 6, 45)                                     # (a ring of 96 cameras: the front tree)
    good = _run_with_library(None, shape, host_loop).split()
    bad = _run_with_library(so, shape, host_loop).split()
    assert good[0] == bad[0] == "RESULT", (good, bad)
    assert good[5] == "0" and bad[5] == "1"                        # spin_timeouts
    assert good[1:5] == bad[1:5] and good[6] == bad[6]             # termination, iterations, accepted steps, cost and cameras: same bits


def _timeout_rank_worker(rank, world, port, out_dir, so_bad):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from sfm_danpipeline_amd import _lib as L, bundle as B, sharding, synth as S
    assert L._lib is None                                          # (nothing has loaded the library in this process yet)
    L.SO = so_bad if rank == 1 else os.path.join(L.HERE, "libsfmhip.so")   # rank 1: the build with the hand-off that never arrives
    ctx = L.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    pb = S.ba_problem(96, 6000, 6, seed=45)
    loc = sharding.local_ba_problem(pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], pb["pts0"], rank, world)
    prob = B.BaProblem(96, len(loc["pts"]), loc["obs_cam"], loc["obs_pt"], loc["obs_xy"], ctx=ctx)
    ar = sharding.StagedAllReduce(device="cuda:0")
    prob.set_allreduce(ar, rank, world)
    prob.set_params(pb["cams0"], loc["pts"], pb["focal0"])
    s = prob.run(B.default_opts())
    c, p, f = prob.get_params()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), c=c, f=f, cost=s.final_cost, it=s.iterations, steps=s.successful_steps,
             term=s.termination, timeouts=s.spin_timeouts, fronts=prob.reduced_tree()["fronts"], n_exchanges=len(ar.counts))
    dist.barrier()
    dist.destroy_process_group()


def test_a_spin_time_out_on_one_rank_stops_every_rank(ctx, tmp_path):
    """A bounded wait that runs out is a scheduling artefact of ONE device; the ranks of a sharded solve must take the same decision
    all the same, or the one that stopped leaves its peers waiting in an all-reduce it never issues (advisor, round 5).  Two ranks
    share the device (gloo, staged exchange); rank 1 loads the diagnostic build whose first front never raises its flag in the
    one-launch up-sweep, rank 0 the product.  The flag travels with the step evaluation's sums: BOTH ranks report the time-out,
    both repeat the solve level by level, both issue the same exchanges, and the solve ends where the single-process solve does,
    the replicated cameras bitwise equal."""
    import socket
    import torch.multiprocessing as mp
    from sfm_danpipeline_amd import build
    so = [p for p in build.build_timeout_diag() if p.endswith("breakfront.so")][0]
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    mp.spawn(_timeout_rank_worker, args=(2, port, str(tmp_path), so), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["fronts"]) >= 3
    assert int(r0["timeouts"]) >= 1 and int(r1["timeouts"]) >= 1, (int(r0["timeouts"]), int(r1["timeouts"]))
    assert int(r0["n_exchanges"]) == int(r1["n_exchanges"])
    assert (int(r0["term"]), int(r0["it"]), int(r0["steps"])) == (int(r1["term"]), int(r1["it"]), int(r1["steps"]))
    assert np.array_equal(r0["c"], r1["c"]) and float(r0["f"]) == float(r1["f"]) and float(r0["cost"]) == float(r1["cost"])
    pb = synth.ba_problem(96, 6000, 6, seed=45)
    c1, p1, f1, s1 = bundle.ba_solve(*_ba_args(pb), opts=bundle.default_opts(), ctx=ctx)
    assert (s1.termination, s1.iterations, s1.successful_steps) == (int(r0["term"]), int(r0["it"]), int(r0["steps"]))
    assert abs(float(r0["cost"]) - s1.final_cost) <= 1e-9 * s1.final_cost
    assert np.allclose(r0["c"], c1, rtol=1e-6, atol=1e-9) and abs(float(r0["f"]) - f1) <= 1e-6 * f1


def test_a_spin_that_times_out_inside_a_workgroup_is_an_error_not_a_trajectory():
    """... and where no fallback exists (an LDS hand-off inside chol_step2, the dense factorisation: a bug, not a scheduling
    artefact) the caller gets SFMHIP_ERR_TIMEOUT instead of an LM run that silently took other steps."""
    from sfm_danpipeline_amd import build
    so = [p for p in build.build_timeout_diag() if p.endswith("breakchol.so")][0]
    shape = (50, 5000, 8, 5)
    assert _run_with_library(None, shape, 0, SFMHIP_BA_ND="0").startswith("RESULT")          # (SFMHIP_BA_ND=0: the dense factorisation)
    for host_loop in (0, 1):
        line = _run_with_library(so, shape, host_loop, SFMHIP_BA_ND="0")
        assert line.startswith("ERROR") and "status -8" in line, line
