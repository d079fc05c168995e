"""CPU: the C oracle's triangulation and BA pieces against golden vectors / numpy / scipy."""
import numpy as np
import pytest

from oracle import np_check as nc
from sfm_danpipeline_amd import synth


def test_triangulate_golden(orc, golden):
    g = golden["triangulate"]
    X, err, keep = orc.triangulate(g["P1"], g["P2"], g["K"], g["dist"], g["xy1"], g["xy2"])
    assert np.array_equal(keep.astype(bool), g["keep"])
    assert np.allclose(X, g["X"], rtol=1e-9, atol=1e-10)
    assert np.allclose(err, g["err"], rtol=1e-5, atol=1e-5)


def test_triangulate_noise_free_recovers_points(orc):
    sc = synth.two_view_scene(50, seed=3, noise_px=0.0, outlier_frac=0.0)
    X, err, keep = orc.triangulate(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"], sc["xy2"])
    assert keep.all() and np.allclose(X, sc["X_true"], atol=1e-8) and err.max() < 1e-6


def test_undistort_iterations_with_distortion(orc):
    # distortion != 0 exercises the 5-iteration undistort + distorting projectPoints; a consistent
    # noise-free scene must still reproject to ~0 error
    sc = synth.two_view_scene(20, seed=4, noise_px=0.0, outlier_frac=0.0)
    dist = np.array([-0.05, 0.01, 1e-4, -2e-4, 0.0])
    K = sc["K"]

    def distort(xy):
        x = (xy[:, 0] - K[0, 2]) / K[0, 0]
        y = (xy[:, 1] - K[1, 2]) / K[1, 1]
        r2 = x * x + y * y
        cd = 1 + dist[0] * r2 + dist[1] * r2 ** 2 + dist[4] * r2 ** 3
        xd = x * cd + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
        yd = y * cd + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y
        return np.stack([xd * K[0, 0] + K[0, 2], yd * K[1, 1] + K[1, 2]], 1)

    X, err, keep = orc.triangulate(sc["P1"], sc["P2"], K, dist, distort(sc["xy1"]), distort(sc["xy2"]))
    assert keep.all() and err.max() < 1e-3 and np.allclose(X, sc["X_true"], atol=1e-4)


def test_rotation_helpers_roundtrip(orc):
    rng = np.random.default_rng(1)
    for _ in range(20):
        aa = rng.normal(size=3) * rng.choice([1e-9, 0.3, 2.5])
        R = orc.angleaxis_to_rotmat(aa)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-8)
        aa2 = orc.rotmat_to_angleaxis(R)
        assert np.allclose(orc.angleaxis_to_rotmat(aa2), R, atol=1e-12)
        x = rng.normal(size=3)
        assert np.allclose(orc.rotate_point(aa, x), nc.rotate_aa(aa, x), atol=1e-14)
    # near-pi rotation takes the trace<0 quaternion branch
    R = orc.angleaxis_to_rotmat(np.array([0.0, 3.1, 0.0]))
    assert np.allclose(orc.angleaxis_to_rotmat(orc.rotmat_to_angleaxis(R)), R, atol=1e-12)


def test_jacobian_golden_both_branches(orc, golden):
    g = golden["ba"]
    for cam, J, r in zip(g["jac_cams"], g["jac_J"], g["jac_r"]):
        r_, Jc, Jp, Jf = orc.ba_residual(cam, g["jac_X"], 1500.0, g["jac_obs"])
        assert np.allclose(r_, r, atol=1e-12)
        assert np.allclose(Jc, J[:, :6], rtol=1e-11, atol=1e-11)
        assert np.allclose(Jp, J[:, 6:9], rtol=1e-11, atol=1e-11)
        assert np.allclose(Jf, J[:, 9], rtol=1e-13, atol=1e-13)


def test_reduced_system_golden(orc, golden):
    g = golden["ba"]
    S, gg, cost, scale = orc.ba_reduced_system(g["cams0"], g["pts0"], float(g["focal0"]), g["obs_cam"], g["obs_pt"],
                                               g["obs_xy"], radius=1e4)
    assert abs(cost - float(g["cost0"])) < 1e-9 * cost
    assert np.allclose(scale, g["scale"], rtol=1e-12)
    assert np.abs(S - g["S"]).max() < 1e-11 * np.abs(g["S"]).max()
    assert np.abs(gg - g["g"]).max() < 1e-10 * np.abs(g["g"]).max()


def test_solve_reaches_scipy_optimum(orc, golden):
    g = golden["ba"]
    opts = orc.default_opts(max_time_s=0.0, function_tolerance=1e-14, parameter_tolerance=1e-14)
    c, p, f, s = orc.ba_solve(g["cams0"], g["pts0"], float(g["focal0"]), g["obs_cam"], g["obs_pt"], g["obs_xy"], opts)
    assert s.final_cost <= float(g["cost_opt"]) * (1 + 1e-6)
    assert s.final_cost >= float(g["cost_opt"]) * (1 - 1e-6)


def test_default_tolerances_terminate_with_convergence(orc, golden):
    g = golden["ba"]
    c, p, f, s = orc.ba_solve(g["cams0"], g["pts0"], float(g["focal0"]), g["obs_cam"], g["obs_pt"], g["obs_xy"],
                              orc.default_opts(max_time_s=0.0))
    assert s.termination == orc.CONVERGENCE and s.iterations < 50
    assert s.final_cost < s.initial_cost


def test_noise_free_problem_reaches_zero_cost(orc):
    pb = synth.ba_problem(6, 80, 4, seed=9, noise_px=0.0)
    c, p, f, s = orc.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"],
                              orc.default_opts(max_time_s=0.0, function_tolerance=1e-16, parameter_tolerance=1e-16,
                                               max_iterations=200))
    assert s.final_cost < 1e-12 * s.initial_cost


def test_max_iterations_gives_no_convergence(orc):
    pb = synth.ba_problem(6, 80, 4, seed=9)
    c, p, f, s = orc.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"],
                              orc.default_opts(max_time_s=0.0, max_iterations=1, function_tolerance=0.0,
                                               parameter_tolerance=0.0))
    assert s.termination == orc.NO_CONVERGENCE and s.iterations == 1


def test_observation_order_is_irrelevant(orc):
    pb = synth.ba_problem(6, 50, 4, seed=10)
    perm = np.random.default_rng(0).permutation(pb["n_obs"])
    a = orc.ba_reduced_system(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    b = orc.ba_reduced_system(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"][perm], pb["obs_pt"][perm],
                              pb["obs_xy"][perm])
    assert np.allclose(a[0], b[0], rtol=1e-12, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 130, 301, 1201])
def test_blocked_cholesky_of_the_cpu_baseline_solves_the_same_system(orc, n):
    """bench.py's timed cpu_baseline leg factors the reduced system with a blocked, vectorised right-looking Cholesky (what
    Eigen's LLT inside Ceres 1.13's DENSE_SCHUR is, reference src/BundleAdjustment.cpp:116) where the checker factors row by
    row: the same solution to 1e-12, against each other and against LAPACK; a matrix that is not positive definite is refused
    by both."""
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n + 5))
    S = M @ M.T + n * np.eye(n)
    b = rng.normal(size=n)
    x_row, x_blk, x_ref = orc.chol_solve(S, b, False), orc.chol_solve(S, b, True), np.linalg.solve(S, b)
    scale = np.abs(x_ref).max()
    assert np.abs(x_blk - x_row).max() <= 1e-12 * scale and np.abs(x_blk - x_ref).max() <= 1e-12 * scale
    S[n // 2, n // 2] = -1.0
    assert orc.chol_solve(S, b, True) is None and orc.chol_solve(S, b, False) is None


def test_blocked_cholesky_changes_only_the_timed_leg(orc):
    """The switch is the bench's: off by default (the parity tests compare with the row-by-row arithmetic), and an LM run with it
    on walks the same trajectory to rounding."""
    from sfm_danpipeline_amd import synth
    pb = synth.ba_problem(8, 400, 5, seed=2)
    args = (pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    a = orc.ba_solve(*args, opts=orc.default_opts(max_time_s=0.0))
    orc.ba_set_blocked_cholesky(True)
    try:
        b = orc.ba_solve(*args, opts=orc.default_opts(max_time_s=0.0))
    finally:
        orc.ba_set_blocked_cholesky(False)
    assert (a[3].termination, a[3].iterations, a[3].successful_steps) == (b[3].termination, b[3].iterations, b[3].successful_steps)
    assert np.allclose(a[0], b[0], rtol=1e-9, atol=1e-12) and np.allclose(a[1], b[1], rtol=1e-9, atol=1e-12)
    flops, seconds = orc.ba_cholesky_stats(reset=True)
    assert flops > 0 and seconds > 0
