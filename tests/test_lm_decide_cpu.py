"""CPU: the trust-region decision of the bundle adjuster (`lm_decide`, csrc/ba.hip) through its host-side test hook
`sfmhip_ba_lm_decide` -- the SAME function the device runs at the end of every step evaluation, compiled for the host.  Checked
against a plain Python restatement of TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy of Ceres 1.13 with the options of
reference src/BundleAdjustment.cpp:115-121: every branch (accepted / rejected / invalid steps, the three
tolerances, the iteration limit, the radius floor, five invalid steps in a row, a time-out of the solve), and random walks through
them, state for state, bit for bit.  No GPU call: the library is only loaded."""
import ctypes as C
import math

import numpy as np
import pytest

from sfm_danpipeline_amd import _lib

RUNNING, CONVERGENCE, NO_CONVERGENCE, FAILURE, TIMEOUT = -1, 0, 1, 2, 100


def new_state(**kw):
    s = _lib.LmState()
    s.gradient_tolerance, s.parameter_tolerance, s.function_tolerance = 1e-10, 1e-8, 1e-6
    s.min_relative_decrease, s.max_radius, s.min_radius = 1e-3, 1e16, 1e-32
    s.max_consecutive_invalid, s.max_iterations, s.timing_only = 5, 500, 0
    s.radius, s.decrease_factor, s.cost, s.gradient_max_norm, s.x_norm = 1e4, 2.0, 100.0, 1.0, 10.0
    s.stop = RUNNING
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def inputs(**kw):
    i = _lib.LmInputs()
    i.lin_cost, i.lin_failed_blocks, i.lin_gradient_max = 100.0, 0.0, 1.0
    i.candidate_cost, i.model_cost_change, i.step_norm2, i.candidate_norm2, i.solve_info = 90.0, 10.0, 1.0, 101.0, 0
    for k, v in kw.items():
        setattr(i, k, v)
    return i


def decide(s, i):
    L = _lib.lib()
    L.sfmhip_ba_lm_decide.argtypes = [C.c_void_p, C.c_void_p]
    assert L.sfmhip_ba_lm_decide(C.addressof(s), C.addressof(i)) == 0
    return s


def restated(s, i):
    """The same step in plain Python (IEEE doubles, no fused operations), on a dict copy of the state."""
    d = {k: getattr(s, k) for k, _ in _lib.LmState._fields_}
    d["accepted"] = 0
    if d["stop"] != RUNNING:
        return d
    if i.solve_info < 0:
        d["stop"] = TIMEOUT
        return d
    d["iterations"] += 1
    t_only = d["timing_only"] != 0
    if d["lin_unread"]:
        d["lin_unread"] = 0
        d["cost"], d["gradient_max_norm"] = i.lin_cost, i.lin_gradient_max
        if not t_only and d["gradient_max_norm"] <= d["gradient_tolerance"]:
            d["iterations"] -= 1
            d["stop"] = CONVERGENCE
            return d
    finite = all(math.isfinite(v) for v in (i.step_norm2, i.model_cost_change, i.candidate_cost))
    bad = i.solve_info != 0 or i.lin_failed_blocks > 0 or not finite
    if bad or not (i.model_cost_change > 0.0):
        d["invalid_steps"] += 1
        if d["invalid_steps"] >= d["max_consecutive_invalid"] and not t_only:
            d["stop"] = FAILURE
            return d
        d["radius"] = d["radius"] / d["decrease_factor"]
        d["decrease_factor"] = d["decrease_factor"] * 2.0
    else:
        d["invalid_steps"] = 0
        step_norm = math.sqrt(i.step_norm2)
        if not t_only:
            if step_norm <= d["parameter_tolerance"] * (d["x_norm"] + d["parameter_tolerance"]):
                d["stop"] = CONVERGENCE
                return d
            if abs(d["cost"] - i.candidate_cost) <= d["function_tolerance"] * d["cost"]:
                d["stop"] = CONVERGENCE
                return d
        rho = (d["cost"] - i.candidate_cost) / i.model_cost_change
        if rho > d["min_relative_decrease"]:
            d["accepted"] = 1
            d["x_norm"] = math.sqrt(i.candidate_norm2)
            d["successful_steps"] += 1
            q = 2.0 * rho - 1.0
            d["radius"] = d["radius"] / max(1.0 / 3.0, 1.0 - q * q * q)
            d["radius"] = min(d["max_radius"], d["radius"])
            d["decrease_factor"] = 2.0
            d["cost"] = i.candidate_cost
            d["lin_unread"] = 1
        else:
            d["radius"] = d["radius"] / d["decrease_factor"]
            d["decrease_factor"] = d["decrease_factor"] * 2.0
    if not t_only:
        if d["iterations"] >= d["max_iterations"]:
            d["stop"] = NO_CONVERGENCE
        elif d["radius"] < d["min_radius"]:
            d["stop"] = CONVERGENCE
    return d


def same(s, d):
    for k, _ in _lib.LmState._fields_:
        a, b = getattr(s, k), d[k]
        if isinstance(b, float):
            assert np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64), (k, a, b)
        else:
            assert a == b, (k, a, b)


def test_an_accepted_step_grows_the_radius_and_leaves_the_linearisation_unread():
    s = decide(new_state(), inputs(candidate_cost=90.0, model_cost_change=10.0))          # rho = 1: radius x 3
    assert (s.accepted, s.stop, s.iterations, s.successful_steps, s.lin_unread) == (1, RUNNING, 1, 1, 1)
    assert s.radius == 3e4 and s.cost == 90.0 and s.decrease_factor == 2.0 and s.x_norm == math.sqrt(101.0)
    s = decide(new_state(radius=9e15), inputs())
    assert s.radius == 1e16                                                                # max_radius caps it


def test_rejected_and_invalid_steps_shrink_the_radius_by_a_growing_factor():
    s = new_state()
    for k, expect in enumerate((5e3, 1.25e3, 156.25)):                                     # / 2, / 4, / 8
        s = decide(s, inputs(candidate_cost=100.0 + 1.0, model_cost_change=10.0))          # rho < 0: rejected
        assert (s.accepted, s.stop, s.iterations, s.invalid_steps) == (0, RUNNING, k + 1, 0) and s.radius == expect
    s = new_state()
    for k in range(4):                                                                     # a pivot that was not positive: invalid
        s = decide(s, inputs(solve_info=3))
        assert (s.stop, s.invalid_steps, s.successful_steps) == (RUNNING, k + 1, 0)
    s = decide(s, inputs(solve_info=3))
    assert (s.stop, s.invalid_steps, s.iterations) == (FAILURE, 5, 5)                      # the fifth in a row: FAILURE
    for bad in (dict(candidate_cost=float("nan")), dict(model_cost_change=-1.0), dict(model_cost_change=0.0),
                dict(lin_failed_blocks=2.0), dict(step_norm2=float("inf"))):
        s = decide(new_state(), inputs(**bad))
        assert (s.accepted, s.invalid_steps) == (0, 1) and s.radius == 5e3
    s = decide(decide(new_state(), inputs(solve_info=3)), inputs())                        # a valid step resets the count
    assert s.invalid_steps == 0 and s.accepted == 1


def test_the_three_tolerances_the_limits_and_the_order_they_are_tested_in():
    s = decide(new_state(), inputs(step_norm2=1e-18))                                      # |step| <= ptol (|x| + ptol)
    assert (s.stop, s.accepted, s.iterations) == (CONVERGENCE, 0, 1)
    s = decide(new_state(), inputs(candidate_cost=100.0 - 5e-5))                           # |dcost| <= ftol cost
    assert (s.stop, s.accepted) == (CONVERGENCE, 0)
    s = decide(new_state(lin_unread=1), inputs(lin_gradient_max=1e-11))                    # the gradient, read late: the iteration does not count
    assert (s.stop, s.iterations, s.lin_unread, s.gradient_max_norm) == (CONVERGENCE, 0, 0, 1e-11)
    s = decide(new_state(lin_unread=1), inputs(lin_cost=95.0, candidate_cost=90.0, model_cost_change=5.0))
    assert s.accepted == 1 and s.cost == 90.0                                              # ... else rho is formed from the linearisation's cost
    s = decide(new_state(max_iterations=1), inputs())
    assert (s.stop, s.accepted, s.iterations) == (NO_CONVERGENCE, 1, 1)                    # the limit is tested behind the step
    s = decide(new_state(min_radius=6e3), inputs(candidate_cost=101.0))
    assert (s.stop, s.radius) == (CONVERGENCE, 5e3)                                        # the radius floor
    s = decide(new_state(max_iterations=1, min_radius=6e3), inputs(candidate_cost=101.0))
    assert s.stop == NO_CONVERGENCE                                                        # the iteration limit first
    t = decide(new_state(timing_only=1, max_iterations=1), inputs(step_norm2=1e-30))       # sfmhip_ba_iterate: no tests at all
    assert (t.stop, t.accepted) == (RUNNING, 1)


def test_a_stop_is_final_and_a_time_out_is_neither_an_iteration_nor_a_step():
    s = decide(new_state(max_iterations=1), inputs())
    before = {k: getattr(s, k) for k, _ in _lib.LmState._fields_}
    s = decide(s, inputs(candidate_cost=1.0))
    before["accepted"] = 0
    same(s, before)
    s = decide(new_state(), inputs(solve_info=-1))
    assert (s.stop, s.iterations, s.invalid_steps, s.radius, s.accepted) == (TIMEOUT, 0, 0, 1e4, 0)


@pytest.mark.parametrize("seed", range(8))
def test_random_walks_equal_the_restatement_state_for_state(seed):
    rng = np.random.default_rng(seed)
    s = new_state(max_iterations=int(rng.integers(20, 80)), timing_only=int(seed == 7))
    for step in range(200):
        kind = rng.random()
        i = inputs(lin_cost=s.cost * (1 + 1e-13 * rng.normal()), lin_gradient_max=float(10 ** rng.uniform(-12, 2)),
                   candidate_cost=s.cost * float(1 - 10 ** rng.uniform(-9, -0.5)) if kind < 0.7 else s.cost * 1.01,
                   model_cost_change=s.cost * float(10 ** rng.uniform(-9, -0.5)) * float(rng.uniform(0.3, 3.0)),
                   step_norm2=float(10 ** rng.uniform(-18, 2)), candidate_norm2=float(rng.uniform(50, 200)),
                   solve_info=int(rng.choice([0, 0, 0, 0, 0, 0, 0, 5, -1])), lin_failed_blocks=float(rng.random() < 0.03))
        want = restated(s, i)
        s = decide(s, i)
        same(s, want)
        if s.stop == TIMEOUT:
            s.stop = RUNNING                   # (what the host does: the solve is repeated level by level)
        elif s.stop != RUNNING:
            break
