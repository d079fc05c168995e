"""Ad-hoc GPU check (run through gpurun): 20 LM iterations at three problem sizes, cost trajectory
and per-phase timing (the reduced solve is solve_s) -- the quick A/B while working on chol_step2.
(NOTE, late round 5: this script iterates ONE solve far past its convergence -- rejected steps, the radius collapsing to 0:
scripts/gpu_ba_radius_probe.py -- so its rates compare builds and shapes like with like but run 3-4 % above an LM iteration's;
scripts/gpu_ba_loop_ab.py and bench.py time iterations of a solve that still moves.)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib

ctx = _lib.default_context()
sizes = ((200, 100000, 10),) if "cfg4" in sys.argv[1:] else ((7, 400, 5), (50, 20000, 10), (200, 100000, 10))
for (nc, npt, k) in sizes:
    pb = synth.ba_problem(nc, npt, k, seed=777)
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    for rep in range(3):
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        t0 = time.time(); s = prob.iterate(20); dt = time.time() - t0
    ctx.set_timing(True)                 # stage breakdown: a second pass with the events on
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(20)
    tm = {k: (round(v * 1e3 / 20, 4) if k != "launches" else v) for k, v in prob.last_timing().items()}
    ctx.set_timing(False)
    print(f"{nc}/{npt}: 20 iterations {dt*1e3:.2f} ms -> {20/dt:.1f} it/s; cost {s.initial_cost:.12e} -> {s.final_cost:.12e}; succ {s.successful_steps}; per-iteration ms with stage timing on: {tm}", flush=True)
    prob.close()
