"""Diagnostic (gpurun): what sfmhip_ba_iterate (no convergence tests) is iterating ON once the solve has converged -- the radius,
cost and gradient after every batch -- and the rate per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
pb = synth.ba_problem(200, 100000, 10, seed=777)
prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
for n in (10, 10, 10, 10, 10, 10, 20, 20, 50, 50, 100, 100):
    ctx.synchronize(); t0 = time.perf_counter(); s = prob.iterate(n); ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"iterate({n:3d}): {dt / n * 1e3:.4f} ms/it  iterations {s.iterations:4d} accepted {s.successful_steps:3d} cost {s.final_cost:.12e} radius {s.final_radius:.3e} gmax {s.gradient_max_norm:.3e}", flush=True)
