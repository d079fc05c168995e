"""Diagnostic (gpurun): bench.py's timed BA region -- iterate(W + 20), then ONE iterate(20) between synchronisations -- repeated,
next to longer batches; SFMHIP_SO selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
pb = synth.ba_problem(200, 100000, 10, seed=777)
prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
prob.iterate(21)
for n in (20, 20, 20, 20, 100, 20, 20):
    torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    s = prob.iterate(n)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"iterate({n}): {dt / n * 1e3:.4f} ms/it  iterations {s.iterations} accepted {s.successful_steps} cost {s.final_cost!r} launches {prob.last_timing().get('launches')}", flush=True)
