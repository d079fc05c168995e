"""Ad-hoc GPU check: LM iteration rate at 200 cameras / 20 000 points / 200 000 observations with RANDOM visibility (each point
seen by ten cameras drawn at random: no separators, the reduced solve stays dense) next to the arcs of cfg3/cfg4's generator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
pb = synth.ba_problem(200, 20000, 10, seed=3)
rng = np.random.default_rng(0)
oc = np.concatenate([np.sort(rng.choice(200, 10, replace=False)) for _ in range(20000)]).astype(np.int32)
for name, cams in (("random", oc), ("arcs", pb["obs_cam"])):
    prob = bundle.BaProblem(200, 20000, cams, pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(3)
    ctx.set_timing(True)
    prob.iterate(2)
    tm = prob.last_timing()
    ctx.set_timing(False)
    t0 = time.time(); s = prob.iterate(20); dt = time.time() - t0
    print(f"{name:7s} visibility: {20/dt:.1f} it/s   layout {prob.reduced_layout()} tree {prob.reduced_tree()}  timing {tm}", flush=True)
    prob.close()
