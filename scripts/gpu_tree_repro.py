"""Ad-hoc GPU check: repeated LM runs of one problem under the front tree; every run must walk the same iterates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib
nc, npt, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (50, 20000, 10)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ctx = _lib.default_context()
pb = synth.ba_problem(nc, npt, k, seed=777)
prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
for rep in range(reps):
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    t0 = time.time(); s = prob.iterate(20); dt = time.time() - t0
    print(f"rep {rep}: {dt*1e3:.2f} ms succ {s.successful_steps} cost {s.final_cost:.12e} tree {prob.reduced_tree()}", flush=True)
