import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
for shape in ((16, 6000, 6, 44), (96, 9000, 5, 45)):
    pb = synth.ba_problem(shape[0], shape[1], shape[2], seed=shape[3])
    for rep in range(2):
        c, p, f, s = bundle.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"],
                                     opts=bundle.default_opts(max_time_s=0.0, max_iterations=8, verbose=1), ctx=ctx)
        print(shape, "steps", s.successful_steps, "cost", s.final_cost, flush=True)
