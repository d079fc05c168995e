import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
ctx.set_timing(True)
pb = synth.ba_problem(200, 20000, 10, seed=3)
rng = np.random.default_rng(0)
# random visibility: each point seen by 10 random distinct cameras (keep xy consistent by re-projecting is overkill: timing only)
oc = np.concatenate([np.sort(rng.choice(200, 10, replace=False)) for _ in range(20000)]).astype(np.int32)
prob = bundle.BaProblem(200, 20000, oc, pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
t0=time.time(); s=prob.iterate(3); print("random visibility 200/20k/200k: 3 iterations", time.time()-t0, prob.last_timing())
prob2 = bundle.BaProblem(200, 20000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob2.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
t0=time.time(); s=prob2.iterate(3); print("arc visibility   200/20k/200k: 3 iterations", time.time()-t0, prob2.last_timing())
