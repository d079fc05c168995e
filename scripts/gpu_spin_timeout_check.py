"""Ad-hoc GPU check of the spin-timeout report (run through gpurun with SFMHIP_SO = a build with -DSFM_CHOL_BREAK_HANDOFF, in
which one LDS hand-off of chol_step2 never arrives): the reduced step must come back flagged (chol_failed = -1) instead of
silently wrong; with the product library the flag stays 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib

ctx = _lib.default_context()
pb = synth.ba_problem(50, 5000, 8, seed=5)
prob = bundle.BaProblem(50, 5000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
t0 = time.time()
z, failed = prob.reduced_step(1e4)
print(os.path.basename(os.environ.get("SFMHIP_SO", "product")), "chol_failed =", failed, "finite z:", bool(np.all(np.isfinite(z))), "%.1f ms" % ((time.time() - t0) * 1e3), flush=True)
