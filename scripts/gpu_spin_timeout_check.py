"""Ad-hoc GPU check of the spin-timeout reports (run through gpurun; tests/test_gpu_geometry.py has the same as tests):
with SFMHIP_SO = sfm_danpipeline_amd/libsfmhip_dbg_breakchol.so (an LDS hand-off of chol_step2 never arrives) the reduced step
comes back flagged (chol_failed = -1) and a run returns SFMHIP_ERR_TIMEOUT (-8) instead of shrinking the radius; with
libsfmhip_dbg_breakfront.so (a child front never raises its flag in the one-launch form) the run reports spin_timeouts = 1 and
goes on level by level; with the product library the flags stay 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib

ctx = _lib.default_context()
name = os.path.basename(os.environ.get("SFMHIP_SO", "product"))
pb = synth.ba_problem(50, 5000, 8, seed=5)
prob = bundle.BaProblem(50, 5000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
t0 = time.time()
z, failed = prob.reduced_step(1e4)
print(name, "reduced_step: chol_failed =", failed, "finite z:", bool(np.all(np.isfinite(z))), "%.1f ms" % ((time.time() - t0) * 1e3), flush=True)
try:
    s = prob.run(bundle.default_opts(max_time_s=0.0, max_iterations=5))
    print(name, "run: termination", s.termination, "iterations", s.iterations, "spin_timeouts", s.spin_timeouts, flush=True)
except _lib.SfmHipError as e:
    print(name, "run:", e, flush=True)
