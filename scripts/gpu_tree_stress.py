"""Ad-hoc GPU check (run through gpurun): the front tree's reduced solve repeated many times on the same system -- every
answer must be the same bit pattern as the first and solve the system (a race between fronts shows as an odd one out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = _lib.default_context()
for (nc, npt, k) in ((200, 20000, 10), (560, 8000, 8), (96, 6000, 6)):
    pb = synth.ba_problem(nc, npt, k, seed=5)
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    S, g, _ = prob.reduced_system(1e4)
    zr = np.linalg.solve(S, g)
    z0, bad, odd = None, 0, 0
    for rep in range(reps):
        z, failed = prob.reduced_step(1e4)
        err = np.abs(z - zr).max() / np.abs(zr).max()
        if failed or not err < 1e-10:
            bad += 1
        if z0 is None:
            z0 = z.copy()
        elif not np.array_equal(z, z0):
            odd += 1
    print(f"{nc} cams: {reps} solves, {bad} wrong, {odd} differ from the first", flush=True)
    prob.close()
