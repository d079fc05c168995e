"""Ad-hoc (gpurun): host-side cost of sfmhip_ba_create / set_params / one-shot solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from sfm_danpipeline_amd import _lib, bundle, synth
ctx = _lib.default_context()
for (nc, npt, k, tag) in [(50, 20000, 10, "cfg3"), (200, 100000, 10, "cfg4")]:
    pb = synth.ba_problem(nc, npt, k, seed=777)
    for rep in range(2):
        t0 = time.perf_counter()
        prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        t1 = time.perf_counter()
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        t2 = time.perf_counter()
        s = prob.run(bundle.default_opts(max_time_s=0.0, max_iterations=20))
        t3 = time.perf_counter()
        c, p, f = prob.get_params()
        t4 = time.perf_counter()
        prob.close()
        print(f"{tag}: create {1e3*(t1-t0):.1f} ms, set_params {1e3*(t2-t1):.1f} ms, run {s.iterations} it {1e3*(t3-t2):.1f} ms, get {1e3*(t4-t3):.1f} ms")
