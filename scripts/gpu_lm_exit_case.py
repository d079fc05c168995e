"""Diagnostic (gpurun): the trajectory of one small solve, device loop against the oracle, iteration by iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import _lib, bundle, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import orc
orc.build()
pb = synth.ba_problem(11, 600, 5, seed=11)
args = (pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
kw = dict(max_iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 6, function_tolerance=0.0, parameter_tolerance=0.0, max_time_s=0.0)
ctx = _lib.default_context()
s = bundle.ba_solve(*args, opts=bundle.default_opts(verbose=1, **kw), ctx=ctx)[3]
print("device", s.termination, s.iterations, s.successful_steps, repr(s.final_cost), s.gradient_max_norm)
so = orc.ba_solve(*args, opts=orc.default_opts(verbose=1, **kw))[3]
print("oracle", so.termination, so.iterations, so.successful_steps, repr(so.final_cost), so.gradient_max_norm)
