import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import orc
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
for nc, npt in ((400, 20000), (640, 12000)):
    pb = synth.ba_problem(nc, npt, 8, seed=5)
    args = (pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    o = dict(max_time_s=0.0, max_iterations=2)
    t0 = time.time(); c, p, f, s = bundle.ba_solve(*args, opts=bundle.default_opts(**o), ctx=ctx); tg = time.time() - t0
    t0 = time.time(); co, po, fo, so = orc.ba_solve(*args, opts=orc.default_opts(**o)); to = time.time() - t0
    print(f"nc={nc}: gpu it {s.iterations} cost {s.final_cost:.12e} ({tg*1e3:.1f} ms) | oracle it {so.iterations} cost {so.final_cost:.12e} ({to:.1f} s) | max|dcam| {np.abs(c-co).max():.2e} max|dpt| {np.abs(p-po).max():.2e}", flush=True)
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(3); t0 = time.time(); prob.iterate(10); print(f"   {10/(time.time()-t0):.1f} it/s")
    prob.close()
