"""Diagnostic (gpurun): one BA problem through the device loop and the host loop with the per-iteration log on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sfm_danpipeline_amd import synth, bundle, _lib
from oracle import orc
ctx = _lib.default_context()
rng = np.random.default_rng(4)
pb = synth.ba_problem(9, 250, 6, seed=11)      # tests/test_gpu_geometry.py::test_mixed_signatures_unsorted_and_repeated_cameras
keep = rng.random(pb["n_obs"]) < 0.7
keep[:6] = True
oc, op, xy = pb["obs_cam"][keep], pb["obs_pt"][keep], pb["obs_xy"][keep]
sel = ~((oc == 8) | (op == 17))
oc, op, xy = oc[sel], op[sel], xy[sel]
oc = np.concatenate([oc, oc[:1]]); op = np.concatenate([op, op[:1]]); xy = np.concatenate([xy, xy[:1] + 0.3])
perm = rng.permutation(len(oc)); oc, op, xy = oc[perm], op[perm], xy[perm]
for host in ("1", "0"):
    os.environ["SFMHIP_BA_HOST_LOOP"] = host
    print("host loop" if host == "1" else "device loop", flush=True)
    c, p, f, s = bundle.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy, opts=bundle.default_opts(max_time_s=0.0, verbose=1), ctx=ctx)
    print("  ->", s.termination, s.iterations, s.successful_steps, repr(s.final_cost), s.spin_timeouts, flush=True)
co, po, fo, so = orc.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy, orc.default_opts(max_time_s=0.0))
print("oracle ->", so.termination, so.iterations, so.successful_steps, repr(so.final_cost))
