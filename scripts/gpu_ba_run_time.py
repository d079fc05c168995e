"""Ad-hoc GPU measurement: sfmhip_ba_run to termination (the product's entry point behind adjustBundle) with the decision on the
device and on the host, cfg3 and cfg4: wall time of the solve, iterations, and what a batch size costs in idle launches behind
the stop.  argv: batch sizes to try for the device loop (default 1 2 4 8)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np, torch
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
for (nc, npt) in ((50, 20000), (200, 100000)):
    pb = synth.ba_problem(nc, npt, 10, seed=777)
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    best = None
    for rep in range(4):
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = prob.run(bundle.default_opts(max_time_s=0.0))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("RESULT", nc, s.termination, s.iterations, s.successful_steps, "%%.3f" %% (best * 1e3), repr(s.final_cost))
    prob.close()
''' % ROOT
variants = [("host-loop", {"SFMHIP_BA_HOST_LOOP": "1"})] + [(f"device B={b}", {"SFMHIP_BA_LM_BATCH": b}) for b in (sys.argv[1:] or ["1", "2", "4", "8"])]
for name, env in variants:
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    for l in out.stdout.splitlines():
        if l.startswith("RESULT"):
            f = l.split()
            print(f"{name:14s} {f[1]:>4s} cameras: termination {f[2]} after {f[3]} iterations ({f[4]} accepted): {f[5]} ms   cost {f[6]}", flush=True)
    if "RESULT" not in out.stdout:
        print(name, "FAILED", out.stderr[-500:])
