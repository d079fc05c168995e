"""Diagnostic (gpurun): the cfg4 iterate() sequence through the device loop and the host loop, per-iteration log of both,
first line that differs."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
pb = synth.ba_problem(200, 100000, 10, seed=777)
prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
for n in (30, 20, 20, 7, 33):
    s = prob.iterate(n)
    print("CALL", n, s.iterations, s.successful_steps, repr(s.final_cost), repr(s.final_radius), file=sys.stderr, flush=True)
''' % ROOT
logs = {}
for name, env in (("device", {}), ("host", {"SFMHIP_BA_HOST_LOOP": "1"})):
    e = dict(os.environ); e.update(env); e["SFMHIP_BA_VERBOSE"] = "1"; e["SFMHIP_BA_VERBOSE_BITS"] = "1"
    out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    logs[name] = [l for l in out.stderr.splitlines() if l.startswith("[sfmhip-ba") or l.startswith("CALL")]
    print(name, len(logs[name]), "lines")
for i, (a, b) in enumerate(zip(logs["device"], logs["host"])):
    if a != b:
        print("first difference at line", i)
        for j in range(max(0, i - 9), min(len(logs["device"]), i + 2)):
            print("  D", logs["device"][j]); print("  H", logs["host"][j])
        break
else:
    print("identical logs")
