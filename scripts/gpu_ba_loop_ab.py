"""Same-box A/B of the cfg4 LM iteration rate (run through gpurun): this build with the loop on the device (default) and on the
host (SFMHIP_BA_HOST_LOOP=1), and -- when sfm_danpipeline_amd/libsfmhip_dbg_base.so exists (scripts/build_rev_variant.sh) -- the
build of another revision; child processes, alternating, three rounds.  Not product."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, os
sys.path.insert(0, %r)
import numpy as np
import torch
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
pb = synth.ba_problem(200, 100000, 10, seed=777)
prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
# (bench.py's region: from the start, 21 iterations untimed, the next 20 timed -- a solve that still accepts steps; past
# convergence sfmhip_ba_iterate iterates on rejected steps and a radius that collapses to 0, 3-4 percent faster and no LM iteration)
r = []
for rep in range(7):
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(21)
    ctx.synchronize()
    t0 = time.perf_counter(); s = prob.iterate(20); ctx.synchronize(); r.append((time.perf_counter() - t0) / 20)
print("RESULT", min(r) * 1e3, float(np.median(r)) * 1e3, s.iterations, s.successful_steps, repr(s.final_cost))
''' % ROOT
variants = [("device-loop", {}), ("host-loop", {"SFMHIP_BA_HOST_LOOP": "1"})]
# EXTRA_VARIANT="name:VAR=value[,VAR=value]": one more variant of THIS build under an environment
if os.environ.get("EXTRA_VARIANT"):
    nm, kv = os.environ["EXTRA_VARIANT"].split(":", 1)
    variants.append((nm, dict(x.split("=", 1) for x in kv.split(","))))
base = os.path.join(ROOT, "sfm_danpipeline_amd", "libsfmhip_dbg_base.so")
if os.path.exists(base):
    variants.append(("base-revision", {"SFMHIP_SO": base}))
for rnd in range(3):
    for name, env in variants:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(name, "FAILED", out.stderr[-800:]); continue
        f = line[0].split()
        print(f"round {rnd} {name:14s} iterations 22-41, 7 solves: min {float(f[1]):.4f} ms/it = {1e3/float(f[1]):7.1f} it/s   median {float(f[2]):.4f} ms/it = {1e3/float(f[2]):7.1f} it/s   "
              f"iterations {f[3]} accepted {f[4]} cost {f[5]}", flush=True)
