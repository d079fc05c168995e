"""Ad-hoc GPU A/B (run through gpurun): the linearisation stage at cfg4 for several workgroup shapes of
ba_eliminate_mfma (SFMHIP_BA_ELIM=waves,points), all in one process, interleaved over rounds (a fresh process
per variant measures the clock ramp, not the kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib

# a variant: "" (default), "waves,points" (SFMHIP_BA_ELIM), optionally prefixed "f0:" (SFMHIP_BA_ELIM_FILL=0: the plain cut)
variants = sys.argv[1:] or ["", "f0:", "8,500", "4,250", "2,100", "4,168", "2,125"]
ctx = _lib.default_context()
pb = synth.ba_problem(200, 100000, 10, seed=777)
probs = {}
for v in variants:
    fill0, shape = v.startswith("f0:"), v[3:] if v.startswith("f0:") else v
    if shape[:1] == "s" and shape[2:3] == ":":       # "sN:": N resident workgroups per CU for the slot-filling cut
        os.environ["SFMHIP_BA_ELIM_SLOTS"], shape = shape[1], shape[3:]
    else:
        os.environ.pop("SFMHIP_BA_ELIM_SLOTS", None)
    os.environ["SFMHIP_BA_ELIM_FILL"] = "0" if fill0 else "1"
    if shape:
        os.environ["SFMHIP_BA_ELIM"] = shape
    else:
        os.environ.pop("SFMHIP_BA_ELIM", None)
    probs[v] = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
res = {v: [] for v in variants}
ctx.set_timing(True)
for rnd in range(6):
    for v in variants:
        p = probs[v]
        p.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        s = p.iterate(20)
        res[v].append((p.last_timing()["eliminate_s"] / 20 * 1e6, s.final_cost))
for v in variants:
    t = [r[0] for r in res[v][1:]]
    print(f"{v or 'default':8s} eliminate stage us/iteration: min {min(t):.1f} median {np.median(t):.1f}  final cost {res[v][-1][1]:.9e}")
