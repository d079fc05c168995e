"""Ad-hoc GPU check: LM iteration rate with the DENSE factorisation of the reduced system (SFMHIP_BA_ND=0) over camera
counts; argv: camera counts (default 200 400 640).  Prints the cost too (an answer check against the other modes)."""
import os, sys, time
os.environ["SFMHIP_BA_ND"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
sizes = {200: 100000, 400: 20000, 640: 12000, 1400: 9000}
for nc in [int(a) for a in sys.argv[1:]] or [200, 400, 640]:
    npt = sizes.get(nc, 10000)
    pb = synth.ba_problem(nc, npt, 10 if nc == 200 else 8, seed=5)
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(3)
    n = 10 if nc <= 640 else 4
    t0 = time.time(); s = prob.iterate(n); dt = time.time() - t0
    print(f"[dense xb={os.environ.get('SFMHIP_BA_DENSE_XB')}] {nc} cameras / {npt} points: {n/dt:.1f} it/s  cost {s.final_cost:.10e}", flush=True)
    prob.close()
