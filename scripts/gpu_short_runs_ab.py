"""cfg4 with m points that lost one observation each (m camera lists of one point: "short runs"): the LM iteration with the short
runs on the pair path (SFMHIP_BA_SHORT_PIECES=0, every build before round 6's last) and as small pieces of the elimination.
usage: gpu_short_runs_ab.py  (re-runs itself per setting: the switch is read once per process)"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:
    import numpy as np
    from sfm_danpipeline_amd import bundle, synth
    m = int(sys.argv[1])
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    oc, op, xy = pb["obs_cam"], pb["obs_pt"], pb["obs_xy"]
    keep = np.ones(len(oc), bool)
    rng = np.random.default_rng(3)
    for p in rng.choice(100000, m, replace=False):
        keep[10 * p + rng.integers(0, 10)] = False      # (point-major, ten observations per point)
    prob = bundle.BaProblem(200, 100000, oc[keep], op[keep], xy[keep])
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(15)
    import time
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s = prob.iterate(20)
        best = min(best, (time.perf_counter() - t0) / 20)
    print(json.dumps({"short_runs": m, "pieces": os.environ.get("SFMHIP_BA_SHORT_PIECES", "512"), "ms_per_iteration": round(best * 1e3, 4),
                      "cost": s.final_cost}))
else:
    for m in (0, 1, 10, 100, 400, 2000):
        for pieces in ("0", "512", "4096"):
            r = subprocess.run([sys.executable, __file__, str(m)], env=dict(os.environ, SFMHIP_BA_SHORT_PIECES=pieces), capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else ("FAILED " + r.stderr[-500:]))
