"""Ad-hoc GPU check (run through gpurun): the reduced system [S | g | cost] of randomly shaped problems against the oracle --
track lengths 2..10 mixed in one problem (several Gram widths), long and short runs (MFMA path and pair path side by side),
ring and random visibility -- in the default (slab epilogue + ba_gather_rows) and in the atomic mode.  Prints the worst
relative deviations; exits non-zero on a miss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib
from oracle import orc

orc.build()
ctx = _lib.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2024)
worst = [0.0, 0.0, 0.0]
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for case in range(n_cases):
    nc = int(rng.integers(12, 70))
    parts = []
    for _ in range(int(rng.integers(1, 4))):                      # blocks of points with their own track length
        k = int(rng.integers(2, min(10, nc - 1) + 1))
        npt = int(rng.choice([40, 300, 1500, 4000]))
        parts.append(synth.ba_problem(nc, npt, k, seed=int(rng.integers(1 << 30))))
    cams0, focal0 = parts[0]["cams0"], parts[0]["focal0"]
    pts0 = np.concatenate([p["pts0"] for p in parts])
    off = np.cumsum([0] + [len(p["pts0"]) for p in parts])
    oc = np.concatenate([p["obs_cam"] for p in parts])
    op = np.concatenate([p["obs_pt"] + off[i] for i, p in enumerate(parts)])
    xy = np.concatenate([p["obs_xy"] for p in parts])
    if case % 3 == 2:                                             # drop observations at random: ragged tracks, short runs
        keep = rng.random(len(oc)) < 0.8
        oc, op, xy = oc[keep], op[keep], xy[keep]
    perm = rng.permutation(len(oc))
    oc, op, xy = oc[perm].astype(np.int32), op[perm].astype(np.int32), xy[perm]
    for mode in ("1", "0"):
        os.environ["SFMHIP_BA_DETERMINISTIC"] = mode
        prob = bundle.BaProblem(nc, len(pts0), oc, op, xy, ctx=ctx)
        prob.set_params(cams0, pts0, focal0)
        for radius in (1e4, 7.0):
            S, g, cost = prob.reduced_system(radius)
            So, go, costo, _ = orc.ba_reduced_system(cams0, pts0, focal0, oc, op, xy, radius=radius)
            e = [abs(cost - costo) / costo, np.abs(S - So).max() / np.abs(So).max(), np.abs(g - go).max() / np.abs(go).max()]
            worst = [max(a, b) for a, b in zip(worst, e)]
            if e[0] > 1e-12 or e[1] > 1e-11 or e[2] > 1e-10:
                print("MISS case", case, "mode", mode, "nc", nc, "points", len(pts0), "obs", len(oc), "radius", radius, e)
                sys.exit(1)
        prob.close()
print(f"{n_cases} problems x 2 modes x 2 radii: worst relative deviation cost {worst[0]:.2e}  S {worst[1]:.2e}  g {worst[2]:.2e}")
