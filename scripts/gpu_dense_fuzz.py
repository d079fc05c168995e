"""Ad-hoc GPU check: the dense factorisation of large reduced systems (SFMHIP_BA_ND=0; X in diagonal blocks, block-by-block
backward substitution, deferred trailing updates) over camera counts around the switches (48 and 100 tile columns, last
blocks of every width), ring and random visibility: (S + D/r) z = g against numpy, twice (same bits)."""
import os, sys, time
os.environ["SFMHIP_BA_ND"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
cams = [250, 256, 261, 267, 272, 277, 283, 288, 336, 341, 342, 346, 352, 357, 363, 368, 373, 380, 528, 533, 534, 539, 544, 555, 640, 700] + [int(rng.integers(250, 760)) for _ in range(6)]
for case, nc in enumerate(cams):
    k = int(rng.integers(4, 10)); npt = int(rng.integers(3000, 9000)); kind = case % 2
    pb = synth.ba_problem(nc, npt, k, seed=100 + case)
    oc = pb["obs_cam"]
    if kind == 1:   # random visibility: k cameras drawn at random per point
        oc = np.concatenate([np.sort(rng.choice(nc, k, replace=False)) for _ in range(npt)]).astype(np.int32)
    prob = bundle.BaProblem(nc, npt, oc, pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    z, failed = prob.reduced_step(1e4)
    z2, _ = prob.reduced_step(1e4)
    S, g, _ = prob.reduced_system(1e4)
    zr = np.linalg.solve(S, g)
    res = np.linalg.norm(S @ z - g) / np.linalg.norm(g)
    err = np.abs(z - zr).max() / np.abs(zr).max()
    ok = failed == 0 and res <= 1e-11 and err <= 1e-8 and np.array_equal(z, z2)
    bad += not ok
    print(f"case {case}: {nc} cams (nt {(-(-(6 * nc + 1) // 64)) * 2}), k {k}, kind {kind}: residual {res:.1e} error {err:.1e} same bits {np.array_equal(z, z2)} {'ok' if ok else 'BAD'}", flush=True)
    prob.close()
print(f"done in {time.time() - t0:.0f} s: {bad} bad")
