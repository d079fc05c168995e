"""Ad-hoc GPU check (run through gpurun): LM trajectories of the device solver against the oracle's on small problems of many
shapes (rings, open bands, irregular and shuffled tracks, repeated cameras, long-range edges) -- same termination, iteration
and successful-step counts, final cost to 1e-9, parameters to the parity tolerances of tests/test_gpu_geometry.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib
from oracle import orc

orc.build()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = _lib.default_context()
rng = np.random.default_rng(seed0)
bad = 0
for case in range(n_cases):
    nc = int(rng.choice([5, 9, 14, 22, 36, 50, 64]))
    k = int(rng.integers(3, 11))
    npt = int(rng.integers(15, 50)) * nc
    pb = synth.ba_problem(nc, npt, k, seed=int(rng.integers(1 << 30)))
    oc, op, xy = pb["obs_cam"].copy(), pb["obs_pt"].copy(), pb["obs_xy"].copy()
    kind = int(rng.integers(0, 5))
    kk = min(k, nc)
    if kind == 1:
        cams = oc.reshape(-1, kk)
        keep = np.repeat((cams[:, -1] - cams[:, 0]) < kk, kk)
        oc, op, xy = oc[keep], op[keep], xy[keep]
    elif kind == 2:
        keep = rng.random(len(oc)) < 0.75
        keep[np.unique(op, return_index=True)[1]] = True
        oc, op, xy = oc[keep], op[keep], xy[keep]
    elif kind == 3:
        perm = rng.permutation(len(oc))
        oc, op, xy = oc[perm], op[perm], xy[perm]
    elif kind == 4:
        m = max(1, len(oc) // 200)
        oc, op, xy = np.concatenate([oc, oc[:m]]), np.concatenate([op, op[:m]]), np.concatenate([xy, xy[:m] + 0.3])
    iters = int(rng.integers(3, 12))
    opts = dict(max_iterations=iters, max_time_s=0.0)
    c, p, f, s = bundle.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy, opts=bundle.default_opts(**opts), ctx=ctx)
    co, po, fo, so = orc.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], oc, op, xy, orc.default_opts(**opts))
    same = (s.termination, s.iterations, s.successful_steps) == (so.termination, so.iterations, so.successful_steps)
    costok = abs(s.final_cost - so.final_cost) <= 1e-9 * max(so.final_cost, 1e-300)
    parok = np.allclose(c, co, rtol=1e-7, atol=1e-9) and np.allclose(p, po, rtol=1e-6, atol=1e-8) and abs(f - fo) <= 1e-8 * abs(fo)
    ok = same and costok and parok
    bad += 0 if ok else 1
    print(f"case {case}: {nc} cams, {npt} pts, k {k}, kind {kind}, {iters} it: device ({s.termination}, {s.iterations}, {s.successful_steps}) oracle ({so.termination}, {so.iterations}, {so.successful_steps}); "
          f"cost rel diff {abs(s.final_cost - so.final_cost) / max(so.final_cost, 1e-300):.1e}{'' if ok else '   <<<<<<<<'}", flush=True)
print(f"{bad} of {n_cases} differ")
