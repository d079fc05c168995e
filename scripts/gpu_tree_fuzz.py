"""Ad-hoc GPU check (run through gpurun): the reduced solve on camera graphs of many shapes -- rings and open bands of varying
width, two rings joined by a few shared points, irregular track lengths -- each solved repeatedly: every answer must solve the
system the solver hands out and be the same bit pattern every time.  Prints the plan each problem got (front tree / chains /
dense)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
ctx = _lib.default_context()
rng = np.random.default_rng(seed0)
bad_total = 0
t_start = time.time()
for case in range(n_cases):
    nc = int(rng.choice([12, 24, 40, 64, 90, 130, 200, 260, 330, 420]))
    k = int(rng.integers(3, 11))
    npt = int(rng.integers(20, 60)) * nc
    pb = synth.ba_problem(nc, npt, k, seed=int(rng.integers(1 << 30)))
    oc, op, xy = pb["obs_cam"].copy(), pb["obs_pt"].copy(), pb["obs_xy"].copy()
    kind = int(rng.integers(0, 4))
    if kind == 1:      # an open band: drop the tracks that wrap around
        first = oc.reshape(-1, min(k, nc))[:, 0][op]
        last = oc.reshape(-1, min(k, nc))[:, -1][op]
        keep = (last - first) < min(k, nc)
        oc, op, xy = oc[keep], op[keep], xy[keep]
    elif kind == 2:    # irregular tracks: drop observations at random (keeps >= 2 per point where it can)
        keep = rng.random(len(oc)) < 0.8
        keep[np.unique(op, return_index=True)[1]] = True
        oc, op, xy = oc[keep], op[keep], xy[keep]
    elif kind == 3:    # every 7th point also seen by a far-away camera: long-range edges (wider separators, or none)
        far = (np.arange(npt) % 7 == 0) & (rng.random(npt) < 0.15)
        idx = np.nonzero(far)[0]
        if len(idx):
            firstcam = oc.reshape(-1, min(k, nc))[idx, 0]
            fc = ((firstcam + nc // 2) % nc).astype(np.int32)
            oc = np.concatenate([oc, fc]); op = np.concatenate([op, idx.astype(np.int32)]); xy = np.concatenate([xy, rng.normal(0, 50, (len(idx), 2))])
    try:
        prob = bundle.BaProblem(nc, npt, oc, op, xy, ctx=ctx)
    except Exception as e:
        print(f"case {case}: {nc} cams k {k} kind {kind}: create failed: {e}")
        continue
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    S, g, _ = prob.reduced_system(1e3)
    try:
        zr = np.linalg.solve(S, g)
    except np.linalg.LinAlgError:
        prob.close()
        continue
    z0, bad, odd, fails = None, 0, 0, 0
    for rep in range(reps):
        z, failed = prob.reduced_step(1e3)
        fails += int(failed != 0)
        if failed:
            continue
        if not np.abs(z - zr).max() <= 1e-8 * max(np.abs(zr).max(), 1e-300):
            bad += 1
        if z0 is None:
            z0 = z.copy()
        elif not np.array_equal(z, z0):
            odd += 1
    tree, lay = prob.reduced_tree(), prob.reduced_layout()
    plan = f"tree {tree['fronts']} fronts / {tree['levels']} levels / T {tree['max_front_tiles']}" if tree["fronts"] else (f"chains {lay['chains']}" if lay["chains"] else "dense")
    flag = "" if (bad == 0 and odd == 0 and fails == 0) else "   <<<<<<<<"
    print(f"case {case}: {nc} cams, k {k}, kind {kind}, cond {np.linalg.cond(S):.1e}: {plan}; {reps} solves: {bad} wrong, {odd} differ, {fails} reported failed{flag}", flush=True)
    bad_total += bad + odd
    prob.close()
print(f"done in {time.time() - t_start:.0f} s: {bad_total} bad answers")
