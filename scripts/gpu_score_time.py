"""Ad-hoc GPU measurement (run through gpurun): sfmhip_score_essential over a batch of pairs shaped like the matches
of cfg2's all-pairs sweep (1225 pairs, a few hundred ratio-test survivors each, a share of wrong matches), next to the
numpy restatement on a sample of the pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import sfm_oracle_score as S
from sfm_danpipeline_amd import scoring, synth, _lib

ctx = _lib.default_context()
K = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1]])
rng = np.random.default_rng(0)
for n_pairs, lo, hi, outl in ((1225, 150, 900, (0.1, 0.5)), (1225, 150, 900, (0.5, 0.8))):
    pairs = []
    for p in range(n_pairs):
        sc = synth.two_view_scene(m=int(rng.integers(lo, hi)), seed=1000 + p, K=K, noise_px=0.4,
                                  outlier_frac=float(rng.uniform(*outl)))
        pairs.append((sc["xy1"], sc["xy2"]))
    scoring.score_essential(pairs[:8], K, ctx=ctx)                      # warm-up (module load, allocations)
    t0 = time.perf_counter()
    inl, _, its = scoring.score_essential(pairs, K, ctx=ctx)
    dt = time.perf_counter() - t0
    nm = sum(len(a) for a, _ in pairs)
    t1 = time.perf_counter()
    ref = [S.find_essential_mat_ransac(a, b, K) for a, b in pairs[:24]]
    dr = (time.perf_counter() - t1) / 24
    same = all((int(inl[i]), int(its[i])) == (ref[i][0], ref[i][3]) for i in range(24))
    print(f"{n_pairs} pairs, {nm} matches, outliers {outl}: {dt*1e3:.1f} ms -> {n_pairs/dt:.0f} pairs/s; iterations mean {its.mean():.1f} "
          f"max {its.max()}; numpy restatement {dr*1e3:.1f} ms per pair ({1/dr:.1f} pairs/s, 1 thread); first 24 pairs identical: {same}", flush=True)
