"""Ad-hoc GPU measurement: wall time of sfmhip_score_essential on the second batch of scripts/gpu_score_time.py (many
iterations per pair), for A/B runs of library builds (SFMHIP_SO)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import scoring, synth, _lib

ctx = _lib.default_context()
K = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1]])
rng = np.random.default_rng(0)
pairs = []
for p in range(400):
    sc = synth.two_view_scene(m=int(rng.integers(150, 900)), seed=1000 + p, K=K, noise_px=0.4, outlier_frac=float(rng.uniform(0.5, 0.8)))
    pairs.append((sc["xy1"], sc["xy2"]))
scoring.score_essential(pairs[:8], K, ctx=ctx)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); inl, _, its = scoring.score_essential(pairs, K, ctx=ctx); ts.append(time.perf_counter() - t0)
print(os.path.basename(os.environ.get("SFMHIP_SO", "product")), f"400 pairs: {min(ts)*1e3:.1f} ms; iterations mean {its.mean():.1f}; inliers sum {int(inl.sum())}", flush=True)
