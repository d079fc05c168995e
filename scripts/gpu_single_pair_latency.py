"""Ad-hoc (gpurun): latency of the one-pair drop-in calls (getMatching / triangulate / merge)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from sfm_danpipeline_amd import _lib, matcher, synth, triangulate, incremental
ctx = _lib.default_context()
imgs = synth.sift_image_set(2, 2000, 128, seed=5)
for _ in range(3):
    matcher.get_matching(imgs[0], imgs[1], ctx=ctx)
t0 = time.perf_counter(); n = 50
for _ in range(n):
    mq, mt, md = matcher.get_matching(imgs[0], imgs[1], ctx=ctx)
print(f"getMatching 2000x2000 SIFT-128, host buffers in/out: {(time.perf_counter()-t0)/n*1e3:.3f} ms per call, {len(mq)} matches")
iset = matcher.ImageSet(imgs, ctx=ctx)
iset.prepare_async()
one = matcher.MatchPlan(iset, [[0, 1]])
for _ in range(3):
    one.set_pairs([[0, 1]]); one.run_async(0.8); one.fetch()
t0 = time.perf_counter()
for i in range(n):
    one.set_pairs([[i & 1, 1 - (i & 1)]]); one.run_async(0.8); c, q_, t_, d_ = one.fetch()
print(f"getMatching over a resident set (set_pairs + run + fetch): {(time.perf_counter()-t0)/n*1e3:.3f} ms per call")
sc = synth.two_view_scene(500, seed=1)
for _ in range(3):
    triangulate.triangulate_points(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"], sc["xy2"], ctx=ctx)
t0 = time.perf_counter()
for _ in range(n):
    triangulate.triangulate_points(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"], sc["xy2"], ctx=ctx)
print(f"triangulate 500 matches: {(time.perf_counter()-t0)/n*1e3:.3f} ms per call")
cloud = np.random.default_rng(0).uniform(-1, 1, (20000, 3)); new = np.random.default_rng(1).uniform(-1, 1, (500, 3))
for _ in range(3):
    incremental.merge_accept(cloud, new, ctx=ctx)
t0 = time.perf_counter()
for _ in range(n):
    incremental.merge_accept(cloud, new, ctx=ctx)
print(f"mergeNewPoints 20000 + 500: {(time.perf_counter()-t0)/n*1e3:.3f} ms per call")
