"""Ad-hoc GPU measurement (run through gpurun): throughput of the Hamming matcher at BASELINE cfg5's
shape (500 x 5000 x ORB-256, 124 750 pairs): one rank's share of the 8-way pair sharding, descriptors
resident in HBM, prepare + knn + compaction per sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from sfm_danpipeline_amd import _lib, matcher, sharding, synth

ctx = _lib.default_context()
imgs = synth.orb_image_set()
pairs = synth.all_pairs(len(imgs))
shards = sharding.shard_pairs(pairs, [len(a) for a in imgs], 8)
mine = pairs[shards[0]]
iset = matcher.ImageSet(imgs, norm=_lib.HAMMING, ctx=ctx)
plan = matcher.MatchPlan(iset, mine)
for rep in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    iset.prepare_async()
    plan.run_async(0.8)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"sweep {rep}: {len(mine)} pairs in {dt*1e3:.1f} ms -> {len(mine)/dt:.0f} pairs/s "
          f"({len(mine)*25e6/dt/1e12:.2f} T distances/s; all 124750 pairs on one GPU: {124750/len(mine)*dt:.2f} s)", flush=True)
ctx.set_timing(True)
iset.prepare_async(); plan.run_async(0.8)
tm = plan.last_timing()
ops = 2.0 * 5000 * 5000 * 256 * len(mine)
print(f"stages {tm}; knn kernel: {ops/tm['knn_s']/1e12:.0f} TOP/s of the 5000 dense i8 peak "
      f"({ops/tm['knn_s']/5e15:.3f}); matches {plan.counts().sum()}")
