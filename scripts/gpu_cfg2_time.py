"""cfg2 sweep timing (diagnostic): stages of N sweeps of the loaded library (SFMHIP_SO selects a build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, matcher, _lib
ctx = _lib.default_context()
ctx.set_timing(True)
imgs = synth.sift_image_set()
s = matcher.ImageSet(imgs, ctx=ctx)
pairs = synth.all_pairs(len(imgs))
pl = matcher.MatchPlan(s, pairs)
ks, kk = [], []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    s.prepare_async(); pl.run_async(0.8); ctx.synchronize()
    t = pl.last_timing(); ks.append(t["knn_s"]); kk.append(t["knn_kernel_s"])
print(os.path.basename(os.environ.get("SFMHIP_SO", "product")), "knn-kernel ms:", " ".join(f"{k*1e3:.3f}" for k in kk), "| stage ms:", " ".join(f"{k*1e3:.3f}" for k in ks), "prepare us %.1f compact us %.1f" % (t["prepare_s"]*1e6, t["compact_s"]*1e6), flush=True)
