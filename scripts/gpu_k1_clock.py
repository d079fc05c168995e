"""Diagnostic (gpurun): the k-NN kernel's time and the shader clock the chip holds WHILE the cfg2 sweeps run (a one-wave sampler on a
second stream, csrc/probe.hip), for the loaded library (SFMHIP_SO selects a build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_danpipeline_amd import _lib, matcher, synth
dev = torch.device("cuda:0")
side = torch.cuda.Stream(dev)
ctx = _lib.default_context()
ctx2 = _lib.Context(0, stream=side.cuda_stream)
imgs = synth.sift_image_set()
s = matcher.ImageSet(imgs, ctx=ctx)
pl = matcher.MatchPlan(s, synth.all_pairs(len(imgs)))
ctx.set_timing(True)
res = []
for rep in range(4):
    for _ in range(10):
        s.prepare_async(); pl.run_async(0.8)
    ctx.synchronize()
    ctx2.probe_clock_start(0.012)
    ks = []
    for _ in range(30):
        s.prepare_async(); pl.run_async(0.8)
    ghz = ctx2.probe_clock_read()
    ctx.synchronize()
    res.append((pl.last_timing()["knn_kernel_s"] * 1e3, ghz))
print(os.path.basename(os.environ.get("SFMHIP_SO", "product")), " ".join(f"[{k:.3f} ms @ {g:.3f} GHz]" for k, g in res), flush=True)
