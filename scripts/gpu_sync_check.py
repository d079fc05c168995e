import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sfm_danpipeline_amd import synth, matcher, _lib
dev = torch.device("cuda:0")
ctx = _lib.Context(0, stream=torch.cuda.current_stream().cuda_stream)
imgs = synth.sift_image_set()
s = matcher.ImageSet(imgs, ctx=ctx)
pairs = synth.all_pairs(len(imgs))
pl = matcher.MatchPlan(s, pairs)
def step():
    s.prepare_async(); pl.run_async(0.8)
for _ in range(300): step()
torch.cuda.synchronize(dev)
for K in (20, 100, 20, 100):
    for mode in ("torch_sync", "ctx_sync", "event"):
        torch.cuda.synchronize(dev)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if mode == "event": e0.record()
        th = []
        for _ in range(K):
            step()
        t_enq = time.perf_counter() - t0
        if mode == "event": e1.record()
        if mode == "ctx_sync": ctx.synchronize()
        else: torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        extra = f" gpu-events {e0.elapsed_time(e1)/K:.4f} ms/step" if mode == "event" else ""
        print(f"K={K} {mode}: host {dt*1e3/K:.4f} ms/step (enqueue done after {t_enq*1e3:.2f} ms of {dt*1e3:.2f}){extra}", flush=True)
