"""Ad-hoc GPU experiment (run through gpurun): cfg2 sweeps of consecutive batches alternating between
two streams (two contexts, two resident image sets) against the same sweeps on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sfm_danpipeline_amd import _lib, matcher, synth

dev = torch.device("cuda:0")
imgs = synth.sift_image_set(50, 2000, 128, seed=1234)
pairs = synth.all_pairs(50)
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ctxs = [_lib.Context(0, stream=s.cuda_stream) for s in streams]
sets, plans, keep = [], [], []
for c in ctxs:
    d = [torch.from_numpy(a).to(dev) for a in imgs]
    keep.append(d)
    s = matcher.ImageSet(n_rows=[2000] * 50, dim=128, dtype=_lib.F32, norm=_lib.L2, ctx=c)
    for i, t in enumerate(d):
        s.adopt_device(i, t.data_ptr(), keepalive=t)
    sets.append(s)
    plans.append(matcher.MatchPlan(s, pairs))

def run(n, nstreams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        j = k % nstreams
        sets[j].prepare_async()
        plans[j].run_async(0.8)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for rep in range(3):
    a = run(24, 1)
    b = run(24, 2)
    c = run(24, 3)
    print(f"one stream {a*1e3:.4f} ms/sweep ({1225/a/1e6:.3f} M pairs/s)   two {b*1e3:.4f} ({1225/b/1e6:.3f})   three {c*1e3:.4f} ({1225/c/1e6:.3f})", flush=True)
c0 = plans[0].counts().sum(); c1 = plans[1].counts().sum()
print("matches", c0, c1)
