"""Diagnostic (gpurun): cfg2 sweep rate with consecutive batches alternating over 1, 2, 3, 4 HIP streams (bench.py uses 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_danpipeline_amd import _lib, matcher, synth
dev = torch.device("cuda:0")
imgs = synth.sift_image_set(50, 2000, 128, seed=1234)
pairs = synth.all_pairs(50)
ctxs, isets, plans, keep = [], [], [], []
for k in range(4):
    st = torch.cuda.Stream(dev)
    c = _lib.Context(0, stream=st.cuda_stream)
    d_imgs = [torch.from_numpy(a).to(dev) for a in imgs]
    s_ = matcher.ImageSet(n_rows=[2000] * 50, dim=128, dtype=_lib.F32, norm=_lib.L2, ctx=c)
    for i, t in enumerate(d_imgs):
        s_.adopt_device(i, t.data_ptr(), keepalive=t)
    ctxs.append(c); isets.append(s_); plans.append(matcher.MatchPlan(s_, pairs)); keep.append((st, d_imgs))
for n in (1, 2, 3, 4, 2, 3):
    def step(i):
        j = i % n
        isets[j].prepare_async(); plans[j].run_async(0.8)
    for i in range(60): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 200
    for i in range(N): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams {n}: {dt / N * 1e3:.4f} ms per sweep, {N * len(pairs) / dt / 1e6:.3f} M pairs/s", flush=True)
