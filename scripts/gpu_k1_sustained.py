"""Diagnostic (gpurun): the cfg2 sweep rate of the loaded library (SFMHIP_SO selects a build) the way bench.py's `sustained` leg
measures it -- consecutive batches alternating between two streams for two seconds -- and over the first 20 sweeps of the process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_danpipeline_amd import _lib, matcher, synth
dev = torch.device("cuda:0")
imgs = synth.sift_image_set(50, 2000, 128, seed=1234)
pairs = synth.all_pairs(50)
isets, plans, keep = [], [], []
for k in range(2):
    st = torch.cuda.Stream(dev)
    c = _lib.Context(0, stream=st.cuda_stream)
    d_imgs = [torch.from_numpy(a).to(dev) for a in imgs]
    s_ = matcher.ImageSet(n_rows=[2000] * 50, dim=128, dtype=_lib.F32, norm=_lib.L2, ctx=c)
    for i, t in enumerate(d_imgs):
        s_.adopt_device(i, t.data_ptr(), keepalive=t)
    isets.append(s_); plans.append(matcher.MatchPlan(s_, pairs)); keep.append((st, c, d_imgs))
NS = int(os.environ.get("K1_STREAMS", "2"))
def step(i):
    isets[i % NS].prepare_async(); plans[i % NS].run_async(0.8)
for i in range(4): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): step(i)
torch.cuda.synchronize()
first = (time.perf_counter() - t0) / 20
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    for i in range(50): step(i)
    n += 50
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(os.path.basename(os.environ.get("SFMHIP_SO", "product")), f"streams {NS}: first 20 sweeps {first * 1e3:.4f} ms each ({len(pairs) / first / 1e6:.3f} M pairs/s); sustained 2 s: {dt / n * 1e3:.4f} ms each ({n * len(pairs) / dt / 1e6:.3f} M pairs/s)", flush=True)
