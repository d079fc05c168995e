"""Ad-hoc GPU measurement: two independent cfg4 problems iterated concurrently (a context + stream + host thread each)
against one -- the reduced solve of an LM iteration is latency-bound on a handful of CUs, so a second problem's
elimination fits beside it.
(NOTE, late round 5: this script iterates ONE solve far past its convergence -- rejected steps, the radius collapsing to 0:
scripts/gpu_ba_radius_probe.py -- so its rates compare builds and shapes like with like but run 3-4 % above an LM iteration's;
scripts/gpu_ba_loop_ab.py and bench.py time iterations of a solve that still moves.)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sfm_danpipeline_amd import _lib, bundle, synth

n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctxs = [_lib.Context(0, stream=torch.cuda.Stream().cuda_stream) for _ in range(n_prob)]
probs = []
for k, c in enumerate(ctxs):
    pb = synth.ba_problem(200, 100000, 10, seed=777 + k)
    p = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=c)
    p.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    p.iterate(30)
    probs.append(p)
torch.cuda.synchronize()
t0 = time.perf_counter()
probs[0].iterate(iters)
torch.cuda.synchronize()
one = iters / (time.perf_counter() - t0)
ths = [threading.Thread(target=lambda p=p: p.iterate(iters)) for p in probs]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
both = n_prob * iters / (time.perf_counter() - t0)
print(f"one problem {one:.0f} it/s; {n_prob} problems concurrently {both:.0f} it/s in total ({both / one:.2f} x)")
