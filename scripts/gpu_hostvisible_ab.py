"""Ad-hoc GPU A/B: host-visible sweep rate of the pipelined fetch for several sizes of the packing launch
(SFMHIP_PIPE_WGS), one process per variant.  usage: python scripts/gpu_hostvisible_ab.py [wgs ...]"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time
    import numpy as np
    import torch
    from sfm_danpipeline_amd import _lib, matcher, synth
    ctx = _lib.default_context()
    imgs = synth.sift_image_set(50, 2000, 128, seed=1234)
    pairs = synth.all_pairs(50)
    s = matcher.ImageSet(imgs, ctx=ctx)
    pl = matcher.MatchPlan(s, pairs)
    def sweep():
        s.prepare_async(); pl.run_async(0.8)
    for _ in range(300): sweep()
    ctx.synchronize() if hasattr(ctx, "synchronize") else torch.cuda.synchronize()
    def rate(fn, n=600):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): fn(i)
        torch.cuda.synchronize(); return n * len(pairs) / (time.perf_counter() - t0)
    r0 = rate(lambda i: sweep())
    pl.pipeline()
    def piped(i):
        sweep()
        if i: pl.fetch_wait(1)
    r1 = rate(piped)
    pl.pipeline(-1)
    r2 = rate(lambda i: sweep())
    print(f"WGS {os.environ.get('SFMHIP_PIPE_WGS','default'):>8s}: device-only {r0/1e6:.3f} M pairs/s, pipelined host-visible {r1/1e6:.3f} ({r1/r0:.3f}), device-only again {r2/1e6:.3f}")
else:
    for w in sys.argv[1:] or ["16", "64", "256", "1225"]:
        env = dict(os.environ, SFMHIP_PIPE_WGS=w)
        subprocess.run([sys.executable, __file__, "--child"], env=env)
