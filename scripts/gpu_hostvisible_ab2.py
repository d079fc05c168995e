"""Same-box A/B of the host-visible sweep rate as bench.py measures it (two plans on two streams, every sweep's lists taken
from the plan's previous run right before its next one): the zero-copy packing of round 3 (SFMHIP_PIPE_ZEROCOPY=1) against
packing on the device + a DMA copy (round 5).  One process per variant, alternating.  Not product."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time
    import numpy as np
    import torch
    from sfm_danpipeline_amd import _lib, matcher, synth
    dev = torch.device("cuda:0")
    imgs = synth.sift_image_set(50, 2000, 128, seed=1234)
    pairs = synth.all_pairs(50)
    streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
    ctxs = [_lib.Context(0, stream=s.cuda_stream) for s in streams]
    sets, plans, keep = [], [], []
    for c in ctxs:
        d = [torch.from_numpy(a).to(dev) for a in imgs]; keep.append(d)
        s_ = matcher.ImageSet(n_rows=[2000] * 50, dim=128, dtype=_lib.F32, norm=_lib.L2, ctx=c)
        for i, t in enumerate(d): s_.adopt_device(i, t.data_ptr(), keepalive=t)
        sets.append(s_); plans.append(matcher.MatchPlan(s_, pairs))
    n = [0]
    def step():
        j = n[0] % 2; n[0] += 1
        sets[j].prepare_async(); plans[j].run_async(0.8)
    def loop(fn, seconds=1.5):
        torch.cuda.synchronize(); k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(50): fn()
            k += 50; torch.cuda.synchronize()
        return k * len(pairs) / (time.perf_counter() - t0)
    loop(step, 0.5)
    r0 = loop(step)
    for p in plans: p.pipeline()
    seen = [0]
    def step_fetch():
        j = n[0] % 2
        if seen[0] >= 2:
            c_, q_, t_, d_ = plans[j].fetch_wait(back=0)
            assert int(c_.sum()) == len(q_)
        step(); seen[0] += 1
    r1 = loop(step_fetch)
    r1b = loop(step)                       # the packing + copy still run behind every sweep, nobody waits for them
    def step_fetch_back1():                # the lists of the plan's run BEFORE its last: two sweeps stay queued behind the one waited for
        j = n[0] % 2
        c_, q_, t_, d_ = plans[j].fetch_wait(back=1)
        assert int(c_.sum()) == len(q_)
        step()
    r1d = loop(step_fetch_back1)
    print(f"INFO fetch of the run before last (back=1): {r1d/1e6:.3f} M pairs/s ({r1d/r0:.3f})", flush=True)
    th = [0.0, 0.0, 0]
    def step_fetch_timed():
        j = n[0] % 2
        t0 = time.perf_counter()
        c_, q_, t_, d_ = plans[j].fetch_wait(back=0)
        t1 = time.perf_counter()
        step()
        th[0] += t1 - t0; th[1] += time.perf_counter() - t1; th[2] += 1
    r1c = loop(step_fetch_timed)
    print(f"INFO pipeline on, no fetch: {r1b/1e6:.3f} M pairs/s; host time per step: fetch_wait {th[0]/th[2]*1e6:.0f} us, enqueue {th[1]/th[2]*1e6:.0f} us (rate {r1c/1e6:.3f})", flush=True)
    ref = [synth.pair_checksums(*p.fetch()) for p in plans]
    for p, r in zip(plans, ref):
        assert np.array_equal(synth.pair_checksums(*p.fetch_wait(back=0)), r)
    for p in plans: p.pipeline(-1)
    r2 = loop(step)
    print(f"RESULT device-only {r0/1e6:.3f} M pairs/s | host-visible {r1/1e6:.3f} ({r1/r0:.3f}) | device-only again {r2/1e6:.3f}", flush=True)
else:
    for rnd in range(1):
        for name, env in (("dma-copy", {}), ("zero-copy", {"SFMHIP_PIPE_ZEROCOPY": "1"})):
            out = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, **env), capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            print(f"round {rnd} {name:10s}", line[0] if line else "FAILED " + out.stderr[-600:], flush=True)
            for l in out.stdout.splitlines():
                if l.startswith("INFO"): print("   ", l, flush=True)
