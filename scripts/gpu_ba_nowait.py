"""Ad-hoc GPU check (measurement build: scripts/build_ba_variant.py nowait -DSFM_DBG_NOWAIT, loaded with SFMHIP_SO, + SFMHIP_DBG_NOWAIT=1): the cfg4 LM loop with the host
taken out of it -- every step accepted unseen, the next linearisation enqueued at once -- against the real loop: what the
publish kernel, the PCIe round trip, the host's decision and the launch latency after it cost per iteration."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
pb = synth.ba_problem(200, 100000, 10, seed=777)
prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
for rep in range(3):
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    n = 256
    t0 = time.time(); s = prob.iterate(n); dt = time.time() - t0
    print(f"nowait={os.environ.get('SFMHIP_DBG_NOWAIT')}: {n} iterations {dt*1e3:.2f} ms -> {n/dt:.1f} it/s ({dt/n*1e6:.1f} us each)", flush=True)
