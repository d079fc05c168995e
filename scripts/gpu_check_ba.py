"""Ad-hoc GPU check (run through gpurun): BA parity against the oracle + first timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import orc
from sfm_danpipeline_amd import synth, bundle, _lib

ctx = _lib.default_context()

ctx.set_timing(True)

def check_problem(nc, npt, k, seed, tag, run=True):
    pb = synth.ba_problem(nc, npt, k, seed=seed)
    args = (pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    S, g, cost = prob.reduced_system(1e4)
    So, go, costo, sc = orc.ba_reduced_system(*args, radius=1e4)
    print(f"{tag}: S rel {np.abs(S-So).max()/np.abs(So).max():.2e} g rel {np.abs(g-go).max()/np.abs(go).max():.2e} cost {cost:.12e} vs {costo:.12e}", flush=True)
    if not run: return
    t0 = time.time()
    s = prob.run(bundle.default_opts(max_time_s=0.0))
    tg = time.time() - t0
    c, p, f = prob.get_params()
    t0 = time.time()
    co, po, fo, so = orc.ba_solve(*args, opts=orc.default_opts(max_time_s=0.0))
    to = time.time() - t0
    print(f"   gpu: term {s.termination} it {s.iterations} succ {s.successful_steps} cost {s.initial_cost:.9e} -> {s.final_cost:.12e}  f={f:.9f}  {tg*1e3:.1f} ms")
    print(f"   orc: term {so.termination} it {so.iterations} succ {so.successful_steps} cost {so.initial_cost:.9e} -> {so.final_cost:.12e}  f={fo:.9f}  {to*1e3:.1f} ms")
    print(f"   max|dcam| {np.abs(c-co).max():.2e} max|dpt| {np.abs(p-po).max():.2e} |df| {abs(f-fo):.2e}  timing {prob.last_timing()}", flush=True)
    prob.close()

check_problem(6, 60, 4, 5, "tiny 6/60/4")
check_problem(8, 300, 2, 6, "two-view tracks 8/300/2")
check_problem(20, 2000, 10, 7, "20/2000/10")
check_problem(12, 500, 12, 8, "n=12 (generic kernel)")
check_problem(50, 20000, 10, 777, "cfg3 50/20k/200k")

# cfg4 timing
pb = synth.ba_problem(200, 100000, 10, seed=777)
prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
for rep in range(2):
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    t0 = time.time(); s = prob.iterate(10); dt = time.time() - t0
    print(f"cfg4 10 iterations: {dt*1e3:.2f} ms total -> {10/dt:.1f} it/s; cost {s.initial_cost:.6e} -> {s.final_cost:.6e}; succ {s.successful_steps}; {prob.last_timing()}", flush=True)
t0 = time.time(); tt, fc = orc.ba_time_iterations(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], 2)
print(f"oracle cfg4 2 iterations: {tt:.2f} s -> {2/tt:.3f} it/s final cost {fc:.6e}")
