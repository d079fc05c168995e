"""Ad-hoc GPU check (run through gpurun): the front tree of the reduced solve (csrc/ba_front.h) -- (S + D/r) z = g against
numpy at several camera counts, then the LM iteration rate of cfg3 / cfg4 under SFMHIP_BA_ND = 2 (tree), 1 (chains), 0 (dense)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def child(mode):
    os.environ["SFMHIP_BA_ND"] = mode
    from sfm_danpipeline_amd import synth, bundle, _lib
    ctx = _lib.default_context()
    if mode == "2":
        for (nc, npt, k) in ((6, 300, 4), (50, 5000, 10), (96, 6000, 6), (200, 20000, 10), (560, 8000, 8)):
            pb = synth.ba_problem(nc, npt, k, seed=5)
            prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
            prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
            z, failed = prob.reduced_step(1e4)
            S, g, _ = prob.reduced_system(1e4)
            zr = np.linalg.solve(S, g)
            print(f"[tree] {nc} cams: {prob.reduced_tree()} failed {failed} residual {np.linalg.norm(S @ z - g) / np.linalg.norm(g):.3e} "
                  f"max |z - z_ref| / max |z_ref| {np.abs(z - zr).max() / np.abs(zr).max():.3e}", flush=True)
            prob.close()
    for (nc, npt, k) in ((50, 20000, 10), (200, 100000, 10)):
        pb = synth.ba_problem(nc, npt, k, seed=777)
        prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        for rep in range(3):
            prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
            t0 = time.time(); s = prob.iterate(20); dt = time.time() - t0
        ctx.set_timing(True)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        prob.iterate(20)
        tm = {k: (round(v * 1e3 / 20, 4) if k != "launches" else v) for k, v in prob.last_timing().items()}
        ctx.set_timing(False)
        print(f"[nd={mode}] {nc}/{npt}: {20/dt:.1f} it/s; cost {s.initial_cost:.12e} -> {s.final_cost:.12e}; succ {s.successful_steps}; "
              f"ms per iteration (events on): {tm}", flush=True)
        prob.close()

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for mode in ("2", "1", "0"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], timeout=600)
            print(f"[nd={mode}] exit {r.returncode}", flush=True)
