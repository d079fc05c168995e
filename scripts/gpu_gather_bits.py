"""The reduced system's bit patterns (S, g, cost) of a few problems, as a digest per problem: run once per library
(SFMHIP_SO=...) and compare -- a change of the slab epilogue that keeps the order of the additions keeps every digest.
usage: gpu_gather_bits.py [time]   (time: also the LM iteration of each problem, iterations 6-25)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import bundle, synth

for nc, npt, k, seed in ((200, 100000, 10, 777), (60, 30000, 10, 19), (43, 2500, 6, 43), (400, 60000, 8, 5), (640, 100000, 10, 6), (1000, 100000, 10, 7)):
    pb = synth.ba_problem(nc, npt, k, seed=seed)
    pr = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    pr.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    S, g, cost = pr.reduced_system(1e4)
    h = hashlib.sha1(S.tobytes() + g.tobytes() + np.float64(cost).tobytes()).hexdigest()[:16]
    line = f"{nc:5d} cams {npt:7d} pts k {k:2d}: {h} cost {cost!r}"
    if len(sys.argv) > 1:
        pr.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        pr.iterate(5)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            pr.iterate(20)
            best = min(best, (time.perf_counter() - t0) / 20)
        line += f"  {best * 1e3:.4f} ms / iteration ({1.0 / best:.0f} it/s)"
    pr.close()
    # the same sums by the per-destination lists (another order of the additions: equal to rounding, not in bits)
    os.environ["SFMHIP_BA_GATHER_ROWS"] = "0"
    pr = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    del os.environ["SFMHIP_BA_GATHER_ROWS"]
    pr.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    S2, g2, cost2 = pr.reduced_system(1e4)
    pr.close()
    line += f"  | vs per-destination lists: S {np.abs(S - S2).max() / np.abs(S2).max():.1e} g {np.abs(g - g2).max() / np.abs(g2).max():.1e} cost {abs(cost - cost2) / cost2:.1e}"
    print(line, flush=True)
