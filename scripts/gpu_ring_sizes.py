"""Ad-hoc GPU check: LM iteration rate over ring captures of growing camera count under the three factorisations of the
reduced system (SFMHIP_BA_ND = 2 front tree, 1 chains + separator, 0 dense), one process per mode."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(mode):
    os.environ["SFMHIP_BA_ND"] = mode
    from sfm_danpipeline_amd import synth, bundle, _lib
    ctx = _lib.default_context()
    for nc, npt in ((200, 100000), (400, 20000), (640, 12000), (1400, 9000)):
        if mode == "0" and nc > 640:
            continue
        pb = synth.ba_problem(nc, npt, 10 if nc == 200 else 8, seed=5)
        prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        prob.iterate(3)
        t0 = time.time(); s = prob.iterate(10); dt = time.time() - t0
        print(f"[nd={mode}] {nc} cameras / {npt} points: {10/dt:.1f} it/s  cost {s.final_cost:.10e}  tree {prob.reduced_tree()} layout {prob.reduced_layout()}", flush=True)
        prob.close()

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for mode in ("2", "1", "0"):
            subprocess.run([sys.executable, os.path.abspath(__file__), mode], timeout=900)
