"""Ad-hoc GPU check: the two slab epilogues (ba_gather_rows / ba_gather_slabs, SFMHIP_BA_GATHER_ROWS=0) over problem shapes:
LM iteration rate and the linearisation stage's time.  One process per setting (the switch is read at plan time)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child():
    from sfm_danpipeline_amd import synth, bundle, _lib
    ctx = _lib.default_context()
    for nc, npt, k in ((200, 100000, 10), (200, 20000, 10), (400, 20000, 8), (400, 60000, 8), (640, 12000, 8), (640, 60000, 8), (1000, 20000, 8)):
        pb = synth.ba_problem(nc, npt, k, seed=5)
        prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
        prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
        prob.iterate(3)
        ctx.set_timing(True); prob.iterate(2); tm = prob.last_timing(); ctx.set_timing(False)
        t0 = time.time(); prob.iterate(20); dt = time.time() - t0
        print(f"[rows={os.environ.get('SFMHIP_BA_GATHER_ROWS', '1')}] {nc:5d} cameras {npt:7d} points: {20/dt:7.1f} it/s  eliminate {tm['eliminate_s']*1e6/2:6.1f} us", flush=True)
        prob.close()

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for v in ("1", "0"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, SFMHIP_BA_GATHER_ROWS=v), timeout=600)
