"""Ad-hoc GPU repro (run through gpurun): the two-rank (gloo, one device) sharded BA of tests/test_gpu_geometry.py run
repeatedly after the parent has used the GPU, with the LM log of both ranks; prints the runs whose log differs from
the single-process log."""
import os, sys, socket, subprocess, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch.multiprocessing as mp
from sfm_danpipeline_amd import synth, bundle, _lib
from tests.test_gpu_geometry import _two_rank_worker


def child(rank, world, port, out_dir):
    sys.stderr = open(os.path.join(out_dir, f"log{rank}.txt"), "w")
    os.dup2(sys.stderr.fileno(), 2)
    _two_rank_worker(rank, world, port, out_dir)


if __name__ == "__main__":
    os.environ["SFMHIP_BA_VERBOSE"] = "1"
    ctx = _lib.default_context()
    pb = synth.ba_problem(16, 6000, 6, seed=44)
    for warm in range(3):        # the parent's own GPU work, as the tests before this one leave it
        big = synth.ba_problem(200, 20000, 10, seed=3 + warm)
        bundle.ba_solve(big["cams0"], big["pts0"], big["focal0"], big["obs_cam"], big["obs_pt"], big["obs_xy"],
                        opts=bundle.default_opts(max_time_s=0.0, max_iterations=3), ctx=ctx)
    bad = 0
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        out = f"/tmp/tr{rep}"
        os.makedirs(out, exist_ok=True)
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        mp.spawn(child, args=(2, port, out), nprocs=2, join=True)
        r0 = np.load(os.path.join(out, "rank0.npz"))
        lines = [l for l in open(os.path.join(out, "log0.txt")) if "sfmhip-ba" in l]
        ok = int(r0["steps"]) == 6
        print(rep, "steps", int(r0["steps"]), "cost %.9e" % float(r0["cost"]), "OK" if ok else "DIFFERENT", flush=True)
        if not ok:
            bad += 1
            print("".join(lines), flush=True)
    print("different runs:", bad)
