"""What a caller of BundleAdjustment::adjustBundle sees (reference include/BundleAdjustment.h:19-20, src/BundleAdjustment.cpp:46-175):
the whole call through the C++ mirror in the reference's containers -- pack, sfmhip_ba_create, solve, write-back -- at BASELINE
cfg3 and cfg4, first call (cold: the plan is built) and repeated calls on the same structure (the plan is kept).
usage: gpu_adjust_bundle_call.py [calls]"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_danpipeline_amd import build, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
exe = build.build_ba_demo()
for tag, (nc, npt, k) in (("cfg3", (50, 20000, 10)), ("cfg4", (200, 100000, 10))):
    pb = synth.ba_problem(nc, npt, k, seed=777)
    with tempfile.TemporaryDirectory() as d:
        synth.write_ba_containers(os.path.join(d, "in.bin"), pb, 960.0, 540.0)
        for env in ({}, {"SFMHIP_BA_PLAN_CACHE": "0"}):
            r = subprocess.run([exe, os.path.join(d, "in.bin"), os.path.join(d, "out.bin")], capture_output=True, text=True,
                               env=dict(os.environ, SFM_BA_SELFTEST_CALLS=str(calls), SFM_BA_SELFTEST_NEW_STRUCTURE="2", **env))
            assert r.returncode == 0, r.stderr[-2000:]
            its = [l for l in r.stdout.splitlines() if l.startswith("Bundle adjustment:")]
            for l in r.stdout.splitlines():
                if l.startswith("{"):
                    rec = json.loads(l)
                    print(tag, "plan cache off" if env else "plan cache on ", json.dumps(rec), "|", its[rec["call"]].split(",")[0])
