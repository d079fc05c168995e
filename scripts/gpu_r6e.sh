cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6e
python - <<PY
import os, subprocess, sys, tempfile
sys.path.insert(0, ".")
from sfm_danpipeline_amd import build, synth
exe = build.build_ba_demo()
pb = synth.ba_problem(200, 100000, 10, seed=777)
d = tempfile.mkdtemp()
synth.write_ba_containers(os.path.join(d, "in.bin"), pb, 960.0, 540.0)
for env in ({"SFMHIP_BA_PLAN_CACHE": "0"}, {"SFMHIP_BA_PLAN_CACHE": "0", "SFMHIP_BA_ARENA": "0"}):
    r = subprocess.run([exe, os.path.join(d, "in.bin"), os.path.join(d, "out.bin")], capture_output=True, text=True,
                       env=dict(os.environ, SFM_BA_SELFTEST_CALLS="3", SFMHIP_PROFILE_CREATE="1", **env))
    print(env); print(r.stderr[-2500:]); print("\n".join(l[:330] for l in r.stdout.splitlines() if l.startswith("{")))
PY
