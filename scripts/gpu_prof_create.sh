set -e
python - $1 <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from sfm_danpipeline_amd import build, synth
exe = build.build_ba_demo()
import sys as _s; cfg = (50, 20000, 10) if len(_s.argv) > 1 else (200, 100000, 10); pb = synth.ba_problem(*cfg, seed=777)
d = tempfile.mkdtemp()
synth.write_ba_containers(os.path.join(d, "in.bin"), pb, 960.0, 540.0)
r = subprocess.run([exe, os.path.join(d, "in.bin"), os.path.join(d, "out.bin")], capture_output=True, text=True,
                   env=dict(os.environ, SFM_BA_SELFTEST_CALLS="3", SFM_BA_SELFTEST_NEW_STRUCTURE="3", SFMHIP_PROFILE_CREATE="1"))
print(r.stdout[-6000:]); print(r.stderr[-9000:])
PY
