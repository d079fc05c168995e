#!/bin/bash
# the native-RCCL self-test N times (one rank declared as rank 0 of 2): an intermittent hang shows up as a time-out here
# usage: gpu_rccl_loop.sh [n]
n=${1:-20}
python3 - <<'PY'
import os, struct, sys
sys.path.insert(0, os.getcwd())
from sfm_danpipeline_amd import build, synth
build.build_rccl()
pb = synth.ba_problem(12, 3000, 6, seed=31)
with open("/tmp/rccl_pb.bin", "wb") as f:
    f.write(struct.pack("<iiii", 12, 3000, len(pb["obs_cam"]), 5))
    f.write(pb["cams0"].astype("<f8").tobytes()); f.write(pb["pts0"].astype("<f8").tobytes())
    f.write(struct.pack("<d", float(pb["focal0"])))
    f.write(pb["obs_cam"].astype("<i4").tobytes()); f.write(pb["obs_pt"].astype("<i4").tobytes()); f.write(pb["obs_xy"].astype("<f8").tobytes())
PY
ok=0; bad=0
for i in $(seq 1 $n); do
  s=$(date +%s.%N)
  if HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 60 sfm_danpipeline_amd/sfm_rccl_selftest /tmp/rccl_pb.bin /tmp/rccl_out.bin > /tmp/rccl.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i: rc $? after $(echo "$(date +%s.%N) - $s" | bc) s"; tail -5 /tmp/rccl.log; fi
done
echo "$ok ok, $bad bad of $n"
