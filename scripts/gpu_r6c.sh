#!/bin/bash
# round 6, job c: the K1 resolve rework + the BA host-side changes: tests, A/B, stamps, the adjustBundle call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_match.py -x -q > $O/match_tests.log 2>&1; tail -3 $O/match_tests.log
scripts/gpu_k1_ab.sh r5 3 > $O/k1_ab.txt 2>&1; cat $O/k1_ab.txt
SFMHIP_KNN_NW=4 SFMHIP_SO=$R/sfm_danpipeline_amd/libsfmhip_dbg4.so python scripts/gpu_knn_stamps.py > $O/stamps.log 2>&1; tail -8 $O/stamps.log
SFMHIP_PROFILE_CREATE=1 python scripts/gpu_ba_create_time.py > $O/create.log 2>&1; tail -22 $O/create.log
python scripts/gpu_adjust_bundle_call.py 3 > $O/adjust_bundle_call.txt 2>&1; cat $O/adjust_bundle_call.txt
python -m pytest tests/test_gpu_geometry.py tests/test_gpu_host_cpp.py -x -q > $O/ba_tests.log 2>&1; tail -15 $O/ba_tests.log
