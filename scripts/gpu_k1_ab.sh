#!/bin/bash
# Same-box A/B of the k-NN kernel (K1): the product library against a build of another revision (scripts/build_rev_variant.sh
# <rev> <name> -> libsfmhip_dbg_<name>.so), interleaved so that both see the same clock history.  usage: gpu_k1_ab.sh <name> [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; rounds=${2:-3}
for i in $(seq $rounds); do
  python3 $R/scripts/gpu_cfg2_time.py 8
  SFMHIP_SO=$R/sfm_danpipeline_amd/libsfmhip_dbg_$name.so python3 $R/scripts/gpu_cfg2_time.py 8
done
