#!/bin/bash
# Diagnostic: libsfmhip built from the sources of another revision (same-box A/B of two builds in ONE gpurun call:
#   SFMHIP_SO=sfm_danpipeline_amd/libsfmhip_dbg_<name>.so python3 scripts/gpu_ba_iter_time.py cfg4).  Not product.
# usage: scripts/build_rev_variant.sh <git-rev> <name>
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" sfm_danpipeline_amd/csrc include | tar -x -C "$tmp"
cd "$tmp/sfm_danpipeline_amd/csrc"
objs=""
for f in context match triangulate incremental score sift probe; do
  [ -f $f.hip ] || continue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -ffp-contract=off -c $f.hip -o $f.o 2>/dev/null &
  objs="$objs $f.o"
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -ffp-contract=fast-honor-pragmas -c ba.hip -o ba.o 2>/dev/null
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/sfm_danpipeline_amd/libsfmhip_dbg_$name.so" $objs ba.o
rm -rf "$tmp"
echo "$root/sfm_danpipeline_amd/libsfmhip_dbg_$name.so"
