#!/bin/bash
# rocprofv3 kernel trace of the cfg2 sweep timing script (diagnostic).  Usage: gpu_prof_cfg2.sh <outdir-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/scripts/gpu_cfg2_time.py 8 > $O/kt.log 2>&1
F=$(find $O -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $F | sed 's/(anonymous namespace):://g' | cut -c1-150 | head -12
tail -2 $O/kt.log
