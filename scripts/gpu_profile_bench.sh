#!/bin/bash
# rocprofv3 evidence for bench.py --lean --match-streams 1 --cfg5-sample 2000 (the sample: 2000 pairs of cfg5 so that K2, knn_keyed_kernel, has rows; the timed regions and per-kernel samples; same kernels, same sizes;
# one stream, so that a kernel's average is not stretched by another batch's kernels running beside it):
# kernel trace + stats, then PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md
# prescribes).  Raw traces are deleted once condensed (gpurun_out is capped at 64 MiB).
# Usage: gpu_profile_bench.sh <tag>   -> gpurun_out/<tag>/{kernel_stats.csv,pmc_summary.csv,bench_profiled.json,bench.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --lean --match-streams 1 --cfg5-sample 2000 > $O/bench_profiled.json 2> $O/kt.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/kt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"  # (alone: with the TCC hit/miss counters the request "exceeds the capabilities of the hardware to collect")
P4="WRITE_SIZE SQ_WAVES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
i=0
FILES=""
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $P --kernel-trace -d $O/p$i -o p$i --output-format csv -- python3 $R/bench.py --lean --match-streams 1 --cfg5-sample 2000 --steps 4 --warmup 1 > $O/p$i.log 2>&1
  F=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp $F $O/p${i}_counters.csv && FILES="$FILES $O/p${i}_counters.csv"
  rm -rf $O/p$i
done
python3 $R/scripts/pmc_summary.py $O/pmc_summary.csv $FILES
rm -f $O/p*_counters.csv
python3 $R/bench.py --no-cpu-baseline --no-cfg5 > $O/bench.json 2> $O/bench.err
head -c 700 $O/pmc_summary.csv; echo; cut -d, -f1-4 $O/kernel_stats.csv | cut -c1-120 | head -8
