#!/bin/bash
# rocprofv3 per-kernel averages of a cfg4 LM iteration under several environments ("-" = none, else "VAR=value[,VAR=value]" per
# argument); run through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  O=$R/gpurun_out/elim_env
  mkdir -p $O
  (
  if [ "$v" != "-" ]; then IFS=',' read -ra KV <<< "$v"; for kv in "${KV[@]}"; do export "$kv"; done; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/scripts/gpu_ba_iter_time.py cfg4 > $O/log.txt 2>&1
  )
  F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  echo "env $v"
  python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("  ", r["Name"].replace("(anonymous namespace)::", "")[:50].ljust(50), r["Calls"], r["AverageNs"])
PY
  tail -2 $O/log.txt
  rm -rf $O/kt
done
