#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes only (separate passes, as MI355X_MICROARCH.md prescribes), condensed per kernel:
#   scripts/gpu_pmc_fetch.sh <tag>  -> gpurun_out/<tag>/pmc_fetch_write.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
FILES=""
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $P --kernel-trace -d $O/$P -o $P --output-format csv -- python3 $R/bench.py --lean --steps 4 --warmup 1 > $O/$P.log 2>&1
  F=$(find $O/$P -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp $F $O/${P}_counters.csv && FILES="$FILES $O/${P}_counters.csv"
  rm -rf $O/$P
done
python3 $R/scripts/pmc_summary.py $O/pmc_fetch_write.csv $FILES
rm -f $O/*_counters.csv
cut -d, -f1-3 $O/pmc_fetch_write.csv | cut -c1-90 | head -5
python3 - "$O/pmc_fetch_write.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["kernel"][:48].ljust(48), {k: r[k] for k in r if k in ("FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_dispatch")})
PY
