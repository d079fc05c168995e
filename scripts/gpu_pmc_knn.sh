#!/bin/bash
# PMC passes over the cfg2 sweep timing script (diagnostic).  Usage: gpu_pmc_knn.sh <outdir-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P3="SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d $O/p$i -o p$i --output-format csv -- python3 $R/scripts/gpu_cfg2_time.py 4 > $O/p$i.log 2>&1
done
find $O -name "*counter_collection.csv" | head
python3 $R/scripts/pmc_summary.py $O/pmc_summary.csv $(find $O -name "*counter_collection.csv") 
head -3 $O/pmc_summary.csv
