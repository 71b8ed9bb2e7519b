#!/bin/bash
# PMC counter averages per kernel (one counter per rocprofv3 pass, never combined with other trace domains):
#   tools/pmc_counters.sh TAG [f32|bf16]   -> gpurun_out/TAG_{f32,bf16}_pmc_kernel_avgs.txt
TAG=${1:-r04}; DT=${2:-f32}; R=$(pwd); O=$R/gpurun_out/${TAG}_pmcavg_$DT; mkdir -p $O; export TMPDIR=/tmp
for C in MfmaUtil SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES VALUBusy; do
  (cd /tmp && PYTHONPATH=$R timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o $C -- python $R/bench.py --dtype $DT \
      --steps 2 --warmup 2 --no-cpu-baseline --no-configs2 --no-configs3 --no-feed --no-graph > $O/$C.log 2>&1)
done
python tools/pmc_kernel_avgs.py $O "" > $R/gpurun_out/${TAG}_${DT}_pmc_kernel_avgs.txt 2>&1
