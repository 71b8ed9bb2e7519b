#!/bin/bash
# PMC passes (one counter per pass) over the single-layer DCN bench
R=$(pwd); O=$R/gpurun_out/pmc_dcn; mkdir -p $O; export TMPDIR=/tmp
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE; do
  (cd /tmp && PYTHONPATH=$R timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o $C -- python $R/tools/bf16_dcn_bench.py 1.5 > $O/$C.log 2>&1)
done
python tools/pmc_kernel_avgs.py $O bf16_ 2>&1 | tail -40
