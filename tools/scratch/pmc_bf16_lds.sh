#!/bin/bash
R=$(pwd); O=$R/gpurun_out/pmc_r03_bf16; mkdir -p $O; export TMPDIR=/tmp
for C in SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE MfmaUtil SQ_BUSY_CYCLES; do
  (cd /tmp && PYTHONPATH=$R timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o $C -- python $R/bench.py --dtype bf16 --steps 2 --warmup 2 --no-cpu-baseline --no-graph > $O/$C.log 2>&1)
done
python tools/pmc_kernel_avgs.py $O bf16_ > $R/gpurun_out/r03_bf16_pmc_kernel_avgs.txt 2>&1
O2=$R/gpurun_out/pmc_r03_f32; mkdir -p $O2
for C in SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS MfmaUtil SQ_BUSY_CYCLES; do
  (cd /tmp && PYTHONPATH=$R timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O2 -o $C -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-configs2 --no-graph > $O2/$C.log 2>&1)
done
python tools/pmc_kernel_avgs.py $O2 "" 2>&1 | grep -A5 -E "wino44|head_mlp_kernel<true|wino_wave|conv_wave_kernel<true, 4>" > $R/gpurun_out/r03_f32_pmc_kernel_avgs.txt
