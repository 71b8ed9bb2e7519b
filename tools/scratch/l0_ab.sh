#!/bin/bash
# fp32 bs 8 step with level0 on the direct MFMA kernel (0) / on F(4x4) (1), interleaved on one box
for r in 1 2 3; do for v in 0 1; do
  M3D_LEVEL0_WINO44=$v python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-configs2 --dump-layers gpurun_out/l0_$v.csv 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level0 wino44 $v', j['value'], j['ms_per_step'])"
done; done
grep "^level0\|^stem\|^level1" gpurun_out/l0_0.csv gpurun_out/l0_1.csv
