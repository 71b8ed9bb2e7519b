#!/bin/bash
# bf16 bs 64 step with / without the 128 x 128 wave-tile 3x3 kernel, interleaved on one box
for r in 1 2; do for v in 0 1; do
  M3D_BF16_WIDE=$v python bench.py --dtype bf16 --steps 40 --warmup 5 --no-cpu-baseline --dump-layers gpurun_out/wide_$v.csv 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide $v', j['value'], j['ms_per_step'])"
done; done
