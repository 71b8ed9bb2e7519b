#!/bin/bash
# step time of the fp32 bs 8 config under the F(4x4) switches: how many launches a folded cache touch may bridge
for sp in 1 2 3 6 0; do
  M3D_WINO44_TOUCH_SPAN=$sp python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-configs2 --dump-layers gpurun_out/w44_span${sp}.csv 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('span $sp', j['value'], j['ms_per_step'])"
done
