#!/bin/bash
# images/s of both precisions as a function of the batch size (does a smaller working set -- MALL residency -- pay more than it costs in occupancy?)
for b in 16 32 48 64 96; do
  python bench.py --dtype bf16 --batch $b --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 bs $b', j['value'], j['ms_per_step'])"
done
for b in 4 8 16 32; do
  python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-configs2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 bs $b', j['value'], j['ms_per_step'])"
done
