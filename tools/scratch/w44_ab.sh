#!/bin/bash
# A/B of two builds of the library on the same box, interleaved: tools/scratch/w44_ab.sh A.so B.so
L=m3dssd_amd/csrc/build/libm3dssd_hip.so
cp $L /tmp/cur.so
for r in 1 2 3; do for v in A B; do
  [ $v = A ] && cp $1 $L || cp $2 $L
  python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-configs2 --dump-layers gpurun_out/ab_${v}.csv 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'])"
done; done
cp /tmp/cur.so $L
