#!/bin/bash
# The round's closing measurement set on one lease: tools/final_pass.sh TAG  (every step under its own timeout; outputs in gpurun_out/)
TAG=${1:-r05p}; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
timeout 1500 bash tools/gpu_profile.sh $TAG both all > $O/${TAG}_profile.log 2>&1
timeout 600 python bench.py --detail $O/${TAG}_bench_detail.json > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
timeout 600 bash tools/pmc_counters.sh $TAG bf16 > $O/${TAG}_pmc_counters.log 2>&1
timeout 600 bash tools/pmc_counters.sh $TAG f32 >> $O/${TAG}_pmc_counters.log 2>&1
timeout 300 python bench.py --dtype bf16 --steps 5 --no-cpu-baseline --no-configs2 --no-configs3 --no-feed --no-dropin --dump-layers $O/${TAG}_layers_bf16.csv > /dev/null 2>&1
timeout 300 python bench.py --dtype f32 --steps 5 --no-cpu-baseline --no-configs2 --no-configs3 --no-feed --no-dropin --dump-layers $O/${TAG}_layers_f32.csv > /dev/null 2>&1
for cfg in "bf16 64" "f32 8"; do set -- $cfg
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_$1 -o t -- python $R/tools/graph_replay.py $1 $2 20 > /dev/null 2>&1)
  f=$(find $O/${TAG}_trace_$1 -name 't_kernel_trace.csv' | head -1)
  python tools/graph_timeline.py $f --full > $O/${TAG}_timeline_$1.txt 2>&1; rm -rf $O/${TAG}_trace_$1
done
timeout 200 python tools/front2_trace.py > $O/${TAG}_front2_trace.txt 2>&1
timeout 200 python tools/bf16_head2_trace.py > $O/${TAG}_head2_trace.txt 2>&1
timeout 200 python tools/conv_wave_third_wave.py > $O/${TAG}_third_wave.txt 2>&1
timeout 100 python tools/anab_pool_bench.py 64 > $O/${TAG}_anab_pool.txt 2>&1; timeout 100 python tools/anab_pool_bench.py 8 f32 >> $O/${TAG}_anab_pool.txt 2>&1
timeout 100 python tools/dcn_radius_hist.py 8 > $O/${TAG}_dcn_radius.txt 2>&1
timeout 300 python tools/bf16_ap_agreement.py 64 8 $O/${TAG}_bf16_ap_agreement.json > $O/${TAG}_bf16_ap_agreement.txt 2>&1
timeout 100 python tools/nms_bench.py 64 3000 > $O/${TAG}_nms.txt 2>&1; M3D_NMS_DIV=1 timeout 100 python tools/nms_bench.py 64 3000 >> $O/${TAG}_nms.txt 2>&1
tail -3 $O/${TAG}_tests.log; python -c "
import json; d=json.load(open('$O/${TAG}_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_stale'], d['configs2_bf16']['ms_per_step'])"
