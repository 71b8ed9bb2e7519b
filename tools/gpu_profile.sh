#!/bin/bash
# Standard measurement set on the GPU box (run through gpurun from the repo root):
#   tools/gpu_profile.sh TAG [f32|bf16|both] [stats|pmc|all]
# kernel stats: rocprofv3 --kernel-trace --stats of `bench.py --steps 10`; PMC: FETCH_SIZE / WRITE_SIZE in SEPARATE passes of
# `bench.py --steps 3 --warmup 2 --no-graph` (counters never combined with other trace domains), aligned per engine family
# with bench.py --dump-launches by tools/pmc_traffic.py.  Everything lands in gpurun_out/TAG_*.
TAG=${1:-r03}; WHAT=${2:-both}; MODE=${3:-all}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for DT in f32 bf16; do
  [ "$WHAT" != both ] && [ "$WHAT" != $DT ] && continue
  if [ "$MODE" = stats ] || [ "$MODE" = all ]; then
    (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_$DT -o $DT -- python $R/bench.py --dtype $DT --steps 10 --warmup 3 \
        --no-cpu-baseline --no-configs2 --no-configs3 --no-feed --no-dropin > $O/${TAG}_${DT}_under_rocprof.json 2> $O/${TAG}_${DT}_under_rocprof.err)
    DB=$(ls $O/${TAG}_prof_$DT/*/*_results.db $O/${TAG}_prof_$DT/*_results.db 2>/dev/null | head -1)
    python tools/rocpd_stats.py "$DB" $O/${TAG}_${DT}_bench_kernel_stats.csv
  fi
  if [ "$MODE" = pmc ] || [ "$MODE" = all ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${TAG}_pmc_$DT -o $C -- python $R/bench.py --dtype $DT \
          --steps 3 --warmup 2 --no-cpu-baseline --no-configs2 --no-configs3 --no-feed --no-dropin --no-graph --dump-launches $O/${TAG}_${DT}_launches.json \
          > $O/${TAG}_pmc_${DT}_$C.log 2>&1)
    done
    D=$O/${TAG}_pmc_$DT; [ -f $D/FETCH_SIZE_counter_collection.csv ] || D=$(dirname $(ls $D/*/FETCH_SIZE_counter_collection.csv | head -1))
    python tools/pmc_traffic.py $D $O/${TAG}_${DT}_hbm_traffic.json $O/${TAG}_${DT}_launches.json | tee $O/${TAG}_${DT}_hbm_traffic.txt
  fi
done
