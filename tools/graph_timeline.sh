R=$PWD; O=$R/gpurun_out/r05i; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "bf16 64" "f32 8"; do set -- $cfg
  python $R/tools/graph_replay.py $1 $2 40 > $O/replay_$1.json 2> $O/replay_$1.err
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1 -o t -- python $R/tools/graph_replay.py $1 $2 20 > $O/trace_$1.json 2> $O/trace_$1.err
  f=$(find $O/trace_$1 -name 't_kernel_trace.csv' | head -1)
  python $R/tools/graph_timeline.py $f --full > $O/timeline_$1.txt 2>&1
  rm -rf $O/trace_$1
done
cat $O/replay_bf16.json $O/replay_f32.json; head -30 $O/timeline_bf16.txt
