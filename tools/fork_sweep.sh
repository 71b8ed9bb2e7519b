R=$PWD; O=$R/gpurun_out/r05j; mkdir -p $O
for f in 0 1 2 3 6; do echo "bf16 fork $f: $(M3D_PIPE_FORK=$f python tools/graph_replay.py bf16 64 60 2>/dev/null | tail -1)"; done | tee $O/fork_bf16.txt
for f in 0 1 2 3 4 8 12; do echo "f32 fork $f: $(M3D_PIPE_FORK=$f python tools/graph_replay.py f32 8 200 2>/dev/null | tail -1)"; done | tee $O/fork_f32.txt
