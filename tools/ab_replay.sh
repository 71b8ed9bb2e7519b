# A/B of pipelined-graph replays under environment switches on ONE lease: tools/ab_replay.sh OUT "ENV1" "ENV2" ... (dtype/batch via DT, BS)
O=$1; shift; mkdir -p $(dirname $O); : > $O
for rep in 1 2; do for e in "$@"; do
  echo "$e : $(env $e python tools/graph_replay.py ${DT:-bf16} ${BS:-64} ${STEPS:-60} 2>/dev/null | tail -1)" | tee -a $O
done; done
