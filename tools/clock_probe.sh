#!/bin/bash
# Samples rocm-smi (sclk, power) while one kernel family runs in a loop: bash tools/clock_probe.sh <python cmd ...>
# usage on the GPU box: bash tools/clock_probe.sh python tools/conv_wave_loop.py 1
"$@" &
PID=$!
sleep 4
for i in 1 2 3 4; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '
    echo
    sleep 0.7
done
wait $PID
