#!/bin/bash
# Samples GPU clocks / power with rocm-smi while bench.py runs (ordinary user, read-only queries).
( for i in $(seq 1 12); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temp" | tr -s ' ' | tr '\n' ';'; echo; sleep 1; done ) > gpurun_out/clocks.log &
SMI=$!
timeout -k 10 200 python bench.py --steps 400 --warmup 5 --cpu-sample 0 > gpurun_out/bench_long.json 2>/dev/null
wait $SMI
tail -1 gpurun_out/bench_long.json | cut -c1-200
cut -c1-400 gpurun_out/clocks.log
