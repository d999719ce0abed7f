#!/bin/bash
# shader / memory clocks and power while bench.py runs (rocm-smi polled in the background every ~0.1 s)
mkdir -p gpurun_out/r03
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' \t' ' ' | tr '\n' '|'; echo; sleep 0.05; done ) > gpurun_out/r03/clock_watch.txt &
W=$!
python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r03/bench_clock.json 2>/dev/null
kill $W 2>/dev/null
sort gpurun_out/r03/clock_watch.txt | uniq -c | sort -rn | head -12
