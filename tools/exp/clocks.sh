#!/bin/bash
# samples GPU clock / power while the bench runs (is the step power-limited?)
mkdir -p gpurun_out/r3
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/r3/clocks.txt &
SM=$!
sleep 1
python bench.py --no-cpu-baseline --no-observe --steps 1200 --warmup 20 > gpurun_out/r3/clocks_bench.json 2> gpurun_out/r3/clocks_bench.err
wait $SM
python -c "
import json; b=json.load(open('gpurun_out/r3/clocks_bench.json')); print(b['value'], b['ms_per_step'])"
cat gpurun_out/r3/clocks.txt | cut -c1-260 | head -60
