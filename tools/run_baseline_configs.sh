#!/bin/bash
# Runs ON THE GPU BOX: one bench line per BASELINE.json config (+ the other estimators at the headline size) into
# gpurun_out/<tag>_baseline_configs.jsonl.   Usage: bash tools/run_baseline_configs.sh r03
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_baseline_configs.jsonl
: > $OUT
B="python bench.py --no-cpu-baseline --no-host-boundary"
timeout -k 10 120 $B --grid 64 64 64 --members 16 --measure pearson | tail -1 >> $OUT
timeout -k 10 200 python bench.py | tail -1 >> $OUT
timeout -k 10 200 $B --measure mi_kraskov --steps 10 --warmup 2 --repeats 3 | tail -1 >> $OUT
timeout -k 10 400 $B --grid 512 512 512 --members 128 --measure spearman --steps 10 --warmup 2 --repeats 3 | tail -1 >> $OUT
timeout -k 10 400 $B --grid 1024 1024 128 --members 256 --measure pearson --steps 10 --warmup 2 --repeats 3 | tail -1 >> $OUT
for m in spearman kendall mi_binned; do timeout -k 10 120 $B --measure $m | tail -1 >> $OUT; done
wc -l $OUT
