#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace statistics of the default benchmark command, then separate
# PMC passes for HBM traffic (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC slot limits, MI355X_MICROARCH.md).
# Usage: bash profiles/collect.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -eo pipefail
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-host-boundary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
python3 "$ROOT/tools/profile_summary.py" "$OUT" $ARGS > "$OUT/summary.md" 2>&1 || true
cat "$OUT/summary.md"
