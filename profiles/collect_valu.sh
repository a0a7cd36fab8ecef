#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): SQ_INSTS_VALU / SQ_WAVES / SQ_INSTS_LDS of the Kraskov kernels at 256^3 x 64, k = 3 for the
# shipped dispatch and for the forced variants (env switches of kernels_kraskov.hip).  PMC pass only (no trace domains
# besides the kernel trace).  Usage: bash profiles/collect_valu.sh   -> gpurun_out/valu/
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/valu
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--measure mi_kraskov --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-host-boundary"
run() {  # tag, env assignments...
    local tag=$1; shift
    for kv in "$@"; do export "$kv"; done
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/$tag" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$tag.log" 2>&1
    for kv in "$@"; do unset "${kv%%=*}"; done
}
run default
run no_table CRF_KRASKOV_DXT=0
run tile_per_wave CRF_KRASKOV_DXT=0 CRF_KRASKOV_SHARE=0
run lds_column CRF_KRASKOV_TILE=1 CRF_KRASKOV_DXT=0
run sorted_column CRF_KRASKOV_SORTED=1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
print("| variant | kernel | launches | SQ_INSTS_VALU per wave | SQ_INSTS_LDS per wave | SQ_WAVES per launch |")
print("|---|---|---|---|---|---|")
for tag in ["default", "no_table", "tile_per_wave", "lds_column", "sorted_column"]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "kraskov" not in k or "prep" in k:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "SQ_WAVES":
                launches[k] += 1
    for k, c in acc.items():
        n = max(launches[k], 1)
        waves = c["SQ_WAVES"] / n
        print(f"| {tag} | {k[:60]} | {n} | {c['SQ_INSTS_VALU'] / max(c['SQ_WAVES'], 1):.0f} | {c['SQ_INSTS_LDS'] / max(c['SQ_WAVES'], 1):.0f} | {waves:.0f} |")
PY
