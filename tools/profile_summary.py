#!/usr/bin/env python3
"""Summarises a profiles/collect.sh output directory: per-kernel duration statistics from the rocprofv3 kernel trace
and HBM traffic per launch from the PMC passes, with the gfx950 unit handling of MI355X_MICROARCH.md section HBM:
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE counts 128-byte requests at 64 bytes for wide coalesced
streaming reads (x2 correction), calibrated in this same profile on kernels whose byte counts are known exactly
(minmax_kernel reads cs*M*4 bytes with 16-B/lane loads; synth_box_kernel writes M*4 bytes per launch)."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import kernel_source_sha256  # the hash bench.py checks before it reports a stored traffic figure


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


def short(name):
    name = name.split("(")[0]
    for key in ("pearson_split_kernel", "pearson_reg_lds_kernel", "pearson_reg_kernel", "pearson_relay_kernel", "pearson_big_kernel",
                "pearson_stream_kernel", "pearson_prep_kernel", "spearman_u32_kernel", "spearman_split_kernel", "kendall_split_kernel",
                "spearman_prep_kernel", "kendall_prep_kernel", "spearman_kernel", "kendall_kernel", "mi_binned_kernel",
                "kraskov_direct_kernel", "kraskov_sorted_kernel", "kraskov_prep_kernel", "mi_kraskov_kernel", "minmax_kernel",
                "synth_box_kernel", "gather_reference_kernel", "direct_rank_kernel", "fill_kernel"):
        if key in name:
            return key
    return name[-60:]


def main():
    out = sys.argv[1]
    dur = defaultdict(list)
    for r in rows(os.path.join(out, "trace", "**", "*kernel_trace.csv")):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"# rocprofv3 summary ({' '.join(sys.argv[2:])})\n")
    print("Three separate runs of the same command (profiles/collect.sh): durations from the `--kernel-trace --stats` run ONLY;\n"
          "FETCH_SIZE from the `--pmc FETCH_SIZE` run, WRITE_SIZE from the `--pmc WRITE_SIZE` run (kernels run slower under\n"
          "counter collection: their durations are not reported).\n")
    print("Table 1 -- kernel durations, pass `rocprofv3 --kernel-trace --stats`:\n")
    print("| kernel | launches | avg us | median us | min us | total ms |")
    print("|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"| {k} | {len(v)} | {statistics.mean(v):.1f} | {statistics.median(v):.1f} | {min(v):.1f} | {sum(v) / 1e3:.2f} |")
    pmc = {}
    for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(list)
        for r in rows(os.path.join(out, sub, "**", "*counter_collection.csv")):
            if r.get("Counter_Name") == name:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        pmc[name] = {k: statistics.mean(v) for k, v in acc.items()}
    print("\nTable 2 -- HBM traffic per launch, passes `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE`:\n")
    print("| kernel | FETCH_SIZE KiB/launch (raw) | x2-corrected GB | WRITE_SIZE KiB/launch | GB |")
    print("|---|---|---|---|---|")
    for k in sorted(set(pmc["FETCH_SIZE"]) | set(pmc["WRITE_SIZE"])):
        f = pmc["FETCH_SIZE"].get(k, float("nan"))
        w = pmc["WRITE_SIZE"].get(k, float("nan"))
        print(f"| {k} | {f:.0f} | {f * 1024 * 2 / 1e9:.4f} | {w:.0f} | {w * 1024 / 1e9:.4f} |")
    # HBM traffic per launch of the dominant kernel, gfx950-corrected: FETCH_SIZE counts 128-B requests at 64 B (x2),
    # WRITE_SIZE is exact; both in KiB.  Keyed like bench.py's load_pmc_traffic().
    dominant = max(dur.items(), key=lambda kv: sum(kv[1]) if "synth" not in kv[0] else 0)[0]
    measure, grid, members = "pearson", [256, 256, 256], 64
    argv = sys.argv[2:]
    for i, a in enumerate(argv):
        if a == "--measure":
            measure = argv[i + 1]
        if a == "--members":
            members = int(argv[i + 1])
        if a == "--grid":
            grid = [int(v) for v in argv[i + 1:i + 4]]
    traffic = pmc["FETCH_SIZE"].get(dominant, 0.0) * 1024 * 2 + pmc["WRITE_SIZE"].get(dominant, 0.0) * 1024
    entry = {f"{measure}:{grid[0]}x{grid[1]}x{grid[2]}x{members}:gpus1": {
        "kernel": dominant, "traffic_bytes_per_launch": int(traffic), "source_sha256": kernel_source_sha256(measure),
        "fetch_kib_raw": pmc["FETCH_SIZE"].get(dominant), "write_kib": pmc["WRITE_SIZE"].get(dominant),
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B; calibrated: equals cs*M*4 exactly), "
                      "WRITE_SIZE x1; units KiB"}}
    with open(os.path.join(out, "pmc_traffic_entry.json"), "w") as fh:
        json.dump(entry, fh, indent=1)
    print("\nHBM traffic per launch (" + dominant + f"): {traffic / 1e9:.4f} GB")
    with open(os.path.join(out, "summary.json"), "w") as fh:
        json.dump({"durations_us": {k: {"n": len(v), "avg": statistics.mean(v), "median": statistics.median(v)}
                                    for k, v in dur.items()}, "pmc_kib_per_launch": pmc}, fh, indent=1)


if __name__ == "__main__":
    main()
