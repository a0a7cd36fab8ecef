#!/usr/bin/env python3
"""profiles/<tag>_baseline_configs.md from gpurun_out/<tag>_baseline_configs.jsonl (tools/run_baseline_configs.sh)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
lines = [json.loads(l) for l in (ROOT / "gpurun_out" / f"{tag}_baseline_configs.jsonl").read_text().splitlines() if l.startswith("{")]
names = ["configs[0] 64^3 x 16 Pearson", "configs[1] 256^3 x 64 Pearson (headline)", "configs[2] 256^3 x 64 Kraskov k = 3",
         "configs[3] 512^3 x 128 Spearman", "configs[4] one rank's slab of 1024^3 x 256 on 8 GPUs (1024 x 1024 x 128 x 256) Pearson",
         "256^3 x 64 Spearman", "256^3 x 64 Kendall", "256^3 x 64 binned MI"]
pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
out = [f"# BASELINE.json configs, end of round {tag[1:].lstrip('0')} (one MI355X; `bash tools/run_baseline_configs.sh {tag}`, lines in {tag}_baseline_configs_bench_lines.jsonl)\n",
       "| config | ms per evaluation | kernel | kernel ms (HIP events) | algorithmic GB/s (of 8000) | Mvoxel-corr/s | HBM traffic (PMC, profiles/pmc_traffic.json) |",
       "|---|---|---|---|---|---|---|"]
for name, d in zip(names, lines):
    r = d["roofline"]
    key = f"{d['config']['measure']}:{'x'.join(str(g) for g in d['config']['grid'])}x{d['config']['members']}:gpus1"
    t = r.get("traffic") or (pmc.get(key, {}).get("traffic_bytes_per_launch") if pmc.get(key, {}).get("kernel", "").endswith(r["kernel"]) or r["kernel"] in pmc.get(key, {}).get("kernel", "") else None)
    traffic = f"{t / 1e9:.3f} GB" if t else "-"
    out.append(f"| {name} | {d['ms_per_step']:.3f} | {r['kernel']} | {r['kernel_ms']:.3f} | {r['achieved']:.0f} ({100 * r['frac']:.1f} %) | {d['value']:.0f} | {traffic} |")
(ROOT / "profiles" / f"{tag}_baseline_configs.md").write_text("\n".join(out) + "\n")
(ROOT / "profiles" / f"{tag}_baseline_configs_bench_lines.jsonl").write_text("\n".join(json.dumps(l) for l in lines) + "\n")
print("\n".join(out))
