#!/usr/bin/env python3
"""Merges gpurun_out/prof_<tag>/pmc_traffic_entry.json files (written on the GPU box by profiles/collect.sh) into the
committed profiles/pmc_traffic.json.  Usage: tools/merge_pmc_entry.py gpurun_out/prof_*/pmc_traffic_entry.json"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
target = ROOT / "profiles" / "pmc_traffic.json"
data = json.loads(target.read_text()) if target.exists() else {}
for f in sys.argv[1:]:
    entry = json.loads(Path(f).read_text())
    data.update(entry)
    print("merged", list(entry))
target.write_text(json.dumps(data, indent=1) + "\n")
