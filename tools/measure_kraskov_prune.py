#!/usr/bin/env python3
"""Kraskov at 256^3 (benchmark box ensemble): the shipped tile-free kernel vs the x-ordered sweeps with early exit
(CRF_KRASKOV_PRUNE=1), 1 / 2 / 4 points per sweep, several member counts and k."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

if __name__ == "__main__":
    members = sys.argv[1] if len(sys.argv) > 1 else "32,64,100,128"
    ks = [int(k) for k in (sys.argv[2] if len(sys.argv) > 2 else "3").split(",")]
    tool = str(ROOT / "tools" / "measure_kraskov_variants.py")
    for k in ks:
        for label, extra in [("shipped", {}), ("pruned x4", {"CRF_KRASKOV_PRUNE": "1", "CRF_KRASKOV_PRUNE_TI": "4"}),
                             ("pruned x2", {"CRF_KRASKOV_PRUNE": "1", "CRF_KRASKOV_PRUNE_TI": "2"}),
                             ("pruned x1", {"CRF_KRASKOV_PRUNE": "1", "CRF_KRASKOV_PRUNE_TI": "1"})]:
            print(f"# {label}", flush=True)
            subprocess.run([sys.executable, tool, "one", str(k), members],
                           env=dict(os.environ, CRF_KRASKOV_DIRECT="1", **extra), check=True)
