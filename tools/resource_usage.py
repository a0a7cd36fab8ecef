#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS summary of one HIP translation unit (compiler remarks, no GPU needed).

usage: tools/resource_usage.py correrender_amd/csrc/kernels_rank.hip [extra hipcc flags]
"""
import re
import subprocess
import sys

FLAGS = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    if "kernels_rank" in src and not extra:
        extra = ["-mllvm", "-enable-misched=0"]     # RANKFLAGS of the Makefile
    err = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, src], capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", err)[1:]
    keys = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
            ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]")]
    for b in blocks:
        name = b.split("\n")[0].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void crf::", "")
        vals = []
        for label, pat in keys:
            m = re.search(pat + r": (\d+)", b)
            vals.append(f"{label} {m.group(1) if m else '?':>6}")
        print(f"{name:64s} " + "  ".join(vals))


if __name__ == "__main__":
    main()
