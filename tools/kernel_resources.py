#!/usr/bin/env python3
"""Diagnostic: registers / spills / scratch / LDS of every kernel in saa_kernels.hip, or of the file named by
``--file=saa_predictor.hip`` (hipcc -Rpass-analysis)."""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "synchronization_avoiding_algorithms_amd", "csrc")
src = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--file=")] or ["saa_kernels.hip"])[0]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-c",
       src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *[a for a in sys.argv[1:] if not a.startswith("--file=")]]
err = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                  r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1).split(" [")[0]] = m.group(2)
print(f"{'kernel':58s} sgpr vgpr sgpr_spill vgpr_spill scratch occ")
for k, v in rows.items():
    print(f"{k[-58:]:58s} {v.get('TotalSGPRs'):>4s} {v.get('VGPRs'):>4s} {v.get('SGPRs Spill'):>10s} {v.get('VGPRs Spill'):>10s} "
          f"{v.get('ScratchSize'):>7s} {v.get('Occupancy'):>3s}")
