#!/usr/bin/env python3
"""Per-kernel register / LDS / spill summary of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py speechcatcher_amd/csrc/decoder_layer.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
       "-fno-fast-math", *sys.argv[3:], "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: \s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                  r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur:
            rows.append(cur)
        cur = {"name": v}
    else:
        cur[k.split(" ")[0] if k.startswith(("Scratch", "Occupancy", "LDS")) else k] = v
if cur:
    rows.append(cur)
print(f"{'kernel':70s} VGPR AGPR SGPR scratch occ vspill")
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("void ", "").split("(")[0]
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('TotalSGPRs','?'):>4s} "
          f"{r.get('ScratchSize','?'):>7s} {r.get('Occupancy','?'):>3s} {r.get('VGPRs Spill','?'):>6s}")
