#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel: parses `hipcc -Rpass-analysis=kernel-resource-usage` (stderr) from a file.
usage: hipcc --offload-arch=gfx950 <flags of _build.py> -c radish_hip.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> usage.txt
       python scripts/kernel_resources.py usage.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        pass
    if flt and flt not in name:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    row = [g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPR Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")]
    print(f"{name[:64]:64s} VGPR {row[0]:>4s} AGPR {row[1]:>3s} SGPR {row[2]:>4s} spill {row[3]:>3s} scratch {row[4]:>4s} occ {row[5]:>2s} LDS {row[6]:>6s}")
