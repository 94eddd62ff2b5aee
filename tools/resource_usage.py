#!/usr/bin/env python
"""Per-kernel register / spill / occupancy table of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resource_usage.py nsdp_amd/csrc/gemm_bf16.hip [extra hipcc flags]
"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
       "-ffp-contract=fast", "-c", src, "-o", "/tmp/_ru.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+([A-Za-z \[\]/]+): (\d+)", line) or re.search(r":\s+([A-Za-z][A-Za-z \[\]/]+): (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
print(f"{'kernel':60s} VGPR AGPR  vspill sspill occ  LDS")
for k, r in rows.items():
    print(f"{k[:60]:60s} {r.get('VGPRs', -1):4d} {r.get('AGPRs', -1):4d} {r.get('VGPRs Spill', -1):6d} {r.get('SGPRs Spill', -1):6d} "
          f"{r.get('Occupancy [waves/SIMD]', -1):3d} {r.get('LDS Size [bytes/block]', -1):5d}")
