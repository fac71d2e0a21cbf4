"""Compile one HIP source for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
name, VGPRs, AGPRs, spills, scratch, occupancy, LDS.   python kres.py <file.hip> [extra hipcc flags]"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I" + csrc,
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = {}
for line in r.stderr.splitlines():
    if "error:" in line:
        print(line)
    m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        if cur:
            print(cur)
        cur = {"name": v[:90]}
    else:
        k = re.sub(r" \[.*", "", k)
        if k in ("VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize", "Occupancy", "LDS Size", "SGPRs"):
            cur[k] = v
if cur:
    print(cur)
