#!/bin/bash
# Register / LDS / spill figures of every kernel of one source file, from the compiler's own remarks (no GPU needed):
#   tools/kernel_resources.sh geometrics_amd/csrc/dense_gemm.hip [name filter]
f=$1; filt=${2:-.}
extra=""
case "$f" in *dense_gemm.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -fno-slp-vectorize $extra \
    -I include -I geometrics_amd/csrc -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
python3 -c '
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?):\s+(\S+)", line) or re.search(r"remark:\s+(.*?):\s+(\S+)\s*\[", line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    if not re.search(sys.argv[1], name): continue
    print("%-60s VGPR %3s AGPR %3s SGPR %3s spill %s/%s LDS %6s occ %s" % (name[:60], v.get("VGPRs","?"), v.get("AGPRs","?"), v.get("SGPRs","?"),
          v.get("VGPRs Spill", v.get("VGPR Spill","?")), v.get("SGPRs Spill", v.get("SGPR Spill","?")), v.get("LDS Size [bytes/block]","?"), v.get("Occupancy [waves/SIMD]","?")))
' "$filt"
