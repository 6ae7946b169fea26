#!/bin/bash
# usage: tools/reg_audit.sh  -- VGPRs / scratch of the pieces of a PLM + HLLD face compiled on their own (tools/reg_audit.hip)
root=$(cd $(dirname $0)/.. && pwd)
d=$(mktemp -d); cd $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-schedule-relaxed-occupancy=true \
  -I $root/athenak_amd/csrc -c $root/tools/reg_audit.hip --save-temps -o ra.o 2>/dev/null
python3 - <<'PY'
import re, subprocess
s = open("reg_audit-hip-amdgcn-amd-amdhsa-gfx950.s").read()
rows = []
for blk in s.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    rows.append((name, int(g("vgpr_count")), int(g("private_segment_fixed_size")), int(g("sgpr_count"))))
# static VALU count per kernel
for name, v, sc, sg in sorted(rows):
    print("%-34s VGPRs %4d   scratch %4d B   SGPRs %3d" % (name, v, sc, sg))
PY
rm -rf $d
