#!/usr/bin/env python
"""Resource table of the kernels in a device assembly file (hipcc --save-temps): VGPRs, AGPRs, SGPRs, scratch, LDS.
usage: tools/kmeta.py FILE.s [name-substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(.*", "", dem)
    if pats and not any(p in dem for p in pats):
        continue
    print("%-70s vgpr %3s agpr %3s sgpr %3s scratch %4s lds %6s" % (dem[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
          g("private_segment_fixed_size"), g("group_segment_fixed_size")))
