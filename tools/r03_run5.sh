#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_pw1.so timeout 300 python - <<PY 2>&1 | tail -2
import sys
sys.path.insert(0, "tests")
import parity_util as pu
for n, mb in ((32, 32), (32, 16), (48, 24)):
    r = pu.compare_run("orszag_tang", n=n, dims=3, mb=mb, cycles=3)
    print("parity persist n=%d mb=%d" % (n, mb), r["max_rel_l1"], r.get("bitwise_equal"))
PY
bash tools/r03_ab1.sh pw0 pw1 pw1:AKMI_ML12=16 pw1:AKMI_ML12=11 pw1:AKMI_ML12=8 pw0:AKMI_ML12=16 pw0 pw1
} > gpurun_out/r03_run5.txt 2>&1
tail -60 gpurun_out/r03_run5.txt
