#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
for n in 24 32; do
AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_v2.so timeout 300 python - <<PY 2>&1 | tail -1
import sys
sys.path.insert(0, "tests")
import parity_util as pu
r = pu.compare_run("orszag_tang", n=$n, dims=3, mb=$n, cycles=3)
print("parity n=$n", r["max_rel_l1"], r.get("bitwise_equal"))
PY
done
echo "### A/B"
bash tools/r03_ab1.sh dev0 u3cpsr v2 v2h v2r v2nb v2nu v2w3 v2
echo "### march lengths on v2"
bash tools/r03_ab1.sh v2:AKMI_ML3=10 v2:AKMI_ML3=12 v2:AKMI_ML3=14 v2:AKMI_ML3=16 v2:AKMI_ML3=18 v2:AKMI_ML3=22 v2:AKMI_ML3=26 v2:AKMI_ML12=26 v2:AKMI_ML12=29 v2:AKMI_ML12=33 v2:AKMI_ML12=37 v2:AKMI_ML12=33,AKMI_ML3=14
echo "### SQ counters v2"
AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_v2.so bash tools/pmc_valu.sh r03v2
} > gpurun_out/r03_run2.txt 2>&1
tail -150 gpurun_out/r03_run2.txt
