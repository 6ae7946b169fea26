#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
{
echo "### selftest + parity of the last variant"
AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_u3cpsr.so timeout 600 python -m pytest tests/test_gpu_fastmath.py -x -q 2>&1 | tail -5
for n in 24 32; do
AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_u3cpsr.so timeout 300 python - <<PY 2>&1 | tail -3
import sys
sys.path.insert(0, "tests")
import parity_util as pu
r = pu.compare_run("orszag_tang", n=$n, dims=3, mb=$n, cycles=3)
print("parity n=$n", r["max_rel_l1"], r.get("bitwise_equal"), r)
PY
done
echo "### A/B"
bash tools/r03_ab1.sh dev0 u0 u1 u2 u3 u3c u3cp u3cps u3cpsr dev0 u3cpsr
echo "### march lengths on u3cpsr"
bash tools/r03_ab1.sh u3cpsr:AKMI_ML12=33 u3cpsr:AKMI_ML12=17 u3cpsr:AKMI_ML3=18 u3cpsr:AKMI_ML3=14 u3cpsr:AKMI_ML3=37
} > gpurun_out/r03_run1.txt 2>&1
tail -120 gpurun_out/r03_run1.txt
