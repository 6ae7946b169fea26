#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_b1.so
{
for cfg in "24 24" "32 32" "32 16" "48 24"; do
set -- $cfg
timeout 300 python - <<PY 2>&1 | tail -1
import sys
sys.path.insert(0, "tests")
import parity_util as pu
r = pu.compare_run("orszag_tang", n=$1, dims=3, mb=$2, cycles=3)
print("parity n=$1 mb=$2", r["max_rel_l1"], r.get("bitwise_equal"))
r = pu.compare_run("orszag_tang", n=$1, dims=3, mb=$2, cycles=3, native=True)
print("parity native n=$1 mb=$2", r["max_rel_l1"], r.get("bitwise_equal"))
PY
done
for b in 0 1 0 1; do
  AKMI_MFBITS=$b bash tools/r03_ab1.sh b1 | sed "s/^== b1/== b1 mfbits=$b/"
done
for mb in 64 32; do for b in 0 1; do
  AKMI_BENCH_NATIVE_CHECK=0 AKMI_MFBITS=$b python bench.py --no-cpu-baseline --steps 10 --mb $mb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mfbits=$b mb=$mb', d['value'])"
done; done
} > gpurun_out/r03_run4.txt 2>&1
tail -60 gpurun_out/r03_run4.txt
