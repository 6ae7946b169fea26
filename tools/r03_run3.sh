#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
echo "### what-if builds (timing only): w3 = x3 march without its stores for CornerE + CornerE without x3 inputs; w4 = no mass-flux arrays; w7 = both"
bash tools/r03_ab1.sh w0 w3 w4 w7 w0
echo "### one ConsToPrim over all cells after the ghost fill instead of c2p(active) + c2p(shell)"
for mb in 0 64 32; do
 for m in 0 1; do
  a=""; [ $mb != 0 ] && a="--mb $mb"
  AKMI_BENCH_NATIVE_CHECK=0 AKMI_MERGE_C2P=$m python bench.py --no-cpu-baseline --steps 10 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge=$m mb=$mb', d['value'], d['roofline']['ms_per_launch'], d['roofline']['halo_bcs_shell_c2p_ms'])"
 done
done
echo "### new independent checks + prolong_primitives on the GPU"
timeout 1200 python -m pytest tests/test_independent_checks.py tests/test_gpu_smr.py tests/test_refine_operators.py -m gpu -x -q 2>&1 | tail -4
} > gpurun_out/r03_run3.txt 2>&1
tail -70 gpurun_out/r03_run3.txt
