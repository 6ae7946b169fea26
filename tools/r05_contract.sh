#!/bin/bash
# round 5, GPU box: the stage kernels built with -ffp-contract=fast (tools/build_variant.sh contract -ffp-contract=fast) against
# north_star's bar (relative L1 <= 1e-12) instead of bit identity: whole GPU suite with AKMI_PARITY_TOL, bench, VALU counters.
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
V=$root/athenak_amd/lib/variants/libakmi_contract.so
rm -f gpurun_out/r05_contract_parity.tsv
( AKMI_LIB=$V AKMI_PARITY_LOG=$root/gpurun_out/r05_contract_parity.tsv timeout 1500 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider --parity-tol=1e-12 ) > gpurun_out/r05_contract_tests.txt 2>&1
tail -3 gpurun_out/r05_contract_tests.txt
{
echo "##### orszag_tang 256^3 (BASELINE config 3)"
bash tools/ab.sh - contract - contract
echo "##### sod 256^3"
BENCH_ARGS="--problem sod --nx 256 --no-other-configs" bash tools/ab.sh - contract
echo "##### sod 128^3 (BASELINE configs[1])"
BENCH_ARGS="--problem sod --nx 128 --no-other-configs" bash tools/ab.sh - contract
} > gpurun_out/r05_contract_bench.txt 2>&1
grep -E "^#####|^==" gpurun_out/r05_contract_bench.txt
AKMI_LIB=$V bash tools/pmc_valu.sh r05_contract 2>&1 | grep -E "^kernel|k_sweep|corner|c2p_newdt" > gpurun_out/r05_contract_valu.txt
cat gpurun_out/r05_contract_valu.txt
