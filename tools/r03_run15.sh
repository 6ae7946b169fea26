#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
timeout 2400 python -m pytest tests/test_gpu_smr.py tests/test_gpu_parity.py tests/test_gpu_schemes.py -m gpu -x -q 2>&1 | tail -3
export AKMI_CONFIG5_CPU=0
P="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
for d in 0 1 0 1; do
echo "## AKMI_FLUX_STREAMS=$d"
AKMI_FLUX_STREAMS=$d python tools/config5.py 40 2>&1 | grep "config 5"
AKMI_FLUX_STREAMS=$d python tools/config5.py 10 $P 2>&1 | grep "config 5"
AKMI_FLUX_STREAMS=$d python bench.py --steps 40 --warmup 5 --no-cpu-baseline --nx 64 --split 2>&1 | tail -1 | cut -c1-140
AKMI_FLUX_STREAMS=$d python bench.py --steps 40 --warmup 5 --no-cpu-baseline --nx 128 --split 2>&1 | tail -1 | cut -c1-140
done
} > $root/gpurun_out/r03_run15.txt 2>&1
cat $root/gpurun_out/r03_run15.txt | cut -c1-150
