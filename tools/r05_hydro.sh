#!/bin/bash
# round 5, GPU box: hydro one-kernel stage with the slope plane -- parity subset, then A/B against the r04 kernel
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "one_kernel or odd_shapes or hydro or passive_scalars or isothermal_multi_d or multi_d_schemes" ) > gpurun_out/r05_hyd_tests.txt 2>&1
tail -4 gpurun_out/r05_hyd_tests.txt
{
for nx in 256 128; do
  echo "##### sod $nx^3"
  BENCH_ARGS="--problem sod --nx $nx --no-other-configs" bash tools/ab.sh "-" "-:AKMI_HS_V1=1" "-" "-:AKMI_HS_V1=1"
done
echo "##### tiles, new kernel, 256^3"
for t in 23x11 28x9 32x8 20x12 26x9 36x7 44x5 16x16 18x14; do
  BENCH_ARGS="--problem sod --nx 256 --no-other-configs" bash tools/ab.sh "-:AKMI_HS_TILE=$t"
done
} > gpurun_out/r05_hyd_ab.txt 2>&1
cat gpurun_out/r05_hyd_ab.txt | cut -c1-160
