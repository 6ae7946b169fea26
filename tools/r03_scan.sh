#!/bin/bash
# chunk lengths of the two marches of the headline stage (env knobs, no rebuild): 10-cycle bench + per-kernel times
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
for rep in 1 2; do
bash tools/r03_ab1.sh default
for ml in 8 12 16 20 24 32 43 64 128; do bash tools/r03_ab1.sh default:AKMI_ML12=$ml 2>&1 | grep -E "^==|sweep12s"; done
for ml in 8 10 12 14 16 20 24 32 64; do bash tools/r03_ab1.sh default:AKMI_ML3=$ml 2>&1 | grep -E "^==|sweep_update"; done
done
} > gpurun_out/r03_scan.txt 2>&1
cat gpurun_out/r03_scan.txt
