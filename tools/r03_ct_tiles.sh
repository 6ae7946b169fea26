#!/bin/bash
# tile shapes of k_corner_ct (AKMI_CT_TILE=tw,th, no rebuild): 10-cycle bench + per-kernel times, the scan twice on one box
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
for rep in 1 2; do
for t in 44,11 53,9 66,7 87,5 130,3 87,4 66,6 130,2 34,15 27,18; do
echo "## AKMI_CT_TILE=$t"; AKMI_CT_TILE=$t bash tools/r03_ab1.sh default 2>&1 | grep -E "^==|corner_ct"
done
done
} > gpurun_out/r03_ct_tiles.txt 2>&1
cat gpurun_out/r03_ct_tiles.txt
