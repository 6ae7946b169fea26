#!/bin/bash
# round 5, GPU box: SQ / LDS counters of the one-kernel hydro stage (sod 256^3), optionally with env of $1 (e.g. AKMI_HS_V1=1)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
tag=$1; shift
for e in "$@"; do export "$e"; done
export AKMI_PMC_PROBLEM=sod
bash tools/pmc_valu.sh $tag 2>&1 | grep -E "kernel|hydro_stage|c2p_newdt"
bash tools/pmc_detail.sh $tag 2>&1 | grep -A28 "hydro_stage3d" | head -40
