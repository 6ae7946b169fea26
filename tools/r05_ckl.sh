#!/bin/bash
# GPU box: chunk length of k_hydro_stage3d at 128^3 (and 64^3-block packs): bench value, stage group ms
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
out=gpurun_out/r05_ckl.txt; : > $out
for r in 1 2; do
  for c in 0 4 5 6 7 8 10 15 16 22 32; do
    echo -n "nx=128 ckl=$c  " >> $out
    AKMI_HS_CKL=$c timeout 300 python bench.py --no-cpu-baseline --problem sod --nx 128 --no-other-configs 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"])' >> $out
  done
done
cat $out
