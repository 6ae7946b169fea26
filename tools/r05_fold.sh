#!/bin/bash
# GPU box: folded boundary functions (akmi_bvals_*_local_bcs) -- parity test, then A/B on the Sod deck (configs[1] and 256^3)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
out=gpurun_out/r05_fold.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gather_with_bcs or task_bcs or sod or golden" 2>&1 | tail -5 >> $out
for r in 1 2 3; do
  for f in 0 1; do
    for nx in 128 256; do
      echo "== fold=$f nx=$nx" >> $out
      AKMI_FOLD_BCS=$f timeout 300 python bench.py --no-cpu-baseline --problem sod --nx $nx --no-other-configs 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["other_host"]["value"] if d.get("other_host") else "", d["config"]["host_check"])' >> $out
    done
  done
done
cat $out
