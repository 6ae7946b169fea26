#!/bin/bash
# GPU box: hydro ConsToPrim writing the ghost images (akmi_hydro_c2p_newdt_halo, AKMI_C2P_HALO) -- option tests, then A/B on the Sod deck
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
out=gpurun_out/r05_halo.txt; : > $out
timeout 1200 python -m pytest tests/test_gpu_options.py -q -m gpu -n 4 -k whole_runs 2>&1 | tail -4 >> $out
for r in 1 2; do
  for h in 0 1; do
    for nx in 128 256; do
      echo "== halo=$h nx=$nx" >> $out
      AKMI_C2P_HALO=$h timeout 300 python bench.py --no-cpu-baseline --problem sod --nx $nx --no-other-configs 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["other_host"]["value"] if d.get("other_host") else "", d["config"]["host_check"])' >> $out
    done
  done
done
cat $out
