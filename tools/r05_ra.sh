#!/bin/bash
# GPU box: run-ahead cycles + folded boundary functions -- parity, then A/B on the Sod deck and the headline
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
out=gpurun_out/r05_ra.txt; : > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_options.py -x -q -m gpu -k "native_cpp or gather_with_bcs or whole_runs or golden" 2>&1 | tail -8 >> $out
for r in 1 2 3; do
  for ra in 0 1; do
    for spec in "sod 128" "sod 256"; do
      set -- $spec
      echo "== run_ahead=$ra $1 nx=$2" >> $out
      AKMI_RUN_AHEAD=$ra timeout 300 python bench.py --no-cpu-baseline --problem $1 --nx $2 --no-other-configs 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["other_host"]["value"] if d.get("other_host") else "", d["config"]["host_check"])' >> $out
    done
    echo "== run_ahead=$ra headline" >> $out
    AKMI_RUN_AHEAD=$ra timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["other_host"]["value"] if d.get("other_host") else "", d["config"]["host_check"])' >> $out
  done
done
cat $out
