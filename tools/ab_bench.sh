#!/bin/bash
# usage (GPU box): tools/ab_bench.sh ROUNDS SPEC...  -- alternating plain bench runs (no profiler), SPEC as in tools/ab.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
    (
      if [ "$v" != "-" ]; then export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so; fi
      IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
      python $root/bench.py --no-cpu-baseline --steps ${STEPS:-20} $BENCH_ARGS 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("'"$spec"'", d["value"], d["roofline"]["ms_per_launch"], d["other_host"]["value"] if d.get("other_host") else "")'
    )
  done
done
