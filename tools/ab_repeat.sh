#!/bin/bash
# usage (GPU box): tools/ab_repeat.sh N VARIANT...  -- bench value of default and variants, alternating, N rounds
n=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for r in $(seq 1 $n); do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so; else unset AKMI_LIB; fi
    val=$(python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"])')
    echo "round $r ${v:-default} $val"
  done
done
