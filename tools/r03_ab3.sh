#!/bin/bash
# like r03_ab1.sh with the bench arguments in BENCH_ARGS (e.g. "--recon ppm4 --ng 4 --split --nx 320 --mb 32")
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  if [ "$v" != "default" ]; then export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so; else unset AKMI_LIB; fi
  rm -rf /tmp/pp
  env $envs rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $root/bench.py --no-cpu-baseline --steps 10 $BENCH_ARGS > /tmp/pp.log 2>&1
  echo "== $spec [$BENCH_ARGS] $(grep "^{\"metric\"" /tmp/pp.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"])' 2>/dev/null || tail -3 /tmp/pp.log)"
  python $root/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E "k_sweep|corner|c2p_newdt|rk_update|k_ct" | cut -c1-120
done
