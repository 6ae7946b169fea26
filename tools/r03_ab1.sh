#!/bin/bash
# round 3 A/B on the GPU: bench (10 cycles) + per-kernel times of library variants built with tools/build_variant.sh
# usage (GPU box): bash tools/r03_ab1.sh VARIANT[:ENV=VAL[,ENV=VAL]]...     ("default" = the committed library)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  if [ "$v" != "default" ]; then export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so; else unset AKMI_LIB; fi
  rm -rf /tmp/pp
  env $envs rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $root/bench.py --no-cpu-baseline --steps 10 > /tmp/pp.log 2>&1
  echo "== $spec $(grep "^{\"metric\"" /tmp/pp.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"])' 2>/dev/null || tail -3 /tmp/pp.log)"
  python $root/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E "k_sweep|corner|c2p_newdt" | cut -c1-120
done
