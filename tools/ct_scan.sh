#!/bin/bash
# usage (GPU box): tools/ct_scan.sh "BENCH ARGS" tile tile ...   -- k_corner_ct tile shapes (AKMI_CT_TILE=tw,th; "default" = ct_tile's choice):
# bench value and the kernel's average time under rocprofv3
root=${GRAFT_REPO_ROOT:-$(pwd)}
args=$1; shift
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  if [ $t = default ]; then unset AKMI_CT_TILE; else export AKMI_CT_TILE=$t; fi
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $root/bench.py --no-cpu-baseline --steps 10 $args > /tmp/pp.log 2>&1
  echo "tile=$t $(grep '^{"metric"' /tmp/pp.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"])') $(python $root/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep corner_ct | awk '{print $(NF-1)}')"
done
