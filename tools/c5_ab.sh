#!/bin/bash
# usage (GPU box): tools/c5_ab.sh SPEC...  with SPEC = NAME[:ENV=VAL[,ENV=VAL...]] -- BASELINE config 5 through the C++ host at deck
# and production size under rocprofv3, twice, the specs alternating: whole-run rate + the kernels matching $C5_KERNELS
root=${GRAFT_REPO_ROOT:-$(pwd)}
PROD="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
pat=${C5_KERNELS:-k_smr|k_rk_update|k_ct|k_c2p|k_sweep|corner}
cd /tmp && export TMPDIR=/tmp AKMI_CONFIG5_CPU=0 AKMI_CONFIG5_HOSTS=c++
for r in 1 2; do for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  ( IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
  for sz in deck prod; do
    args="40"; [ $sz = prod ] && args="10 $PROD"
    rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py $args > /tmp/pp5.log 2>&1
    echo "=== $v $r $sz $(grep 'config 5' /tmp/pp5.log | cut -d'|' -f4,5)"
    python $root/tools/kernel_stats.py /tmp/pp5 | grep -E "$pat" | cut -c1-100
  done )
done; done
