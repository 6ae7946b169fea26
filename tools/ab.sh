#!/bin/bash
# usage (GPU box): tools/ab.sh SPEC...   with SPEC = VARIANT[:ENV=VAL[,ENV=VAL...]]  ("-" = the default library)
# bench (10 cycles) under rocprofv3 --kernel-trace per spec: Mcell-updates/s and the stage kernels' average times.
# VARIANT = athenak_amd/lib/variants/libakmi_<VARIANT>.so (tools/build_variant.sh).  BENCH_ARGS adds bench arguments.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  (
    if [ "$v" != "-" ]; then export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so; fi
    IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $root/bench.py --no-cpu-baseline --steps 10 $BENCH_ARGS > /tmp/pp.log 2>&1
    echo "== $spec $(grep "^{\"metric\"" /tmp/pp.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"])' 2>/dev/null || tail -3 /tmp/pp.log)"
    python $root/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E "k_sweep|corner|c2p_newdt|march3|hydro_stage|mhd_stage|k_ct|rk_update|ghost_fill" | cut -c1-120
  )
done
