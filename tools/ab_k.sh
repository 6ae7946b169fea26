#!/bin/bash
# usage (GPU box): tools/ab_k.sh SPEC...  like tools/ab.sh but 6 cycles, MHD one-kernel stage switched on, stage kernels only
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  (
    export AKMI_MHD_ONE_KERNEL=1
    if [ "$v" != "-" ]; then export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so; fi
    IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $root/bench.py --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 $BENCH_ARGS > /tmp/pp.log 2>&1
    echo "== $spec $(grep "^{\"metric\"" /tmp/pp.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"])' 2>/dev/null || tail -3 /tmp/pp.log)"
    python $root/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E "k_sweep|mhd_stage" | cut -c1-120
  )
done
