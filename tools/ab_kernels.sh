# usage (GPU box): tools/ab_kernels.sh VARIANT...   -- bench + per-kernel times of the default library and of
# athenak_amd/lib/variants/libakmi_<VARIANT>.so (built beforehand with -D switches), default run twice
cd /tmp && export TMPDIR=/tmp
for v in "" "$@" ""; do
  if [ -n "$v" ]; then export AKMI_LIB=$GRAFT_REPO_ROOT/athenak_amd/lib/variants/libakmi_$v.so; else unset AKMI_LIB; fi
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 > /tmp/pp.log 2>&1
  echo "== ${v:-default} $(grep "^{\"metric\"" /tmp/pp.log | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"
  python $GRAFT_REPO_ROOT/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E "k_sweep|corner|c2p_newdt" | cut -c1-120
done
