cd /tmp && export TMPDIR=/tmp
for v in "" CJ4 CJ16 K64 K16 OLD ""; do
  if [ -n "$v" ]; then export AKMI_LIB=$GRAFT_REPO_ROOT/athenak_amd/lib/variants/libakmi_$v.so; else unset AKMI_LIB; fi
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 > /tmp/pp.log 2>&1
  echo "== ${v:-default} $(tail -1 /tmp/pp.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"
  python $GRAFT_REPO_ROOT/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E "corner|ct_copy|k_sweep<0" | cut -c1-110
done
