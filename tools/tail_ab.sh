cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in "" "--mb 64" "--problem sod" "--problem sod --nx 128 --steps 30" "--nx 128 --steps 30"; do
for v in 0 1 0 1; do
  export AKMI_TAIL=$v
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --steps 8 $a > /tmp/pp.log 2>&1
  echo "tail=$v [$a] $(grep '^{"metric"' /tmp/pp.log | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])') | $(python $R/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E 'k_sweep_update|corner|hydro_stage' | awk '{printf "%s ", $(NF-1)}')"
done; done
