# usage (GPU box): BENCH_ARGS="--recon ppm4 --ng 4" tools/ab_kernels_args.sh VARIANT...  -- as ab_kernels.sh with extra bench arguments
cd /tmp && export TMPDIR=/tmp
for v in "" "$@" ""; do
  if [ -n "$v" ]; then export AKMI_LIB=$GRAFT_REPO_ROOT/athenak_amd/lib/variants/libakmi_$v.so; else unset AKMI_LIB; fi
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 $BENCH_ARGS > /tmp/pp.log 2>&1
  echo "== ${v:-default} $(grep "^{\"metric\"" /tmp/pp.log | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"
  python $GRAFT_REPO_ROOT/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E "k_sweep|corner|c2p_newdt|hydro_stage" | cut -c1-120
done
