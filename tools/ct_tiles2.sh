cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --steps 6 $1 > /tmp/pp.log 2>&1; echo "tile=${AKMI_CT_TILE:-auto} $1: $(grep '^{"metric"' /tmp/pp.log | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])') Mcell-updates/s, k_corner_ct $(python $R/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E 'corner' | awk '{print $4}') us"; }
scan() { for t in $2; do if [ "$t" != auto ]; then export AKMI_CT_TILE=$t; else unset AKMI_CT_TILE; fi; run "$1"; done; unset AKMI_CT_TILE; }
scan "" "auto 64,8 52,9 44,11 36,14 32,16 44,7 64,5 52,6 auto"
scan "--mb 64" "auto 34,15 66,7"
scan "--mb 32" "auto 34,15 34,7"
