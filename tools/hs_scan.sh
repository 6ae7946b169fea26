# usage (GPU box): bash tools/hs_scan.sh -- k_hydro_stage3d for register targets / prefetch options (variant libraries
# libakmi_w<waves>p<prefetch>.so built with -DAKMI_HS_WAVES= -DAKMI_HS_PREFETCH= in athenak_amd/lib/variants/) x
# workgroup-size caps (AKMI_HS_MAXT)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in w3p0 w3p1 w3p2 w3p3 w2p3; do
  export AKMI_LIB=$R/athenak_amd/lib/variants/libakmi_$lib.so
  for mt in 512 256 192; do
    export AKMI_HS_MAXT=$mt
    rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --problem sod --steps 6 $1 > /tmp/pp.log 2>&1
    echo "lib=$lib maxt=$mt $1: $(grep '^{"metric"' /tmp/pp.log | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])') Mcell-updates/s, k_hydro_stage3d $(python $R/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E 'hydro_stage3d' | awk '{print $5}') us"
  done
done
