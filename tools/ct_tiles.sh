# usage (GPU box): bash tools/ct_tiles.sh  -- k_corner_ct average time for pinned tile shapes (AKMI_CT_TILE=tw,th)
# and for the launcher's own choice, on one 256^3 MeshBlock and on packs of 64^3 / 32^3 MeshBlocks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --steps 6 $1 > /tmp/pp.log 2>&1; echo "tile=${AKMI_CT_TILE:-auto} $1: $(grep '^{"metric"' /tmp/pp.log | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])') Mcell-updates/s, k_corner_ct $(python $R/tools/kernel_stats.py /tmp/pp x 2>/dev/null | grep -E 'corner' | awk '{print $4}') us"; }
scan() { for t in $2; do if [ "$t" != auto ]; then export AKMI_CT_TILE=$t; else unset AKMI_CT_TILE; fi; run "$1"; done; unset AKMI_CT_TILE; }
scan "" "64,8 auto 44,11 34,15 66,7"
scan "--mb 128" "64,8 auto"
scan "--mb 64" "64,8 auto 34,15"
scan "--mb 32" "64,8 auto"
