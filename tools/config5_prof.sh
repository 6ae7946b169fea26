#!/bin/bash
# usage (GPU box): tools/config5_prof.sh TAG [ENV=VAL ...] -- BASELINE config 5 at deck size (120 x 16^3) and at production
# size (960 x 32^3): whole-run rates through both hosts, then rocprofv3 kernel stats of the C++ host at both sizes.
# Output: gpurun_out/TAG_config5.txt
tag=${1:-c5}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
for e in "$@"; do export "$e"; done
out=$root/gpurun_out/${tag}_config5.txt
PROD="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
cd /tmp && export TMPDIR=/tmp AKMI_CONFIG5_CPU=0
{
echo "# config 5, env: $*"
python $root/tools/config5.py 60 2>&1 | grep "config 5"
python $root/tools/config5.py 20 $PROD 2>&1 | grep "config 5"
export AKMI_CONFIG5_HOSTS=c++
for sz in deck prod; do
  args="40"; [ $sz = prod ] && args="10 $PROD"
  rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -- python $root/tools/config5.py $args > /tmp/pp5.log 2>&1
  echo "# $sz size, C++ host, rocprofv3 --kernel-trace --stats (5 + ${args%% *} cycles)"; grep "config 5" /tmp/pp5.log
  python $root/tools/kernel_stats.py /tmp/pp5 | head -45
done
} > $out 2>&1
cat $out
