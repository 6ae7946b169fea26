#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# kernel-trace + stats of bench.py; summary goes to gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -- python $root/bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
python $root/tools/kernel_stats.py $out "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline $*" > $root/gpurun_out/${tag}_kernel_stats.txt
grep "^{\"metric\"" $out/bench.log | tail -1 > $root/gpurun_out/${tag}_bench.json
cat $root/gpurun_out/${tag}_kernel_stats.txt | head -24
