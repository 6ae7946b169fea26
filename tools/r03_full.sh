#!/bin/bash
# usage (GPU box): bash tools/r03_full.sh TAG  -- full GPU suite, default bench line, kernel stats under rocprofv3
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -5 gpurun_out/${tag}_gpu_tests.txt
python bench.py > gpurun_out/${tag}_bench_full.log 2>&1
grep '^{"metric"' gpurun_out/${tag}_bench_full.log | tail -1 > gpurun_out/${tag}_bench_full.json
cat gpurun_out/${tag}_bench_full.json | cut -c1-600
bash tools/prof.sh $tag --steps 10 | head -16
