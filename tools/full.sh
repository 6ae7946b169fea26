#!/bin/bash
# usage (GPU box): bash tools/full.sh TAG [quick]  -- GPU suite, default bench line, kernel stats, PMC traffic, SQ counters;
# everything lands in gpurun_out/TAG_* (copy what is to be kept into profiles/)
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
if [ "$2" = "quick" ]; then
  ( time timeout 1200 python -m pytest tests/test_gpu_rccl_self.py tests/test_abi_caller.py tests/test_gpu_fastmath.py tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/${tag}_gpu_tests.txt 2>&1
else
  ( time timeout 2400 python -m pytest tests -m gpu -q -n 4 ) > gpurun_out/${tag}_gpu_tests.txt 2>&1
fi
tail -5 gpurun_out/${tag}_gpu_tests.txt
bash tools/pmc.sh $tag > /dev/null 2>&1
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic_latest.json
bash tools/pmc_valu.sh $tag > gpurun_out/${tag}_valu_counters.txt 2>&1
cp gpurun_out/${tag}_valu_counters.json profiles/valu_counters_latest.json
cat gpurun_out/${tag}_valu_counters.txt | tail -8
python bench.py > gpurun_out/${tag}_bench_full.log 2>&1
grep '^{"metric"' gpurun_out/${tag}_bench_full.log | tail -1 > gpurun_out/${tag}_bench_full.json
cat gpurun_out/${tag}_bench_full.json | cut -c1-2500
tail -3 gpurun_out/${tag}_bench_full.log | cut -c1-300
bash tools/prof.sh $tag --steps 10 | head -12
cp profiles/pmc_traffic_latest.json gpurun_out/${tag}_pmc_traffic_latest.json
cp profiles/valu_counters_latest.json gpurun_out/${tag}_valu_counters_latest.json
