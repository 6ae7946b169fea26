#!/bin/bash
# round 5, GPU box: the committed hydro kernel -- kernel stats (256^3 and 128^3), PMC traffic and SQ counters (sod 256^3)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
export AKMI_PMC_PROBLEM=sod
bash tools/pmc.sh r05_hydro > /dev/null 2>&1
cp gpurun_out/r05_hydro_pmc_traffic.json profiles/pmc_traffic_hydro_latest.json
bash tools/pmc_valu.sh r05_hydro > gpurun_out/r05_hydro_valu_counters.txt 2>&1
cp gpurun_out/r05_hydro_valu_counters.json profiles/valu_counters_hydro_latest.json
cp profiles/pmc_traffic_hydro_latest.json gpurun_out/; cp profiles/valu_counters_hydro_latest.json gpurun_out/
tail -5 gpurun_out/r05_hydro_valu_counters.txt
unset AKMI_PMC_PROBLEM
bash tools/prof.sh r05_hydro256 --problem sod --nx 256 --no-other-configs | head -8
bash tools/prof.sh r05_hydro128 --problem sod --nx 128 --no-other-configs | head -8
python bench.py --problem sod --nx 256 --no-other-configs --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/r05_hydro256_bench_full.json
python bench.py --problem sod --nx 128 --no-other-configs --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/r05_hydro128_bench_full.json
cut -c1-400 gpurun_out/r05_hydro256_bench_full.json gpurun_out/r05_hydro128_bench_full.json
