#!/bin/bash
# round 6, GPU box: everything the committed library is judged on -- GPU suite (4 workers), MHD profiles (kernel stats, PMC traffic,
# SQ counters) + default bench line, hydro profiles, PPM4 line, config 5.  usage: bash tools/r05_final.sh TAG
tag=${1:-r06_v7}
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -n 4 ) > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -4 gpurun_out/${tag}_gpu_tests.txt
bash tools/pmc.sh $tag > /dev/null 2>&1
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic_latest.json
bash tools/pmc_valu.sh $tag > gpurun_out/${tag}_valu_counters.txt 2>&1
cp gpurun_out/${tag}_valu_counters.json profiles/valu_counters_latest.json
tail -7 gpurun_out/${tag}_valu_counters.txt
export AKMI_PMC_PROBLEM=sod
bash tools/pmc.sh ${tag}_hydro > /dev/null 2>&1
cp gpurun_out/${tag}_hydro_pmc_traffic.json profiles/pmc_traffic_hydro_latest.json
bash tools/pmc_valu.sh ${tag}_hydro > gpurun_out/${tag}_hydro_valu_counters.txt 2>&1
cp gpurun_out/${tag}_hydro_valu_counters.json profiles/valu_counters_hydro_latest.json
tail -4 gpurun_out/${tag}_hydro_valu_counters.txt
unset AKMI_PMC_PROBLEM
for f in pmc_traffic_latest pmc_traffic_hydro_latest valu_counters_latest valu_counters_hydro_latest; do cp profiles/$f.json gpurun_out/${tag}_$f.json; done
python bench.py > gpurun_out/${tag}_bench_full.log 2>&1
grep '^{"metric"' gpurun_out/${tag}_bench_full.log | tail -1 > gpurun_out/${tag}_bench_full.json
cut -c1-3000 gpurun_out/${tag}_bench_full.json
bash tools/prof.sh $tag --steps 10 | head -12
bash tools/prof.sh ${tag}_hydro256 --problem sod --nx 256 --no-other-configs | head -6
bash tools/prof.sh ${tag}_hydro128 --problem sod --nx 128 --no-other-configs | head -6
python bench.py --problem sod --nx 256 --no-other-configs --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/${tag}_hydro256_bench_full.json
python bench.py --problem sod --nx 128 --no-other-configs --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/${tag}_hydro128_bench_full.json
python bench.py --recon ppm4 --no-other-configs --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/${tag}_bench_ppm4.json
cut -c1-330 gpurun_out/${tag}_hydro256_bench_full.json gpurun_out/${tag}_hydro128_bench_full.json gpurun_out/${tag}_bench_ppm4.json
bash tools/config5_prof.sh ${tag} > /dev/null 2>&1; head -12 gpurun_out/${tag}_config5.txt
