#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <tag>
# HBM traffic of every kernel of the bench from rocprofv3 PMC counters, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes (TCC slots), kernel-trace only (no sys/hip/hsa trace together with --pmc), plus a
# calibration kernel of known byte count run in the same process.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_$c
  mkdir -p $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python $root/tools/pmc_workload.py > $out/log.txt 2>&1
done
python $root/tools/pmc_summary.py $root/gpurun_out/pmc_${tag}_FETCH_SIZE $root/gpurun_out/pmc_${tag}_WRITE_SIZE $tag > $root/gpurun_out/${tag}_pmc_traffic.json
cat $root/gpurun_out/${tag}_pmc_traffic.json | head -60
