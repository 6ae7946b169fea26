#!/bin/bash
# usage (GPU box): tools/pmc_detail.sh <tag>: LDS / vector-memory / L2 counters of the stage kernels, one counter group per pass
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TD_BUSY_avr"; do
  i=$((i+1))
  out=$root/gpurun_out/pmcd_${tag}_$i
  mkdir -p $out
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- python $root/tools/pmc_workload.py > $out/log.txt 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$root/gpurun_out/pmcd_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for c in acc.values() for n in c})
for k, c in acc.items():
    if not k.startswith("akmi::k_") or "ghost" in k or "init" in k or "calib" in k or "shell" in k or "newdt<" in k and "c2p" not in k: continue
    print(k[:60])
    for n in names:
        if n in c: print("   %-32s %12.4g" % (n, sum(c[n])/len(c[n])))
PY
