#!/bin/bash
# usage (GPU box): tools/pmc_mem.sh <tag>: address-translation / L1 / L2-to-memory stall counters of the stage kernels
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  out=$root/gpurun_out/pmcm_${tag}_$i
  mkdir -p $out
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- python $root/tools/pmc_workload.py > $out/log.txt 2>&1 || echo "group $i: timed out or failed"
  tail -2 $out/log.txt | grep -i "error\|invalid" | head -2
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$root/gpurun_out/pmcm_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for c in acc.values() for n in c})
ks = [k for k in acc if k.startswith("akmi::k_") and not any(x in k for x in ("ghost", "init", "shell", "k_newdt"))]
print("%-34s" % "counter" + "".join("%16s" % k.replace("akmi::", "")[:15] for k in ks))
for n in names:
    print("%-34s" % n + "".join("%16.4g" % (sum(acc[k][n])/len(acc[k][n]) if n in acc[k] else float("nan")) for k in ks))
PY
