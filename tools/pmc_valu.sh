#!/bin/bash
# usage (GPU box): tools/pmc_valu.sh <tag> : SQ/GRBM counters of the stage kernels (VALU utilisation)
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmcv_$tag
mkdir -p $out
AKMI_PMC_NX=${AKMI_PMC_NX:-256} rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -- python $root/tools/pmc_workload.py > $out/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-44s %10s %12s %12s %12s %10s %10s" % ("kernel", "GUI_ACTIVE", "INSTS_VALU", "ACT_VALU", "WAVE_CYC", "WAIT_INST", "WAIT_ANY"))
for k, c in acc.items():
    if not k.startswith("akmi::k_") or "ghost" in k or "init" in k or "calib" in k: continue
    m = {n: sum(v)/len(v) for n, v in c.items()}
    print("%-44s %10.3g %12.4g %12.4g %12.4g %10.3g %10.3g" % (k[:44], m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_INSTS_VALU", 0), m.get("SQ_ACTIVE_INST_VALU", 0), m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_INST_ANY", 0), m.get("SQ_WAIT_ANY", 0)))
PY
python $root/tools/valu_summary.py $out $tag $root/gpurun_out/${tag}_valu_counters.json
