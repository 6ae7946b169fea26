#!/bin/bash
# usage (GPU box): tools/cnt.sh TAG "COUNTER COUNTER ..." [kernel-regex]   (AKMI_LIB / AKMI_* pass through)
# per-kernel averages of arbitrary rocprofv3 PMC counters over tools/pmc_workload.py (2 RK2 cycles, AKMI_PMC_NX^3)
tag=$1; ctrs=$2; pat=${3:-akmi::k_}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=/tmp/cnt_$tag
rm -rf $out; mkdir -p $out
AKMI_PMC_NX=${AKMI_PMC_NX:-256} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $root/tools/pmc_workload.py > $out/log.txt 2>&1
python - "$out" "$pat" <<'PY'
import csv, glob, collections, re, sys
out, pat = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(open(out + "/log.txt").read()[-2000:]); sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = []
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
print("%-44s %5s " % ("kernel", "n") + " ".join("%14s" % n[:14] for n in names))
for k, c in acc.items():
    if not re.search(pat, k) or "calib" in k or "init" in k: continue
    print("%-44s %5d " % (k[:44], len(c[names[0]])) + " ".join("%14.5g" % (sum(c[n])/max(1, len(c[n]))) for n in names))
PY
