#!/bin/bash
# round 5, GPU box: HBM-side bytes and L2 hit rates of the memory-bound kernels of config 5 at production size (960 x 32^3)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp AKMI_CONFIG5_CPU=0 AKMI_CONFIG5_HOSTS=c++
PROD="mesh/nx1=256 mesh/nx2=256 mesh/nx3=256 meshblock/nx1=32 meshblock/nx2=32 meshblock/nx3=32"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1)); out=$root/gpurun_out/c5pmc_$i; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- python $root/tools/config5.py 2 $PROD > $out/log.txt 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$root/gpurun_out/c5pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-44s %6s %10s %10s %9s %9s %9s %10s %10s" % ("kernel", "calls", "FETCH GB", "WRITE GB", "L2 hit", "L2 miss", "hit %", "TCP rd req", "TCP acc"))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    if not k.startswith("akmi::k_"): continue
    m = {n: sum(v)/len(v) for n, v in c.items()}
    n = len(c.get("FETCH_SIZE", []))
    # guide: FETCH_SIZE unit 1024 B nominal, ~2048 B for wide coalesced read streams on gfx950 (calibrated so in tools/pmc.sh); WRITE 1024 B
    print("%-44s %6d %10.3f %10.3f %9.3g %9.3g %8.1f%% %10.3g %10.3g" % (k[:44], n, m.get("FETCH_SIZE", 0)*2048/1e9, m.get("WRITE_SIZE", 0)*1024/1e9,
          m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0), 100*m.get("TCC_HIT_sum", 0)/max(m.get("TCC_REQ_sum", 1), 1), m.get("TCP_TCC_READ_REQ_sum", 0), m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)))
PY
