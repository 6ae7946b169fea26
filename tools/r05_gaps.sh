#!/bin/bash
# GPU box: kernel timeline of the 128^3 Sod deck (configs[1]): where the time between the kernels of a cycle goes
root=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp; rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $root/bench.py --no-cpu-baseline --problem sod --nx ${NX:-128} --no-other-configs > /tmp/gp.log 2>&1
python - > $root/gpurun_out/r05_gaps_${NX:-128}.txt <<'PY'
import csv, glob
f = glob.glob('/tmp/gp/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
# the first host's timed loop: take stage kernels 150..170
idx = [i for i, r in enumerate(rows) if "k_hydro_stage3d" in r[2]]
a = idx[60]; b = idx[66]
t0 = rows[a][0]
prev_end = None
print("%10s %9s %9s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
for s, e, n in rows[a:b]:
    print("%10.1f %9.1f %9.1f  %s" % ((s - t0)/1e3, (e - s)/1e3, (s - prev_end)/1e3 if prev_end else 0.0, n[:70]))
    prev_end = e
# totals over 40 cycles
a = idx[40]; b = idx[120]
busy = sum(e - s for s, e, n in rows[a:b]); span = rows[b][0] - rows[a][0]
print("40 cycles: span %.1f us/cycle, busy %.1f us/cycle" % (span/40e3, busy/40e3))
PY
cat $root/gpurun_out/r05_gaps_${NX:-128}.txt
