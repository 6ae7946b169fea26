#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
export AKMI_CONFIG5_CPU=0 AKMI_FACE_SWEEPS=1
for v in tf0 tf tf0 tf; do
export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so
echo "## variant $v"
python tools/config5.py 40 2>&1 | grep "config 5"
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --nx 64 --split 2>&1 | tail -1 | cut -c1-140
done
cd /tmp && export TMPDIR=/tmp
for v in tf0 tf; do
export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_$v.so
rm -rf /tmp/pp6; rocprofv3 --kernel-trace --stats -d /tmp/pp6 -- python $root/tools/config5.py 40 > /tmp/pp6.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp6 "deck-size run, $v" | head -6
done
python -c "
import sys; sys.path.insert(0,'$root'); sys.path.insert(0,'$root/tests')
import parity_util as pu
for kw in (dict(), dict(recon='plm', ng=2)):
    r = pu.compare_run('blast_smr', (32,32,32), 3, (8,8,8), cycles=2, **kw)
    print('parity', r['bitwise_equal'])
r = pu.compare_run('orszag_tang', n=32, dims=3, mb=16, cycles=2, fused=False)
print('parity', r['bitwise_equal'])
r = pu.compare_run('sod', n=32, dims=3, mb=16, cycles=2, fused=False)
print('parity', r['bitwise_equal'])
" 2>&1 | tail -5
} > $root/gpurun_out/r03_run19.txt 2>&1
cat $root/gpurun_out/r03_run19.txt | cut -c1-150
