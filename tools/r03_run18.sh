#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
{
export AKMI_CONFIG5_CPU=0
export AKMI_LIB=$root/athenak_amd/lib/variants/libakmi_tf.so
for d in 0 1 0 1; do
echo "## AKMI_FACE_SWEEPS=$d (variant tf)"
AKMI_FACE_SWEEPS=$d python tools/config5.py 40 2>&1 | grep "config 5"
AKMI_FACE_SWEEPS=$d python bench.py --steps 40 --warmup 5 --no-cpu-baseline --nx 64 --split 2>&1 | tail -1 | cut -c1-140
AKMI_FACE_SWEEPS=$d python bench.py --steps 40 --warmup 5 --no-cpu-baseline --nx 64 --mb 16 --split --recon ppm4 --ng 4 2>&1 | tail -1 | cut -c1-140
AKMI_FACE_SWEEPS=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --nx 128 --mb 16 --split --recon ppm4 --ng 4 2>&1 | tail -1 | cut -c1-140
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp6; AKMI_FACE_SWEEPS=1 rocprofv3 --kernel-trace --stats -d /tmp/pp6 -- python $root/tools/config5.py 40 > /tmp/pp6.log 2>&1
python $root/tools/kernel_stats.py /tmp/pp6 "deck-size run, thread-per-face k_sweep<1|2>" | head -10
python -c "
import sys; sys.path.insert(0,'$root'); sys.path.insert(0,'$root/tests')
import parity_util as pu
r = pu.compare_run('blast_smr', (32,32,32), 3, (8,8,8), cycles=2)
print('parity', r['bitwise_equal'])
r = pu.compare_run('orszag_tang', n=32, dims=3, mb=16, cycles=2, fused=False)
print('parity', r['bitwise_equal'])
" 2>&1 | tail -3
} > $root/gpurun_out/r03_run18.txt 2>&1
cat $root/gpurun_out/r03_run18.txt | cut -c1-150
