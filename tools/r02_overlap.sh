#!/bin/bash
# A/B of the k-slab pipeline with the sweeps' residency capped by unused LDS (GPU box)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
run() {
  echo "== $*"
  env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["ms_per_step"])'
}
run A=0
run AKMI_SLAB_CELLS=64
run AKMI_SLAB_CELLS=64 AKMI_MARCH_LDS=11264 AKMI_X1_LDS=49152 AKMI_CT_TILE=64,4
run AKMI_SLAB_CELLS=32 AKMI_MARCH_LDS=11264 AKMI_X1_LDS=49152 AKMI_CT_TILE=64,4
run AKMI_SLAB_CELLS=128 AKMI_MARCH_LDS=11264 AKMI_X1_LDS=49152 AKMI_CT_TILE=64,4
run AKMI_SLAB_CELLS=64 AKMI_MARCH_LDS=11264 AKMI_CT_TILE=64,4
run AKMI_SLAB_CELLS=64 AKMI_MARCH_LDS=11264 AKMI_X1_LDS=49152
run AKMI_MARCH_LDS=11264 AKMI_X1_LDS=49152
run A=0
