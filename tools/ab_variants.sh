# usage (GPU box): build variant libraries into athenak_amd/lib/variants/ first, e.g.
#   cd athenak_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fPIC -shared -o ../lib/variants/libakmi_fast.so akmi_tasks.hip akmi_bvals.hip akmi_stage.hip akmi_host.cpp
# (same with -ffp-contract=off -freciprocal-math -> libakmi_recip.so); result: profiles/r01_fastmath_ab.txt
for v in "" athenak_amd/lib/variants/libakmi_fast.so athenak_amd/lib/variants/libakmi_recip.so; do
  if [ -n "$v" ]; then export AKMI_LIB=$v; else unset AKMI_LIB; fi
  echo "== ${v:-default}"
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['ms_per_launch'])"
  timeout 300 python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import parity_util as pu
for case in [("orszag_tang",32,3,16,6,dict(cfl=0.3)),("blast",24,3,12,4,{}),("sod",32,3,16,6,dict(cfl=0.3))]:
    r=pu.compare_run(*case[:5],**case[5]); print(case[0], "max_rel_l1 %.3e"%r["max_rel_l1"], r["bitwise_equal"], r["time"][0]==r["time"][1])
PY
done
