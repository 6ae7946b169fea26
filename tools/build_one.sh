#!/bin/bash
# usage: tools/build_one.sh akmi_stage.hip [akmi_tasks.hip ...]  -- recompile the named translation units and relink
# libakmi.so (the flags of __graft_entry__.build; the other objects must exist from a full build)
root=$(cd $(dirname $0)/.. && pwd)
cs=$root/athenak_amd/csrc; od=$root/athenak_amd/lib/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -mllvm -amdgpu-schedule-relaxed-occupancy=true $AKMI_EXTRA_FLAGS"
# a header newer than an object that is NOT being rebuilt: that object is stale (class layouts!) -- refuse
for h in $cs/*.hpp $root/include/akmi.h; do
  for o in $od/*.o; do
    b=$(basename $o .o); skip=0; for f in "$@"; do [ "$f" = "$b" ] && skip=1; done
    if [ $skip = 0 ] && [ $h -nt $o ]; then echo "build_one: $(basename $h) is newer than $b.o -- name $b too, or run __graft_entry__.build(force=True)"; exit 1; fi
  done
done
pids=()
for f in "$@"; do ( cd $cs && /opt/rocm/bin/hipcc $FLAGS -c $f -o $od/$f.o ) & pids+=($!); done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc = 0 ] || { echo "compile failed"; exit 1; }
cd $cs && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/athenak_amd/lib/libakmi.so $od/*.o -ldl && echo linked
