#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFLAG ...]   -> athenak_amd/lib/variants/libakmi_NAME.so
# (akmi_stage.hip recompiled with the extra flags, the other objects of the default build reused;
#  -DAKMI_DEV_FAST restricts the scheme dispatch to PLM + HLLD/HLLC, ideal gas: seconds instead of minutes;
#  AKMI_ISA=1 keeps the device assembly as /tmp/akmi_var_NAME/akmi_stage-hip-amdgcn-amd-amdhsa-gfx950.s)
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/athenak_amd/lib/variants /tmp/akmi_var_$name
cd $root/athenak_amd/csrc
extra=""
[ -n "$AKMI_ISA" ] && extra="--save-temps=obj"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value \
  -mllvm -amdgpu-schedule-relaxed-occupancy=true $extra "$@" -c akmi_stage.hip -o /tmp/akmi_var_$name/akmi_stage.hip.o
objs=$(ls $root/athenak_amd/lib/obj/*.o | grep -v akmi_stage)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/athenak_amd/lib/variants/libakmi_$name.so $objs /tmp/akmi_var_$name/akmi_stage.hip.o -ldl
echo built $name
