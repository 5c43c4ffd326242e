#!/bin/bash
# Diagnostic: build a what-if variant of the library into tools/variants/NAME.so with extra defines for mimo_kernels.hip
# (the other objects are reused from the regular build):   tools/build_variant.sh NAME "-DMIMO_ASYM_PRIO=1" [file.hip ...]
set -e
cd "$(dirname "$0")/../mimo_amd/csrc"
name=$1; defs=$2; shift 2
files=${@:-mimo_kernels.hip}
mkdir -p ../../tools/variants /tmp/variant_$name
objs=""
for o in mimo_kernels mimo_small mimo_rowwave mimo_wide mimo_narrow mimo_narrow_table mimo_narrow_grouped mimo_narrow_big mimo_mid mimo_predict; do
  if [[ " $files " == *" $o.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-honor-nans $defs -c $o.hip -o /tmp/variant_$name/$o.o
    objs="$objs /tmp/variant_$name/$o.o"
  else
    objs="$objs $o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread $objs mimo_abi.o mimo_comm.o mimo_host.o -ldl -o ../../tools/variants/$name.so
echo built tools/variants/$name.so
