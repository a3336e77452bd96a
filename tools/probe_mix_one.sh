#!/bin/bash
# Build timing-probe variants of libdmn_hip.so (mix.hip with -DDMN_MIX_ONE_PROBE=n, every other object as built)
# into tools/probe/libdmn_mixone_pN.so; run on the GPU box as
#   DMN_HIP_LIB=tools/probe/libdmn_mixone_p1.so DMN_MIX_BWD_ONE=1 python tools/mix_bwd_bench.py
set -e
cd "$(dirname "$0")/.."
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDMN_MIX_ONE_PROBE=$n -c dms_amd/csrc/mix.hip -o tools/probe/mix_p$n.o
  objs=$(ls dms_amd/build/*.o | grep -v "/mix.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libdmn_mixone_p$n.so $objs tools/probe/mix_p$n.o
done
