#!/bin/bash
# A/B build of ONE experiment kernel: tools/exp_build.sh <name> <stem in csrc/experiments> "<extra -D flags>"
#   -> uninext_amd/lib/abl/libmsda_<name>.so = the experiments library's objects (make -C uninext_amd/csrc experiments first)
#      with experiments/<stem>.hip recompiled with the extra flags; run with MSDA_HIP_LIB=<that file>
set -e
name=$1; stem=$2; flags=$3
cd /root/repo/uninext_amd/csrc
mkdir -p ../lib/abl /tmp/abl
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-strict-aliasing -Wno-unused-parameter -DMSDA_EXPERIMENTS $flags -c experiments/$stem.hip -o /tmp/abl/${stem}_$name.o
objs=$(ls experiments/obj/*.o | grep -v "/$stem.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/abl/libmsda_$name.so $objs msda_host.o /tmp/abl/${stem}_$name.o
echo built ../lib/abl/libmsda_$name.so
