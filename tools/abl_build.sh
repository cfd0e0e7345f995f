#!/bin/bash
# ablation build: tools/abl_build.sh <name> <source-stem> "<extra -D flags>"  ->  uninext_amd/lib/abl/libmsda_<name>.so
# (the product objects with ONE translation unit recompiled with extra flags; run with MSDA_HIP_LIB=<that file>)
set -e
name=$1; stem=$2; flags=$3
cd /root/repo/uninext_amd/csrc
mkdir -p ../lib/abl /tmp/abl
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-strict-aliasing -Wno-unused-parameter $flags -c $stem.hip -o /tmp/abl/${stem}_$name.o
objs=$(ls *.o | grep -v "^$stem.o$" | grep -v "_prof.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/abl/libmsda_$name.so $objs /tmp/abl/${stem}_$name.o
echo built ../lib/abl/libmsda_$name.so
