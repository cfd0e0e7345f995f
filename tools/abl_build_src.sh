#!/bin/bash
# A/B build from ANOTHER source file: tools/abl_build_src.sh <name> <stem> <source file> ["<extra flags>"]
#   -> uninext_amd/lib/abl/libmsda_<name>.so = the product objects with <stem>.o replaced by <source file> compiled with the product flags
# (e.g. the committed version of a kernel beside the working tree's: git show HEAD:uninext_amd/csrc/x.hip > /tmp/x_old.hip)
set -e
name=$1; stem=$2; src=$3; flags=$4
cd /root/repo/uninext_amd/csrc
mkdir -p ../lib/abl /tmp/abl
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-strict-aliasing -Wno-unused-parameter -I. $flags -x hip -c $src -o /tmp/abl/${stem}_$name.o
objs=$(ls *.o | grep -v "^$stem.o$" | grep -v "_prof.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/abl/libmsda_$name.so $objs /tmp/abl/${stem}_$name.o
echo built ../lib/abl/libmsda_$name.so
