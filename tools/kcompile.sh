#!/bin/bash
# compile one kernel file with resource remarks; ISA lands next to the object (gitignored *.s? no: removed after)
cd /root/repo/uninext_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-strict-aliasing -Wall -Wextra -Wno-unused-parameter -c $1.hip -o $1.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -v "^$" | grep -i "error\|warning\|Function Name\|SGPRs\|VGPRs\|Scratch\|Occupancy\|LDS Size" 
mkdir -p /tmp/isa && mv /root/repo/uninext_amd/csrc/$1-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/isa/ 2>/dev/null; rm -f /root/repo/uninext_amd/csrc/$1-h*.{bc,hipi,hipfb,s,o,out,ll} /root/repo/uninext_amd/csrc/$1-hip-* /root/repo/uninext_amd/csrc/$1-host-* 2>/dev/null; true
