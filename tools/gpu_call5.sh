#!/bin/bash
# round 5, call 5: where do msda_bwd_win's +30 us come from -- item order or transition?  new / no carry / round-4 order / round-4 kernel
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
O=gpurun_out/c5
for rep in 1 2; do
for lib in new bwnocarry bwstrided bwinold; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_bwin.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model --variants-fwd 0 --variants-bwd 4 --reps 30 --rotate 3 2>&1 | grep -v amdgpu.ids | grep bwd >> $O/kbench_bwin.txt
done; done
cat $O/kbench_bwin.txt
echo "== round-4 kernel" >> $O/bwin_prof.txt
MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_bwinoldprof.so timeout 120 python tools/bwin_prof_r04.py model 2>&1 | grep -v amdgpu.ids >> $O/bwin_prof.txt
cat $O/bwin_prof.txt
