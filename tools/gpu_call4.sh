#!/bin/bash
# round 5, call 4: bwd_win transition placement A/B (late = behind barrier #3, early = in front of it), timelines, full suite
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
O=gpurun_out/c4
for rep in 1 2; do
for lib in new bwearly bwinold; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_bwin.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model --variants-fwd 0 --variants-bwd 4 --reps 30 --rotate 3 2>&1 | grep -v amdgpu.ids | grep bwd >> $O/kbench_bwin.txt
done; done
cat $O/kbench_bwin.txt
for lib in bwprof bwprofearly; do
  echo "== $lib" >> $O/bwin_prof.txt
  MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so timeout 120 python tools/bwin_prof.py model 2>&1 | grep -v amdgpu.ids >> $O/bwin_prof.txt
done
unset MSDA_HIP_LIB
cat $O/bwin_prof.txt
MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_bwearly.so timeout 300 python tools/bwin_check.py msda_bwd_win > $O/bwin_check_early.txt 2>&1; tail -2 $O/bwin_check_early.txt
unset MSDA_HIP_LIB
timeout 900 python -m pytest tests -m gpu -q --timeout 200 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^\s*$" $O/pytest.log | tail -30
