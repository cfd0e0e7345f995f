#!/bin/bash
# round 5, call 3: bwd_win (per-item scale, LDS half / memory half of the transition): parity, A/B, timeline; full -m gpu suite
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
O=gpurun_out/c3
timeout 300 python tools/bwin_check.py msda_bwd_win > $O/bwin_check.txt 2>&1; echo "bwin_check rc $?" >> $O/bwin_check.txt
grep -c MISMATCH $O/bwin_check.txt; tail -4 $O/bwin_check.txt
for rep in 1 2; do
for lib in new bwinold; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_bwin.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model,wide --variants-fwd 0 --variants-bwd 4 --reps 30 --rotate 3 2>&1 | grep -v amdgpu.ids | grep bwd >> $O/kbench_bwin.txt
done; done
cat $O/kbench_bwin.txt
MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_bwprof.so timeout 120 python tools/bwin_prof.py model > $O/bwin_prof.txt 2>&1
unset MSDA_HIP_LIB
cat $O/bwin_prof.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 200 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -40 $O/pytest.log
