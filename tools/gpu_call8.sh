#!/bin/bash
# round 5, call 8: window geometry 14x20 / 10x14 / 8x12 / 8x10 (was 14x22 / 10x14 / 8x10 / 7x8) in msda_fwd_win and msda_bwd_win
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
O=gpurun_out/c8
for rep in 1 2; do
for lib in new fwdgeo0; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_geo.txt
  timeout 200 python tools/kbench.py --kinds encoder --flavours model,wide --variants-fwd 9 --no-bwd --reps 30 --rotate 6 2>&1 | grep fwd >> $O/kbench_geo.txt
done
for lib in new bwdgeo0; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_geo.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model --variants-fwd 0 --variants-bwd 4 --reps 30 --rotate 3 2>&1 | grep bwd >> $O/kbench_geo.txt
done; done
unset MSDA_HIP_LIB
cat $O/kbench_geo.txt
python tools/far_probe.py > $O/far_probe.txt 2>&1; tail -5 $O/far_probe.txt
timeout 200 python tools/win_check.py > $O/win_check.txt 2>&1; tail -3 $O/win_check.txt
timeout 300 python tools/bwin_check.py msda_bwd_win > $O/bwin_check.txt 2>&1; tail -2 $O/bwin_check.txt
