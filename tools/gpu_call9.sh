#!/bin/bash
# round 5, call 9: window geometry G2 = 12x20 / 10x14 / 10x12 / 10x10 (608 slots) against A = 14x20 / 10x14 / 8x12 / 8x10 (600) and round 4's (592)
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
O=gpurun_out/c9
for rep in 1 2; do
for lib in new fwdgeoA fwdgeo0; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_geo.txt
  timeout 200 python tools/kbench.py --kinds encoder --flavours model --variants-fwd 9 --no-bwd --reps 30 --rotate 6 2>&1 | grep fwd >> $O/kbench_geo.txt
  timeout 200 python tools/kbench.py --kinds encoder --flavours model --sigma 2.0 --variants-fwd 9,7 --no-bwd --reps 30 --rotate 6 2>&1 | grep fwd | sed 's/model /sigma2/' >> $O/kbench_geo.txt
  timeout 200 python tools/kbench.py --kinds encoder --flavours model --sigma 3.0 --variants-fwd 9 --no-bwd --reps 30 --rotate 6 2>&1 | grep fwd | sed 's/model /sigma3/' >> $O/kbench_geo.txt
done
for lib in new bwdgeoA; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_geo.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model --variants-fwd 0 --variants-bwd 4 --reps 30 --rotate 3 2>&1 | grep bwd >> $O/kbench_geo.txt
done; done
unset MSDA_HIP_LIB
cat $O/kbench_geo.txt
python tools/far_probe.py > $O/far_probe.txt 2>&1; tail -4 $O/far_probe.txt
timeout 200 python tools/win_check.py > $O/win_check.txt 2>&1; tail -2 $O/win_check.txt
timeout 300 python tools/bwin_check.py msda_bwd_win > $O/bwin_check.txt 2>&1; tail -2 $O/bwin_check.txt
