#!/bin/bash
# round 5, call 6: timing-only ablations (WRONG results by construction; experiments/*.patch) -- where msda_bwd_dec's and msda_fwd_win's time goes
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
O=gpurun_out/c6
for lib in product decnodirect decnoflush decnolds decnoloads decnoatomics decnothing; do
  if [ $lib = product ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib" >> $O/dec_ablations.txt
  timeout 120 python tools/kbench.py --workloads r50_train_decoder --flavours model --variants-fwd 0 --variants-bwd 5 --reps 30 --rotate 3 2>&1 | grep bwd >> $O/dec_ablations.txt
done
for rep in 1 2; do
for lib in product winnofar; do
  if [ $lib = product ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/win_nofar.txt
  timeout 120 python tools/kbench.py --kinds encoder --flavours model --variants-fwd 9 --no-bwd --reps 30 --rotate 6 2>&1 | grep fwd >> $O/win_nofar.txt
done; done
unset MSDA_HIP_LIB
cat $O/dec_ablations.txt $O/win_nofar.txt
