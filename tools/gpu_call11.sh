#!/bin/bash
# round 5, call 11: msda_bwd_tiled with level-dependent window margins against the uniform margin
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
O=gpurun_out/c11
for rep in 1 2; do
for lib in new tiledgeo0; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_tiled.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model,wide --variants-fwd 0 --variants-bwd 3 --reps 24 --rotate 3 2>&1 | grep bwd >> $O/kbench_tiled.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model --sigma 2.0 --variants-fwd 0 --variants-bwd 3,4 --reps 24 --rotate 3 2>&1 | grep bwd | sed 's/model /sigma2/' >> $O/kbench_tiled.txt
  timeout 200 python tools/kbench.py --workloads r50_train_encoder --flavours model --sigma 3.0 --variants-fwd 0 --variants-bwd 3,6 --reps 24 --rotate 3 2>&1 | grep bwd | sed 's/model /sigma3/' >> $O/kbench_tiled.txt
done; done
unset MSDA_HIP_LIB
cat $O/kbench_tiled.txt
timeout 600 python -m pytest tests -m gpu -q --timeout 200 -x -k "tiled or backward or workload or gradcheck" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^\s*$" $O/pytest.log | tail -6
