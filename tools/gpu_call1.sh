#!/bin/bash
# round 5, call 1: full -m gpu suite on the new tree + decoder backward A/B (old = lib/abl/libmsda_decold.so)
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1/pytest.log
tail -5 gpurun_out/c1/pytest.log
for rep in 1 2; do
for lib in new decold; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> gpurun_out/c1/kbench_dec.txt
  timeout 300 python tools/kbench.py --workloads r50_train_decoder,r50_train_encoder --flavours model --variants-fwd 0 --variants-bwd 5,4 --reps 30 --rotate 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/c1/kbench_dec.txt
  timeout 300 python tools/kbench.py --kinds decoder --flavours model,uniform --variants-fwd 0 --variants-bwd 5 --reps 30 --rotate 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/c1/kbench_dec.txt
done; done
unset MSDA_HIP_LIB
cat gpurun_out/c1/kbench_dec.txt
