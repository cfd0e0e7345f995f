#!/bin/bash
# round 5, call 7: msda_bwd_dec with wave-group roles (A: levels 2 / 3 in LDS + flush, B: levels 0 / 1 direct atomics) against the units kernel
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
O=gpurun_out/c7
for rep in 1 2; do
for lib in new decunits; do
  if [ $lib = new ]; then unset MSDA_HIP_LIB; else export MSDA_HIP_LIB=$PWD/uninext_amd/lib/abl/libmsda_$lib.so; fi
  echo "== $lib $rep" >> $O/kbench_dec.txt
  timeout 200 python tools/kbench.py --workloads r50_train_decoder --flavours model,wide --variants-fwd 0 --variants-bwd 5 --reps 30 --rotate 3 2>&1 | grep bwd >> $O/kbench_dec.txt
  timeout 200 python tools/kbench.py --kinds decoder --flavours model,uniform --variants-fwd 0 --variants-bwd 5 --reps 30 --rotate 3 2>&1 | grep bwd >> $O/kbench_dec.txt
done; done
unset MSDA_HIP_LIB
cat $O/kbench_dec.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 200 -x -k "decoder or dec or workload or reference_selftest or smoke or ddp" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^\s*$" $O/pytest.log | tail -15
