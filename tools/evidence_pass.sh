#!/bin/bash
# The round's evidence pass (GPU box): bench line, rocprofv3 kernel traces of the bench, PMC traffic + SQ counters of the contract
# launches, kbench tables, the caller-side benches.  Everything lands under gpurun_out/ev/; copy what is to be judged into profiles/.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
# traffic first: bench.py quotes profiles/traffic.json when kernel names and source hash match its own run
timeout 900 python tools/measure_traffic.py --out $O/traffic.json --sq $O/sq_pmc.txt --workdir $O/traffic_prof > $O/measure_traffic.log 2>&1
cp $O/traffic.json profiles/traffic.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_noextras -- python $R/bench.py --no-cpu-baseline --no-extras > $O/trace_noextras.json 2> $O/trace_noextras.err )
python tools/rocprof_summary.py trace $(find $O/trace_noextras -name "*_results.db" | head -1) > $O/bench_kernel_trace_noextras.txt 2>&1
cat $O/trace_noextras.json >> $O/bench_kernel_trace_noextras.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_full -- python $R/bench.py --no-cpu-baseline --extras-only flavours,backward,train > $O/trace_full.json 2> $O/trace_full.err )
python tools/rocprof_summary.py trace $(find $O/trace_full -name "*_results.db" | head -1) > $O/bench_kernel_trace.txt 2>&1
cat $O/trace_full.json >> $O/bench_kernel_trace.txt
timeout 600 python tools/kbench.py --reps 18 --rotate 6 --variants-fwd 7,9 --variants-bwd 3,4,5,6,8 --flavours model,wide,uniform > $O/kbench_final.txt 2>&1
timeout 900 python tools/kbench.py --reps 12 --rotate 3 --workloads all --flavours model,wide > $O/kbench_workloads.txt 2>&1
timeout 200 python tools/ota_bench.py > $O/ota_bench.txt 2>&1
# round 5: the dynamic mask head under autograd (time / peak memory against the PyTorch composition), the option-A stand-in
timeout 300 python -m pytest tests/test_dynmask_gpu.py -q -s -k "training_shape or match_the_reference_under_autograd" > $O/dynmask_backward.txt 2>&1
# decoder backward: query slices per (image, head) with the (query, level) units of round 5 (experiments build: the hook is not in the product)
if [ -f uninext_amd/lib/libmsda_hip_exp.so ]; then
  for sl in 0 8 12 24 32; do
    echo "== MSDA_BWD_DEC_SLICES=$sl (0: the default rule = 16)" >> $O/dec_slices.txt
    MSDA_HIP_LIB=$R/uninext_amd/lib/libmsda_hip_exp.so MSDA_BWD_DEC_SLICES=$sl timeout 120 python tools/kbench.py --workloads r50_train_decoder --flavours model --variants-fwd 0 --variants-bwd 5 --reps 30 --rotate 3 2>&1 | grep bwd >> $O/dec_slices.txt
  done
fi
timeout 200 python tools/module_bench.py --reps 30 --rotate 3 > $O/module_bench.txt 2>&1
# round 6: the static mask head in exact fp32, own MFMA convolution against MIOpen
timeout 200 python tools/maskhead_exact_ab.py > $O/maskhead_exact_ab.txt 2>&1
rm -rf $O/trace_noextras $O/trace_full $O/traffic_prof
ls -la $O
