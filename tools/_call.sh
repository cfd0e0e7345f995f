set -x
mkdir -p gpurun_out/c1
python -m pytest tests/test_ddp_gpu.py tests/test_msda_gpu.py::test_module_under_inference_mode -x -q > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1/pytest.log
python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; echo "bench rc $?" >> gpurun_out/c1/bench.err
python tools/measure_traffic.py --out gpurun_out/c1/traffic.json --sq gpurun_out/c1/sq_pmc.txt > gpurun_out/c1/traffic.log 2>&1; echo "traffic rc $?" >> gpurun_out/c1/traffic.log
python tools/kbench.py --reps 18 --rotate 6 --variants-fwd 2,7 --variants-bwd 1,3 > gpurun_out/c1/kbench.txt 2>&1
tail -5 gpurun_out/c1/pytest.log; cat gpurun_out/c1/bench.json; tail -3 gpurun_out/c1/bench.err; tail -30 gpurun_out/c1/traffic.log; cat gpurun_out/c1/kbench.txt
