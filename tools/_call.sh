mkdir -p gpurun_out/c6
timeout 300 python tools/win_check.py --quick > gpurun_out/c6/check.txt 2>&1; echo "rc $?" >> gpurun_out/c6/check.txt
timeout 120 python tools/kbench.py --reps 18 --rotate 6 --variants-fwd 7,9 --kinds encoder --no-bwd --flavours model > gpurun_out/c6/kbench.txt 2>&1
timeout 120 python tools/kbench.py --reps 18 --variants-fwd 7,9 --kinds encoder --no-bwd --flavours model >> gpurun_out/c6/kbench.txt 2>&1
timeout 200 python tools/win_prof.py model > gpurun_out/c6/prof.txt 2>&1
MSDA_HIP_FWD_VARIANT=9 timeout 300 python tools/measure_traffic.py --out gpurun_out/c6/traffic.json --sq gpurun_out/c6/sq.txt > gpurun_out/c6/traffic.log 2>&1
grep -c "^ok" gpurun_out/c6/check.txt; grep -v "^ok" gpurun_out/c6/check.txt; cat gpurun_out/c6/kbench.txt gpurun_out/c6/prof.txt; head -30 gpurun_out/c6/traffic.json; head -12 gpurun_out/c6/sq.txt
