mkdir -p gpurun_out/c10
timeout 300 python tools/win_check.py > gpurun_out/c10/check.txt 2>&1; echo "rc $?" >> gpurun_out/c10/check.txt
timeout 120 python tools/kbench.py --reps 18 --rotate 6 --variants-fwd 7,9 --kinds encoder --no-bwd > gpurun_out/c10/kbench.txt 2>&1
MSDA_WIN_PERSIST=1 timeout 120 python tools/kbench.py --reps 18 --rotate 6 --variants-fwd 9 --kinds encoder --no-bwd --flavours model >> gpurun_out/c10/kbench.txt 2>&1
timeout 120 python tools/kbench.py --reps 18 --variants-fwd 7,9 --kinds encoder --no-bwd --flavours model >> gpurun_out/c10/kbench.txt 2>&1
timeout 200 python tools/win_prof.py model > gpurun_out/c10/prof.txt 2>&1
MSDA_HIP_FWD_VARIANT=9 timeout 300 python tools/measure_traffic.py --out gpurun_out/c10/traffic.json --sq gpurun_out/c10/sq.txt > gpurun_out/c10/traffic.log 2>&1
grep -c "^ok" gpurun_out/c10/check.txt; grep -v "^ok" gpurun_out/c10/check.txt | head -20; cat gpurun_out/c10/kbench.txt gpurun_out/c10/prof.txt; head -18 gpurun_out/c10/traffic.json | tail -8; head -12 gpurun_out/c10/sq.txt
