mkdir -p gpurun_out/r3q
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3q/gpu_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3q/gpu_tests.log
python bench.py > gpurun_out/r3q/bench.json 2> gpurun_out/r3q/bench.err
tools/rocprof_stats.sh r3q_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/r3q/stats.log 2>&1
python tools/step_profile.py > gpurun_out/r3q/step_profile.txt 2>&1
tools/rocprof_pmc.sh r3q_pmc python /root/repo/tools/msda_microbench.py 5 > gpurun_out/r3q/pmc.log 2>&1
tail -3 gpurun_out/r3q/gpu_tests.log; cut -c1-400 gpurun_out/r3q/bench.json
