mkdir -p gpurun_out/r3m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3m/gpu_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3m/gpu_tests.log
python bench.py > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err
tools/rocprof_stats.sh r3m_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/r3m/stats.log 2>&1
python tools/step_profile.py > gpurun_out/r3m/step_profile.txt 2>&1
tools/rocprof_pmc.sh r3m_pmc python /root/repo/tools/msda_microbench.py 5 > gpurun_out/r3m/pmc.log 2>&1
tail -3 gpurun_out/r3m/gpu_tests.log; cat gpurun_out/r3m/bench.json | cut -c1-400
