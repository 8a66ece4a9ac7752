mkdir -p gpurun_out/r3v
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3v/gpu_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3v/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3v/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r3v/smoke.log
python bench.py > gpurun_out/r3v/bench.json 2> gpurun_out/r3v/bench.err
tools/rocprof_stats.sh r3v_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/r3v/stats.log 2>&1
python tools/step_profile.py > gpurun_out/r3v/step_profile.txt 2>&1
python tools/host_profile.py --rows 12 > gpurun_out/r3v/host_profile.txt 2>&1
python bench.py --model s --img 640 --batch 16 --dtype fp32 --cpu-steps 0 --steps 20 --warmup 5 > gpurun_out/r3v/bench_s_fp32.json 2>/dev/null
tail -3 gpurun_out/r3v/gpu_tests.log; tail -1 gpurun_out/r3v/smoke.log; cut -c1-330 gpurun_out/r3v/bench.json; cut -c1-250 gpurun_out/r3v/bench_s_fp32.json; grep un-profiled gpurun_out/r3v/host_profile.txt
