mkdir -p gpurun_out/r3t
python bench.py > gpurun_out/r3t/bench.json 2> gpurun_out/r3t/bench.err
tools/rocprof_stats.sh r3t_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/r3t/stats.log 2>&1
python tools/step_profile.py > gpurun_out/r3t/step_profile.txt 2>&1
python tools/host_profile.py --rows 12 > gpurun_out/r3t/host_profile.txt 2>&1
cut -c1-330 gpurun_out/r3t/bench.json; grep -v amdgpu gpurun_out/r3t/step_profile.txt | head -24; grep un-profiled gpurun_out/r3t/host_profile.txt
