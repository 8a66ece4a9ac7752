mkdir -p gpurun_out/r7
(timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4) > gpurun_out/r7/gputests.log
python bench.py > gpurun_out/r7/bench.log 2>&1
tools/rocprof_stats.sh r7_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/r7/stats_cmd.log 2>&1
STEP_PROFILE_TOP=40 python tools/step_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/r7/step_profile.log
tools/rocprof_pmc.sh r7_pmc python /root/repo/tools/msda_microbench.py 5 > gpurun_out/r7/pmc_cmd.log 2>&1
python tools/host_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | head -60 > gpurun_out/r7/host_profile.log
python tools/stream_timeline.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/r7/timeline.log
python tools/host_window.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/r7/host_window.log
python tools/msda_microbench.py 20 2>&1 | tail -1 > gpurun_out/r7/msda_microbench.log
MSDA_PADS=1 python tools/msda_acc_bench.py 10 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/r7/msda_acc_pads.log
