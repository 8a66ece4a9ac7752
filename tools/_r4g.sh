mkdir -p gpurun_out/r4g
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r4g/tests.log 2>&1; tail -3 gpurun_out/r4g/tests.log
python bench.py > gpurun_out/r4g/bench_line.json 2> gpurun_out/r4g/bench.err; tail -1 gpurun_out/r4g/bench.err | cut -c1-200
bash tools/rocprof_stats.sh r4g_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/r4g/rocprof.log 2>&1; rm -f gpurun_out/r4g_stats/*kernel_trace.csv
python tools/step_profile.py 2>/dev/null > gpurun_out/r4g/step_profile.txt; head -22 gpurun_out/r4g/step_profile.txt
python tools/host_profile.py 2>/dev/null > gpurun_out/r4g/host_profile.txt; head -2 gpurun_out/r4g/host_profile.txt
python tools/conv_survey.py > gpurun_out/r4g/conv_survey.txt 2>&1; tail -1 gpurun_out/r4g/conv_survey.txt
(python tools/ab_step.py hip.WGRAD_STREAM; python tools/ab_step.py env:DFINE_GRAD_FANIN; python tools/ab_step.py env:DFINE_CONV1X1_XIMG) 2>/dev/null | grep -v amdgpu > gpurun_out/r4g/ab_switches.txt; cat gpurun_out/r4g/ab_switches.txt
python bench.py --model x --img 960 --batch 8 --mask 1 --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null > gpurun_out/r4g/bench_x_mask.json; python -c "
import json; d=json.loads(open('gpurun_out/r4g/bench_x_mask.json').read().strip().splitlines()[-1]); print('x mask', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
python bench.py --model s --img 640 --batch 16 --dtype fp32 --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null > gpurun_out/r4g/bench_s_fp32.json; python -c "
import json; d=json.loads(open('gpurun_out/r4g/bench_s_fp32.json').read().strip().splitlines()[-1]); print('s fp32', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
