mkdir -p gpurun_out/r4b
python bench.py > gpurun_out/r4b/bench_line.json 2> gpurun_out/r4b/bench.err; tail -2 gpurun_out/r4b/bench.err | cut -c1-300
bash tools/rocprof_stats.sh r4b_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/r4b/rocprof.log 2>&1; tail -3 gpurun_out/r4b/rocprof.log | cut -c1-200
python tools/step_profile.py 2>/dev/null > gpurun_out/r4b/step_profile.txt; head -22 gpurun_out/r4b/step_profile.txt
python tools/host_profile.py 2>/dev/null > gpurun_out/r4b/host_profile.txt; head -3 gpurun_out/r4b/host_profile.txt
python bench.py --model x --img 960 --batch 8 --mask 1 --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null > gpurun_out/r4b/bench_x_mask.json; python -c "
import json; d=json.loads(open('gpurun_out/r4b/bench_x_mask.json').read().strip().splitlines()[-1]); print('x mask', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
python bench.py --model s --img 640 --batch 16 --dtype fp32 --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null > gpurun_out/r4b/bench_s_fp32.json; python -c "
import json; d=json.loads(open('gpurun_out/r4b/bench_s_fp32.json').read().strip().splitlines()[-1]); print('s fp32', d['value'], d['ms_per_step'], d['median_ms_per_step'])"
