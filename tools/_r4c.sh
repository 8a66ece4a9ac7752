mkdir -p gpurun_out/r4c
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r4c/tests.log 2>&1; tail -4 gpurun_out/r4c/tests.log
python tools/host_profile.py 2>/dev/null > gpurun_out/r4c/host_profile.txt; head -2 gpurun_out/r4c/host_profile.txt
python tools/ab_step.py hip.WGRAD_STREAM 2>/dev/null | tail -3
python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"
python tools/step_aten_shapes.py "" 30 2>/dev/null > gpurun_out/r4c/aten_all.txt
