mkdir -p gpurun_out/r3t
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r3t/tests.log 2>&1; tail -4 gpurun_out/r3t/tests.log
for cfg in 1 0 1 0; do DFINE_WGRAD_STREAM=$cfg python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream $cfg', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
python tools/gpu_gaps.py > gpurun_out/r3t/gaps.txt 2>&1; tail -30 gpurun_out/r3t/gaps.txt
