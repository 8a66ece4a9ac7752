mkdir -p gpurun_out/r3u
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r3u/tests.log 2>&1; tail -4 gpurun_out/r3u/tests.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_optim_gpu.py -x -q 2>&1 | tail -1; done
for cfg in 1 0 1; do DFINE_WGRAD_STREAM=$cfg python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream $cfg', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
