mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_dist_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r3v/t1.log 2>&1; tail -4 gpurun_out/r3v/t1.log
for cfg in 48 0 24 96 48 0; do DFINE_EARLY_REDUCE=$cfg python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early_reduce $cfg', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
