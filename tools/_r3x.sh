mkdir -p gpurun_out/r3x
timeout 900 python -m pytest tests/test_gemm_attn_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r3x/t1.log 2>&1; tail -4 gpurun_out/r3x/t1.log
for cfg in 1 0 1 0; do DFINE_ATTN_BWD_FORK=$cfg python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('attn_fork $cfg', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
