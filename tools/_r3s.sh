mkdir -p gpurun_out/r3s
timeout 900 python -m pytest tests/test_conv_epilogue_gpu.py tests/test_optim_gpu.py tests/test_dist_gpu.py tests/test_model_gpu.py tests/test_conv_mfma_gpu.py -x -q > gpurun_out/r3s/t1.log 2>&1; tail -8 gpurun_out/r3s/t1.log
for cfg in "1 12" "0 12" "1 6" "1 1000" "1 12" "0 12"; do set -- $cfg; DFINE_WGRAD_STREAM=$1 DFINE_WGRAD_GROUP_AT=$2 python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream $1 group_at $2', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
python tools/step_profile.py 2>/dev/null > gpurun_out/r3s/step_profile.txt; head -24 gpurun_out/r3s/step_profile.txt
