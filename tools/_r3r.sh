mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_conv_epilogue_gpu.py -x -q > gpurun_out/r3r/t1.log 2>&1; tail -15 gpurun_out/r3r/t1.log
timeout 1500 python -m pytest tests/test_conv_units_gpu.py tests/test_conv_mfma_gpu.py tests/test_model_gpu.py tests/test_optim_gpu.py tests/test_dist_gpu.py -x -q > gpurun_out/r3r/t2.log 2>&1; tail -5 gpurun_out/r3r/t2.log
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1"; do set -- $cfg; DFINE_CONV_BN_STATS=$1 DFINE_GRAD_FANIN=$2 python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stats $1 fanin $2', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
python tools/step_profile.py 2>/dev/null > gpurun_out/r3r/step_profile.txt; head -24 gpurun_out/r3r/step_profile.txt
