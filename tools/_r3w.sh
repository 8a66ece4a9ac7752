mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_conv_mfma_gpu.py tests/test_conv_epilogue_gpu.py tests/test_conv_units_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r3w/t1.log 2>&1; tail -4 gpurun_out/r3w/t1.log
python tools/conv_survey.py > gpurun_out/r3w/conv_survey.txt 2>&1; tail -38 gpurun_out/r3w/conv_survey.txt | cut -c1-200
for cfg in 1 0 1 0; do DFINE_CONV1X1_XIMG=$cfg python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ximg $cfg', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
