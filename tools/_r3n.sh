mkdir -p gpurun_out/r3p
timeout 1500 python -m pytest tests/test_conv_units_gpu.py tests/test_model_gpu.py tests/test_optim_gpu.py tests/test_dist_gpu.py tests/test_mask_path.py -x -q > gpurun_out/r3p/tests.log 2>&1; tail -3 gpurun_out/r3p/tests.log
for i in 1 2; do for f in 1 0; do DFINE_FUSE_CONV_BN=$f python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse', $f, d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done; done
